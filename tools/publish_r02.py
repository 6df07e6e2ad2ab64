#!/usr/bin/env python
"""Copies the summaries of gpurun_out/prof_r02 (tools/profile_round.sh r02 + tools/profile_counters.sh r02) into profiles/r02_* (the analysis header of the counters file is
kept from the committed version; only its table is refreshed)."""
import json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "prof_r02"); P = os.path.join(ROOT, "profiles")
d = json.load(open(O + "/bench_n1.json"))
json.dump(d, open(P + "/r02_bench_n1.json", "w"), indent=1)
lines = open(O + "/kernel_stats.md").read().splitlines()
body = [l for l in lines[2:] if ("dmv::" in l or "__amd_rocclr_copyBuffer" in l)]
head = open(P + "/r02_kernel_stats.md").read().split("| kernel |")[0]
open(P + "/r02_kernel_stats.md", "w").write(head + "\n".join(lines[:2] + body[:70]) + "\n")
def filt(path):
    return [l for l in open(path).read().splitlines() if l.startswith("| kernel") or l.startswith("|---") or "dmv::" in l]
head = open(P + "/r02_pmc_hbm_traffic.md").read().split("## FETCH_SIZE")[0]
open(P + "/r02_pmc_hbm_traffic.md", "w").write(head + "## FETCH_SIZE\n" + "\n".join(filt(O + "/pmc_FETCH_SIZE.md")) + "\n\n## WRITE_SIZE\n" + "\n".join(filt(O + "/pmc_WRITE_SIZE.md")) + "\n")
rows = []
for g in ("sq_insts", "sq_time", "tcp", "tcp2", "tlb", "tcc", "ta"):
    rows += [l for l in open(O + "/counters_%s.md" % g).read().splitlines() if "k_track_lm<256" in l or "k_build_pyramids" in l]
head = open(P + "/r02_counters_k_track_lm.md").read().split("| kernel | counter |")[0]
open(P + "/r02_counters_k_track_lm.md", "w").write(head + "| kernel | counter | dispatches | avg | min | max |\n|---|---|---|---|---|---|\n" + "\n".join(rows) + "\n")
out = ["# r02 — BA: host-side split of the GN iteration (DMVIO_HIP_BA_TIMING=1, no synchronisation added) and kernel timeline (rocprofv3 --kernel-trace of tools/ba_loop.py, 1x MI355X)", ""]
out += [l for l in open(O + "/ba_timing.log").read().splitlines() if l.startswith("[dmvio_hip_ba]")]
out += ["", "## tools/ba_loop.py under rocprofv3 (one dmvio_hip_ba_gn_iteration call per iteration from Python; optimize(6) on fresh windows; the profiler adds ~15 % to these host-clock figures)"]
out += [l for l in open(O + "/ba_loop.log").read().splitlines() if ("GN-iter" in l or "decision" in l or "optimize(" in l)]
out += ["", "## kernel timeline (us): start, duration, gap to the previous kernel, workgroups — set-up, then accepted iterations of the first optimize"]
out += open(O + "/ba_timeline.txt").read().splitlines()
open(P + "/r02_ba_host_split_and_timeline.txt", "w").write("\n".join(out) + "\n")
vals = {}
for l in rows:
    c = [x.strip() for x in l.split("|")]
    if "k_track_lm" in c[1]: vals[c[2]] = float(c[4])
print(json.dumps({k: d[k] for k in ("value", "ms_per_step")}), json.dumps(d["roofline"])[:400])
print("ba", d["ba"]["value"], d["ba"]["value_per_call_api"], d["ba"]["optimize6_ms"], d["ba"]["value_single_threaded_order"], d["ba"]["cpu_baseline"]["value"])
print("live", d["live"]["value"], d["live"]["ms_track"], "vio", d["vio_handoff"]["handoff"]["ms_per_frame"], d["vio_handoff"]["device_lm"]["ms_per_frame"], "pcie", d["pcie"]["value"], "cpu", d["cpu_baseline"]["value"], "trace", d["trace"]["value"])
print("sweep", [(p["frames_per_step"], p["value"], p["track_kernel_us_per_frame"]) for p in d["batch_sweep"]["points"]])
print("counters", {k: vals[k] for k in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "TCP_GATE_EN1_sum", "TCP_GATE_EN2_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCC_EA0_RDREQ_sum") if k in vals})
