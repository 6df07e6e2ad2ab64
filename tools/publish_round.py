#!/usr/bin/env python
"""Copies the summaries of gpurun_out/prof_<round> (tools/profile_round.sh <round> + the driver's own bench command) into profiles/<round>_*.
usage: python tools/publish_round.py r06 [--force]
Refuses to run when the sources the GPU box profiled (content hash of dm-vio_amd/csrc + include, written by tools/profile_round.sh into gpurun_out/prof_<round>/csrc_hash.txt)
are not the sources of this tree: published profiles are profiles of HEAD.  Every published markdown file carries the hash and the commit under its title."""
import json, os, sys, subprocess, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "r06"
O = os.path.join(ROOT, "gpurun_out", "prof_" + R); P = os.path.join(ROOT, "profiles")


def csrc_hash(root=ROOT):
    h = hashlib.sha256()
    files = []
    for top in ("dm-vio_amd/csrc", "include"):
        for dp, _, fs in os.walk(os.path.join(root, top)):
            files += [os.path.join(dp, f) for f in fs]
    for f in sorted(files):
        h.update(os.path.relpath(f, root).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__" and "--hash" in sys.argv:
    print(csrc_hash()); sys.exit(0)
here = csrc_hash()
there = open(O + "/csrc_hash.txt").read().split()[0] if os.path.exists(O + "/csrc_hash.txt") else None
if there != here and "--force" not in sys.argv:
    sys.exit("publish_round: gpurun_out/prof_%s was profiled on sources %s, this tree's dm-vio_amd/csrc + include hash to %s — re-run tools/profile_round.sh %s" % (R, there, here, R))
try:
    HEAD = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    DIRTY = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "dm-vio_amd/csrc", "include"], capture_output=True, text=True).stdout.strip() != ""
except Exception:
    HEAD, DIRTY = "?", False
STAMP = "(sources: dm-vio_amd/csrc + include content hash %s; tree at commit %s%s)" % (here, HEAD, " + uncommitted changes" if DIRTY else "")


def publish(name, text):
    """profiles/<round>_<name>: the title line, the source stamp, the rest"""
    lines = text.split("\n", 1)
    open(os.path.join(P, R + "_" + name), "w").write(lines[0] + "\n\n" + STAMP + "\n" + (lines[1] if len(lines) > 1 else ""))


d = json.loads(open(O + "/bench_n1.json").read().strip().splitlines()[-1])
d["_published_from"] = STAMP; json.dump(d, open(P + "/" + R + "_bench_n1.json", "w"), indent=1)
dd = json.loads(open(O + "/bench_driver_cmd.json").read().strip().splitlines()[-1])
dd["_published_from"] = STAMP; json.dump(dd, open(P + "/" + R + "_bench_n1_driver_command.json", "w"), indent=1)
rf = d["roofline"]; ba = d["ba"]
lines = open(O + "/kernel_stats.md").read().splitlines()
body = [l for l in lines[2:] if ("dmv::" in l or "__amd_rocclr_copyBuffer" in l)]
head = ("# " + R + " — `rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu --no-traffic` (defaults: 4096 frames per step, 200 steps, batch sweep, PCIe legs, BA / trace / overlap / live / "
        "VIO legs), 1x MI355X\n\nProduced by `tools/profile_round.sh <round>` + `tools/publish_round.py`; durations in microseconds from the rocpd database (`tools/rocprof_summary.py`).  The dominant kernel of "
        "the headline step is `k_track_lm<256, 4>` with 4096 workgroups (one per frame): avg below vs %.4f ms by HIP events on its stream in the un-profiled run (`profiles/<round>_bench_n1.json`: "
        "%.0f frames/s, algorithmic fraction %.3f, HBM-counter fraction %.3f).  BA kernels: `k_ba_linearize` avg below vs %.1f us by HIP events incl. the gap to the next launch "
        "(`ba.roofline.chain_us`); `ba.value` = %.0f accepted GN iterations/s on fresh windows (optimize(6) = %.3f ms), %.0f/s on the converged (reject-dominated) loop.\n\n"
        % (rf["kernel_ms"], d["value"], rf["frac"], rf.get("frac_hbm_counter", float("nan")), ba["roofline"]["kernel_us"], ba["value"], ba["optimize6_ms"], ba["value_converged_loop"]))
kt = [l for l in body if "k_track_lm<256, 4, false>" in l and "| 4096 |" in l]
kp = [l for l in body if "k_build_pyramids_reg<true>" in l and "| 131072 |" in l]
def col(l, i): return float([x.strip() for x in l.strip().strip("|").split("|")][i])
tail = ""
if kt and kp:
    tail = ("\n## The two clocks\n\n`k_track_lm<256,4>` (4096 workgroups): rocprofv3 avg %.0f us / min %.0f / max %.0f over %d dispatches vs %.0f us by HIP events in the un-profiled run.  The profiler's "
            "figure is each dispatch's own begin -> end stamp under its instrumentation, over ALL launches of the process — the warm-up and the settle phase, during which the clocks ramp "
            "(the tail up to the max), included; the HIP-event figure is (event behind the launch - event in front of it) on the kernel's stream, averaged over the K timed steps only.  The "
            "profiler's min is the steady-state figure and agrees with the events to ~1 %%; `roofline.achieved` uses the events (the contract's clock), this table is the cross-check.  "
            "`k_build_pyramids_reg<true>` (131072 workgroups): rocprofv3 avg %.0f us; the line's step time minus the tracker's kernel time (%.0f us here) is NOT this kernel's duration: it also "
            "holds the gap between the two launches of a step and whatever of the host's unpacking of the previous batch is not hidden behind them.  The build's own bandwidth is quoted from the "
            "profiler's duration (and from HIP events around the build alone in `pcie.raw_u8.resident` / `tools/time_pyramids.py`).\n"
            % (col(kt[0], 4), col(kt[0], 5), col(kt[0], 6), int(col(kt[0], 2)), 1e3 * rf["kernel_ms"], col(kp[0], 4), 1e3 * (d["ms_per_step"] - rf["kernel_ms"])))
publish("kernel_stats.md", head + "\n".join(lines[:2] + body[:80]) + "\n" + tail)
def filt(path):
    return [l for l in open(path).read().splitlines() if l.startswith("| kernel") or l.startswith("|---") or "dmv::" in l]
head = ("# " + R + " — HBM traffic counters (`rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`, separate passes, no other tracing), 1x MI355X\n\n"
        "`bench.py --no-cpu --no-traffic --no-sweep --no-pcie --steps 3 --warmup 1 --ba-iters 20`.  Values are KiB as the counter reports them; on gfx950 FETCH_SIZE tallies 128-byte reads at half "
        "their size (MI355X_MICROARCH.md): the in-run calibration on `k_build_pyramids` (reads exactly B x w x h x 4 bytes per launch) gives the factor 2.000 that `bench.py` applies "
        "(`roofline.traffic_source`).  `k_track_lm<256,4>` with 4096 workgroups: %.2f GB per launch by the counter = %.2fx its %.2f GB of algorithmic bytes.\n\n"
        % (rf["traffic"] / 1e9, rf["traffic"] / rf["algorithmic_bytes_per_launch"], rf["algorithmic_bytes_per_launch"] / 1e9))
publish("pmc_hbm_traffic.md", head + "## FETCH_SIZE\n" + "\n".join(filt(O + "/pmc_FETCH_SIZE.md")) + "\n\n## WRITE_SIZE\n" + "\n".join(filt(O + "/pmc_WRITE_SIZE.md")) + "\n")
import subprocess, sys
summary = subprocess.run([sys.executable, ROOT + "/tools/ba_split_summary.py",
                          R + " — BA: host-side split of the GN iteration (DMVIO_HIP_BA_TIMING=1, no synchronisation added) and kernel timeline (rocprofv3 --kernel-trace of tools/ba_loop.py), 1x MI355X",
                          O + "/ba_timing.log", O + "/ba_timeline.txt", O + "/ba_loop.log"], capture_output=True, text=True, check=True).stdout
publish("ba_host_split_and_timeline.md", summary)
bw = ba.get("batched_windows")
if bw and "error" not in bw:
    rows = ["# " + R + " — `ba.batched_windows`: dmvio_hip_ba_optimize_batch, W windows per launch sequence on the device-resident loop (`bench.py`, 1x MI355X)", "", bw.get("what", ""), "",
            "| W | accepted it/s, default accumulation order (4 partial accumulators per bucket) | accepted it/s, single-threaded order (`set_accumulators(1)`: the bit-exact replay) | wall ms per optimize(6) of the batch (default order) | device ms | stepped linearisation us | k_ba_linearize_b: TB/s algorithmic | of 8 TB/s |", "|---|---|---|---|---|---|---|---|"]
    for r in bw["sweep"]:
        rows.append("| %d | %.0f | %.0f | %.3f | %.3f | %.1f | %.3f | %.4f |" % (r["windows"], r["value"], r.get("value_single_threaded_order", float("nan")), r["wall_ms"], r["device_ms"], r["k_ba_linearize_b_us"], r["k_ba_linearize_b_GBs"] / 1e3, r["k_ba_linearize_b_frac"]))
    rows += ["", "Roofline kernel `%s`: %.1f us for the stepped linearisation of all %d windows = %.1f GB/s algorithmic = %.4f of 8 TB/s." % (bw["roofline"]["kernel"], bw["roofline"]["kernel_us"], bw["at_windows"], bw["roofline"]["achieved"], bw["roofline"]["frac"])]
    rows += ["", "Single window, host-driven loop (the default of dmvio_hip_ba_optimize): %.3f ms per optimize(6) = %.0f accepted it/s." % (ba["optimize6_ms"], ba["value"])]
    publish("ba_batched_windows.md", "\n".join(rows) + "\n")
di = d.get("drop_in")
if di and "error" not in di:
    rows = ["# " + R + " — the reference's own FullSystem, all-CPU vs with its hot-path members on libdmvio_hip.so (`bench.py` -> `drop_in`, 1x MI355X box)", "",
            di["what"] + ".", "",
            "| run | wall clock of the %d addActiveFrame calls (s) | ms per frame after initialisation |" % di["frames"], "|---|---|---|",
            "| all-CPU, the reference's default threading (multiThreading = true, its own thread-pooled initialiser) | %.3f | %.2f |" % (di["all_cpu_s"], di["ms_per_frame_after_initialisation"]["all_cpu"]),
            "| all-CPU, single-threaded, sequential initialiser (bit-reproducible: the trajectory baseline) | %.3f | %.2f |" % (di["all_cpu_single_threaded_s"], di["ms_per_frame_after_initialisation"]["all_cpu_single_threaded"]),
            "| HIP-backed (seven members + makeKeyFrame through tests/dropin/dmvio_hip_adapter.cpp) | %.3f | %.2f |" % (di["hip_backed_s"], di["ms_per_frame_after_initialisation"]["hip_backed"]),
            "", "Trajectory HIP-backed vs the single-threaded baseline: rmse %.2e m, max %.2e m (bar 1e-3 m); the two all-CPU runs against each other: rmse %.2e m.  %d keyframe optimisations, adapter failures %d."
            % (di["traj_rmse_m"], di["traj_max_m"], di["reference_own_spread_rmse_m"], di["keyframe_optimisations"], di["adapter_failures"]), "",
            "## inclusive seconds under the reference's own profiler labels (util/TimeMeasurement scopes)", "", "| scope | all-CPU (default threading) | HIP-backed |", "|---|---|---|"]
    for k in di["scopes_all_cpu"]:
        rows.append("| %s | %.4f | %.4f |" % (k, di["scopes_all_cpu"][k], di["scopes_hip_backed"].get(k, 0.0)))
    rows += ["", "## seconds inside the replaced members, HIP-backed run", "", "| member | seconds |", "|---|---|"]
    for k, v in di["seconds_in_replaced_members"].items():
        rows.append("| %s | %.4f |" % (k, v))
    if "adapter_ms_per_keyframe" in di:
        a = di["adapter_ms_per_keyframe"]; am = di.get("adapter_ms_per_keyframe_default_threading", {})
        rows += ["", "## FullSystem::optimize member of the adapter, ms per keyframe (window graph resident: `dmvio_hip_graph_*`, %d forwarded EnergyFunctional mutations, %d resyncs)" % (di["window_graph"]["forwarded_mutations"], di["window_graph"]["resyncs"]), "",
                 "| part | single-threaded run | multiThreading = true |", "|---|---|---|"]
        for k in ("hand_over", "dmvio_hip_ba_optimize", "write_back"):
            rows.append("| %s | %.3f | %.3f |" % (k, a[k], am.get(k, float("nan"))))
        rows += ["", "HIP-backed with the reference's default threading for what it keeps doing itself: %.3f s (%.2fx the all-CPU default)." % (di["hip_backed_default_threading_s"], di["speedup_vs_reference_default_same_threading"])]
    v = di.get("vio")
    if v and "error" not in v:
        rows += ["", "## the reference's DEFAULT configuration (setting_useIMU = setting_useGTSAMIntegration = true) live through the adapter", "", v["what"] + ".", "",
                 "all-CPU (single-threaded) %.3f s, HIP-backed %.3f s; trajectory rmse %.2e m, max %.2e m; adapter failures %d, lost %s." % (v["all_cpu_single_threaded_s"], v["hip_backed_s"], v["traj_rmse_m"], v["traj_max_m"], v["adapter_failures"], v["lost"]),
                 "adapter calls: " + ", ".join("%s %d" % kv for kv in v["adapter_calls"].items()) + ".",
                 "facade member calls all-CPU %s / HIP-backed %s." % (v["facade_calls_all_cpu"], v["facade_calls_hip_backed"]),
                 "optimize member per keyframe (ms): " + ", ".join("%s %.3f" % kv for kv in v["adapter_ms_per_keyframe"].items()) + "."]
    t = di.get("track_new_coarse")
    if t:
        rows += ["", "FullSystem::trackNewCoarse adapter member: " + ", ".join("%s %d" % kv for kv in t.items()) + "."]
    rows += ["", di["note"]]
    publish("fullsystem_scopes.md", "\n".join(rows) + "\n")
print(json.dumps({k: d[k] for k in ("value", "ms_per_step")}), json.dumps(rf)[:300])
print("driver cmd:", dd["value"], dd["ms_per_step"], dd["roofline"]["frac"])
for k in ("value", "optimize6_ms", "value_converged_loop", "value_per_call_api", "value_single_threaded_order", "gtsam_handoff"):
    print("ba", k, ba.get(k))
print("ba roofline", json.dumps(ba["roofline"])[:500]); print("ba cpu", json.dumps(ba["cpu_baseline"])[:300])
print("live", d["live"]["value"], "vio", d["vio_handoff"]["handoff"]["ms_per_frame"], "pcie", d["pcie"]["value"], d["pcie"]["raw_u8"]["value"], "cpu", d["cpu_baseline"]["value"], "trace", d["trace"]["value"], d["trace"]["cpu_baseline"]["value"])
