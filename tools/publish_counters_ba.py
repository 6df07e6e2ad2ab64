#!/usr/bin/env python
"""profiles/<round>_counters_ba.md from the outputs of tools/profile_counters_ba.sh (gpurun_out/prof_<dir>/counters_ba_*.md): hardware counters of the batched bundle
adjustment's kernels at 64 windows per launch.   usage: python tools/publish_counters_ba.py r06 <dir_before> <dir_after>
The values used are the MAX over the dispatches of a kernel: the 64-window launch of the probe's profiled repetition (one group, one stream, alone on the device); the
smaller dispatches (the three stream groups of the other repetitions) are in the raw tables."""
import os, sys, glob
R = sys.argv[1]; DB = sys.argv[2]; DA = sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(d):
    C = {}
    raw = []
    for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "prof_" + d, "counters_ba_*.md"))):
        for l in open(f):
            p = [x.strip() for x in l.strip().strip("|").split("|")]
            if len(p) < 6 or not p[0].startswith("dmv::k_ba_"):
                continue
            k = p[0].split("(")[0].replace("dmv::", "")
            C.setdefault(k, {})[p[1]] = float(p[5])
            raw.append((k, p[1], p[2], p[3], p[4], p[5]))
    return C, raw


CB, rawB = load(DB); CA, rawA = load(DA)
RES = 64 * 12865          # residuals of the 64-window launch (bench window: 12 865 residuals)
out = ["# %s — hardware counters of the batched bundle adjustment at W = 64 windows per launch (`tools/profile_counters_ba.sh`, `tools/ba_batch_probe.py 64`, 1x MI355X)" % R, "",
       "One `rocprofv3 --kernel-trace --pmc <group>` run per counter group (no other tracing).  Values: the 64-window dispatch of each kernel (max over its dispatches; per-dispatch",
       "sums over the device).  SQ_* cycle counters count quad-cycles (4 clocks).  `before` = the round's first measurement (one lane per residual, every pattern pixel tapping the",
       "image itself, the point's step formed inside the kernel); `after` = HEAD (the step in `k_ba_resubstitute_b`, the taps of four pattern pixels from a 6x8 window per lane in LDS).", ""]


def section(k, C, label, wg_waves):
    c = C[k]
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    o = ["### `%s` — %s" % (k, label), ""]
    o.append("* duration under the profiler: GRBM_GUI_ACTIVE / 8 XCDs = %.0f k clocks = %.0f us at 2.4 GHz; SQ_BUSY_CU_CYCLES = 256 CUs x %.0f k." % (cyc / 1e3, cyc / 2400.0, c["SQ_BUSY_CU_CYCLES"] / 256e3))
    o.append("* %.0f wavefronts; resident waves per SIMD on average = 4 x SQ_WAVE_CYCLES / (1024 SIMDs x clocks) = **%.2f**; waiting for an instruction's operands (SQ_WAIT_INST_ANY) %.0f %% of the wave-cycles."
             % (c["SQ_WAVES"], 4 * c["SQ_WAVE_CYCLES"] / (1024 * cyc), 100 * c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]))
    o.append("* instructions per wavefront: VALU %.0f, SALU %.0f, LDS %.0f, vector-memory reads %.0f, writes %.0f; VALU busy = 4 x SQ_ACTIVE_INST_VALU / (1024 x clocks) = **%.0f %%**."
             % (c["SQ_INSTS_VALU"] / c["SQ_WAVES"], c["SQ_INSTS_SALU"] / c["SQ_WAVES"], c["SQ_INSTS_LDS"] / c["SQ_WAVES"], c["SQ_INSTS_VMEM_RD"] / c["SQ_WAVES"], c["SQ_INSTS_VMEM_WR"] / c["SQ_WAVES"],
                100 * 4 * c["SQ_ACTIVE_INST_VALU"] / (1024 * cyc)))
    o.append("* vector L1 (TCP): %.1f M tag look-ups = %.1f per vector-memory instruction = **%.1f per residual**; busy (TCP_GATE_EN2 / EN1) **%.0f %%**, stalled on outstanding misses (TCP_PENDING_STALL) %.0f %% of the CU-cycles;"
             % (c["TCP_TOTAL_CACHE_ACCESSES_sum"] / 1e6, c["TCP_TOTAL_CACHE_ACCESSES_sum"] / (c["SQ_INSTS_VMEM_RD"] + c["SQ_INSTS_VMEM_WR"]), c["TCP_TOTAL_CACHE_ACCESSES_sum"] / RES,
                100 * c["TCP_GATE_EN2_sum"] / c["TCP_GATE_EN1_sum"], 100 * c["TCP_PENDING_STALL_CYCLES_sum"] / c["SQ_BUSY_CU_CYCLES"]))
    o.append("  read requests to the L2 %.1f M = **%.1f per residual**, write requests %.1f M = %.1f per residual; tag-conflict stalls %.0f %%, data-return stalls (TA) %.0f %% of the CU-cycles."
             % (c["TCP_TCC_READ_REQ_sum"] / 1e6, c["TCP_TCC_READ_REQ_sum"] / RES, c["TCP_TCC_WRITE_REQ_sum"] / 1e6, c["TCP_TCC_WRITE_REQ_sum"] / RES,
                100 * c["TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"] / c["SQ_BUSY_CU_CYCLES"], 100 * c["TCP_TCP_TA_DATA_STALL_CYCLES_sum"] / c["SQ_BUSY_CU_CYCLES"]))
    o.append("* LDS: %.1f M index-active quad-cycles, bank conflicts %.1f %% of them." % (c["SQ_LDS_IDX_ACTIVE"] / 1e6, 100 * c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1)))
    o.append("")
    return o


out += ["## `k_ba_linearize_b1` (one lane per residual; 823 360 residuals x 8 bilinear interpolations per launch; SURVEY 8(d): 464 algorithmic bytes per residual)", ""]
out += section("k_ba_linearize_b1", CB, "before", 4)
out += section("k_ba_linearize_b1", CA, "after (HEAD)", 4)
cb, ca = CB["k_ba_linearize_b1"], CA["k_ba_linearize_b1"]
out += ["What the counters said and what was done about it.  The kernel was NOT short of waves (3 per SIMD) and not issue-bound (VALU busy 20 %%): the vector L1 was busy %.0f %% of the time —"
        % (100 * cb["TCP_GATE_EN2_sum"] / cb["TCP_GATE_EN1_sum"]),
        "one tag look-up per distinct 128-byte line per instruction, and with one residual per lane every lane of a load touches its own line (%.0f look-ups per vector-memory instruction)."
        % (cb["TCP_TOTAL_CACHE_ACCESSES_sum"] / (cb["SQ_INSTS_VMEM_RD"] + cb["SQ_INSTS_VMEM_WR"])),
        "%.0f look-ups per residual: 32 for the 8 x 4 tap rows (%.0f of them went on to the L2: every tap row a miss — the 64 target images under a wavefront's lanes share nothing and the"
        % (cb["TCP_TOTAL_CACHE_ACCESSES_sum"] / RES, cb["TCP_TCC_READ_REQ_sum"] / RES),
        "rows a lane touches again for its next pattern pixel have been evicted by then), ~20 for the 208-byte record a lane writes in sixteen-byte pieces, ~20 for the point / pair / residual",
        "metadata and the point's back-substitution, which every residual of a point repeated.  Changes (all bit-exact: `tests/test_ba_batch_gpu.py`): (i) the back-substitution + point step",
        "moved into a kernel of its own in the eight-lanes-per-point form (`k_ba_resubstitute_b`: the loads of a point's residuals side by side instead of one dependent pair after the other",
        "inside every wavefront's critical path); (ii) the pattern is sorted by rows, so each half of it (4 pixels) fits a window of 6 rows x 8 columns: the lane fetches that window once into",
        "LDS (12 row requests per residual instead of 32) and interpolates from there; a first form with ONE 8 x 8 window per residual halved the L2 requests too but cost a wave of",
        "occupancy (16.6 KB of LDS per wavefront) and was slower (292 vs 266 us); the decision pass's staging arrays moved into the same LDS so that three workgroups per CU still fit.",
        "Result: L2 read requests per residual %.1f -> %.1f, tag look-ups %.0f -> %.0f, the stepped linearisation of 64 windows 266 -> 210-214 us (HIP events, un-profiled) ="
        % (cb["TCP_TCC_READ_REQ_sum"] / RES, ca["TCP_TCC_READ_REQ_sum"] / RES, cb["TCP_TOTAL_CACHE_ACCESSES_sum"] / RES, ca["TCP_TOTAL_CACHE_ACCESSES_sum"] / RES),
        "**0.18 -> 0.22-0.23 of the 8 TB/s peak by algorithmic bytes**.  The round-5 verdict's 0.35 is not reached: the vector L1 is still busy %.0f %% of the time with %.0f look-ups per residual, of which"
        % (100 * ca["TCP_GATE_EN2_sum"] / ca["TCP_GATE_EN1_sum"], ca["TCP_TOTAL_CACHE_ACCESSES_sum"] / RES),
        "the record stores (one 16-byte piece per lane and instruction: 13 + 8 accesses per residual) and the two-per-row window loads (24) are what is left; both are a property of 'one residual per",
        "lane against 64 different images', which only a residual order sorted by (target, image tile) would change — that order is the graph's (residuals of a point contiguous: the per-point",
        "sums and the Schur member lists are built on it).", ""]
out += ["## `k_ba_accumulate_b` (50 176 workgroups per 64-window launch: per window 16 calibration + 256 (host, target) + 512 Schur workgroups, one wavefront per bucket part)", ""]
out += section("k_ba_accumulate_b", CB, "at HEAD (round 6: global instead of flat loads, 16-byte staging loads of the records)", 4)
c = CB["k_ba_accumulate_b"]
out += ["The same picture, more so: the vector L1 is busy %.0f %% of the time with %.0f M tag look-ups (%.1f per vector-memory instruction: the members of a bucket are gathered one per lane — 2 x 32 bytes"
        % (100 * c["TCP_GATE_EN2_sum"] / c["TCP_GATE_EN1_sum"], c["TCP_TOTAL_CACHE_ACCESSES_sum"] / 1e6, c["TCP_TOTAL_CACHE_ACCESSES_sum"] / c["SQ_INSTS_VMEM_RD"]),
        "of `JpJdF` out of two 208-byte records plus the point's scalars), VALU busy %.0f %%, LDS bank conflicts %.1f %%.  Every residual's `JpJdF` is fetched by 2F Schur buckets; what would cut"
        % (100 * 4 * c["SQ_ACTIVE_INST_VALU"] / (1024 * c["GRBM_GUI_ACTIVE"] / 8.0), 100 * c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]),
        "the look-ups is a per-host-keyframe form (a workgroup walks a keyframe's points once, keeps the 64 (t1, t2) blocks of that host in registers and adds each point's outer products in",
        "point order — the reference's order inside a bucket): designed (DESIGN.md section 7), not built this round.", ""]
out += ["## `k_ba_stitch_b`", ""] + section("k_ba_stitch_b", CB, "at HEAD", 8)
out += ["## Raw values, HEAD (`%s`): avg / min / max per dispatch" % DA, "", "| kernel | counter | dispatches | avg | min | max |", "|---|---|---|---|---|---|"]
out += ["| %s | %s | %s | %s | %s | %s |" % r for r in rawA]
out += ["", "## Raw values, before (`%s`; also the source of the `k_ba_accumulate_b` / `k_ba_stitch_b` sections)" % DB, "", "| kernel | counter | dispatches | avg | min | max |", "|---|---|---|---|---|---|"]
out += ["| %s | %s | %s | %s | %s | %s |" % r for r in rawB]
open(os.path.join(ROOT, "profiles", R + "_counters_ba.md"), "w").write("\n".join(out) + "\n")
print("published", R + "_counters_ba.md")
