#!/usr/bin/env python
"""Micro-benchmarks of the tracker kernels (diagnostics, not the headline bench)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
import torch
pkg = graft.load_package()
import dmvio_amd.synth as synth

def main():
    w = h = 512
    B = int(os.environ.get("MB_BATCH", "1"))
    case = synth.tracking_case(w, h, n_ref=2000, n_frames=2, xi_jitter=0.3)
    ctx = pkg.Context(w, h, n_slots=B + 2)
    stream = torch.cuda.Stream(); ctx.set_stream(stream.cuda_stream)
    trk = pkg.CoarseTrackerHip(ctx); trk.makeK(case["K4"])
    ctx.frame_upload(0, case["ref_img"])
    for i in range(B): ctx.frame_upload(1 + i, case["frames"][i % 2]["img"])
    trk.setCoarseTrackingRef(0, case["u"], case["v"], case["idepth"], case["hdiF"])
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    # single-eval path (2 kernels + D2H + sync): host wall time
    for lvl in range(4):
        for _ in range(3): trk.eval(lvl, 1, case["frames"][0]["pose7"], (0, 0))
        t0 = time.perf_counter()
        for _ in range(50): trk.eval(lvl, 1, case["frames"][0]["pose7"], (0, 0))
        dt = (time.perf_counter() - t0) / 50
        print("eval lvl %d n=%d: host wall %.1f us per call" % (lvl, trk.pc_n(lvl), dt * 1e6))
    slots = [1 + i for i in range(B)]
    trk.stage(slots, [ident] * B, [(0, 0)] * B)
    for rep in range(3):
        ts = []
        for _ in range(20):
            e0.record(stream); trk.launch(); e1.record(stream); e1.synchronize(); ts.append(e0.elapsed_time(e1))
        r = trk.fetch(); ev, pe = trk.last_work(); tk = trk.last_ticks()
        print("track_lm B=%d: kernel %.1f us (min %.1f)  evals %d  in-kernel us/problem: control %.1f eval %.1f" % (B, 1e3 * np.mean(ts), 1e3 * np.min(ts), ev, tk[0] / 100 / B, tk[1] / 100 / B))
    # whole call (stage + launch + fetch): host wall time
    for rep in range(2):
        t0 = time.perf_counter()
        for _ in range(100): trk.stage(slots, [ident] * B, [(0, 0)] * B); trk.launch(); trk.fetch()
        print("track call B=%d: host wall %.1f us" % (B, (time.perf_counter() - t0) / 100 * 1e6))

main()
