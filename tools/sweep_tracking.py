#!/usr/bin/env python
"""Bandwidth-asymptote sweep of the device-resident tracker: template size N_ref x batch of B alignment problems.
Prints a markdown table: kernel time (HIP events on the context stream, median of 10 launches), point-evaluations per launch (counted
in-kernel) and the algorithmic rate 64 B x point-evals / time against the 8 TB/s HBM peak.

usage (GPU box): python tools/sweep_tracking.py > gpurun_out/sweep.md
                 python tools/sweep_tracking.py --cluster >> gpurun_out/sweep.md   (kernel time for every cluster size C, the data behind
                                                                                 clusterSize() in csrc/capi.hip)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
import torch
pkg = graft.load_package()
import dmvio_amd.synth as synth


def main():
    w = h = 512
    distinct = 8
    batches = [1, 8, 31, 256, 1024]
    print("| N_ref | pc_n (lvl 0..3) | B | launch mode | kernel ms | point-evals / launch | algorithmic GB/s | frac of 8 TB/s | frames/s (kernel only) |")
    print("|---|---|---|---|---|---|---|---|---|")
    for n_ref, min_grad in [(2000, 8.0), (8000, 8.0), (32000, 4.0), (128000, 0.5), (504 * 504, -1.0)]:
        case = synth.tracking_case(w, h, n_ref=n_ref, n_frames=distinct, xi_jitter=0.3, min_grad=min_grad)
        Bmax = max(b for b in batches if b * n_ref <= 64 * 504 * 504)   # bound the largest launches to a few 10 ms
        ctx = pkg.Context(w, h, n_slots=Bmax + 1)
        stream = torch.cuda.Stream(); ctx.set_stream(stream.cuda_stream)
        trk = pkg.CoarseTrackerHip(ctx); trk.makeK(case["K4"])
        ctx.frame_upload(0, case["ref_img"])
        for i in range(Bmax): ctx.frame_upload(1 + i, case["frames"][i % distinct]["img"])
        trk.setCoarseTrackingRef(0, case["u"], case["v"], case["idepth"], case["hdiF"])
        pcn = [trk.pc_n(l) for l in range(4)]
        ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        for B in [b for b in batches if b <= Bmax]:
            trk.stage(list(range(1, B + 1)), [ident] * B, [(0, 0)] * B)
            ts = []
            for _ in range(12):
                e0.record(stream); trk.launch(); e1.record(stream); e1.synchronize(); ts.append(e0.elapsed_time(e1))
            ms = float(np.median(ts[2:]))
            r = trk.fetch(); ev, pe = trk.last_work()
            gbs = 64.0 * pe / (ms * 1e-3) / 1e9
            Cc, Tt = trk.last_launch()
            mode = "%d x %d threads" % (Cc, Tt)
            print("| %d | %s | %d | %s | %.3f | %d | %.0f | %.3f | %.0f |" % (n_ref, pcn, B, mode, ms, pe, gbs, gbs / 8000.0, B / (ms * 1e-3)))
            sys.stdout.flush()


def cluster_sweep():
    w = h = 512
    distinct = 8
    Cs = [1, 2, 4, 8, 16, 32, 64]
    print("\nKernel time in us per launch for C workgroups per problem (C = 1: one 1024-thread workgroup; C > 1: 256-thread workgroups); `-`: B x C > 1024\n")
    print("| N_ref | pc_n[0] | B | " + " | ".join("C=%d" % c for c in Cs) + " | chosen |")
    print("|---|---|---|" + "---|" * (len(Cs) + 1))
    for n_ref, min_grad in [(2000, 8.0), (8000, 8.0), (32000, 4.0), (504 * 504, -1.0)]:
        case = synth.tracking_case(w, h, n_ref=n_ref, n_frames=distinct, xi_jitter=0.3, min_grad=min_grad)
        Bmax = 128
        ctx = pkg.Context(w, h, n_slots=Bmax + 1)
        stream = torch.cuda.Stream(); ctx.set_stream(stream.cuda_stream)
        ctx.frame_upload(0, case["ref_img"])
        for i in range(Bmax): ctx.frame_upload(1 + i, case["frames"][i % distinct]["img"])
        ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        for B in [1, 4, 8, 16, 31, 64, 128]:
            row = []; chosen = 0; pc0 = 0
            for C in [0] + Cs:   # 0: the library's own choice
                if B * C > 1024: row.append("-"); continue
                trk = pkg.CoarseTrackerHip(ctx); trk.makeK(case["K4"])
                trk.set_launch_shape(lm_cluster=C)   # 0: the library's own choice
                trk.setCoarseTrackingRef(0, case["u"], case["v"], case["idepth"], case["hdiF"])
                pc0 = trk.pc_n(0)
                trk.stage(list(range(1, B + 1)), [ident] * B, [(0, 0)] * B)
                ts = []
                for _ in range(8):
                    e0.record(stream); trk.launch(); e1.record(stream); e1.synchronize(); ts.append(e0.elapsed_time(e1))
                trk.fetch()
                if C: row.append("%.0f" % (1e3 * float(np.median(ts[2:]))))
                else: chosen = "C=%d: %.0f" % (trk.last_launch()[0], 1e3 * float(np.median(ts[2:])))
                del trk
            print("| %d | %d | %d | %s | %s |" % (n_ref, pc0, B, " | ".join(row), chosen)); sys.stdout.flush()


if "--cluster" in sys.argv: cluster_sweep()
else: main()
