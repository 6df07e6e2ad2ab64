#!/usr/bin/env python
"""k_track_lm launch shapes side by side (measurement only): python tools/track_shapes.py [B]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
import torch
pkg = graft.load_package()
import dmvio_amd.synth as synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
w = h = 512
distinct = 16
case = synth.tracking_case(w, h, n_ref=2000, n_frames=distinct, xi_jitter=0.3, min_grad=8.0)
ctx = pkg.Context(w, h, n_slots=B + 1)
stream = torch.cuda.Stream(); ctx.set_stream(stream.cuda_stream)
trk = pkg.CoarseTrackerHip(ctx); trk.makeK(case["K4"])
ctx.frame_upload(0, case["ref_img"])
for i in range(B): ctx.frame_upload(1 + i, case["frames"][i % distinct]["img"])
ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
import ctypes as C
L = trk.L
L.dmvio_hip_tracker_set_template_order.argtypes = [C.c_void_p, C.c_int]; L.dmvio_hip_tracker_set_batch_kernel.argtypes = [C.c_void_p, C.c_int]
for name, shape, bk, order in (("256 x 4 (default)", (0, 0, 0, 0), 0, 0), ("256 x 2", (0, 256, 2, 0), 0, 0), ("512 x 4", (0, 512, 4, 0), 0, 0), ("512 x 2", (0, 512, 2, 0), 0, 0),
                               ("two problems per workgroup", (0, 0, 0, 0), 1, 0), ("row-major template", (0, 0, 0, 0), 0, 1)):
    trk.set_launch_shape(*shape); L.dmvio_hip_tracker_set_batch_kernel(trk.p, bk); L.dmvio_hip_tracker_set_template_order(trk.p, order)
    trk.setCoarseTrackingRef(0, case["u"], case["v"], case["idepth"], case["hdiF"])
    trk.stage(list(range(1, B + 1)), [ident] * B, [(0, 0)] * B)
    ts = []
    for _ in range(10):
        e0.record(stream); trk.launch(); e1.record(stream); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts[2:]))
    r = trk.fetch(); ev, pe = trk.last_work()
    print("%-28s B=%d: %.3f ms, %.0f GB/s algorithmic = %.3f of 8 TB/s, launch %s" % (name, B, ms, 64.0 * pe / (ms * 1e-3) / 1e9, 64.0 * pe / (ms * 1e-3) / 8e12, trk.last_launch()), flush=True)
