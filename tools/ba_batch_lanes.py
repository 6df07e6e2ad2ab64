import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as graft
pkg = graft.load_package()
import torch
import dmvio_amd.synth as synth
case = synth.ba_case(512, 512, n_frames=8, n_points=2000, seed=synth.SEED)
F = 8
ctx = pkg.Context(512, 512, n_slots=F)
for k in range(F):
    ctx.frame_upload(k, case["imgs"][k])
B = pkg.BundleAdjusterBatch(ctx, 64)
pool = [pkg.BundleAdjusterHip(ctx) for _ in range(64)]
fl = B.L.dmvio_hip_ba_batch_set_linearize_lanes; fl.argtypes = [C.c_void_p, C.c_int]; fl.restype = C.c_int
fs = B.L.dmvio_hip_ba_batch_set_streams; fs.argtypes = [C.c_void_p, C.c_int]; fs.restype = C.c_int
for W in (8, 16, 24, 32, 64):
    for lanes in (1, 8):
        for streams in (3, 4):
            fl(B.p, lanes); fs(B.p, streams)
            walls = []
            for rep in range(6):
                for h in pool[:W]:
                    h.set_case(case, list(range(F)))
                torch.cuda.synchronize()
                t0 = time.perf_counter(); rs = B.optimize(pool[:W], 6); walls.append(time.perf_counter() - t0)
            acc = sum(int(r["trace"][1:, 3].sum()) for r in rs)
            w = np.median(walls[1:])
            print("W=%3d lanes=%d streams=%d: wall %.3f ms -> %.1f it/s" % (W, lanes, streams, 1e3 * w, acc / w), flush=True)
