"""Quick probe of dmvio_hip_ba_optimize_batch: wall / device time per W (tools: not part of the bench line)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()
import torch
import dmvio_amd.synth as synth
case = synth.ba_case(512, 512, n_frames=8, n_points=2000, seed=synth.SEED)
F = 8
ctx = pkg.Context(512, 512, n_slots=F)
for k in range(F):
    ctx.frame_upload(k, case["imgs"][k])
Ws = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1,4,16,64".split(","))]
B = pkg.BundleAdjusterBatch(ctx, max(Ws))
pool = [pkg.BundleAdjusterHip(ctx) for _ in range(max(Ws))]
host = pkg.BundleAdjusterHip(ctx); host.set_case(case, list(range(F)))
ts = []
for _ in range(6):
    host.set_case(case, list(range(F)))
    t0 = time.perf_counter(); r = host.optimize(6); ts.append(time.perf_counter() - t0)
print("host-driven loop: optimize(6) %.3f ms, accepted %d" % (1e3 * np.median(ts), int(r["trace"][1:, 3].sum())))
for W in Ws:
    walls = []; dev = []
    for rep in range(6):
        for h in pool[:W]:
            h.set_case(case, list(range(F)))
        torch.cuda.synchronize()
        B.set_profile(rep == 5)
        t0 = time.perf_counter(); rs = B.optimize(pool[:W], 6); walls.append(time.perf_counter() - t0)
        dev.append(B.last_ms())
    acc = sum(int(r["trace"][1:, 3].sum()) for r in rs)
    w = np.median(walls[1:5]); d = np.median([x[0] + x[1] for x in dev[1:5]])
    import ctypes as C
    tk = (C.c_int * 12)(); B.L.dmvio_hip_ba_batch_last_solve_ticks.argtypes = [C.c_void_p, C.c_void_p]; B.L.dmvio_hip_ba_batch_last_solve_ticks(B.p, tk)
    print("   k_ba_solve timeline (us):", [round(t / 100.0, 1) for t in tk])
    print("W=%3d: wall %.3f ms, device %.3f ms, accepted %d -> %.1f it/s, %.1f us per accepted iteration per window-batch; stepped linearisation %.1f us" % (W, 1e3 * w, d, acc, acc / w, 1e6 * w / acc * W, 1e3 * dev[5][2]))
