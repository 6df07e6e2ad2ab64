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
Ws = [int(x) for x in (sys.argv[2].split(',') if len(sys.argv) > 2 else '16,64'.split(','))]
B = pkg.BundleAdjusterBatch(ctx, max(Ws))
ACC = int(os.environ.get('BA_ACC', '0')) or None
pool = [pkg.BundleAdjusterHip(ctx, accumulators=ACC) for _ in range(max(Ws))]
fn = B.L.dmvio_hip_ba_batch_set_streams; fn.argtypes = [C.c_void_p, C.c_int]; fn.restype = C.c_int
for single in [int(x) for x in (sys.argv[1].split(',') if len(sys.argv) > 1 else '1,2,3,4'.split(','))]:
    fn(B.p, single)
    for W in Ws:
        walls = []; dev = []
        for rep in range(6):
            for h in pool[:W]:
                h.set_case(case, list(range(F)))
            torch.cuda.synchronize()
            t0 = time.perf_counter(); rs = B.optimize(pool[:W], 6); walls.append(time.perf_counter() - t0)
            dev.append(B.last_ms())
        acc = sum(int(r["trace"][1:, 3].sum()) for r in rs)
        w = np.median(walls[1:]); d = np.median([x[0] + x[1] for x in dev[1:]])
        print("streams=%d W=%3d: wall %.3f ms, device %.3f ms -> %.1f it/s   host phases (us) %s" % (single, W, 1e3 * w, d, acc / w, [round(x) for x in B.last_host_us()]), flush=True)
fl = B.L.dmvio_hip_ba_batch_set_linearize_lanes; fl.argtypes = [C.c_void_p, C.c_int]; fl.restype = C.c_int
Rn = len(case["res_point"])
for lanes in (8, 1):
    fl(B.p, lanes)
    for W in Ws:
        lins = []
        for rep in range(4):
            for h in pool[:W]:
                h.set_case(case, list(range(F)))
            torch.cuda.synchronize()
            B.set_profile(True); B.optimize(pool[:W], 6); B.set_profile(False)
            lins.append(B.last_ms()[2])
        us = 1e3 * np.median(lins[1:])
        print("lanes=%d W=%3d: stepped linearisation of all windows %.1f us = %.3f TB/s algorithmic = %.4f of 8 TB/s" % (lanes, W, us, W * Rn * 464 / us / 1e6, W * Rn * 464 / us / 1e6 / 8), flush=True)
fl(B.p, 1)

for lanes in (8, 1):
    fl(B.p, lanes)
    for W in Ws:
        walls = []
        for rep in range(6):
            for h in pool[:W]:
                h.set_case(case, list(range(F)))
            torch.cuda.synchronize()
            t0 = time.perf_counter(); rs = B.optimize(pool[:W], 6); walls.append(time.perf_counter() - t0)
        acc = sum(int(r["trace"][1:, 3].sum()) for r in rs)
        print("lanes=%d W=%3d: wall %.3f ms -> %.1f it/s" % (lanes, W, 1e3 * np.median(walls[1:]), acc / np.median(walls[1:])), flush=True)
fl(B.p, 1)
