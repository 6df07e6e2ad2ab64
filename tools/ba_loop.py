#!/usr/bin/env python
"""GN iterations of the bench's BA window in a loop (for rocprofv3 --kernel-trace): python tools/ba_loop.py [iters]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()
import dmvio_amd.synth as synth
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 300
case = synth.ba_case(512, 512, n_frames=8, n_points=2000, seed=synth.SEED)
ctx = pkg.Context(512, 512, n_slots=8)
for k in range(8):
    ctx.frame_upload(k, case["imgs"][k])
ba = pkg.BundleAdjusterHip(ctx)
ba.set_case(case, list(range(8)))
ba.activate_all(); e = ba.linearize_all(False); ba.apply_res()
lam, lastE = 1e-5, [e, 0.0, 0.0]
for it in range(12):
    _, lam, lastE = ba.gn_iteration(it % 6, lam, lastE)
t0 = time.perf_counter(); acc = 0
for it in range(n_it):
    a, lam, lastE = ba.gn_iteration(it % 6, lam, lastE); acc += int(a)
dt = time.perf_counter() - t0
print("%.1f GN-iterations/s, %.1f us per iteration, %d of %d accepted" % (n_it / dt, 1e6 * dt / n_it, acc, n_it))
import ctypes as C
tk = (C.c_int * 4)()
ba.L.dmvio_hip_ba_last_decide_ticks.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
ba.L.dmvio_hip_ba_last_decide_ticks(ba.p, tk)
print("last decision pass, us since its workgroup started: begin %.2f, energy %.2f, keys %.2f, done %.2f" % tuple(t / 100.0 for t in tk))
# FullSystem::optimize as the reference calls it once per keyframe: 6 GN iterations from the perturbed window + the final fix-linearisation
ts = []; accs = 0
for rep in range(30):
    ba.set_case(case, list(range(8)))
    t0 = time.perf_counter(); r = ba.optimize(6); ts.append(time.perf_counter() - t0); accs += int(np.sum(r["trace"][1:, 3]))
ts = np.array(ts[5:])
print("optimize(6) on the fresh window: %.1f us per call (median), %.1f us per GN iteration if the final linearisation is charged to them; %d of %d steps accepted"
      % (1e6 * np.median(ts), 1e6 * np.median(ts) / 6, accs, 30 * 6))
