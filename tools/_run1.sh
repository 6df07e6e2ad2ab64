cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/t1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t1.log
grep -E "^E |passed|failed|rc=|Error" gpurun_out/t1.log | cut -c1-300 | head -20
cat > /tmp/ovl.py <<'PY'
import sys, time, threading, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package()
import dmvio_amd.synth as synth
w = h = 512
bcase = synth.ba_case(w, h, n_frames=8, n_points=2000, seed=17)
tcase = synth.tracking_case(w, h, n_ref=2000, n_frames=4, xi_jitter=0.3)
B = 256
ctx = pkg.Context(w, h, n_slots=9 + B)
for k in range(8): ctx.frame_upload(k, bcase["imgs"][k])
ctx.frame_upload(8, tcase["ref_img"])
for i in range(B): ctx.frame_upload(9 + i, tcase["frames"][i % 4]["img"])
trk = pkg.CoarseTrackerHip(ctx); trk.makeK(tcase["K4"]); trk.setCoarseTrackingRef(8, tcase["u"], tcase["v"], tcase["idepth"], tcase["hdiF"])
ba = pkg.BundleAdjusterHip(ctx); ba.set_case(bcase, list(range(8)))
ident = np.tile(np.array([0, 0, 0, 0, 0, 0, 1.0]), (B, 1)); aff0 = np.zeros((B, 2))
slots = list(range(9, 9 + B))
def T(n):
    for _ in range(n): trk.track_batch(slots, ident.copy(), aff0.copy())
def M(n):
    ba.activate_all(); e = ba.linearize_all(False); ba.apply_res(); lam, lastE = 1e-5, [e, 0.0, 0.0]
    for it in range(n): _, lam, lastE = ba.gn_iteration(it % 6, lam, lastE)
T(3); M(20)
t0 = time.perf_counter(); T(40); tT = time.perf_counter() - t0
t0 = time.perf_counter(); M(200); tM = time.perf_counter() - t0
a = threading.Thread(target=T, args=(40,)); b = threading.Thread(target=M, args=(200,))
t0 = time.perf_counter(); a.start(); b.start(); a.join(); b.join(); tP = time.perf_counter() - t0
print("tracking alone %.1f ms, BA alone %.1f ms, sequential %.1f ms, overlapped on two threads %.1f ms" % (1e3 * tT, 1e3 * tM, 1e3 * (tT + tM), 1e3 * tP))
PY
timeout 300 python /tmp/ovl.py 2>&1 | tail -2
