"""profiles/r05_tracker_floor.md, part 1: what the LM control step costs on the critical path of the full tracking batch.  The bench batch (4096 frames of 512x512, every
frame its own render, 2000 reference points) is tracked normally (k_track_lm, HIP events on the context's stream), its evaluations are recorded, and then run again
WITHOUT the control steps between them (k_track_replay: same points, same taps, same fused reductions per evaluation).  Run through gpurun from the repo root."""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()
import torch
import dmvio_amd.synth as synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
w = h = 512
dev = torch.device("cuda", 0)
case = synth.tracking_case(w, h, n_ref=2000, seed=synth.SEED, n_frames=1, xi_jitter=0.35)
rngx = np.random.RandomState(synth.SEED + 2)
xi0 = case["frames"][0]["xi"]
metas = []
for k in range(B):
    xi = xi0 if k == 0 else xi0 * (1.0 + 0.35 * rngx.standard_normal(6))
    R, t = synth.se3_exp(xi)
    metas.append((R, t))
ctx = pkg.Context(w, h, n_slots=B + 1)
stream = torch.cuda.Stream(device=dev)
ctx.set_stream(stream.cuda_stream)
trk = pkg.CoarseTrackerHip(ctx)
trk.makeK(case["K4"])
ctx.frame_upload(0, case["ref_img"])
trk.setCoarseTrackingRef(0, case["u"], case["v"], case["idepth"], case["hdiF"])
raw = synth.render_batch_torch(case["world"], case["K4"], [m[0] for m in metas], [m[1] for m in metas], w, h, dev).contiguous()
slots = np.arange(1, B + 1, dtype=np.int32)
rng = np.random.RandomState(99)
poses0 = np.zeros((B, 7)); poses0[:, 6] = 1.0
for i in range(B):
    R, t = synth.se3_exp(rng.normal(0, 0.002, 6))
    poses0[i] = synth.pose7(R, t)
affs0 = np.zeros((B, 2))
ctx.frames_attach_device_batch(slots, raw.data_ptr(), w * h * 4)
L = ctx.L
L.dmvio_hip_tracker_debug_record_replay.argtypes = [C.c_void_p, C.c_int]


def timed(n):
    ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(stream); trk.launch(); e1.record(stream); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return ts


trk.stage(slots, poses0, affs0)
for _ in range(20):
    trk.launch()
torch.cuda.synchronize()
t_norm = timed(8)
r = trk.fetch()
ev, pe = trk.last_work()
tk = trk.last_ticks() if hasattr(trk, "last_ticks") else None
L.dmvio_hip_tracker_debug_record_replay(trk.p, 1)
t_rec = timed(3)
trk.fetch()
L.dmvio_hip_tracker_debug_record_replay(trk.p, 2)
t_rep = timed(8)
L.dmvio_hip_tracker_debug_record_replay(trk.p, 0)
t_norm2 = timed(4)
trk.fetch()
a, b, c = np.median(t_norm), np.median(t_rep), np.median(t_norm2)
print("batch %d: %d evaluations, %d point-evaluations per launch (%.1f evaluations per frame), all good: %s" % (B, ev, pe, ev / B, bool(r["good"].all())))
print("k_track_lm           %.3f ms per launch (HIP events, median of 8; again after the experiment: %.3f)" % (a, c))
print("k_track_lm recording %.3f ms" % np.median(t_rec))
print("k_track_replay       %.3f ms per launch: the same evaluations without the LM control steps = %.3f of the full kernel" % (b, b / a))
print("=> everything the control step costs on the critical path of this batch: %.1f %% of the launch" % (100 * (1 - b / a)))
print("algorithmic bytes %.3f GB per launch: %.2f TB/s full kernel, %.2f TB/s evaluations alone" % (64 * pe / 1e9, 64 * pe / (a * 1e-3) / 1e12, 64 * pe / (b * 1e-3) / 1e12))
# ---- part 2: the template stored in the reference's row-major order instead of 8x8 tiles (tools/template_lines.py counts the cache lines per load for both)
L.dmvio_hip_tracker_set_template_order.argtypes = [C.c_void_p, C.c_int]
for order, name in ((1, "row-major"), (0, "8x8 tiles (default)")):
    L.dmvio_hip_tracker_set_template_order(trk.p, order)
    trk.setCoarseTrackingRef(0, case["u"], case["v"], case["idepth"], case["hdiF"])
    trk.stage(slots, poses0, affs0)
    for _ in range(6):
        trk.launch()
    torch.cuda.synchronize()
    tt = timed(8)
    rr = trk.fetch()
    ev2, pe2 = trk.last_work()
    truth = np.stack([synth.pose7(m[0], m[1]) for m in metas])
    err = np.linalg.norm(rr["pose7"][:, :3] - truth[:, :3], axis=1).max()
    print("template order %-20s k_track_lm %.3f ms per launch (median of 8), %d point-evaluations, all good: %s, max pose error %.2e m -> %.2f TB/s algorithmic"
          % (name, np.median(tt), pe2, bool(rr["good"].all()), err, 64 * pe2 / (np.median(tt) * 1e-3) / 1e12))
# ---- part 3: the control step off the evaluation waves' critical path (k_track_lm_pp)
L.dmvio_hip_tracker_set_batch_kernel.argtypes = [C.c_void_p, C.c_int]
trk.stage(slots, poses0, affs0)
trk.launch(); base = trk.fetch()
L.dmvio_hip_tracker_set_batch_kernel(trk.p, 1)
trk.stage(slots, poses0, affs0)
for _ in range(6):
    trk.launch()
torch.cuda.synchronize()
tp = timed(8)
rp = trk.fetch()
dpose = np.abs(rp["pose7"][:, :3] - base["pose7"][:, :3]).max(); dres = np.nanmax(np.abs(rp["lastResiduals"] - base["lastResiduals"]) / np.abs(base["lastResiduals"]))
evp, pep = trk.last_work()
print("k_track_lm_pp        %.3f ms per launch (two problems per workgroup, the solving wave evaluates less; median of 8) = %.3f of k_track_lm; all good: %s, same iteration counts: %s, "
      "poses within %.2e m, level residuals within %.2e relative of k_track_lm's; %d point-evaluations -> %.2f TB/s algorithmic"
      % (np.median(tp), np.median(tp) / a, bool(rp["good"].all()), bool(np.array_equal(rp["iterations"], base["iterations"])), dpose, dres, pep, 64 * pep / (np.median(tp) * 1e-3) / 1e12))
L.dmvio_hip_tracker_set_batch_kernel(trk.p, 0)
if tk:
    print("in-kernel ticks summed over problems (100 MHz): control steps %.3g, evaluations %.3g -> %.1f %% of a workgroup's time in the control step" % (tk[0], tk[1], 100.0 * tk[0] / (tk[0] + tk[1])))
