"""Timeline of the LM control step: needs a library built with -DDMV_LM_TICKS (make -C dm-vio_amd/csrc HIPFLAGS="... -DDMV_LM_TICKS"); see profiles/r02_lm_control_step.md."""
import sys, ctypes as C
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import __graft_entry__ as g
P = g.load_package()
import dmvio_amd.synth as synth
case = synth.tracking_case(512, 512, n_ref=2000, seed=synth.SEED, n_frames=4, xi_jitter=0.35)
for B in (1, 64, 1024):
    ctx = P.Context(512, 512, n_slots=B + 1)
    trk = P.CoarseTrackerHip(ctx); trk.makeK(case["K4"])
    ctx.frame_upload(0, case["ref_img"])
    for k in range(B): ctx.frame_upload(1 + k, case["frames"][k % 4]["img"])
    trk.setCoarseTrackingRef(0, case["u"], case["v"], case["idepth"], case["hdiF"])
    ident = np.tile(np.array([0, 0, 0, 0, 0, 0, 1.0]), (B, 1)); aff = np.zeros((B, 2))
    slots = np.arange(1, B + 1, dtype=np.int32)
    trk.track_batch(slots, ident, aff)
    out = (C.c_double * 8)()
    ctx.L.dmvio_hip_debug_lm_ticks(out, 1)
    for _ in range(3): r = trk.track_batch(slots, ident, aff)
    ctx.L.dmvio_hip_debug_lm_ticks(out, 1)
    o = np.array(out[:])
    print("B=%d: per control step (us): whole %.2f | lane-0 part 1 %.2f | solve %.2f (per solve) | lane-0 part 2 (exp, mul, makeEvalP) %.2f ; steps %d solves %d" %
          (B, o[3] / o[5] / 100, o[1] / o[5] / 100, o[0] / o[4] / 100, o[2] / o[5] / 100, o[5], o[4]))
