import os, sys
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import __graft_entry__ as graft
import torch
pkg = graft.load_package()
import dmvio_amd.synth as synth
w = h = 512; distinct = 8
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
        row = []
        for C in [1, 2, 4, 8, 16, 32, 64]:
            if B * C > 1024: row.append("   -  "); continue
            os.environ["DMVIO_HIP_LM_CLUSTER"] = str(C)
            trk = pkg.CoarseTrackerHip(ctx); trk.makeK(case["K4"])
            trk.setCoarseTrackingRef(0, case["u"], case["v"], case["idepth"], case["hdiF"])
            trk.stage(list(range(1, B + 1)), [ident] * B, [(0, 0)] * B)
            ts = []
            for _ in range(8):
                e0.record(stream); trk.launch(); e1.record(stream); e1.synchronize(); ts.append(e0.elapsed_time(e1))
            trk.fetch()
            row.append("%6.0f" % (1e3 * float(np.median(ts[2:]))))
            del trk
        print("N_ref %6d pc0 %6d B %3d | C=1,2,4,..,64 us: %s" % (n_ref, 0, B, " ".join(row))); sys.stdout.flush()
