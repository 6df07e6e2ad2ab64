import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
import torch
pkg = graft.load_package()
import dmvio_amd.synth as synth
w = h = 512; B = 1024; distinct = 8
dev = torch.device("cuda", 0)
case = synth.tracking_case(w, h, n_ref=2000, seed=synth.SEED, n_frames=distinct, xi_jitter=0.35)
ctx = pkg.Context(w, h, n_slots=B + 1)
s1 = torch.cuda.Stream(device=dev); s2 = torch.cuda.Stream(device=dev)
ctx.set_stream(s1.cuda_stream)
trk = pkg.CoarseTrackerHip(ctx); trk.makeK(case["K4"])
ctx.frame_upload(0, case["ref_img"])
trk.setCoarseTrackingRef(0, case["u"], case["v"], case["idepth"], case["hdiF"])
raw = torch.empty((B, h, w), dtype=torch.float32, device=dev)
raw_distinct = torch.from_numpy(np.stack([f["img"] for f in case["frames"]])).to(dev)
raw.copy_(raw_distinct[torch.arange(B, device=dev) % distinct]); torch.cuda.synchronize()
slots = np.arange(1, B + 1, dtype=np.int32)
poses0 = np.zeros((B, 7)); poses0[:, 6] = 1.0; affs0 = np.zeros((B, 2))
raw_ptr = raw.data_ptr(); fb = w * h * 4

def run(two_streams, K=20):
    def step(have_prev):
        if two_streams: ctx.set_stream(s2.cuda_stream)
        ctx.frames_from_device_batch(slots, raw_ptr, fb)
        if two_streams: ctx.set_stream(s1.cuda_stream)
        if have_prev: trk.fetch_begin()
        trk.stage(slots, poses0, affs0); trk.launch()
        return trk.fetch() if have_prev else None
    for _ in range(3): step(False); trk.fetch()
    torch.cuda.synchronize(); step(False); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K): r = step(True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("two_streams=%d: %.4f ms/step  %.0f frames/s  good=%d" % (two_streams, 1e3 * dt / K, B * K / dt, int(r["good"].sum())))

for rep in range(2):
    run(0); run(1)
