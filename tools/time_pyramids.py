"""Kernel times of the pyramid builds for 4096 frames of 512x512 (HIP events on the context's stream): attached fp32 frames (k_build_pyramids, level 0 in place), copied fp32
frames (level 0 written too), 8-bit raw frames (k_build_pyramids_raw<u8>, passthrough geometry, with and without photometric tables)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
P = g.load_package()
w = h = 512; B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
ctx = P.Context(w, h, n_slots=B)
stream = torch.cuda.Stream(device=dev); ctx.set_stream(stream.cuda_stream)
raw = torch.rand((B, h, w), device=dev) * 255
raw8 = (torch.rand((B, h, w), device=dev) * 255).to(torch.uint8)
slots = np.arange(B, dtype=np.int32)
ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
def t(fn, reps=8):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        ev0.record(stream); fn(); ev1.record(stream); ev1.synchronize(); ts.append(ev0.elapsed_time(ev1))
    return float(np.median(ts))
lv = sum(4 * (w >> l) * (h >> l) for l in range(1, ctx.levels))
for variant, vname in ((0, "LDS-tile build"), (1, "register build")):
    P.set_raw_batch_kernel(ctx, variant)
    ta = t(lambda: ctx.frames_attach_device_batch(slots, raw.data_ptr(), w * h * 4))
    print("fp32 attached in place, %s: %.3f ms  %.2f TB/s (read %d + write %d B/frame)" % (vname, ta, B * (4 * w * h + lv) / ta / 1e9, 4 * w * h, lv))
    tc = t(lambda: ctx.frames_from_device_batch(slots, raw.data_ptr(), w * h * 4))
    print("fp32 copied, %s: %.3f ms  %.2f TB/s (read %d + write %d B/frame)" % (vname, tc, B * (8 * w * h + lv) / tc / 1e9, 4 * w * h, 4 * w * h + lv))
und = P.UndistorterHip(ctx, w, h, 8)
G = np.linspace(0, 255, 256).astype(np.float32); vig = np.ones((h, w), np.float32)
und2 = P.UndistorterHip(ctx, w, h, 8, G, vig)
for variant, vname in ((0, "LDS-tile build"), (1, "register build")):
    for tiled in (0, 1):
        P.set_raw_batch_kernel(ctx, variant); P.set_raw_batch_layout(ctx, bool(tiled))
        tr = t(lambda: und.from_raw_device_batch(slots, raw8.data_ptr(), w * h, factor=1.0))
        tg = t(lambda: und2.from_raw_device_batch(slots, raw8.data_ptr(), w * h, factor=1.0))
        print("raw u8, %s, level 0 %s: factor %.3f ms = %.2f TB/s moved (read %d + write %d B/frame), %.2f TB/s written;  G + vignette %.3f ms = %.2f TB/s moved"
              % (vname, "8x4 tiles" if tiled else "row-major", tr, B * (w * h + 4 * w * h + lv) / tr / 1e9, w * h, 4 * w * h + lv, B * (4 * w * h + lv) / tr / 1e9,
                 tg, B * (w * h + 4 * w * h + 4 * w * h + lv) / tg / 1e9))
