"""What a pure streaming kernel reaches on this chip, by direction: fill (write only), copy (1 read + 1 write per byte moved), sum (read only) over 4 GiB with torch's own kernels —
the ceiling k_build_pyramids_raw (1 B read : 5.3 B written) and k_build_pyramids (4 B read : 1.3-5.3 B written) have to be judged against."""
import torch, time
dev = torch.device("cuda", 0)
n = 1 << 30   # floats: 4 GiB
a = torch.empty(n, dtype=torch.float32, device=dev); b = torch.empty(n, dtype=torch.float32, device=dev)
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
tf = t(lambda: a.fill_(1.0)); tc = t(lambda: b.copy_(a)); ts = t(lambda: a.sum())
GB = 4 * n / 1e9
print("fill (write only):   %.2f TB/s written" % (GB / tf / 1e3))
print("copy (read + write): %.2f TB/s moved = %.2f read + %.2f written" % (2 * GB / tc / 1e3, GB / tc / 1e3, GB / tc / 1e3))
print("sum  (read only):    %.2f TB/s read" % (GB / ts / 1e3))
u8 = torch.empty(n, dtype=torch.uint8, device=dev)
tu = t(lambda: torch.mul(u8, 1.0, out=a))      # 1 B read : 4 B written, the ratio of the raw-image build's level 0
print("u8 -> f32 (1 B read : 4 B written): %.2f TB/s moved, %.2f TB/s written" % (5 * n / tu / 1e12, 4 * n / tu / 1e12))
