"""profiles/r05_tracker_floor.md, part 2 (CPU only): cache lines per 64-lane tap load of the coarse tracker as a function of the ORDER of the template points.
A tap load instruction reads, for each of its 64 lanes (= 64 consecutive template points), four floats of ONE image row around the projected point (csrc/interp.hpp: 4 row
loads per tap); a 128-byte line holds 32 pixels of a row.  Lines per load = distinct (row, 32-pixel segment[s]) over the wave's points.  For the bench template (2000
reference points dilated by makeCoarseDepthL0) at an identity warp this counts the lines for: the library's stored order (8x8 tiles, Z order inside 16x16 blocks, blocks
row-major: csrc/ref_kernels.hpp), the reference's row-major order, Morton order, 32x8 'line-shaped' tiles — and the lower bound (every distinct line touched by the template
must be loaded by at least one wave: distinct lines / number of waves)."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()
import dmvio_amd.synth as synth
O = graft.load_oracle()

w = h = 512
case = synth.tracking_case(w, h, n_ref=2000, seed=synth.SEED, n_frames=1, xi_jitter=0.35)
dI, _ = O.make_images(case["ref_img"], w, h)
T = O.Tracker(w, h)
T.make_k(case["K4"])
T.set_ref(dI, case["u"], case["v"], case["idepth"], case["hdiF"])


def lines_per_load(x, y, wl):
    """mean distinct 128-byte lines per 64-lane row load, over the four row loads of a tap (rows y-1 .. y+2, pixels x-1 .. x+2)"""
    n = len(x)
    tot = 0.0; cnt = 0
    for s in range(0, n, 64):
        xs, ys = x[s:s + 64], y[s:s + 64]
        for dy in (-1, 0, 1, 2):
            a0 = ((ys + dy) * wl + xs - 1) * 4
            l0, l1 = a0 // 128, (a0 + 15) // 128
            tot += len(np.unique(np.concatenate([l0, l1]))); cnt += 1
    return tot / cnt


def morton(x, y):
    def part(v):
        v = v.astype(np.uint64)
        v = (v | (v << 8)) & 0x00FF00FF; v = (v | (v << 4)) & 0x0F0F0F0F; v = (v | (v << 2)) & 0x33333333; v = (v | (v << 1)) & 0x55555555
        return v
    return part(x) | (part(y) << np.uint64(1))


print("| level | template points | waves | distinct lines (dy = 0) | lower bound per load | stored order (8x8 tiles, Z in 16x16) | row-major (reference) | Morton | 32x8 tiles | 32x4 tiles |")
print("|---|---|---|---|---|---|---|---|---|---|")
for lvl in range(4):
    u, v, _, _ = T.get_pc(lvl)
    wl = w >> lvl
    x = np.round(u).astype(np.int64); y = np.round(v).astype(np.int64)
    n = len(x)
    waves = (n + 63) // 64
    a0 = (y * wl + x - 1) * 4
    distinct = len(np.unique(np.concatenate([a0 // 128, (a0 + 15) // 128])))
    lb = distinct / waves
    orders = {}
    orders["row"] = np.lexsort((x, y))
    key_blk = (y // 16) * ((wl + 15) // 16) + (x // 16)
    key_tile = ((y % 16) // 8) * 2 + ((x % 16) // 8)
    orders["stored"] = np.lexsort((x % 8, y % 8, key_tile, key_blk))
    orders["morton"] = np.argsort(morton(x, y), kind="stable")
    orders["t32x8"] = np.lexsort((x % 32, y % 8, x // 32, y // 8))
    orders["t32x4"] = np.lexsort((x % 32, y % 4, x // 32, y // 4))
    r = {k: lines_per_load(x[o], y[o], wl) for k, o in orders.items()}
    print("| %d | %d | %d | %d | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f |" % (lvl, n, waves, distinct, lb, r["stored"], r["row"], r["morton"], r["t32x8"], r["t32x4"]))
