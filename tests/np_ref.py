"""Independent float64 NumPy restatement of the tracker's per-point math (vectorised, no shared code with
oracle/*.cpp or the HIP kernels).  Used to validate the oracle BY CONSTRUCTION, since the reference holds no
golden vectors for this path: finite-difference Jacobians, H = sum w J J^T, identity cases.

Follows CoarseTracker::calcRes / calcGSSSE (src/dso/FullSystem/CoarseTracker.cpp:361-517, 299-356) but in
float64 and with plain sums, so agreement with the fp32 oracle is expected to ~1e-5 relative, not bitwise.
"""
import numpy as np

HUBER = 9.0


def quat_to_R(p7):
    x, y, z, w = p7[3:7]
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def hat(o):
    return np.array([[0, -o[2], o[1]], [o[2], 0, -o[0]], [-o[1], o[0], 0.0]])


def se3_exp_matrix(xi):
    """4x4 matrix exponential of hat(xi), xi = [upsilon; omega], via its power series (independent check)."""
    A = np.zeros((4, 4)); A[:3, :3] = hat(xi[3:]); A[:3, 3] = xi[:3]
    out = np.eye(4); term = np.eye(4)
    for k in range(1, 30):
        term = term @ A / k
        out = out + term
    return out


def level_K(K4, lvl):
    fx, fy, cx, cy = [float(v) for v in K4]
    s = 2.0 ** lvl
    return fx / s, fy / s, (cx + 0.5) / s - 0.5, (cy + 0.5) / s - 0.5


def bilinear3(img, x, y):
    """img [h,w,3]; x,y float arrays (inside the image) -> [n,3]"""
    ix = np.floor(x).astype(int); iy = np.floor(y).astype(int)
    dx = (x - ix)[:, None]; dy = (y - iy)[:, None]
    return (dx * dy * img[iy + 1, ix + 1] + (dy - dx * dy) * img[iy + 1, ix] + (dx - dx * dy) * img[iy, ix + 1]
            + (1 - dx - dy + dx * dy) * img[iy, ix])


def calc_res_gs(K4, lvl, pc, img_new, pose7, aff=(0.0, 0.0), cutoff=20.0, ref_aff=(0.0, 0.0), exposures=(1.0, 1.0)):
    """pc = (u, v, idepth, color) arrays of the level; img_new [h_l, w_l, 3] float.
    Returns dict(E, n, nsat, H[8,8], b[8], r, w, J[n',8]) with the reference's scaling (SCALE_A=10, SCALE_B=1000)."""
    fx, fy, cx, cy = level_K(K4, lvl)
    u0, v0, idp, col = [np.asarray(a, dtype=np.float64) for a in pc]
    img = np.asarray(img_new, dtype=np.float64)
    h, w = img.shape[:2]
    R = quat_to_R(pose7); t = np.asarray(pose7[:3], dtype=np.float64)
    Ki = np.array([[1 / fx, 0, -cx / fx], [0, 1 / fy, -cy / fy], [0, 0, 1.0]])
    pt = (R @ Ki @ np.stack([u0, v0, np.ones_like(u0)])).T + t[None, :] * idp[:, None]
    u = pt[:, 0] / pt[:, 2]; v = pt[:, 1] / pt[:, 2]
    Ku = fx * u + cx; Kv = fy * v + cy
    nid = idp / pt[:, 2]
    ok = (Ku > 2) & (Kv > 2) & (Ku < w - 3) & (Kv < h - 3) & (nid > 0)
    a_rel = np.exp(aff[0] - ref_aff[0]) * exposures[1] / exposures[0]
    b_rel = aff[1] - a_rel * ref_aff[1]
    hit = bilinear3(img, Ku[ok], Kv[ok])
    res = hit[:, 0] - (a_rel * col[ok] + b_rel)
    ar = np.abs(res)
    hw = np.where(ar < HUBER, 1.0, HUBER / np.maximum(ar, 1e-30))
    sat = ar > cutoff
    maxE = 2 * HUBER * cutoff - HUBER * HUBER
    E = np.sum(np.where(sat, maxE, hw * res * res * (2 - hw)))
    inl = ~sat
    uu, vv, idd = u[ok][inl], v[ok][inl], nid[ok][inl]
    dxx = hit[inl, 1] * fx; dyy = hit[inl, 2] * fy
    J = np.stack([idd * dxx, idd * dyy, -idd * (uu * dxx + vv * dyy), -(uu * vv * dxx + dyy * (1 + vv * vv)),
                  uu * vv * dyy + dxx * (1 + uu * uu), uu * dyy - vv * dxx, a_rel * (ref_aff[1] - col[ok][inl]), -np.ones_like(uu)], 1)
    r = res[inl]; wgt = hw[inl]
    n4 = (len(r) + 3) // 4 * 4
    H = (J * wgt[:, None]).T @ J / n4
    b = (J * wgt[:, None]).T @ r / n4
    sc = np.array([1, 1, 1, 1, 1, 1, 10.0, 1000.0])
    return dict(E=E, n=int(ok.sum()), nsat=int(sat.sum()), H=H * sc[:, None] * sc[None, :], b=b * sc, r=r, w=wgt, J=J, ok=ok, sat=sat)


def make_images(color):
    """FrameHessian::makeImages in float32 NumPy (same operation order, vectorised): list of [h,w,3]."""
    out = []
    I = np.asarray(color, dtype=np.float32)
    while True:
        h, w = I.shape
        d = np.zeros((h, w, 3), dtype=np.float32)
        d[..., 0] = I
        flat = I.reshape(-1)
        idx = np.arange(w, w * (h - 1))
        dx = np.float32(0.5) * (flat[idx + 1] - flat[idx - 1])
        dy = np.float32(0.5) * (flat[idx + w] - flat[idx - w])
        d.reshape(-1, 3)[idx, 1] = dx
        d.reshape(-1, 3)[idx, 2] = dy
        out.append(d)
        if not (w % 2 == 0 and h % 2 == 0 and w * h > 5000 and len(out) < 6):
            break
        I = (np.float32(0.25) * (((I[0::2, 0::2] + I[0::2, 1::2]) + I[1::2, 0::2]) + I[1::2, 1::2])).astype(np.float32)
    return out
