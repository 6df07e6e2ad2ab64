"""Deterministic synthetic inputs for the photometric-alignment hot path (tests + bench.py).

Plane-world renderer: two textured planes viewed by a pinhole camera, rendered
analytically (ray/plane intersection + closed-form texture), so photo-consistency and inverse
depth ground truth are exact.  Intrinsics default to the TUM-VI rectified pinhole the reference's
configs/tumvi_calib/camera02.txt:3-4 + Undistort.cpp:897-900 give at 512x512:
fx = fy = 0.2*512, cx = cy = 0.499*512 - 0.5.

This is input generation only — no part of the product path.
"""
import numpy as np

SEED = 20250204


def default_intrinsics(w=512, h=512):
    return np.array([0.2 * w, 0.2 * h, 0.499 * w - 0.5, 0.499 * h - 0.5], dtype=np.float64)


def quat_to_R(q):  # q = (qx,qy,qz,qw)
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=np.float64)


def R_to_quat(R):
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    x = np.sqrt(max(0.0, 1 + R[0, 0] - R[1, 1] - R[2, 2])) / 2
    y = np.sqrt(max(0.0, 1 - R[0, 0] + R[1, 1] - R[2, 2])) / 2
    z = np.sqrt(max(0.0, 1 - R[0, 0] - R[1, 1] + R[2, 2])) / 2
    x = np.copysign(x, R[2, 1] - R[1, 2]); y = np.copysign(y, R[0, 2] - R[2, 0]); z = np.copysign(z, R[1, 0] - R[0, 1])
    q = np.array([x, y, z, w]); return q / np.linalg.norm(q)


def se3_exp(xi):
    """xi = [upsilon(3), omega(3)] -> (R, t); float64 closed form (independent of the oracle's code)."""
    xi = np.asarray(xi, dtype=np.float64)
    u, om = xi[:3], xi[3:]
    th = np.linalg.norm(om)
    Om = np.array([[0, -om[2], om[1]], [om[2], 0, -om[0]], [-om[1], om[0], 0]])
    if th < 1e-10:
        R = np.eye(3) + Om + 0.5 * Om @ Om
        V = np.eye(3) + 0.5 * Om + Om @ Om / 6
    else:
        R = np.eye(3) + np.sin(th) / th * Om + (1 - np.cos(th)) / th ** 2 * Om @ Om
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * Om + (th - np.sin(th)) / th ** 3 * Om @ Om
    return R, V @ u


def pose7(R, t):
    """(R,t) -> [tx,ty,tz,qx,qy,qz,qw] (the reference's result.txt convention, FullSystem.cpp:256-298)."""
    q = R_to_quat(R)
    return np.array([t[0], t[1], t[2], q[0], q[1], q[2], q[3]], dtype=np.float64)


def pose7_to_Rt(p):
    return quat_to_R(p[3:7]), np.asarray(p[:3], dtype=np.float64)


class PlaneWorld:
    """Two textured planes: a slanted wall  n1.X = d1  and a floor  n2.X = d2  (camera y points down)."""

    def __init__(self, seed=SEED, n_waves=24, depth=4.0, fmax=22.0):
        rng = np.random.RandomState(seed)
        n1 = np.array([-0.25, 0.05, 1.0]); n1 /= np.linalg.norm(n1)
        n2 = np.array([0.0, 1.0, 0.08]); n2 /= np.linalg.norm(n2)
        self.planes = [(n1, depth * n1[2]), (n2, 1.6)]
        # in-plane bases
        self.bases = []
        for n, _ in self.planes:
            a = np.cross(n, [0.0, 0.0, 1.0]) if abs(n[2]) < 0.9 else np.cross(n, [0.0, 1.0, 0.0])
            a /= np.linalg.norm(a); b = np.cross(n, a)
            self.bases.append((a, b))
        self.waves = []
        for _ in self.planes:
            f = np.exp(rng.uniform(np.log(min(0.8, fmax / 4)), np.log(fmax), n_waves))      # rad / m
            ang = rng.uniform(0, np.pi, n_waves)
            amp = rng.uniform(4.0, 20.0, n_waves) / np.sqrt(np.maximum(f, 1.0)) * 2.2
            ph = rng.uniform(0, 2 * np.pi, n_waves)
            self.waves.append((f * np.cos(ang), f * np.sin(ang), amp, ph))

    def render(self, K4, R_cw, t_cw, w=512, h=512, aff=(0.0, 0.0)):
        """Render irradiance seen by camera with world->cam (R_cw, t_cw). Returns (img f32 [h,w], idepth f64 [h,w])."""
        fx, fy, cx, cy = K4
        xs, ys = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
        d_c = np.stack([(xs - cx) / fx, (ys - cy) / fy, np.ones_like(xs)], -1)      # cam rays, z=1
        c_w = -R_cw.T @ t_cw
        d_w = d_c @ R_cw                                                             # R^T d  (row-vector form)
        best_s = np.full((h, w), np.inf)
        img = np.full((h, w), 128.0)
        for (n, d0), (a, b), (fa, fb, amp, ph) in zip(self.planes, self.bases, self.waves):
            denom = d_w @ n
            with np.errstate(divide='ignore', invalid='ignore'):
                s = (d0 - n @ c_w) / denom
            ok = (s > 0.05) & (s < best_s) & np.isfinite(s)
            X = c_w + s[..., None] * d_w
            pa = X @ a; pb = X @ b
            val = np.full((h, w), 128.0)
            for k in range(len(fa)):
                val += amp[k] * np.sin(fa[k] * pa + fb[k] * pb + ph[k])
            img = np.where(ok, val, img)
            best_s = np.where(ok, s, best_s)
        img = np.exp(aff[0]) * img + aff[1]
        img = np.clip(img, 1.0, 254.0)
        idepth = np.where(np.isfinite(best_s), 1.0 / best_s, 0.0)
        return img.astype(np.float32), idepth


def select_points(img, n, rng, border=4, min_grad=8.0):
    """n seeded pixel positions with |grad I| > min_grad, uniform over border<=x<w-border."""
    h, w = img.shape
    gx = np.zeros_like(img); gy = np.zeros_like(img)
    gx[:, 1:-1] = 0.5 * (img[:, 2:] - img[:, :-2]); gy[1:-1, :] = 0.5 * (img[2:, :] - img[:-2, :])
    g = np.sqrt(gx * gx + gy * gy)
    ok = g > min_grad
    ok[:border, :] = False; ok[-border:, :] = False; ok[:, :border] = False; ok[:, -border:] = False
    ys, xs = np.nonzero(ok)
    if len(xs) < n:
        raise RuntimeError("not enough textured pixels: %d < %d" % (len(xs), n))
    sel = rng.choice(len(xs), n, replace=False)
    sel.sort()
    return xs[sel].astype(np.float32), ys[sel].astype(np.float32)


def tracking_case(w=512, h=512, n_ref=2000, seed=SEED, xi_true=(0.03, -0.02, 0.04, 0.01, -0.015, 0.008),
                  aff_new=(0.0, 0.0), n_frames=1, xi_jitter=0.0, fmax=22.0, min_grad=8.0):
    """Config-2 style case: reference KF at identity with n_ref points of exact idepth; new frame(s) at xi_true."""
    world = PlaneWorld(seed, fmax=fmax)
    K4 = default_intrinsics(w, h)
    rng = np.random.RandomState(seed + 1)
    ref_img, ref_id = world.render(K4, np.eye(3), np.zeros(3), w, h)
    u, v = select_points(ref_img, n_ref, rng, min_grad=min_grad)
    idepth = ref_id[v.astype(int), u.astype(int)].astype(np.float32)
    frames = []
    for k in range(n_frames):
        xi = np.asarray(xi_true, dtype=np.float64)
        if k > 0 and xi_jitter > 0:
            xi = xi * (1.0 + xi_jitter * rng.standard_normal(6))
        R, t = se3_exp(xi)
        img, _ = world.render(K4, R, t, w, h, aff=aff_new)
        frames.append(dict(img=img, R=R, t=t, pose7=pose7(R, t), xi=xi))
    return dict(K4=K4, w=w, h=h, ref_img=ref_img, u=u, v=v, idepth=idepth,
                hdiF=np.full(n_ref, 1e-4, dtype=np.float32), frames=frames, world=world)
