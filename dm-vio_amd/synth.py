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
    # float-exact values: in the reference every calibration passes through float globals (globalCalib.cpp:79-82) before it is widened
    # to CalibHessian::value_scaled, so intrinsics that are not representable in fp32 never reach the path
    return np.array([0.2 * w, 0.2 * h, 0.499 * w - 0.5, 0.499 * h - 0.5], dtype=np.float32).astype(np.float64)


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


def render_batch_torch(world, K4, Rs, ts, w, h, device, chunk=16, aff=(0.0, 0.0)):
    """PlaneWorld.render for a batch of cameras, evaluated with torch on `device` (fp64, same formula; plumbing for benchmarks that need thousands
    of distinct frames — the numpy renderer takes 0.25 s per frame).  Returns a float32 tensor [n, h, w] on the device."""
    import torch
    fx, fy, cx, cy = [float(x) for x in K4]
    f64 = dict(dtype=torch.float64, device=device)
    ys, xs = torch.meshgrid(torch.arange(h, **f64), torch.arange(w, **f64), indexing="ij")
    d_c = torch.stack([(xs - cx) / fx, (ys - cy) / fy, torch.ones_like(xs)], -1)
    n = len(Rs)
    out = torch.empty((n, h, w), dtype=torch.float32, device=device)
    for c0 in range(0, n, chunk):
        R = torch.as_tensor(np.stack(Rs[c0:c0 + chunk]), **f64)
        t = torch.as_tensor(np.stack(ts[c0:c0 + chunk]), **f64)
        m = R.shape[0]
        c_w = -torch.einsum("mji,mj->mi", R, t)
        d_w = torch.stack([sum(d_c[None, :, :, j] * R[:, j, i, None, None] for j in range(3)) for i in range(3)], -1)
        best_s = torch.full((m, h, w), float("inf"), **f64)
        img = torch.full((m, h, w), 128.0, **f64)
        for (nrm, d0), (a, b), (fa, fb, amp, ph) in zip(world.planes, world.bases, world.waves):
            nrm_t = torch.as_tensor(nrm, **f64); a_t = torch.as_tensor(a, **f64); b_t = torch.as_tensor(b, **f64)
            dot3 = lambda V, q: V[..., 0] * q[0] + V[..., 1] * q[1] + V[..., 2] * q[2]      # elementwise (a matmul with a 3-vector lands on a slow fp64 gemv)
            denom = dot3(d_w, nrm_t)
            s = ((d0 - dot3(c_w, nrm_t))[:, None, None]) / denom
            ok = (s > 0.05) & (s < best_s) & torch.isfinite(s)
            X = c_w[:, None, None, :] + s[..., None] * d_w
            pa = dot3(X, a_t); pb = dot3(X, b_t)
            val = torch.full((m, h, w), 128.0, **f64)
            for k in range(len(fa)):
                val += float(amp[k]) * torch.sin(float(fa[k]) * pa + float(fb[k]) * pb + float(ph[k]))
            img = torch.where(ok, val, img)
            best_s = torch.where(ok, s, best_s)
        img = float(np.exp(aff[0])) * img + float(aff[1])
        out[c0:c0 + m] = img.clamp(1.0, 254.0).to(torch.float32)
    return out


PATTERN8 = np.array([[0, -2], [-1, -1], [1, -1], [-2, 0], [0, 0], [2, 0], [-1, 1], [0, 2]], dtype=np.int32)  # settings.cpp:296, pattern 8


def ba_case(w=512, h=512, n_frames=8, n_points=2000, seed=SEED, idepth_noise=0.05, trans_noise=0.005, rot_noise=0.0035,
            step_t=0.1, step_r=np.deg2rad(2.0), hosts_share=(400, 350, 300, 300, 250, 250, 150, 0)):
    """Sliding-window BA case: F keyframes on a smooth trajectory, N points hosted in the older
    keyframes, each observed in every other frame where its centre projects in bounds.  Returns ground truth, the
    perturbed initial state (poses as evalPT, idepths) and the residual graph."""
    world = PlaneWorld(seed)
    K4 = default_intrinsics(w, h)
    rng = np.random.RandomState(seed + 7)
    fx, fy, cx, cy = K4
    # smooth trajectory: sideways + forward drift, gentle yaw
    Rs, ts, imgs, ids = [], [], [], []
    for k in range(n_frames):
        xi = np.array([step_t * k, 0.02 * k * step_t / 0.1, 0.3 * step_t * k, 0.2 * step_r * k, -step_r * k, 0.1 * step_r * k])
        R, t = se3_exp(xi)
        Rs.append(R); ts.append(t)
        img, idm = world.render(K4, R, t, w, h)
        imgs.append(img); ids.append(idm)
    share = np.array(hosts_share[:n_frames], dtype=np.float64)
    if share.sum() <= 0:
        share = np.ones(n_frames); share[-1] = 0
    counts = np.floor(share / share.sum() * n_points).astype(int)
    counts[0] += n_points - counts.sum()
    pts = []  # (host, u, v, idepth_true)
    for hf in range(n_frames):
        if counts[hf] == 0:
            continue
        u, v = select_points(imgs[hf], counts[hf], rng, border=8)
        for a, b in zip(u, v):
            pts.append((hf, float(a), float(b), float(ids[hf][int(b), int(a)])))
    # colors / weights exactly as ImmaturePoint's constructor samples them (ImmaturePoint.cpp:40-56): pattern pixels of the
    # host image at integer positions; weights = sqrt(c / (c + |grad|^2)), c = setting_outlierTHSumComponent = 50^2
    grads = []
    for img in imgs:
        gx = np.zeros_like(img); gy = np.zeros_like(img)
        flat = img.reshape(-1); idx = np.arange(w, w * (h - 1))
        gx.reshape(-1)[idx] = np.float32(0.5) * (flat[idx + 1] - flat[idx - 1])
        gy.reshape(-1)[idx] = np.float32(0.5) * (flat[idx + w] - flat[idx - w])
        grads.append((gx, gy))
    N = len(pts)
    host = np.array([p[0] for p in pts], dtype=np.int32)
    u = np.array([p[1] for p in pts], dtype=np.float32); v = np.array([p[2] for p in pts], dtype=np.float32)
    idepth_true = np.array([p[3] for p in pts], dtype=np.float32)
    color = np.zeros((N, 8), dtype=np.float32); weights = np.zeros((N, 8), dtype=np.float32)
    for i in range(N):
        hf = host[i]
        px = u[i].astype(int) + PATTERN8[:, 0]; py = v[i].astype(int) + PATTERN8[:, 1]
        color[i] = imgs[hf][py, px]
        g2 = grads[hf][0][py, px] ** 2 + grads[hf][1][py, px] ** 2
        weights[i] = np.sqrt(np.float32(2500.0) / (np.float32(2500.0) + g2))
    # residual graph: point i observed in frame t != host when its centre projects well inside the image
    res_point, res_target = [], []
    for i in range(N):
        hf = host[i]
        X_h = np.array([(u[i] - cx) / fx, (v[i] - cy) / fy, 1.0]) / idepth_true[i]
        X_w = Rs[hf].T @ (X_h - ts[hf])
        for t in range(n_frames):
            if t == hf:
                continue
            X_t = Rs[t] @ X_w + ts[t]
            if X_t[2] <= 0.1:
                continue
            pu = fx * X_t[0] / X_t[2] + cx; pv = fy * X_t[1] / X_t[2] + cy
            if 6 < pu < w - 7 and 6 < pv < h - 7:
                res_point.append(i); res_target.append(t)
    # perturbed initial state
    idepth0 = (idepth_true * (1.0 + idepth_noise * rng.standard_normal(N))).astype(np.float32)
    poses0 = []
    for k in range(n_frames):
        d = np.concatenate([rng.normal(0, trans_noise, 3), rng.normal(0, rot_noise, 3)]) if k > 0 else np.zeros(6)
        dR, dt = se3_exp(d)
        R0 = dR @ Rs[k]; t0 = dR @ ts[k] + dt
        poses0.append(pose7(R0, t0))
    return dict(K4=K4, w=w, h=h, imgs=imgs, poses_true=[pose7(R, t) for R, t in zip(Rs, ts)], poses0=poses0, host=host, u=u, v=v,
                idepth_true=idepth_true, idepth0=idepth0, color=color, weights=weights,
                res_point=np.array(res_point, dtype=np.int32), res_target=np.array(res_target, dtype=np.int32), n_frames=n_frames)
