"""A small keyframe-based visual odometry loop assembled ONLY from the entry points of the hot path — tracking
(FullSystem::trackNewCoarse), immature-point tracing and activation (traceNewCoarse, optimizeImmaturePoint), sliding-window bundle
adjustment (FullSystem::optimize) and marginalisation (marginalizePointsF, marginalizeFrame) — driven identically through the HIP
library (C ABI) and through the CPU oracle.  The policies around those calls (keyframe every K frames, which candidates to
activate, the oldest keyframe is marginalised) are simplified versions of the reference's and are the same for both back ends; what
is compared is the trajectory both produce on the same synthetic sequence.  Test infrastructure only."""
import numpy as np

IDENT = np.array([0, 0, 0, 0, 0, 0, 1.0])


# ---------------------------------------------------------------------------------------------------- pose helpers (float64, numpy)
def p7_to_T(p):
    x, y, z, w = p[3:7]
    n = np.sqrt(x * x + y * y + z * z + w * w); x, y, z, w = x / n, y / n, z / n, w / n
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = p[:3]
    return T


def T_to_p7(T):
    R = T[:3, :3]
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    if w > 1e-6:
        x = (R[2, 1] - R[1, 2]) / (4 * w); y = (R[0, 2] - R[2, 0]) / (4 * w); z = (R[1, 0] - R[0, 1]) / (4 * w)
    else:                                           # never reached by the small rotations of the test sequence
        x = np.sqrt(max(0.0, 1 + R[0, 0] - R[1, 1] - R[2, 2])) / 2; y = (R[0, 1] + R[1, 0]) / (4 * x); z = (R[0, 2] + R[2, 0]) / (4 * x); w = (R[2, 1] - R[1, 2]) / (4 * x)
    return np.array([T[0, 3], T[1, 3], T[2, 3], x, y, z, w])


def p7_mul(a, b):
    return T_to_p7(p7_to_T(a) @ p7_to_T(b))


def p7_inv(a):
    return T_to_p7(np.linalg.inv(p7_to_T(a)))


# ---------------------------------------------------------------------------------------------------- back ends
class HipBackend:
    name = "hip"

    def __init__(self, pkg, w, h, K4, n_slots=64, accumulators=1):
        self.pkg, self.w, self.h, self.K4 = pkg, w, h, K4
        self.ctx = pkg.Context(w, h, n_slots=n_slots)
        self.trk = pkg.CoarseTrackerHip(self.ctx); self.trk.makeK(K4)
        self.imm = pkg.ImmaturePointsHip(self.ctx, capacity=32768)
        self.ba = pkg.BundleAdjusterHip(self.ctx, accumulators=accumulators)   # 1 = the oracle's (single-threaded) summation order
        self.slot_of = {}

    def upload(self, fid, img):
        slot = self.slot_of.setdefault(fid, len(self.slot_of) % self.ctx.n_slots)
        self.ctx.frame_upload(slot, img)

    def tracker_set_ref(self, fid, u, v, idepth, hdiF):
        self.trk.setCoarseTrackingRef(self.slot_of[fid], u, v, idepth, hdiF)

    def track(self, fid, pose7, aff):
        r = self.trk.trackNewestCoarse(self.slot_of[fid], pose7, aff)
        return dict(pose7=np.array(r["pose7"]), aff=np.array(r["aff"]), good=bool(r["good"]), lastRes=np.array(r["lastResiduals"]))

    # immature points: list of (fid, u, v) groups in window order -> one device set; states round-trip through the host
    def imm_rebuild(self, groups, state):
        self.imm.clear()
        for tag, (fid, u, v) in enumerate(groups):
            if len(u):
                self.imm.add_points(tag, self.slot_of[fid], u, v)
        if state is not None and self.imm.n:
            self.imm.set_state(state["idepth_min"], state["idepth_max"], state["quality"], state["lastTraceStatus"])

    def imm_trace(self, new_fid, new_w2c, host_c2w):
        if self.imm.n:
            self.imm.traceNewCoarse(self.slot_of[new_fid], new_w2c, np.stack(host_c2w), self.K4)

    def imm_state(self):
        return self.imm.get_state()

    def imm_static(self):
        return self.imm.get_static()

    def imm_optimize(self, fids, w2c, select):
        return self.imm.optimize([self.slot_of[f] for f in fids], np.stack(w2c), self.K4, select=select, min_obs=1)

    def ba_window(self, case, fids, states, prior):
        self.ba.set_case(case, [self.slot_of[f] for f in fids])
        for k, st in enumerate(states):
            if st is not None:
                self.ba.set_frame_state(k, st)
        if prior is not None:
            self.ba.set_marg_prior(*prior)
        return self.ba

    def ba_marg_points(self, ba, cand):
        dec, H, b, _ = ba.marginalize_points(cand, update_prior=True)
        return dec

    def ba_prior(self, ba):
        return ba.get_marg_prior()


class OracleBackend:
    name = "oracle"

    def __init__(self, oracle, w, h, K4):
        self.O, self.w, self.h, self.K4 = oracle, w, h, K4
        self.dI = {}
        self.T = oracle.Tracker(w, h); self.T.make_k(K4)
        self.groups = []; self.sets = []

    def upload(self, fid, img):
        self.dI[fid] = self.O.make_images(img, self.w, self.h)[0]

    def tracker_set_ref(self, fid, u, v, idepth, hdiF):
        self.T.set_ref(self.dI[fid], u, v, idepth, hdiF)

    def track(self, fid, pose7, aff):
        self.T.set_new(self.dI[fid])
        r = self.T.track(pose7, aff)
        return dict(pose7=np.array(r["pose7"]), aff=np.array(r["aff"]), good=bool(r["good"]), lastRes=np.array(r["lastResiduals"]))

    def imm_rebuild(self, groups, state):
        self.groups = groups; self.sets = []
        o = 0
        for fid, u, v in groups:
            P = self.O.ImmaturePoints(self.dI[fid][0], self.w, self.h, u, v)
            if state is not None and P.n:
                sl = slice(o, o + P.n)
                P.idepth_min[:] = state["idepth_min"][sl]; P.idepth_max[:] = state["idepth_max"][sl]; P.quality[:] = state["quality"][sl]
                P.lastTraceStatus[:] = state["lastTraceStatus"][sl]
            o += P.n
            self.sets.append(P)

    def imm_trace(self, new_fid, new_w2c, host_c2w):
        for P, c2w in zip(self.sets, host_c2w):
            if P.n:
                KRKi, Kt, aff = self.O.trace_precalc(new_w2c, c2w, self.K4)
                P.trace_on(self.dI[new_fid][0], KRKi, Kt, aff)

    def _cat(self, name, width=None):
        arrs = [getattr(P, name) for P in self.sets if P.n]
        if not arrs:
            return np.zeros((0,) if width is None else (0, width), np.float32)
        return np.concatenate(arrs)

    def imm_state(self):
        return dict(idepth_min=self._cat("idepth_min"), idepth_max=self._cat("idepth_max"), quality=self._cat("quality"), lastTraceUV=self._cat("lastTraceUV", 2),
                    lastTracePixelInterval=self._cat("lastTracePixelInterval"), lastTraceStatus=self._cat("lastTraceStatus").astype(np.int32))

    def imm_static(self):
        host = np.concatenate([np.full(P.n, k, np.int32) for k, P in enumerate(self.sets)]) if self.sets else np.zeros(0, np.int32)
        return dict(u=self._cat("u"), v=self._cat("v"), host=host, color=self._cat("color", 8), weights=self._cat("weights", 8), gradH=self._cat("gradH", 4),
                    energyTH=self._cat("energyTH"))

    def imm_optimize(self, fids, w2c, select):
        F = len(fids)
        res, idp, rst = [], [], []
        o = 0
        for hI, P in enumerate(self.sets):
            if not P.n:
                continue
            others = [t for t in range(F) if t != hI]
            c2w = p7_inv(w2c[hI])
            pre = [self.O.pair_precalc(w2c[t], c2w) for t in others]
            r, d, s = self.O.immature_optimize(P, self.K4, [self.dI[fids[t]][0] for t in others], np.stack([p[0] for p in pre]), np.stack([p[1] for p in pre]),
                                               np.stack([p[2] for p in pre]), min_obs=1)
            full = -np.ones((P.n, F), np.int32); full[:, others] = s
            sel = select[o:o + P.n].astype(bool)
            r = np.where(sel, r, 0)
            res.append(r); idp.append(d); rst.append(full); o += P.n
        if not res:
            return np.zeros(0, np.int32), np.zeros(0, np.float32), np.zeros((0, F), np.int32)
        return np.concatenate(res), np.concatenate(idp), np.concatenate(rst)

    def ba_window(self, case, fids, states, prior):
        case = dict(case); case["dI0"] = [self.dI[f][0] for f in fids]
        W = self.O.BAWindow(case)
        for k, st in enumerate(states):
            if st is not None:
                W.set_frame_state(k, st)
        if prior is not None:
            W.set_marg_prior(*prior)
        self._prior = None if prior is None else (np.array(prior[0]), np.array(prior[1]))
        return W

    def ba_marg_points(self, ba, cand):
        dec, H, b, _ = ba.marginalize_points(cand)
        n = ba.n
        HM, bM = self._prior if self._prior is not None else (np.zeros((n, n)), np.zeros(n))
        self._prior = (HM + H, bM + b)
        ba.set_marg_prior(*self._prior)
        return dec

    def ba_prior(self, ba):
        n = ba.n
        return self._prior if self._prior is not None else (np.zeros((n, n)), np.zeros(n))


# ---------------------------------------------------------------------------------------------------- the loop
class MiniVO:
    def __init__(self, backend, synth, K4, w, h, kf_every=3, max_kf=4, n_new=350, seed=1):
        self.B, self.synth, self.K4, self.w, self.h = backend, synth, np.asarray(K4, dtype=np.float64), w, h
        self.kf_every, self.max_kf, self.n_new = kf_every, max_kf, n_new
        self.rng = np.random.RandomState(seed)
        self.kfs = []          # dicts: fid, frameID, evalPT (w2c pose7), aff0, state10 (or None), w2c (current), imm_u, imm_v
        self.pts = dict(host=np.zeros(0, np.int32), u=np.zeros(0, np.float32), v=np.zeros(0, np.float32), idepth=np.zeros(0, np.float32),
                        color=np.zeros((0, 8), np.float32), weights=np.zeros((0, 8), np.float32), prior=np.zeros(0, np.uint8))
        self.res = []          # (point index, target kf index)
        self.prior = None
        self.imm_state = None
        self.traj = {}         # fid -> camToWorld pose7
        self.last_rel = IDENT.copy()   # refToNew of the previous frame
        self.ref_w2c = None
        self.n_frames = 0
        self.log = []

    def _select(self, img, n):
        u, v = self.synth.select_points(img, n, self.rng, min_grad=8.0)
        u = u.astype(np.int32); v = v.astype(np.int32)
        keep = (u >= 6) & (v >= 6) & (u < self.w - 7) & (v < self.h - 7)
        return u[keep], v[keep]

    def _imm_groups(self):
        return [(k["fid"], k["imm_u"], k["imm_v"]) for k in self.kfs]

    # ---- first frame: keyframe at identity, depths from an external initialiser (ground truth + noise here)
    def init(self, fid, img, idepth_map, noise=0.03):
        self.B.upload(fid, img)
        u, v = self._select(img, 2 * self.n_new)
        idp = (idepth_map[v, u] * (1 + noise * self.rng.standard_normal(len(u)))).astype(np.float32)
        self.kfs = [dict(fid=fid, frameID=0, evalPT=IDENT.copy(), aff0=np.zeros(2), state10=None, w2c=IDENT.copy(), imm_u=np.zeros(0, np.int32), imm_v=np.zeros(0, np.int32))]
        # colours / weights of the new points come from the immature-point constructor of the back end
        self.B.imm_rebuild([(fid, u, v)], None)
        st = self.B.imm_static()
        self.pts = dict(host=np.zeros(len(u), np.int32), u=u.astype(np.float32), v=v.astype(np.float32), idepth=idp, color=st["color"], weights=st["weights"],
                        prior=np.ones(len(u), np.uint8))
        self.res = []
        self.kfs[0]["imm_u"], self.kfs[0]["imm_v"] = self._select(img, self.n_new)
        self.B.imm_rebuild(self._imm_groups(), None)
        self.imm_state = self.B.imm_state()
        self.B.tracker_set_ref(fid, u.astype(np.float32), v.astype(np.float32), idp, np.full(len(u), 1e-3, np.float32))
        self.ref_w2c = IDENT.copy()
        self.traj[fid] = IDENT.copy()
        self.n_frames = 1

    def add_frame(self, fid, img):
        B = self.B
        B.upload(fid, img)
        # ---- FullSystem::trackNewCoarse (constant-motion guess; one hypothesis is enough on this sequence)
        guess = p7_mul(self.last_rel, self.last_ref_to_prev if hasattr(self, "last_ref_to_prev") else IDENT)
        r = B.track(fid, guess, (0.0, 0.0))
        if not r["good"]:
            r = B.track(fid, self.last_ref_to_prev if hasattr(self, "last_ref_to_prev") else IDENT, (0.0, 0.0))
        ref_to_new = r["pose7"]
        prev = getattr(self, "last_ref_to_prev", IDENT)
        self.last_rel = p7_mul(ref_to_new, p7_inv(prev))
        self.last_ref_to_prev = ref_to_new
        w2c = p7_mul(ref_to_new, self.ref_w2c)
        self.traj[fid] = p7_inv(w2c)
        self.n_frames += 1
        # ---- FullSystem::traceNewCoarse
        B.imm_rebuild(self._imm_groups(), self.imm_state)
        B.imm_trace(fid, w2c, [p7_inv(k["w2c"]) for k in self.kfs])
        self.imm_state = B.imm_state()
        if (self.n_frames - 1) % self.kf_every == 0:
            self._make_keyframe(fid, img, w2c, r["aff"])
        return w2c

    def _make_keyframe(self, fid, img, w2c, aff):
        B = self.B
        F0 = len(self.kfs)
        self.kfs.append(dict(fid=fid, frameID=self.n_frames - 1, evalPT=w2c.copy(), aff0=np.array(aff, dtype=np.float64), state10=None, w2c=w2c.copy(),
                             imm_u=np.zeros(0, np.int32), imm_v=np.zeros(0, np.int32)))
        F = F0 + 1
        # every active point gets a residual to the new keyframe (FullSystem::makeKeyFrame, FullSystem.cpp:1292-1303)
        for pi in range(len(self.pts["u"])):
            self.res.append((pi, F0))
        # ---- activation: candidates traced well (simplified canActivate of FullSystem::activatePointsMT, FullSystem.cpp:650-700)
        st = self.imm_state
        n_imm = len(st["lastTraceStatus"])
        if n_imm:
            ok = np.isin(st["lastTraceStatus"], (0, 3, 4)) & np.isfinite(st["idepth_max"]) & (st["lastTracePixelInterval"] < 8) & (st["quality"] > 3) & \
                ((st["idepth_min"] + st["idepth_max"]) > 0)
            fids = [k["fid"] for k in self.kfs]
            result, idepth, rstate = B.imm_optimize(fids, [k["w2c"] for k in self.kfs], ok.astype(np.uint8))
            stat = B.imm_static()
            act = result == 1
            base = len(self.pts["u"])
            if act.any():
                for name, src in (("host", stat["host"][act].astype(np.int32)), ("u", stat["u"][act]), ("v", stat["v"][act]), ("idepth", idepth[act]),
                                  ("color", stat["color"][act]), ("weights", stat["weights"][act]), ("prior", np.zeros(int(act.sum()), np.uint8))):
                    self.pts[name] = np.concatenate([self.pts[name], src])
                for j, i in enumerate(np.nonzero(act)[0]):
                    for t in range(F):
                        if rstate[i, t] == 0:
                            self.res.append((base + j, t))
            # activated and deleted candidates leave the immature set; OOB ones too (they would be dropped by the reference)
            keep = ~(act | (result == -1) | (st["lastTraceStatus"] == 1))
            o = 0
            for k in self.kfs[:-1]:
                n = len(k["imm_u"]); kk = keep[o:o + n]
                k["imm_u"], k["imm_v"] = k["imm_u"][kk], k["imm_v"][kk]; o += n
            self.imm_state = {name: val[keep] for name, val in st.items()}
            self.log.append(dict(fid=fid, activated=int(act.sum()), deleted=int((result == -1).sum()), candidates=int(ok.sum())))
        # ---- FullSystem::optimize
        ba = self._window()
        out = ba.optimize(6)
        self._read_back(ba)
        self.log[-1 if self.log else 0:] and self.log[-1].update(energy=float(out["finalEnergy"]), iterations=int(out["iterations"]))
        # ---- marginalise the oldest keyframe when the window is full (FullSystem::flagFramesForMarginalization keeps it simpler)
        if len(self.kfs) > self.max_kf:
            self._marginalize_oldest(ba)
        # ---- new candidates on the new keyframe (FullSystem::makeNewTraces), new tracking reference (CoarseTracker::setCoarseTrackingRef)
        self.kfs[-1]["imm_u"], self.kfs[-1]["imm_v"] = self._select(img, self.n_new)
        nn = len(self.kfs[-1]["imm_u"])
        self.imm_state = dict(idepth_min=np.concatenate([self.imm_state["idepth_min"], np.zeros(nn, np.float32)]),
                              idepth_max=np.concatenate([self.imm_state["idepth_max"], np.full(nn, np.nan, np.float32)]),
                              quality=np.concatenate([self.imm_state["quality"], np.full(nn, 10000.0, np.float32)]),
                              lastTraceUV=np.concatenate([self.imm_state["lastTraceUV"], np.zeros((nn, 2), np.float32)]),
                              lastTracePixelInterval=np.concatenate([self.imm_state["lastTracePixelInterval"], np.zeros(nn, np.float32)]),
                              lastTraceStatus=np.concatenate([self.imm_state["lastTraceStatus"], np.full(nn, 5, np.int32)]))
        self._set_tracking_ref()

    def _case(self):
        F = len(self.kfs)
        order = np.argsort([r[0] for r in self.res], kind="stable")      # residuals sorted by point, window order inside a point
        rp = np.array([self.res[i][0] for i in order], np.int32); rt = np.array([self.res[i][1] for i in order], np.int32)
        return dict(K4=self.K4, w=self.w, h=self.h, n_frames=F, poses0=np.stack([k["evalPT"] for k in self.kfs]), aff=np.stack([k["aff0"] for k in self.kfs]),
                    exposure=np.ones(F, np.float32), frameIDs=np.array([k["frameID"] for k in self.kfs], np.int32), host=self.pts["host"], u=self.pts["u"],
                    v=self.pts["v"], idepth0=self.pts["idepth"], color=self.pts["color"], weights=self.pts["weights"], hasDepthPrior=self.pts["prior"],
                    res_point=rp, res_target=rt)

    def _window(self):
        case = self._case()
        self._rp, self._rt = case["res_point"], case["res_target"]
        prior = self.prior
        if prior is not None and prior[0].shape[0] != 4 + 8 * len(self.kfs):     # the newest keyframe enters with zero prior rows
            n = 4 + 8 * len(self.kfs); HM = np.zeros((n, n)); bM = np.zeros(n)
            m = prior[0].shape[0]; HM[:m, :m] = prior[0]; bM[:m] = prior[1]
            prior = (HM, bM)
        return self.B.ba_window(case, [k["fid"] for k in self.kfs], [k["state10"] for k in self.kfs], prior)

    def _read_back(self, ba):
        for k, kf in enumerate(self.kfs):
            pose, aff, st = ba.frame_pose(k)
            kf["w2c"] = np.array(pose)
            if kf["state10"] is None and k == len(self.kfs) - 1 and len(self.kfs) > 1:
                kf["evalPT"] = np.array(pose); kf["aff0"] = np.array(aff)       # optimize() re-anchors the newest frame (FullSystemOptimize.cpp:596-609)
            kf["state10"] = np.array(st)
            self.traj[kf["fid"]] = p7_inv(kf["w2c"])
        self.pts["idepth"] = np.array(ba.point_state()[0])

    def _marginalize_oldest(self, ba):
        cand = (self.pts["host"] == 0).astype(np.uint8)
        self.B.ba_marg_points(ba, cand)                       # decisions 1 / 2: the points leave the window either way
        HM, bM = ba.marginalize_frame(0)
        self.prior = (np.array(HM), np.array(bM))
        keep = cand == 0
        remap = -np.ones(len(keep), np.int64); remap[keep] = np.arange(int(keep.sum()))
        for name in self.pts:
            self.pts[name] = self.pts[name][keep]
        self.pts["host"] = self.pts["host"] - 1
        self.res = [(int(remap[p]), t - 1) for p, t in self.res if keep[p] and t != 0]
        n0 = len(self.kfs[0]["imm_u"])
        self.imm_state = {name: val[n0:] for name, val in self.imm_state.items()}
        self.kfs = self.kfs[1:]

    def _set_tracking_ref(self):
        """CoarseTracker::setCoarseTrackingRef: points whose residual to the newest keyframe is IN, at their projected position."""
        ba = self._window()
        ba.activate_all(); ba.linearize_all(True)
        rs = ba.res_state(); hd = ba.point_acc()["HdiF"] if False else None
        ba.accumulate()
        hdi = ba.point_acc()["HdiF"]
        F = len(self.kfs)
        sel = (self._rt == F - 1) & (np.asarray(rs["isActive"]) != 0)
        c = rs["center"][sel]
        new = self.kfs[-1]
        own = self.pts["host"] == F - 1                        # points hosted in the newest keyframe enter at their own pixel (none right after creation)
        u = np.concatenate([c[:, 0], self.pts["u"][own]]).astype(np.float32); v = np.concatenate([c[:, 1], self.pts["v"][own]]).astype(np.float32)
        idp = np.concatenate([c[:, 2], self.pts["idepth"][own]]).astype(np.float32)
        w = np.concatenate([hdi[self._rp[sel]], hdi[own]]).astype(np.float32)
        self.B.tracker_set_ref(new["fid"], u, v, idp, w)
        self.ref_w2c = new["w2c"].copy()
        self.last_ref_to_prev = IDENT.copy()
        self.log and self.log[-1].update(ref_points=int(len(u)))


def make_sequence(synth, w, h, n, seed=3, motion="drift", device=None):
    """Smooth motion over the plane world; returns images, inverse depth of frame 0, ground-truth camToWorld poses.
    motion: "drift" = forward / sideways drift (short sequences); "orbit" = a closed, bounded path with the same ~3 cm step per frame (long sequences)."""
    world = synth.PlaneWorld(synth.SEED + seed, fmax=20.0)
    K4 = synth.default_intrinsics(w, h)
    imgs, c2w, id0, Rs, ts = [], [], None, [], []
    for k in range(n):
        if motion == "orbit":
            xi = np.array([0.6 * np.sin(0.05 * k), -0.08 * np.sin(0.08 * k), 0.4 * (1 - np.cos(0.05 * k)),
                           0.03 * np.sin(0.07 * k), -0.06 * np.sin(0.05 * k), 0.02 * np.sin(0.09 * k)])
        else:
            xi = np.array([0.035 * k, -0.012 * k + 0.01 * np.sin(0.5 * k), 0.015 * k, 0.004 * np.sin(0.4 * k), -0.003 * k, 0.002 * k])
        R, t = synth.se3_exp(xi)
        Rs.append(R); ts.append(t); c2w.append(p7_inv(synth.pose7(R, t)))
        if device is None or k == 0:
            img, idm = world.render(K4, R, t, w, h)
            imgs.append(img)
            if k == 0:
                id0 = idm
    if device is not None:     # long sequences: the frames rendered with torch on the device (same formula; both back ends get the same images)
        imgs = list(synth.render_batch_torch(world, K4, Rs, ts, w, h, device).cpu().numpy())
    return K4, imgs, id0, c2w


def run(backend, synth, K4, imgs, id0, w, h, **kw):
    vo = MiniVO(backend, synth, K4, w, h, **kw)
    vo.init(0, imgs[0], id0)
    for k in range(1, len(imgs)):
        vo.add_frame(k, imgs[k])
    return vo
