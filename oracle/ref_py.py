"""ctypes binding of oracle/_ref/libref.so — the REFERENCE'S OWN SOURCES compiled unmodified (oracle/Makefile.ref) — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The library is built in this container
(where /root/reference exists) and travels to the GPU box as a prebuilt file; it is never built or needed by the product path.
The classes mirror oracle_py's (Tracker, BAWindow, ...) so that a test can run one case through the reference, the restatement and
the HIP library.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
REFERENCE_ROOT = "/root/reference"

c_f = C.POINTER(C.c_float)
c_d = C.POINTER(C.c_double)
c_i = C.POINTER(C.c_int)
vp = C.c_void_p


def so_path():
    return os.path.join(_HERE, "_ref", "libref.so")


def available():
    """True when the library exists or can be built here (the reference sources are present)."""
    return os.path.exists(so_path()) or os.path.isdir(os.path.join(REFERENCE_ROOT, "src"))


def build(force=False):
    so = so_path()
    if os.path.isdir(os.path.join(REFERENCE_ROOT, "src")):
        # make decides what is stale (shim headers, glue, reference sources)
        subprocess.check_call(["make", "-C", _HERE, "-f", "Makefile.ref", "-s", "-j8"] + (["-B"] if force else []))
    if not os.path.exists(so):
        raise RuntimeError("oracle/_ref/libref.so is missing and %s is not present to build it from" % REFERENCE_ROOT)
    return so


def enumerate_tries(on):
    """ENUMERATE mode of the trackNewestCoarse hook (oracle/ref_trackhook.cpp): every try of FullSystem::trackNewCoarse records its initial guess and fails."""
    L = lib()
    L.ref_enumerate_tries.argtypes = [C.c_int]
    L.ref_enumerate_tries(1 if on else 0)


def enumerated_tries(max_tries=64):
    L = lib()
    L.ref_enumerated_tries.argtypes = [c_d, C.c_int]
    out = np.zeros((max_tries, 7))
    n = L.ref_enumerated_tries(_d(out), max_tries)
    return out[:min(n, max_tries)].copy()


_SOPHUS = None


def sophus_pin():
    """oracle/_ref/libsophus_pin.so: the reference's vendored Sophus (so3.hpp / se3.hpp) compiled unmodified (oracle/sophus_pin.cpp); poses = t(3), q xyzw."""
    global _SOPHUS
    if _SOPHUS is None:
        build()
        L = C.CDLL(os.path.join(_HERE, "_ref", "libsophus_pin.so"))
        for name, args in (("sophus_se3_exp", [c_d, c_d]), ("sophus_se3_log", [c_d, c_d]), ("sophus_se3_mul", [c_d, c_d, c_d]), ("sophus_se3_inverse", [c_d, c_d]),
                           ("sophus_se3_adj", [c_d, c_d]), ("sophus_se3_matrix3x4", [c_d, c_d]), ("sophus_se3_transform", [c_d, c_d, c_d]),
                           ("sophus_se3_from_quaternion", [c_d, c_d]), ("sophus_se3_from_matrix", [c_d, c_d, c_d]), ("sophus_so3_exp", [c_d, c_d, c_d])):
            getattr(L, name).argtypes = args
        L.sophus_se3_log.restype = C.c_int

        class S:
            @staticmethod
            def _call(fn, n_out, *ins):
                out = np.zeros(n_out)
                fn(*[_d(np.ascontiguousarray(a, dtype=np.float64)) for a in ins], _d(out))
                return out
            exp = staticmethod(lambda a: S._call(L.sophus_se3_exp, 7, a))
            mul = staticmethod(lambda a, b: S._call(L.sophus_se3_mul, 7, a, b))
            inverse = staticmethod(lambda a: S._call(L.sophus_se3_inverse, 7, a))
            adj = staticmethod(lambda a: S._call(L.sophus_se3_adj, 36, a).reshape(6, 6))
            matrix3x4 = staticmethod(lambda a: S._call(L.sophus_se3_matrix3x4, 12, a).reshape(3, 4))
            transform = staticmethod(lambda a, p: S._call(L.sophus_se3_transform, 3, a, p))
            from_quaternion = staticmethod(lambda a: S._call(L.sophus_se3_from_quaternion, 7, a))
            from_matrix = staticmethod(lambda Rm, t: S._call(L.sophus_se3_from_matrix, 7, Rm, t))

            @staticmethod
            def log(a):
                out = np.zeros(6)
                if L.sophus_se3_log(_d(np.ascontiguousarray(a, dtype=np.float64)), _d(out)) != 0:
                    raise ValueError("SophusException")
                return out
        _SOPHUS = S
    return _SOPHUS


def _f(a):
    return a.ctypes.data_as(c_f)


def _d(a):
    return a.ctypes.data_as(c_d)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        L.ref_pyr_levels.argtypes = [C.c_int, C.c_int, c_f]
        L.ref_get_global_calib.argtypes = [C.c_int, c_f, c_f, c_i]
        L.ref_set_affine_opt_mode.argtypes = [C.c_double, C.c_double]
        L.ref_interp33.argtypes = [c_f, C.c_int, C.c_int, c_f, c_f, c_f]
        L.ref_interp31.argtypes = [c_f, C.c_int, C.c_int, c_f, c_f, c_f]
        L.ref_interp33_bilin.argtypes = [c_f, C.c_int, C.c_int, c_f, c_f, c_f]
        L.ref_project_point_short.argtypes = [C.c_float, C.c_float, C.c_float, c_f, c_f, c_f]
        L.ref_project_point_long.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, c_f, c_f, c_f, c_f]
        L.ref_aff_from_to.argtypes = [C.c_float, C.c_float, C.c_double, C.c_double, C.c_double, C.c_double, c_d]
        L.ref_acc9_stream.argtypes = [C.c_int, c_f, c_f, C.c_int, c_f, c_f, C.c_int, c_f, C.POINTER(C.c_long)]
        L.ref_accapprox_stream.argtypes = [C.c_int, c_f, C.c_int, c_f, C.POINTER(C.c_long)]
        L.ref_accxx_stream.argtypes = [C.c_int, c_f, C.c_int, c_f, c_f, c_f, C.POINTER(C.c_long)]
        L.ref_make_images.argtypes = [c_f, C.c_int, C.c_int, c_f, c_f, C.POINTER(c_f), C.POINTER(c_f)]
        L.ref_tracker_create.restype = vp
        L.ref_tracker_create.argtypes = [C.c_int, C.c_int, c_f]
        L.ref_tracker_destroy.argtypes = [vp]
        L.ref_tracker_levels.argtypes = [vp]
        L.ref_tracker_get_k.argtypes = [vp, C.c_int, c_f, c_f]
        L.ref_tracker_set_ref.argtypes = [vp, c_f, C.c_float, C.c_double, C.c_double, C.c_int, c_f, c_f, c_f, c_f]
        L.ref_tracker_set_new.argtypes = [vp, c_f, C.c_float]
        L.ref_tracker_get_dIp.argtypes = [vp, C.c_int, C.c_int, c_f]
        L.ref_tracker_pc_n.argtypes = [vp, C.c_int]
        L.ref_tracker_get_pc.argtypes = [vp, C.c_int, c_f, c_f, c_f, c_f]
        L.ref_tracker_get_idepth.argtypes = [vp, C.c_int, c_f, c_f]
        L.ref_tracker_calc_res.argtypes = [vp, C.c_int, c_d, c_d, C.c_float, c_d]
        L.ref_tracker_warped_n.argtypes = [vp]
        L.ref_tracker_get_warped.argtypes = [vp, c_f]
        L.ref_tracker_calc_gs.argtypes = [vp, C.c_int, c_d, c_d, c_d]
        L.ref_tracker_track.argtypes = [vp, c_d, c_d, C.c_int, c_d, C.c_int, c_d, c_d, c_d, c_d]
        L.ref_tracker_vio_calls.argtypes = [vp, c_i, c_i]
        L.ref_tracker_vio_log.argtypes = [vp, c_d]
        L.ref_tracker_vio_visual.argtypes = [vp, c_d]
    return _LIB


# ------------------------------------------------------------------------------------------------------------ primitives
def set_multithreading(on):
    """settings.cpp `multiThreading`: linearizeAll, applyRes and the accumulators of EnergyFunctional on NUM_THREADS (6) workers."""
    lib().ref_set_multithreading(1 if on else 0)


def pyr_levels(w, h, K4=(100.0, 100.0, 50.0, 50.0)):
    return lib().ref_pyr_levels(w, h, _f(_f32(K4)))


def global_calib(lvl):
    k = np.zeros(4, np.float32); ki = np.zeros(9, np.float32); wh = np.zeros(2, np.int32)
    lib().ref_get_global_calib(lvl, _f(k), _f(ki), wh.ctypes.data_as(c_i)); return k, ki.reshape(3, 3), wh


def interp33(img3, x, y, kind="33"):
    img3 = _f32(img3); width = img3.shape[1]
    x = _f32(x); y = _f32(y)
    if kind == "31":
        out = np.zeros(len(x), np.float32); lib().ref_interp31(_f(img3), width, len(x), _f(x), _f(y), _f(out)); return out
    out = np.zeros((len(x), 3), np.float32)
    (lib().ref_interp33 if kind == "33" else lib().ref_interp33_bilin)(_f(img3), width, len(x), _f(x), _f(y), _f(out))
    return out


def project_point_short(u, v, idepth, KRKi, Kt):
    out = np.zeros(2, np.float32)
    ok = lib().ref_project_point_short(u, v, idepth, _f(_f32(KRKi).reshape(-1)), _f(_f32(Kt)), _f(out))
    return bool(ok), out


def project_point_long(u, v, idepth, dx, dy, K4, R, t):
    out = np.zeros(9, np.float32)
    ok = lib().ref_project_point_long(u, v, idepth, dx, dy, _f(_f32(K4)), _f(_f32(R).reshape(-1)), _f(_f32(t)), _f(out))
    return bool(ok), out


def aff_from_to(eF, eT, aF, bF, aT, bT):
    out = np.zeros(2); lib().ref_aff_from_to(eF, eT, aF, bF, aT, bT, _d(out)); return out


def acc9_stream(J, w, Js=None, ws=None, reps=1, _fn=None):
    """Accumulator9: `reps` times J [4k, 9] through updateSSE_eighted, then Js [m, 9] through updateSingleWeighted -> (H 9x9, num)."""
    J = _f32(J); w = _f32(w); n4 = len(w) // 4
    Js = np.zeros((0, 9), np.float32) if Js is None else _f32(Js); ws = np.zeros(0, np.float32) if ws is None else _f32(ws)
    H = np.zeros((9, 9), np.float32); num = C.c_long(0)
    (_fn or lib().ref_acc9_stream)(n4, _f(J), _f(w), len(ws), _f(Js), _f(ws), reps, _f(H), C.byref(num))
    return H, num.value


def accapprox_stream(rec35, reps=1, _fn=None):
    rec = _f32(rec35); H = np.zeros((13, 13), np.float32); num = C.c_long(0)
    (_fn or lib().ref_accapprox_stream)(len(rec), _f(rec), reps, _f(H), C.byref(num)); return H, num.value


def accxx_stream(rec21, reps=1, _fn=None):
    rec = _f32(rec21); a88 = np.zeros((8, 8), np.float32); a84 = np.zeros((8, 4), np.float32); a8 = np.zeros(8, np.float32); num = C.c_long(0)
    (_fn or lib().ref_accxx_stream)(len(rec), _f(rec), reps, _f(a88), _f(a84), _f(a8), C.byref(num)); return a88, a84, a8, num.value


def make_images(color, w, h, K4=(100.0, 100.0, 50.0, 50.0), B=None):
    """FrameHessian::makeImages of the reference -> (dIp levels [h_l, w_l, 3], absSquaredGrad levels [h_l, w_l])."""
    L = lib()
    levels = pyr_levels(w, h, K4)
    color = _f32(color).reshape(-1)
    dI = [np.zeros(((h >> l), (w >> l), 3), np.float32) for l in range(levels)]
    ab = [np.zeros(((h >> l), (w >> l)), np.float32) for l in range(levels)]
    dp = (c_f * levels)(*[_f(a) for a in dI]); ap = (c_f * levels)(*[_f(a) for a in ab])
    Bp = None if B is None else _f(_f32(B))
    L.ref_make_images(_f(color), w, h, _f(_f32(K4)), Bp, dp, ap)
    return dI, ab


# ------------------------------------------------------------------------------------------------------------ input edge
class Undistorter:
    """The reference's Undistort (+ its PhotometricUndistorter) built by Undistort::getUndistorterForFile from a camera file, a response file and a vignette image
    (registered in memory: the OpenCV reader is replaced by oracle/ref_imagerw.cpp)."""

    def __init__(self, config_txt, gamma_txt, vignette16):
        L = self.L = lib()
        c_u16 = C.POINTER(C.c_ushort)
        L.ref_register_image16.argtypes = [C.c_char_p, c_u16, C.c_int, C.c_int]
        L.ref_undistort_create.restype = vp; L.ref_undistort_create.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
        L.ref_undistort_destroy.argtypes = [vp]; L.ref_undistort_info.argtypes = [vp, c_i]
        L.ref_undistort_tables.argtypes = [vp, c_f, c_f, c_f, c_f, c_d]
        L.ref_undistort_run.argtypes = [vp, C.c_void_p, C.c_int, C.c_float, C.c_float, c_f, c_f]
        vig = np.ascontiguousarray(vignette16, dtype=np.uint16)
        name = b"vignette@%d" % id(self)
        L.ref_register_image16(name, vig.ctypes.data_as(c_u16), vig.shape[1], vig.shape[0])
        self.p = vp(L.ref_undistort_create(str(config_txt).encode(), str(gamma_txt).encode(), name))
        assert self.p, "getUndistorterForFile failed"
        info = np.zeros(7, np.int32); L.ref_undistort_info(self.p, info.ctypes.data_as(c_i))
        self.w, self.h, self.wOrg, self.hOrg, self.GDepth, self.valid, self.passthrough = [int(x) for x in info]
        self.remapX = np.zeros((self.h, self.w), np.float32); self.remapY = np.zeros((self.h, self.w), np.float32)
        self.G = np.zeros(self.GDepth, np.float32); self.vignetteMapInv = np.zeros((self.hOrg, self.wOrg), np.float32); self.K = np.zeros(9)
        L.ref_undistort_tables(self.p, _f(self.remapX), _f(self.remapY), _f(self.G), _f(self.vignetteMapInv), _d(self.K))

    def __del__(self):
        if getattr(self, "p", None):
            self.L.ref_undistort_destroy(self.p); self.p = None

    def undistort(self, raw, exposure=1.0, factor=1.0):
        raw = np.ascontiguousarray(raw)
        bits = 8 if raw.dtype == np.uint8 else 16
        out = np.zeros((self.h, self.w), np.float32); e = C.c_float(0)
        self.L.ref_undistort_run(self.p, raw.ctypes.data_as(C.c_void_p), bits, exposure, factor, _f(out), C.byref(e))
        return out, e.value


# ------------------------------------------------------------------------------------------------------------ CoarseInitializer
def init_make_k(w, h, K4, lvl):
    """CoarseInitializer::makeK: (fx, fy, cx, cy) of level lvl and Ki[lvl] (3x3) as float64."""
    L = lib()
    L.ref_init_make_k.argtypes = [C.c_int, C.c_int, c_f, C.c_int, c_d, c_d]
    out4 = np.zeros(4); Ki = np.zeros(9)
    assert L.ref_init_make_k(w, h, _f(_f32(K4)), lvl, _d(out4), _d(Ki)) == 0
    return out4, Ki.reshape(3, 3)


def init_calc_res_and_gs(img_ref, img_new, w, h, K4, lvl, refToNew7, aff_ab, pts, idepth_new, alphaW=150 * 150, alphaK=2.5 * 2.5, couplingWeight=1.0, priorY=0.0, priorX=0.0):
    """The reference's CoarseInitializer::calcResAndGS on level lvl; same outputs as oracle_py.init_calc_res_and_gs."""
    L = lib()
    c_u8 = C.POINTER(C.c_ubyte)
    L.ref_init_calc_res_and_gs.argtypes = [c_f, c_f, C.c_int, C.c_int, c_f, C.c_int, c_d, C.c_double, C.c_double, C.c_int, c_f, c_f, c_f, c_f, c_u8, c_f, c_f,
                                           C.c_float, C.c_float, C.c_float, C.c_double, C.c_double, c_f, c_f, c_f, c_f, c_f, c_f, c_u8, c_f, c_f, c_f]
    n = len(pts["u"])
    u, v, iR, en, oth, idn = _f32(pts["u"]), _f32(pts["v"]), _f32(pts["iR"]), _f32(pts["energy"]), _f32(pts["outlierTH"]), _f32(idepth_new)
    good = np.ascontiguousarray(pts["isGood"], dtype=np.uint8)
    o = dict(H=np.zeros((8, 8), np.float32), b=np.zeros(8, np.float32), Hsc=np.zeros((8, 8), np.float32), bsc=np.zeros(8, np.float32), res3=np.zeros(3, np.float32),
             energy_new=np.zeros((n, 2), np.float32), isGood_new=np.zeros(n, np.uint8), maxstep=np.zeros(n, np.float32), lastHessian_new=np.zeros(n, np.float32),
             JbBuffer_new=np.zeros((n, 10), np.float32))
    r = L.ref_init_calc_res_and_gs(_f(_f32(img_ref).reshape(-1)), _f(_f32(img_new).reshape(-1)), w, h, _f(_f32(K4)), lvl, _d(_f64(refToNew7)), float(aff_ab[0]), float(aff_ab[1]), n,
                                   _f(u), _f(v), _f(idn), _f(iR), good.ctypes.data_as(c_u8), _f(en), _f(oth), alphaW, alphaK, couplingWeight, priorY, priorX,
                                   _f(o["H"]), _f(o["b"]), _f(o["Hsc"]), _f(o["bsc"]), _f(o["res3"]), _f(o["energy_new"]), o["isGood_new"].ctypes.data_as(c_u8), _f(o["maxstep"]),
                                   _f(o["lastHessian_new"]), _f(o["JbBuffer_new"]))
    assert r == 0
    return o


# ------------------------------------------------------------------------------------------------------------ CoarseTracker
class Tracker:
    """The reference's CoarseTracker (CoarseTracker.h:46-129).  Same surface as oracle_py.Tracker except that frames are given as
    raw images (the reference builds its own pyramids with FrameHessian::makeImages)."""

    def __init__(self, w, h, K4):
        self.L = lib(); self.w, self.h = w, h
        self.K4 = _f32(K4)
        self.p = vp(self.L.ref_tracker_create(w, h, _f(self.K4)))
        self.levels = self.L.ref_tracker_levels(self.p)

    def __del__(self):
        if getattr(self, "p", None):
            self.L.ref_tracker_destroy(self.p); self.p = None

    def get_k(self, lvl):
        k = np.zeros(4, np.float32); ki = np.zeros(9, np.float32)
        self.L.ref_tracker_get_k(self.p, lvl, _f(k), _f(ki)); return k, ki.reshape(3, 3)

    def set_ref(self, img, u, v, idepth, hdiF, exposure=1.0, aff=(0.0, 0.0)):
        u, v, idepth, hdiF = _f32(u), _f32(v), _f32(idepth), _f32(hdiF)
        self.L.ref_tracker_set_ref(self.p, _f(_f32(img).reshape(-1)), exposure, aff[0], aff[1], len(u), _f(u), _f(v), _f(idepth), _f(hdiF))

    def set_new(self, img, exposure=1.0):
        self.L.ref_tracker_set_new(self.p, _f(_f32(img).reshape(-1)), exposure)

    def dIp(self, which, lvl):
        o = np.zeros(((self.h >> lvl), (self.w >> lvl), 3), np.float32); self.L.ref_tracker_get_dIp(self.p, which, lvl, _f(o)); return o

    def pc_n(self, lvl):
        return self.L.ref_tracker_pc_n(self.p, lvl)

    def get_pc(self, lvl):
        n = self.pc_n(lvl); out = [np.zeros(n, np.float32) for _ in range(4)]
        self.L.ref_tracker_get_pc(self.p, lvl, *[_f(a) for a in out]); return out

    def get_idepth(self, lvl):
        n = (self.w >> lvl) * (self.h >> lvl); a = np.zeros(n, np.float32); b = np.zeros(n, np.float32)
        self.L.ref_tracker_get_idepth(self.p, lvl, _f(a), _f(b)); return a, b

    def calc_res(self, lvl, pose7, aff, cutoff=20.0):
        rs = np.zeros(6); self.L.ref_tracker_calc_res(self.p, lvl, _d(_f64(pose7)), _d(_f64(aff)), cutoff, _d(rs)); return rs

    def get_warped(self):
        n = self.L.ref_tracker_warped_n(self.p); out = np.zeros((8, n), np.float32)
        self.L.ref_tracker_get_warped(self.p, _f(out)); return out

    def calc_gs(self, lvl, aff):
        H = np.zeros(64); b = np.zeros(8); self.L.ref_tracker_calc_gs(self.p, lvl, _d(_f64(aff)), _d(H), _d(b)); return H.reshape(8, 8), b

    def track(self, pose7, aff, coarsest=None, min_res=None, modeA=1e12, modeB=1e8, vio=False):
        pose = np.array(pose7, dtype=np.float64); a = np.array(aff, dtype=np.float64)
        if coarsest is None:
            coarsest = self.levels - 1
        mr = np.full(5, np.nan) if min_res is None else np.array(min_res, dtype=np.float64)
        lr = np.zeros(5); fl = np.zeros(3); H = np.zeros(64); b = np.zeros(8)
        self.L.ref_set_affine_opt_mode(modeA, modeB)
        good = self.L.ref_tracker_track(self.p, _d(pose), _d(a), coarsest, _d(mr), 1 if vio else 0, _d(lr), _d(fl), _d(H), _d(b))
        r = dict(good=bool(good), pose7=pose, aff=a, lastResiduals=lr, flow=fl, H=H.reshape(8, 8), b=b)
        if vio:
            acc = C.c_int(0); vis = C.c_int(0)
            n = self.L.ref_tracker_vio_calls(self.p, C.byref(acc), C.byref(vis))
            log = np.zeros((n, 74)); self.L.ref_tracker_vio_log(self.p, _d(log))
            r.update(vio_calls=n, vio_accepts=acc.value, vio_visual=vis.value, vio_log=log)
            if vis.value:
                vv = np.zeros(73); self.L.ref_tracker_vio_visual(self.p, _d(vv)); r["vio_visual_Hb"] = vv
        return r


# ------------------------------------------------------------------------------------------------------------ sliding-window BA
_BA_SIG = False


def _ba_sig(L):
    global _BA_SIG
    if _BA_SIG:
        return
    c_u8 = C.POINTER(C.c_ubyte)
    L.ref_ba_create.restype = vp; L.ref_ba_create.argtypes = [C.c_int, C.c_int, c_d]
    L.ref_ba_destroy.argtypes = [vp]
    L.ref_ba_add_frame.argtypes = [vp, c_d, C.c_double, C.c_double, C.c_float, C.c_int, c_f]
    L.ref_ba_perturb_frame.argtypes = [vp, C.c_int, c_d]
    L.ref_ba_set_frame_state.argtypes = [vp, C.c_int, c_d]
    L.ref_ba_set_frame_zero.argtypes = [vp, C.c_int, c_d]
    L.ref_ba_set_frame_energy_th.argtypes = [vp, c_f]
    L.ref_ba_set_calib_values.argtypes = [vp, c_d, c_d]
    c_i = C.POINTER(C.c_int)
    L.ref_ba_immature_add.argtypes = [vp, C.c_int, C.c_int, c_i, c_i]
    L.ref_ba_immature_get.argtypes = [vp, C.c_int, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i]
    L.ref_ba_immature_set_interval.argtypes = [vp, C.c_int, c_f, c_f]
    L.ref_ba_trace_new_coarse.argtypes = [vp, C.c_int]
    L.ref_ba_optimize_immature.argtypes = [vp, C.c_int, C.c_int, c_i, c_f, c_i]
    L.ref_ba_add_point.argtypes = [vp, C.c_int, C.c_float, C.c_float, C.c_float, c_f, c_f, C.c_int, c_f, c_f]
    L.ref_ba_add_residual.argtypes = [vp, C.c_int, C.c_int]
    L.ref_ba_finalize.argtypes = [vp]
    L.ref_ba_set_marg_prior.argtypes = [vp, c_d, c_d]
    L.ref_ba_activate_all.argtypes = [vp]
    L.ref_ba_linearize_all.restype = C.c_double; L.ref_ba_linearize_all.argtypes = [vp, C.c_int]
    L.ref_ba_apply_res.argtypes = [vp]
    L.ref_ba_get_res_state.argtypes = [vp, c_i, c_d, c_d, c_i, c_f]
    L.ref_ba_get_J.argtypes = [vp, C.c_int, C.c_int, c_f, c_f]
    L.ref_ba_get_res_to_zero.argtypes = [vp, C.c_int, c_f, c_i]
    L.ref_ba_get_frame_energy_th.argtypes = [vp, c_f]
    L.ref_ba_get_precalc.argtypes = [vp, C.c_int, C.c_int, c_f]
    L.ref_ba_get_adjoints.argtypes = [vp, c_d, c_d, c_f]
    L.ref_ba_accumulate.argtypes = [vp, c_d, c_d, c_d, c_d, c_d, c_d, c_i]
    L.ref_ba_get_point_acc.argtypes = [vp, c_f, c_f, c_f, c_f, c_f]
    L.ref_ba_solve.argtypes = [vp, C.c_int, C.c_double, c_d]
    L.ref_ba_get_last_system.argtypes = [vp, c_d, c_d]
    L.ref_ba_resubstitute.argtypes = [vp, c_d]
    L.ref_ba_get_point_state.argtypes = [vp, c_f, c_f]
    L.ref_ba_get_frame_pose.argtypes = [vp, C.c_int, c_d, c_d, c_d]
    L.ref_ba_get_frame_step.argtypes = [vp, C.c_int, c_d]
    L.ref_ba_get_calib.argtypes = [vp, c_d]
    L.ref_ba_get_nullspaces.argtypes = [vp, c_d]
    L.ref_ba_orthogonalize.argtypes = [vp, c_d]
    L.ref_ba_calc_lenergy.restype = C.c_double; L.ref_ba_calc_lenergy.argtypes = [vp]
    L.ref_ba_calc_menergy.restype = C.c_double; L.ref_ba_calc_menergy.argtypes = [vp]
    L.ref_ba_backup_state.argtypes = [vp, C.c_int]
    L.ref_ba_load_state_backup.argtypes = [vp]
    L.ref_ba_do_step_from_backup.argtypes = [vp] + [C.c_float] * 5
    L.ref_ba_optimize.restype = C.c_float; L.ref_ba_optimize.argtypes = [vp, C.c_int, c_i, c_d]
    L.ref_ba_log.argtypes = [vp, C.c_char_p, C.c_int]
    L.ref_ba_marginalize_points.argtypes = [vp, c_u8, c_u8, c_d, c_d]
    L.ref_ba_marginalize_frame.argtypes = [vp, C.c_int, c_d, c_d]
    L.ref_facade_create.restype = vp; L.ref_facade_create.argtypes = [C.c_int, c_d, C.c_double, C.c_double, C.c_double]
    L.ref_facade_destroy.argtypes = [vp]
    L.ref_facade_update_values.argtypes = [vp, c_d]
    L.ref_facade_compute.argtypes = [vp, c_d, c_d, C.c_double, c_d, c_d, c_d]
    L.ref_facade_energy.restype = C.c_double; L.ref_facade_energy.argtypes = [vp, C.c_int]
    L.ref_facade_accept.argtypes = [vp, C.c_double]
    L.ref_facade_weight.restype = C.c_double; L.ref_facade_weight.argtypes = [vp, C.c_double, C.c_double, C.c_int]
    L.ref_facade_can_break.argtypes = [vp]
    L.ref_facade_post.argtypes = [vp, c_d]
    L.ref_facade_n_events.argtypes = [vp]; L.ref_facade_get_events.argtypes = [vp, c_d]
    L.ref_facade_n_log.argtypes = [vp]; L.ref_facade_get_log.argtypes = [vp, c_d]
    L.ref_facade_get_goal.argtypes = [vp, c_d]
    L.ref_ba_set_marg_prior_gtsam.argtypes = [vp, c_d, c_d]
    L.ref_ba_set_resInA.argtypes = [vp, C.c_int]; L.ref_ba_get_resInA.argtypes = [vp]
    L.ref_ba_optimize_gtsam.restype = C.c_float; L.ref_ba_optimize_gtsam.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, c_i, c_d]
    _BA_SIG = True


class GtsamFacade:
    """Stand-in for the GTSAM graph behind dmvio::BAGTSAMIntegration (oracle/ref_glue.cpp, GtsamFacade): one quadratic factor over the stacked window state
    [calib value (4) | per keyframe state (8)], solved together with the photometric system the way BAGTSAMIntegration::computeBAUpdate does.  The reference calls it
    through the shim's hooks (BAWindow.optimize_gtsam); its methods have the signatures dmvio_amd.BundleAdjusterHip.optimize_vio expects of `hooks`, so the very same
    arithmetic answers the HIP library's callbacks.  events(): one row [code, a, b, c] per hook call; handovers(): what computeBAUpdate received."""

    def __init__(self, n, w, goal_offset=0.0, dyn_weight=1.0, break_below=0.0):
        self.L = lib(); _ba_sig(self.L)
        self.n = n
        self.p = vp(self.L.ref_facade_create(n, _d(_f64(w)), float(goal_offset), float(dyn_weight), float(break_below)))

    def __del__(self):
        if getattr(self, "p", None):
            self.L.ref_facade_destroy(self.p); self.p = None

    def _states(self, frames, calib):
        s = np.zeros(self.n); s[:4] = calib
        for f in frames:
            s[4 + 8 * f["index"]:12 + 8 * f["index"]] = f["state"][:8]
        return s

    # hooks of BundleAdjusterHip.optimize_vio
    def computeBAUpdate(self, HPassed, b, lam, HNoLambda, frames, calib):
        x = np.zeros(self.n)
        self.L.ref_facade_compute(self.p, _d(_f64(HPassed)), _d(_f64(b)), float(lam), _d(_f64(HNoLambda)), _d(self._states(frames, calib)), _d(x))
        return x

    def updateBAValues(self, frames, calib):
        self.L.ref_facade_update_values(self.p, _d(self._states(frames, calib)))

    def postOptimization(self, frames, calib):
        self.L.ref_facade_post(self.p, _d(self._states(frames, calib)))

    def getBAEnergy(self, useNew):
        return self.L.ref_facade_energy(self.p, 1 if useNew else 0)

    def acceptBAUpdate(self, e):
        self.L.ref_facade_accept(self.p, float(e))

    def updateDynamicWeight(self, e, rmse, good):
        return self.L.ref_facade_weight(self.p, float(e), float(rmse), 1 if good else 0)

    def canBreak(self):
        return self.L.ref_facade_can_break(self.p) != 0

    def events(self):
        n = self.L.ref_facade_n_events(self.p); o = np.zeros((n, 4))
        if n:
            self.L.ref_facade_get_events(self.p, _d(o))
        return o

    def handovers(self):
        k = self.L.ref_facade_n_log(self.p); n = self.n; w = 1 + 2 * n * n + n
        o = np.zeros((k, w))
        if k:
            self.L.ref_facade_get_log(self.p, _d(o))
        return [dict(lam=r[0], HPassed=r[1:1 + n * n].reshape(n, n), b=r[1 + n * n:1 + n * n + n], HNoLambda=r[1 + n * n + n:].reshape(n, n)) for r in o]


class BAWindow:
    """The reference's FullSystem with a window filled from a synth.ba_case (same constructor arguments as oracle_py.BAWindow)."""

    def __init__(self, case, poses=None, idepth=None, use_case_color=True):
        self.L = lib(); _ba_sig(self.L)
        self.case = case
        K4 = _f64(case["K4"])
        self.p = vp(self.L.ref_ba_create(case["w"], case["h"], _d(K4)))
        poses = case["poses0"] if poses is None else poses
        idepth = case["idepth0"] if idepth is None else idepth
        F = case["n_frames"]
        aff = np.zeros((F, 2)) if case.get("aff") is None else np.asarray(case["aff"], dtype=np.float64)
        expo = np.ones(F) if case.get("exposure") is None else np.asarray(case["exposure"], dtype=np.float64)
        fids = np.arange(F) if case.get("frameIDs") is None else np.asarray(case["frameIDs"])
        for k in range(F):
            img = _f32(case["imgs"][k]).reshape(-1)
            self.L.ref_ba_add_frame(self.p, _d(_f64(poses[k])), float(aff[k, 0]), float(aff[k, 1]), float(expo[k]), int(fids[k]), _f(img))
        col = _f32(case["color"]); wts = _f32(case["weights"])
        N = len(case["u"])
        hdp = np.zeros(N, np.uint8) if case.get("hasDepthPrior") is None else np.asarray(case["hasDepthPrior"], dtype=np.uint8)
        self.sampled_color = np.zeros((N, 8), np.float32); self.sampled_weights = np.zeros((N, 8), np.float32)
        for i in range(N):
            self.L.ref_ba_add_point(self.p, int(case["host"][i]), float(case["u"][i]), float(case["v"][i]), float(idepth[i]),
                                    _f(col[i]) if use_case_color else None, _f(wts[i]) if use_case_color else None, int(hdp[i]),
                                    _f(self.sampled_color[i]), _f(self.sampled_weights[i]))
        for pi, ti in zip(case["res_point"], case["res_target"]):
            self.L.ref_ba_add_residual(self.p, int(pi), int(ti))
        self.L.ref_ba_finalize(self.p)
        self.F = F; self.N = N; self.R = len(case["res_point"]); self.n = 4 + 8 * F

    def __del__(self):
        if getattr(self, "p", None):
            self.L.ref_ba_destroy(self.p); self.p = None

    # ---- immature points (ImmaturePoint constructor, FullSystem::traceNewCoarse, FullSystem::optimizeImmaturePoint — the reference's own members)
    def immature_add(self, host, u, v):
        u = np.ascontiguousarray(u, dtype=np.int32); v = np.ascontiguousarray(v, dtype=np.int32)
        c_i = C.POINTER(C.c_int)
        self._n_imm = getattr(self, "_n_imm", {})
        self._n_imm[host] = self.L.ref_ba_immature_add(self.p, host, len(u), u.ctypes.data_as(c_i), v.ctypes.data_as(c_i))

    def immature_get(self, host):
        n = self._n_imm[host]
        o = dict(color=np.zeros((n, 8), np.float32), weights=np.zeros((n, 8), np.float32), gradH=np.zeros((n, 4), np.float32), energyTH=np.zeros(n, np.float32),
                 idepth_min=np.zeros(n, np.float32), idepth_max=np.zeros(n, np.float32), quality=np.zeros(n, np.float32), lastTraceUV=np.zeros((n, 2), np.float32),
                 lastTracePixelInterval=np.zeros(n, np.float32), lastTraceStatus=np.zeros(n, np.int32))
        self.L.ref_ba_immature_get(self.p, host, _f(o["color"]), _f(o["weights"]), _f(o["gradH"]), _f(o["energyTH"]), _f(o["idepth_min"]), _f(o["idepth_max"]),
                                   _f(o["quality"]), _f(o["lastTraceUV"]), _f(o["lastTracePixelInterval"]), o["lastTraceStatus"].ctypes.data_as(C.POINTER(C.c_int)))
        return o

    def immature_set_interval(self, host, idepth_min, idepth_max):
        self.L.ref_ba_immature_set_interval(self.p, host, _f(_f32(idepth_min)), _f(_f32(idepth_max)))

    def immature_reset(self):
        self.L.ref_ba_immature_reset.argtypes = [vp]
        self.L.ref_ba_immature_reset(self.p)

    def trace_new_coarse(self, target):
        self.L.ref_ba_trace_new_coarse(self.p, target)

    def optimize_immature(self, host, min_obs=1):
        n = self._n_imm[host]; nres = self.F - 1
        result = np.zeros(n, np.int32); idepth = np.zeros(n, np.float32); st = np.zeros((n, nres), np.int32)
        c_i = C.POINTER(C.c_int)
        self.L.ref_ba_optimize_immature(self.p, host, min_obs, result.ctypes.data_as(c_i), _f(idepth), st.ctypes.data_as(c_i))
        return result, idepth, st

    def set_frame_state(self, k, state10):
        self.L.ref_ba_set_frame_state(self.p, k, _d(_f64(state10)))

    def set_frame_zero(self, k, state_zero10):
        self.L.ref_ba_set_frame_zero(self.p, k, _d(_f64(state_zero10)))

    def set_frame_energy_th(self, th):
        self.L.ref_ba_set_frame_energy_th(self.p, _f(_f32(th)))

    def set_calib_values(self, value4, value_zero4):
        self.L.ref_ba_set_calib_values(self.p, _d(_f64(value4)), _d(_f64(value_zero4)))

    def perturb_frame(self, k, d8):
        self.L.ref_ba_perturb_frame(self.p, k, _d(_f64(d8)))

    def activate_all(self):
        self.L.ref_ba_activate_all(self.p)

    def linearize_all(self, fix=False):
        return self.L.ref_ba_linearize_all(self.p, 1 if fix else 0)

    def apply_res(self):
        self.L.ref_ba_apply_res(self.p)

    def res_state(self):
        ns = np.zeros(self.R, np.int32); ne = np.zeros(self.R); nw = np.zeros(self.R); ia = np.zeros(self.R, np.int32); cp = np.zeros((self.R, 3), np.float32)
        self.L.ref_ba_get_res_state(self.p, ns.ctypes.data_as(c_i), _d(ne), _d(nw), ia.ctypes.data_as(c_i), _f(cp))
        return dict(newState=ns, newEnergy=ne, newEnergyWO=nw, isActive=ia, center=cp)

    def get_J(self, i, which=0):
        j = np.zeros(74, np.float32); jp = np.zeros(8, np.float32)
        self.L.ref_ba_get_J(self.p, i, which, _f(j), _f(jp))
        o = 0; out = {}
        for name, shape in (("resF", (8,)), ("Jpdxi", (2, 6)), ("Jpdc", (2, 4)), ("Jpdd", (2,)), ("JIdx", (2, 8)), ("JabF", (2, 8)),
                            ("JIdx2", (2, 2)), ("JabJIdx", (2, 2)), ("Jab2", (2, 2))):
            n = int(np.prod(shape)); out[name] = j[o:o + n].reshape(shape); o += n
        out["JpJdF"] = jp; out["raw"] = j
        return out

    def frame_energy_th(self):
        o = np.zeros(self.F, np.float32); self.L.ref_ba_get_frame_energy_th(self.p, _f(o)); return o

    def precalc(self, h, t):
        o = np.zeros(37, np.float32); self.L.ref_ba_get_precalc(self.p, h, t, _f(o))
        return dict(KRKi=o[0:9].reshape(3, 3), Kt=o[9:12], R0=o[12:21].reshape(3, 3), t0=o[21:24], aff=o[24:26], b0=o[26], R=o[27:36].reshape(3, 3), raw=o)

    def adjoints(self):
        n = self.F * self.F
        ah = np.zeros((n, 8, 8)); at = np.zeros((n, 8, 8)); d = np.zeros((n, 8), np.float32)
        self.L.ref_ba_get_adjoints(self.p, _d(ah), _d(at), _f(d)); return ah, at, d

    def accumulate(self):
        n = self.n
        m = [np.zeros((n, n)), np.zeros(n), np.zeros((n, n)), np.zeros(n), np.zeros((n, n)), np.zeros(n)]
        r = C.c_int(0)
        self.L.ref_ba_accumulate(self.p, *[_d(a) for a in m], C.byref(r))
        return dict(HA=m[0], bA=m[1], HL=m[2], bL=m[3], Hsc=m[4], bsc=m[5], resInA=r.value)

    def set_marg_prior(self, HM, bM):
        self.L.ref_ba_set_marg_prior(self.p, _d(_f64(HM)), _d(_f64(bM)))

    def point_acc(self):
        N = self.N
        o = [np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 4), np.float32), np.zeros(N, np.float32), np.zeros(N, np.float32)]
        self.L.ref_ba_get_point_acc(self.p, *[_f(a) for a in o])
        return dict(Hdd=o[0], bd=o[1], Hcd=o[2], HdiF=o[3], bdSumF=o[4])

    def solve(self, iteration, lam):
        x = np.zeros(self.n); self.L.ref_ba_solve(self.p, iteration, lam, _d(x)); return x

    def last_system(self):
        H = np.zeros((self.n, self.n)); b = np.zeros(self.n); self.L.ref_ba_get_last_system(self.p, _d(H), _d(b)); return H, b

    def resubstitute(self, x):
        self.L.ref_ba_resubstitute(self.p, _d(_f64(x)))

    def point_state(self):
        a = np.zeros(self.N, np.float32); b = np.zeros(self.N, np.float32)
        self.L.ref_ba_get_point_state(self.p, _f(a), _f(b)); return a, b

    def frame_pose(self, k):
        p = np.zeros(7); a = np.zeros(2); s = np.zeros(10)
        self.L.ref_ba_get_frame_pose(self.p, k, _d(p), _d(a), _d(s)); return p, a, s

    def nullspaces(self):
        o = np.zeros((7, self.n)); self.L.ref_ba_get_nullspaces(self.p, _d(o)); return o

    def orthogonalize(self, x):
        x = np.array(x, dtype=np.float64); self.L.ref_ba_orthogonalize(self.p, _d(x)); return x

    def lenergy(self):
        return self.L.ref_ba_calc_lenergy(self.p)

    def fix_linearization(self, mask):
        """EFResidual::fixLinearizationF of the active residuals with mask != 0; returns the number of linearised residuals of the window"""
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        self.L.ref_ba_fix_linearization.argtypes = [C.c_void_p, C.c_char_p]; self.L.ref_ba_fix_linearization.restype = C.c_int
        return self.L.ref_ba_fix_linearization(self.p, m.tobytes())

    def menergy(self):
        return self.L.ref_ba_calc_menergy(self.p)

    def optimize(self, its=6):
        n = C.c_int(0); tr = np.zeros((64, 2))
        rmse = self.L.ref_ba_optimize(self.p, its, C.byref(n), _d(tr))
        tr = tr[:n.value]
        return dict(rmse=rmse, trace=tr, iterations=int(np.sum(tr[:, 1] >= 0)), finalEnergy=float(tr[tr[:, 1] != 0][-1, 0]) if len(tr) else float("nan"))

    def optimize_quiet(self, its=6):
        """FullSystem::optimize, nothing captured or parsed (timing)."""
        self.L.ref_ba_optimize_quiet.restype = C.c_float; self.L.ref_ba_optimize_quiet.argtypes = [vp, C.c_int]
        return self.L.ref_ba_optimize_quiet(self.p, int(its))

    def optimize_gtsam(self, its, facade, update_during=False, tracking_was_good=True, min_opt_its=-1):
        """FullSystem::optimize on the reference's default branch (setting_useGTSAMIntegration) with `facade` (GtsamFacade) behind BAGTSAMIntegration."""
        n = C.c_int(0); tr = np.zeros((64, 2))
        rmse = self.L.ref_ba_optimize_gtsam(self.p, its, facade.p, 1 if update_during else 0, 1 if tracking_was_good else 0, int(min_opt_its), C.byref(n), _d(tr))
        tr = tr[:n.value]
        return dict(rmse=rmse, trace=tr, iterations=int(np.sum(tr[:, 1] >= 0)), finalEnergy=float(tr[tr[:, 1] != 0][-1, 0]) if len(tr) else float("nan"))

    def set_marg_prior_gtsam(self, HM, bM):
        self.L.ref_ba_set_marg_prior_gtsam(self.p, _d(_f64(HM)), _d(_f64(bM)))

    def resInA(self, v=None):
        if v is not None:
            self.L.ref_ba_set_resInA(self.p, int(v))
        return self.L.ref_ba_get_resInA(self.p)

    def log(self):
        buf = C.create_string_buffer(1 << 20); self.L.ref_ba_log(self.p, buf, len(buf)); return buf.value.decode(errors="replace")

    def set_num_good_residuals(self, v):
        self.L.ref_ba_set_num_good_residuals.argtypes = [vp, C.c_int]
        self.L.ref_ba_set_num_good_residuals(self.p, int(v))

    def marginalize_points(self, flagged_frames):
        n = self.n
        fl = np.ascontiguousarray(flagged_frames, dtype=np.uint8)
        dec = np.zeros(self.N, np.uint8); H = np.zeros((n, n)); b = np.zeros(n)
        c_u8 = C.POINTER(C.c_ubyte)
        nres = self.L.ref_ba_marginalize_points(self.p, fl.ctypes.data_as(c_u8), dec.ctypes.data_as(c_u8), _d(H), _d(b))
        return dec, H, b, nres

    def marginalize_frame(self, k):
        n = self.n - 8
        H = np.zeros((n, n)); b = np.zeros(n)
        self.L.ref_ba_marginalize_frame(self.p, k, _d(H), _d(b))
        return H, b


# ------------------------------------------------------------------------------------------------------------ the whole FullSystem
class System:
    """The reference's visual-only FullSystem, frame by frame (FullSystem::addActiveFrame), with the inputs / outputs of its
    trackNewCoarse, optimize and setCoarseTrackingRef calls recorded (see oracle/ref_glue.cpp, oracle/ref_timing.cpp)."""

    def __init__(self, w, h, K4, point_density=1000, max_frames=7, max_opt_its=6, min_opt_its=1):
        self.L = lib(); L = self.L
        L.ref_system_create.restype = vp
        L.ref_system_create.argtypes = [C.c_int, C.c_int, c_f, C.c_float, C.c_int, C.c_int, C.c_int]
        L.ref_system_destroy.argtypes = [vp]
        L.ref_system_add_frame.argtypes = [vp, c_f, C.c_float, C.c_double, C.c_int, c_i, C.c_char_p, C.c_int]
        L.ref_system_get_trajectory.argtypes = [vp, c_d, c_i, c_i, c_i, c_d]
        L.ref_system_print_result.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]
        L.ref_system_n_events.argtypes = [vp]
        L.ref_system_event_sizes.argtypes = [vp, C.c_int, c_i, c_i, c_i, c_i]
        L.ref_system_event_data.argtypes = [vp, C.c_int, c_d, c_f, c_i]
        self.w, self.h = w, h
        self.K4 = _f32(K4)
        self.p = vp(L.ref_system_create(w, h, _f(self.K4), point_density, max_frames, max_opt_its, min_opt_its))
        self.n = 0
        self.logs = []

    def __del__(self):
        if getattr(self, "p", None):
            self.L.ref_system_destroy(self.p); self.p = None

    def add_frame(self, img, exposure=1.0, timestamp=None):
        st = np.zeros(5, np.int32); buf = C.create_string_buffer(1 << 16)
        self.L.ref_system_add_frame(self.p, _f(_f32(img).reshape(-1)), exposure, float(self.n * 0.05 if timestamp is None else timestamp), self.n, st.ctypes.data_as(c_i), buf, len(buf))
        self.n += 1
        self.logs.append(buf.value.decode(errors="replace"))
        return dict(initialized=bool(st[0]), isLost=bool(st[1]), initFailed=bool(st[2]), window=int(st[3]), frames=int(st[4]))

    def trajectory(self):
        n = self.n
        P = np.zeros((n, 7)); v = np.zeros(n, np.int32); kf = np.zeros(n, np.int32); tr = np.zeros(n, np.int32); aff = np.zeros((n, 2))
        m = self.L.ref_system_get_trajectory(self.p, _d(P), v.ctypes.data_as(c_i), kf.ctypes.data_as(c_i), tr.ctypes.data_as(c_i), _d(aff))
        return dict(camToWorld=P[:m], valid=v[:m], keyframeId=kf[:m], trackingRef=tr[:m], aff=aff[:m])

    def shells(self):
        n = self.n
        self.L.ref_system_get_shells.argtypes = [vp, c_d, c_i, c_d, c_d]
        ts = np.zeros(n); nm = np.zeros(n, np.int32); cr = np.zeros((n, 7)); fp = np.zeros(7)
        m = self.L.ref_system_get_shells(self.p, _d(ts), nm.ctypes.data_as(c_i), _d(cr), _d(fp))
        return dict(timestamp=ts[:m], never_marginalized=nm[:m], camToTrackingRef=cr[:m], firstPose=fp)

    def print_result(self, path, only_kf=False, use_cam_to_tracking_ref=True):
        self.L.ref_system_print_result(self.p, str(path).encode(), 1 if only_kf else 0, 1 if use_cam_to_tracking_ref else 0)

    def events(self):
        """The recorded calls, decoded into dictionaries (kinds: setref, track_in, track_out, opt_in, opt_out)."""
        out = []
        for k in range(self.L.ref_system_n_events(self.p)):
            kind = C.c_int(0); nd = C.c_int(0); nf = C.c_int(0); ni = C.c_int(0)
            self.L.ref_system_event_sizes(self.p, k, C.byref(kind), C.byref(nd), C.byref(nf), C.byref(ni))
            d = np.zeros(max(nd.value, 1)); f = np.zeros(max(nf.value, 1), np.float32); i = np.zeros(max(ni.value, 1), np.int32)
            self.L.ref_system_event_data(self.p, k, _d(d), _f(f), i.ctypes.data_as(c_i))
            out.append(decode_event(kind.value, d[:nd.value], f[:nf.value], i[:ni.value]))
        return out


def decode_event(kind, d, f, i):
    if kind == 1:
        pts = f.reshape(-1, 4)
        return dict(kind="setref", ref_id=int(i[0]), exposure=float(d[0]), aff=d[1:3].copy(), K4=d[3:7].copy(), u=pts[:, 0].copy(), v=pts[:, 1].copy(), idepth=pts[:, 2].copy(), hdiF=pts[:, 3].copy())
    if kind == 2:
        return dict(kind="track_in", frame_id=int(i[0]), ref_id=int(i[1]), n_history=int(i[2]), poses_valid=bool(i[3]), slast_c2w=d[0:7].copy(), sprelast_c2w=d[7:14].copy(),
                    lastF_c2w=d[14:21].copy(), aff_last=d[21:23].copy(), lastCoarseRMSE=d[23:28].copy(), reTrackThreshold=float(d[28]))
    if kind == 3:
        return dict(kind="track_out", frame_id=int(i[0]), ref_id=int(i[1]), good=bool(i[2]), refToNew=d[0:7].copy(), aff=d[7:9].copy(), lastCoarseRMSE=d[9:14].copy(),
                    lastResiduals=d[14:19].copy(), flow=d[19:22].copy())
    F, N, Rn = int(i[0]), int(i[1]), int(i[2])
    e = dict(kind="opt_in" if kind == 4 else "opt_out", F=F, N=N, R=Rn, calib=d[0:4].copy(), calib_zero=d[4:8].copy(), calib_value=d[8:12].copy())
    o = 12; ii = 3
    fr = []
    for k in range(F):
        fr.append(dict(shell_id=int(i[ii]), frameID=int(i[ii + 1]), flagged=bool(i[ii + 2]), evalPT=d[o:o + 7].copy(), w2c=d[o + 7:o + 14].copy(), state=d[o + 14:o + 24].copy(),
                       state_zero=d[o + 24:o + 34].copy(), exposure=float(d[o + 34]), frameEnergyTH=float(d[o + 35])))
        o += 36; ii += 3
    e["frames"] = fr
    n = 4 + 8 * F
    if kind == 4:
        e["HM"] = d[o:o + n * n].reshape(n, n).copy(); o += n * n
        e["bM"] = d[o:o + n].copy(); o += n
    else:
        e["rmse"] = float(d[o]); e["resInA"] = int(d[o + 1]); o += 2
    host = np.zeros(N, np.int32); prior = np.zeros(N, np.uint8); nres = np.zeros(N, np.int32); ngood = np.zeros(N, np.int32)
    pf = f.reshape(N, 22) if N else np.zeros((0, 22), np.float32)
    rp, rt, rs, rl, ra = [], [], [], [], []
    for p_ in range(N):
        host[p_], prior[p_], nres[p_], ngood[p_] = i[ii], i[ii + 1], i[ii + 2], i[ii + 3]; ii += 4
        for _ in range(nres[p_]):
            rp.append(p_); rt.append(int(i[ii])); rs.append(int(i[ii + 1])); rl.append(int(i[ii + 2])); ra.append(int(i[ii + 3])); ii += 4
    e.update(host=host, hasDepthPrior=prior, numGoodResiduals=ngood, u=pf[:, 0].copy(), v=pf[:, 1].copy(), idepth=pf[:, 2].copy(), idepth_zero=pf[:, 3].copy(),
             color=pf[:, 4:12].copy(), weights=pf[:, 12:20].copy(), idepth_hessian=pf[:, 20].copy(), maxRelBaseline=pf[:, 21].copy(),
             res_point=np.array(rp, np.int32), res_target=np.array(rt, np.int32), res_state=np.array(rs, np.int32), res_linearized=np.array(rl, np.int32),
             res_active=np.array(ra, np.int32))
    return e
