"""ctypes binding of oracle/_build/liboracle.so — ORACLE = TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
PARITY UNPINNED (see oracle/tracker_oracle.cpp header and DESIGN.md).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_f = C.POINTER(C.c_float)
c_d = C.POINTER(C.c_double)


def build(force=False):
    so = os.path.join(_HERE, "_build", "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".h")) or f == "Makefile"]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        L.orc_tracker_create.restype = C.c_void_p
        L.orc_tracker_create.argtypes = [C.c_int, C.c_int]
        L.orc_tracker_destroy.argtypes = [C.c_void_p]
        L.orc_tracker_levels.argtypes = [C.c_void_p]
        L.orc_tracker_make_k.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float]
        L.orc_tracker_get_k.argtypes = [C.c_void_p, C.c_int, c_f, c_f]
        L.orc_tracker_set_ref.argtypes = [C.c_void_p, C.POINTER(c_f), C.c_float, C.c_double, C.c_double, C.c_int, c_f, c_f, c_f, c_f]
        L.orc_tracker_set_new.argtypes = [C.c_void_p, C.POINTER(c_f), C.c_float]
        L.orc_tracker_pc_n.argtypes = [C.c_void_p, C.c_int]
        L.orc_tracker_get_pc.argtypes = [C.c_void_p, C.c_int, c_f, c_f, c_f, c_f]
        L.orc_tracker_get_idepth.argtypes = [C.c_void_p, C.c_int, c_f, c_f]
        L.orc_tracker_calc_res.argtypes = [C.c_void_p, C.c_int, c_d, c_d, C.c_float, c_d]
        L.orc_tracker_warped_n.argtypes = [C.c_void_p]
        L.orc_tracker_get_warped.argtypes = [C.c_void_p, c_f]
        L.orc_tracker_calc_gs.argtypes = [C.c_void_p, C.c_int, c_d, c_d, c_d]
        L.orc_tracker_track.argtypes = [C.c_void_p, c_d, c_d, C.c_int, c_d, C.c_float, C.c_float, c_d, c_d, c_d, c_d, C.POINTER(C.c_int)]
        L.orc_tracker_stats.argtypes = [C.c_void_p, C.POINTER(C.c_long)]
        L.orc_make_images.argtypes = [c_f, C.c_int, C.c_int, C.c_int, C.POINTER(c_f), C.POINTER(c_f)]
        L.orc_pyr_levels.argtypes = [C.c_int, C.c_int]
        for name, n_in in (("orc_se3_exp", 1), ("orc_se3_log", 1), ("orc_se3_inv", 1), ("orc_se3_adj", 1)):
            getattr(L, name).argtypes = [c_d, c_d]
        L.orc_se3_mul.argtypes = [c_d, c_d, c_d]
        L.orc_se3_matrix.argtypes = [c_d, c_d, c_d]
        L.orc_ldlt_solve.argtypes = [c_d, c_d, c_d, C.c_int]
    return _LIB


def _f(a):
    return a.ctypes.data_as(c_f)


def _d(a):
    return a.ctypes.data_as(c_d)


def pyr_levels(w, h):
    return lib().orc_pyr_levels(w, h)


def make_images(color, w, h):
    """FrameHessian::makeImages -> (list of [h_l,w_l,3] f32, list of [h_l,w_l] f32 absSquaredGrad)."""
    L = lib()
    levels = L.orc_pyr_levels(w, h)
    color = np.ascontiguousarray(color, dtype=np.float32).reshape(-1)
    dI = [np.zeros(((h >> l), (w >> l), 3), dtype=np.float32) for l in range(levels)]
    ab = [np.zeros(((h >> l), (w >> l)), dtype=np.float32) for l in range(levels)]
    dp = (c_f * levels)(*[_f(a) for a in dI])
    ap = (c_f * levels)(*[_f(a) for a in ab])
    L.orc_make_images(_f(color), w, h, levels, dp, ap)
    return dI, ab


def se3_exp(xi):
    out = np.zeros(7); lib().orc_se3_exp(_d(np.ascontiguousarray(xi, dtype=np.float64)), _d(out)); return out


def se3_log(p7):
    out = np.zeros(6); lib().orc_se3_log(_d(np.ascontiguousarray(p7, dtype=np.float64)), _d(out)); return out


def se3_mul(a, b):
    out = np.zeros(7)
    lib().orc_se3_mul(_d(np.ascontiguousarray(a, dtype=np.float64)), _d(np.ascontiguousarray(b, dtype=np.float64)), _d(out))
    return out


def se3_inv(a):
    out = np.zeros(7); lib().orc_se3_inv(_d(np.ascontiguousarray(a, dtype=np.float64)), _d(out)); return out


def se3_matrix(a):
    R = np.zeros(9); t = np.zeros(3)
    lib().orc_se3_matrix(_d(np.ascontiguousarray(a, dtype=np.float64)), _d(R), _d(t)); return R.reshape(3, 3), t


def se3_adj(a):
    A = np.zeros(36); lib().orc_se3_adj(_d(np.ascontiguousarray(a, dtype=np.float64)), _d(A)); return A.reshape(6, 6)


def ldlt_solve(A, rhs):
    A = np.ascontiguousarray(A, dtype=np.float64); rhs = np.ascontiguousarray(rhs, dtype=np.float64)
    x = np.zeros_like(rhs); lib().orc_ldlt_solve(_d(A), _d(rhs), _d(x), len(rhs)); return x


class Tracker:
    """Oracle mirror of CoarseTracker (CoarseTracker.h:46-129)."""

    def __init__(self, w, h):
        self.L = lib()
        self.w, self.h = w, h
        self.p = C.c_void_p(self.L.orc_tracker_create(w, h))
        self.levels = self.L.orc_tracker_levels(self.p)
        self._keep = {}

    def __del__(self):
        if getattr(self, "p", None):
            self.L.orc_tracker_destroy(self.p); self.p = None

    def make_k(self, K4):
        self.L.orc_tracker_make_k(self.p, float(K4[0]), float(K4[1]), float(K4[2]), float(K4[3]))

    def get_k(self, lvl):
        k = np.zeros(4, dtype=np.float32); ki = np.zeros(9, dtype=np.float32)
        self.L.orc_tracker_get_k(self.p, lvl, _f(k), _f(ki)); return k, ki.reshape(3, 3)

    def set_ref(self, dIp_ref, u, v, idepth, hdiF, exposure=1.0, aff=(0.0, 0.0)):
        self._keep["ref"] = [np.ascontiguousarray(a, dtype=np.float32) for a in dIp_ref]
        dp = (c_f * self.levels)(*[_f(a) for a in self._keep["ref"]])
        u = np.ascontiguousarray(u, dtype=np.float32); v = np.ascontiguousarray(v, dtype=np.float32)
        idepth = np.ascontiguousarray(idepth, dtype=np.float32); hdiF = np.ascontiguousarray(hdiF, dtype=np.float32)
        self.L.orc_tracker_set_ref(self.p, dp, exposure, aff[0], aff[1], len(u), _f(u), _f(v), _f(idepth), _f(hdiF))

    def set_new(self, dIp_new, exposure=1.0):
        self._keep["new"] = [np.ascontiguousarray(a, dtype=np.float32) for a in dIp_new]
        dp = (c_f * self.levels)(*[_f(a) for a in self._keep["new"]])
        self.L.orc_tracker_set_new(self.p, dp, exposure)

    def pc_n(self, lvl):
        return self.L.orc_tracker_pc_n(self.p, lvl)

    def get_pc(self, lvl):
        n = self.pc_n(lvl)
        out = [np.zeros(n, dtype=np.float32) for _ in range(4)]
        self.L.orc_tracker_get_pc(self.p, lvl, *[_f(a) for a in out]); return out

    def get_idepth(self, lvl):
        n = (self.w >> lvl) * (self.h >> lvl)
        a = np.zeros(n, dtype=np.float32); b = np.zeros(n, dtype=np.float32)
        self.L.orc_tracker_get_idepth(self.p, lvl, _f(a), _f(b)); return a, b

    def calc_res(self, lvl, pose7, aff, cutoff=20.0):
        rs = np.zeros(6)
        self.L.orc_tracker_calc_res(self.p, lvl, _d(np.ascontiguousarray(pose7, dtype=np.float64)),
                                    _d(np.ascontiguousarray(aff, dtype=np.float64)), cutoff, _d(rs))
        return rs

    def get_warped(self):
        n = self.L.orc_tracker_warped_n(self.p)
        out = np.zeros((8, n), dtype=np.float32)
        self.L.orc_tracker_get_warped(self.p, _f(out)); return out

    def calc_gs(self, lvl, aff):
        H = np.zeros(64); b = np.zeros(8)
        self.L.orc_tracker_calc_gs(self.p, lvl, _d(np.ascontiguousarray(aff, dtype=np.float64)), _d(H), _d(b))
        return H.reshape(8, 8), b

    def track(self, pose7, aff, coarsest=None, min_res=None, modeA=1e12, modeB=1e8):
        pose = np.array(pose7, dtype=np.float64); a = np.array(aff, dtype=np.float64)
        if coarsest is None:
            coarsest = self.levels - 1
        mr = np.full(5, np.nan) if min_res is None else np.array(min_res, dtype=np.float64)
        lr = np.zeros(5); fl = np.zeros(3); H = np.zeros(64); b = np.zeros(8); it = C.c_int(0)
        good = self.L.orc_tracker_track(self.p, _d(pose), _d(a), coarsest, _d(mr), modeA, modeB, _d(lr), _d(fl), _d(H), _d(b), C.byref(it))
        return dict(good=bool(good), pose7=pose, aff=a, lastResiduals=lr, flow=fl, H=H.reshape(8, 8), b=b, iterations=it.value)

    def stats(self):
        o = (C.c_long * 3)(); self.L.orc_tracker_stats(self.p, o); return list(o)
