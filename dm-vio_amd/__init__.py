"""dm-vio_amd — MI355X-native photometric direct alignment (DM-VIO hot path).

This Python module is only the test/bench harness side of the C ABI in include/dmvio_hip.h
(ctypes over dm-vio_amd/lib/libdmvio_hip.so).  The product is the shared library; there is NO
CPU fallback: if the HIP library is missing or no GPU is visible, every compute entry raises.

The directory name contains a hyphen (it mirrors the reference's name), so import it through
`__graft_entry__.load_package()` / tests/conftest.py, which register it as module `dmvio_amd`.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdmvio_hip.so")
INCLUDE_PATH = os.path.join(os.path.dirname(_HERE), "include", "dmvio_hip.h")

c_f = C.POINTER(C.c_float)
c_d = C.POINTER(C.c_double)
c_i = C.POINTER(C.c_int)

_lib = None


class HipLibraryError(RuntimeError):
    pass


def _sig(L):
    vp = C.c_void_p
    L.dmvio_hip_last_error.restype = C.c_char_p
    L.dmvio_hip_device_count.restype = C.c_int
    L.dmvio_hip_create.restype = vp
    L.dmvio_hip_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    L.dmvio_hip_destroy.argtypes = [vp]
    L.dmvio_hip_pyr_levels.argtypes = [vp]
    L.dmvio_hip_set_stream.argtypes = [vp, vp]
    L.dmvio_hip_synchronize.argtypes = [vp]
    L.dmvio_hip_frame_upload.argtypes = [vp, C.c_int, c_f]
    L.dmvio_hip_frame_from_device.argtypes = [vp, C.c_int, vp]
    L.dmvio_hip_frames_from_device_batch.argtypes = [vp, C.c_int, c_i, vp, C.c_size_t]
    L.dmvio_hip_frame_download.argtypes = [vp, C.c_int, C.c_int, c_f]
    L.dmvio_hip_tracker_create.restype = vp
    L.dmvio_hip_tracker_create.argtypes = [vp]
    L.dmvio_hip_tracker_destroy.argtypes = [vp]
    L.dmvio_hip_tracker_set_settings.argtypes = [vp, c_f]
    L.dmvio_hip_tracker_make_k.argtypes = [vp, c_f]
    L.dmvio_hip_tracker_set_ref.argtypes = [vp, C.c_int, C.c_float, C.c_double, C.c_double, C.c_int, c_f, c_f, c_f, c_f]
    L.dmvio_hip_tracker_pc_n.argtypes = [vp, C.c_int]
    L.dmvio_hip_tracker_get_pc.argtypes = [vp, C.c_int, c_f, c_f, c_f, c_f]
    L.dmvio_hip_tracker_eval.argtypes = [vp, C.c_int, C.c_int, C.c_float, c_d, c_d, C.c_float, c_d, c_d, c_d]
    L.dmvio_hip_tracker_track.argtypes = [vp, C.c_int, C.c_float, c_d, c_d, C.c_int, c_d, c_d, c_d, c_d, c_d, c_i]
    L.dmvio_hip_tracker_track_batch.argtypes = [vp, C.c_int, c_i, c_f, c_d, c_d, C.c_int, c_d, c_d, c_d, c_d, c_d, c_i, c_i]
    L.dmvio_hip_tracker_track_batch_stage.argtypes = [vp, C.c_int, c_i, c_f, c_d, c_d, C.c_int, c_d]
    L.dmvio_hip_tracker_track_batch_launch.argtypes = [vp]
    L.dmvio_hip_tracker_track_batch_fetch.argtypes = [vp, c_d, c_d, c_d, c_d, c_d, c_d, c_i, c_i]
    L.dmvio_hip_tracker_last_ticks.argtypes = [vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    L.dmvio_hip_tracker_last_work.argtypes = [vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]


def load_library():
    """dlopen libdmvio_hip.so.  Raises HipLibraryError when it has not been built — never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError("libdmvio_hip.so not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
        # torch wheels bundle their own libamdhip64; when torch shares the process (bench.py uses it for device
        # memory, streams and RCCL) it must be loaded FIRST so that both sides run on one HIP runtime — two
        # runtimes in one process leave the second one without devices ("No HIP GPUs are available").
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        _lib = C.CDLL(LIB_PATH)
        _sig(_lib)
    return _lib


def declared_symbols():
    """Entry points declared in include/dmvio_hip.h (parsed from the header text)."""
    import re
    txt = open(INCLUDE_PATH).read()
    return sorted(set(re.findall(r"\b(dmvio_hip_[a-z0-9_]+)\s*\(", txt)))


def _chk(L, r, what):
    if r is None or (isinstance(r, int) and r < 0):
        raise HipLibraryError("%s: %s" % (what, (L.dmvio_hip_last_error() or b"").decode()))
    return r


def _f(a):
    return a.ctypes.data_as(c_f)


def _d(a):
    return a.ctypes.data_as(c_d)


def _i(a):
    return a.ctypes.data_as(c_i)


class Context:
    """dmvio_hip_ctx: device, stream and the resident image pyramids (frame slots)."""

    def __init__(self, w, h, n_slots=16, device=0):
        self.L = load_library()
        if self.L.dmvio_hip_device_count() <= 0:
            raise HipLibraryError("no HIP device visible: the dm-vio_amd hot path has no CPU fallback")
        p = self.L.dmvio_hip_create(device, w, h, n_slots)
        if not p:
            raise HipLibraryError("dmvio_hip_create: " + (self.L.dmvio_hip_last_error() or b"").decode())
        self.p = C.c_void_p(p)
        self.w, self.h, self.n_slots = w, h, n_slots
        self.levels = self.L.dmvio_hip_pyr_levels(self.p)

    def close(self):
        if getattr(self, "p", None):
            self.L.dmvio_hip_destroy(self.p)
            self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        _chk(self.L, self.L.dmvio_hip_set_stream(self.p, C.c_void_p(stream_ptr)), "set_stream")

    def synchronize(self):
        _chk(self.L, self.L.dmvio_hip_synchronize(self.p), "synchronize")

    def frame_upload(self, slot, img):
        img = np.ascontiguousarray(img, dtype=np.float32)
        assert img.size == self.w * self.h
        _chk(self.L, self.L.dmvio_hip_frame_upload(self.p, slot, _f(img)), "frame_upload")

    def frame_from_device(self, slot, dev_ptr):
        _chk(self.L, self.L.dmvio_hip_frame_from_device(self.p, slot, C.c_void_p(dev_ptr)), "frame_from_device")

    def frames_from_device_batch(self, slots, dev_ptr, stride_bytes):
        slots = np.ascontiguousarray(slots, dtype=np.int32)
        _chk(self.L, self.L.dmvio_hip_frames_from_device_batch(self.p, len(slots), _i(slots), C.c_void_p(dev_ptr), stride_bytes), "frames_from_device_batch")

    def frame_download(self, slot, lvl):
        out = np.zeros(((self.h >> lvl), (self.w >> lvl), 3), dtype=np.float32)
        _chk(self.L, self.L.dmvio_hip_frame_download(self.p, slot, lvl, _f(out)), "frame_download")
        return out


class CoarseTrackerHip:
    """Mirror of the reference's CoarseTracker public surface (CoarseTracker.h:46-129) over the C ABI."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.L = ctx.L
        p = self.L.dmvio_hip_tracker_create(ctx.p)
        if not p:
            raise HipLibraryError("tracker_create: " + (self.L.dmvio_hip_last_error() or b"").decode())
        self.p = C.c_void_p(p)
        self.lastResiduals = np.full(5, np.nan)
        self.lastFlowIndicators = np.full(3, 1000.0)

    def close(self):
        if getattr(self, "p", None):
            self.L.dmvio_hip_tracker_destroy(self.p)
            self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_settings(self, huberTH=9.0, coarseCutoffTH=20.0, affineOptModeA=1e12, affineOptModeB=1e8):
        s = np.array([huberTH, coarseCutoffTH, affineOptModeA, affineOptModeB], dtype=np.float32)
        _chk(self.L, self.L.dmvio_hip_tracker_set_settings(self.p, _f(s)), "set_settings")

    def makeK(self, K4):
        k = np.ascontiguousarray(K4, dtype=np.float32)
        _chk(self.L, self.L.dmvio_hip_tracker_make_k(self.p, _f(k)), "makeK")

    def setCoarseTrackingRef(self, ref_slot, u, v, idepth, hdiF, ref_exposure=1.0, ref_aff=(0.0, 0.0)):
        u = np.ascontiguousarray(u, dtype=np.float32); v = np.ascontiguousarray(v, dtype=np.float32)
        idepth = np.ascontiguousarray(idepth, dtype=np.float32); hdiF = np.ascontiguousarray(hdiF, dtype=np.float32)
        _chk(self.L, self.L.dmvio_hip_tracker_set_ref(self.p, ref_slot, ref_exposure, ref_aff[0], ref_aff[1], len(u),
                                                      _f(u), _f(v), _f(idepth), _f(hdiF)), "setCoarseTrackingRef")

    def pc_n(self, lvl):
        return _chk(self.L, self.L.dmvio_hip_tracker_pc_n(self.p, lvl), "pc_n")

    def get_pc(self, lvl):
        n = self.pc_n(lvl)
        out = [np.zeros(n, dtype=np.float32) for _ in range(4)]
        _chk(self.L, self.L.dmvio_hip_tracker_get_pc(self.p, lvl, *[_f(a) for a in out]), "get_pc")
        return out

    def eval(self, lvl, new_slot, pose7, aff, cutoffTH=20.0, new_exposure=1.0):
        """calcRes + calcGSSSE at refToNew=pose7 -> (res6, H[8,8], b[8])."""
        pose7 = np.ascontiguousarray(pose7, dtype=np.float64); aff = np.ascontiguousarray(aff, dtype=np.float64)
        rs = np.zeros(6); H = np.zeros(64); b = np.zeros(8)
        _chk(self.L, self.L.dmvio_hip_tracker_eval(self.p, lvl, new_slot, new_exposure, _d(pose7), _d(aff), cutoffTH, _d(rs), _d(H), _d(b)), "eval")
        return rs, H.reshape(8, 8), b

    def trackNewestCoarse(self, new_slot, pose7, aff, coarsestLvl=None, minResForAbort=None, new_exposure=1.0):
        r = self.track_batch([new_slot], [pose7], [aff], coarsestLvl, None if minResForAbort is None else [minResForAbort], [new_exposure])
        self.lastResiduals = r["lastResiduals"][0]
        self.lastFlowIndicators = r["flow"][0]
        return dict(good=bool(r["good"][0]), pose7=r["pose7"][0], aff=r["aff"][0], lastResiduals=r["lastResiduals"][0],
                    flow=r["flow"][0], H=r["H"][0], b=r["b"][0], iterations=int(r["iterations"][0]))

    def _batch_inputs(self, slots, poses, affs, coarsestLvl, minRes, exposures):
        B = len(slots)
        slots = np.ascontiguousarray(slots, dtype=np.int32)
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(B, 7).copy()
        affs = np.ascontiguousarray(affs, dtype=np.float64).reshape(B, 2).copy()
        exposures = np.ones(B, dtype=np.float32) if exposures is None else np.ascontiguousarray(exposures, dtype=np.float32)
        mr = np.full((B, 5), np.nan) if minRes is None else np.ascontiguousarray(minRes, dtype=np.float64).reshape(B, 5)
        if coarsestLvl is None:
            coarsestLvl = self.ctx.levels - 1
        return B, slots, poses, affs, exposures, mr, coarsestLvl

    def track_batch(self, slots, poses, affs, coarsestLvl=None, minRes=None, exposures=None):
        B, slots, poses, affs, exposures, mr, coarsestLvl = self._batch_inputs(slots, poses, affs, coarsestLvl, minRes, exposures)
        lr = np.zeros((B, 5)); fl = np.zeros((B, 3)); H = np.zeros((B, 64)); b = np.zeros((B, 8))
        good = np.zeros(B, dtype=np.int32); its = np.zeros(B, dtype=np.int32)
        _chk(self.L, self.L.dmvio_hip_tracker_track_batch(self.p, B, _i(slots), _f(exposures), _d(poses), _d(affs), coarsestLvl, _d(mr),
                                                          _d(lr), _d(fl), _d(H), _d(b), _i(good), _i(its)), "track_batch")
        return dict(good=good, pose7=poses, aff=affs, lastResiduals=lr, flow=fl, H=H.reshape(B, 8, 8), b=b, iterations=its)

    def stage(self, slots, poses, affs, coarsestLvl=None, minRes=None, exposures=None):
        B, slots, poses, affs, exposures, mr, coarsestLvl = self._batch_inputs(slots, poses, affs, coarsestLvl, minRes, exposures)
        self._B = B
        _chk(self.L, self.L.dmvio_hip_tracker_track_batch_stage(self.p, B, _i(slots), _f(exposures), _d(poses), _d(affs), coarsestLvl, _d(mr)), "stage")

    def launch(self):
        _chk(self.L, self.L.dmvio_hip_tracker_track_batch_launch(self.p), "launch")

    def fetch(self):
        B = self._B
        poses = np.zeros((B, 7)); affs = np.zeros((B, 2))
        lr = np.zeros((B, 5)); fl = np.zeros((B, 3)); H = np.zeros((B, 64)); b = np.zeros((B, 8))
        good = np.zeros(B, dtype=np.int32); its = np.zeros(B, dtype=np.int32)
        _chk(self.L, self.L.dmvio_hip_tracker_track_batch_fetch(self.p, _d(poses), _d(affs), _d(lr), _d(fl), _d(H), _d(b), _i(good), _i(its)), "fetch")
        return dict(good=good, pose7=poses, aff=affs, lastResiduals=lr, flow=fl, H=H.reshape(B, 8, 8), b=b, iterations=its)

    def last_ticks(self):
        a = C.c_longlong(0); b = C.c_longlong(0)
        _chk(self.L, self.L.dmvio_hip_tracker_last_ticks(self.p, C.byref(a), C.byref(b)), "last_ticks")
        return a.value, b.value

    def last_work(self):
        a = C.c_longlong(0); b = C.c_longlong(0)
        _chk(self.L, self.L.dmvio_hip_tracker_last_work(self.p, C.byref(a), C.byref(b)), "last_work")
        return a.value, b.value
