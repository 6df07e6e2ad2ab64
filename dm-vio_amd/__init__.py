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



_ALLREDUCE_F64 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_size_t)
_ALLGATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)


class CommCallbacks(C.Structure):
    """dmvio_hip_comm_callbacks (include/dmvio_hip.h): a caller-provided transport for the sharded BA iteration."""
    _fields_ = [("user", C.c_void_p), ("allreduce_sum_f64", _ALLREDUCE_F64), ("allgather", _ALLGATHER)]


class BAFrameView(C.Structure):
    """dmvio_hip_ba_frame_view (include/dmvio_hip.h): what the BA hooks read of a keyframe."""
    _fields_ = [("frameID", C.c_int), ("index", C.c_int), ("PRE_worldToCam7", C.c_double * 7), ("worldToCam_evalPT7", C.c_double * 7),
                ("state10", C.c_double * 10), ("state_zero10", C.c_double * 10)]


_BA_COMPUTE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, c_d, c_d, C.c_double, c_d, C.c_int, C.POINTER(BAFrameView), c_d, c_d)
_BA_ACCEPT = C.CFUNCTYPE(None, C.c_void_p, C.c_double)
_BA_ENERGY = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_int)
_BA_VALUES = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(BAFrameView), c_d)
_BA_WEIGHT = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_double, C.c_double, C.c_int)
_BA_BREAK = C.CFUNCTYPE(C.c_int, C.c_void_p)


class BACallbacks(C.Structure):
    """dmvio_hip_ba_callbacks: the members of dmvio::BAGTSAMIntegration the BA loop calls."""
    _fields_ = [("user", C.c_void_p), ("computeBAUpdate", _BA_COMPUTE), ("acceptBAUpdate", _BA_ACCEPT), ("getBAEnergy", _BA_ENERGY), ("updateBAValues", _BA_VALUES),
                ("updateDynamicWeight", _BA_WEIGHT), ("canBreak", _BA_BREAK), ("postOptimization", _BA_VALUES)]


class BAVioOptions(C.Structure):
    _fields_ = [("coarseTrackingWasGood", C.c_int), ("updateDynamicWeightDuringOptimization", C.c_int), ("minOptIterations", C.c_int), ("resInA_at_entry", C.c_int),
                ("HMForGTSAM", c_d), ("bMForGTSAM", c_d)]


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
    L.dmvio_hip_frames_attach_device_batch.argtypes = [vp, C.c_int, c_i, vp, C.c_size_t]
    L.dmvio_hip_frame_download.argtypes = [vp, C.c_int, C.c_int, c_f]
    L.dmvio_hip_frame_abs_squared_grad.argtypes = [vp, C.c_int, C.c_int, c_f, C.POINTER(c_f)]
    L.dmvio_hip_frame_mark_unclean.argtypes = [vp, C.c_int]
    L.dmvio_hip_selftest_divide.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    L.dmvio_hip_write_result_txt.argtypes = [C.c_char_p, C.c_int, vp, vp, vp, vp, vp, vp]
    L.dmvio_hip_tracker_create.restype = vp
    L.dmvio_hip_tracker_create.argtypes = [vp]
    L.dmvio_hip_tracker_destroy.argtypes = [vp]
    L.dmvio_hip_tracker_set_settings.argtypes = [vp, c_f]
    L.dmvio_hip_tracker_make_k.argtypes = [vp, c_f]
    L.dmvio_hip_tracker_set_ref.argtypes = [vp, C.c_int, C.c_float, C.c_double, C.c_double, C.c_int, c_f, c_f, c_f, c_f]
    L.dmvio_hip_tracker_pc_n.argtypes = [vp, C.c_int]
    L.dmvio_hip_tracker_get_pc.argtypes = [vp, C.c_int, c_f, c_f, c_f, c_f]
    L.dmvio_hip_tracker_get_idepth_map.argtypes = [vp, C.c_int, c_f, c_f]
    L.dmvio_hip_tracker_eval.argtypes = [vp, C.c_int, C.c_int, C.c_float, c_d, c_d, C.c_float, c_d, c_d, c_d]
    L.dmvio_hip_tracker_track.argtypes = [vp, C.c_int, C.c_float, c_d, c_d, C.c_int, c_d, c_d, c_d, c_d, c_d, c_i]
    L.dmvio_hip_tracker_track_batch.argtypes = [vp, C.c_int, c_i, c_f, c_d, c_d, C.c_int, c_d, c_d, c_d, c_d, c_d, c_i, c_i]
    L.dmvio_hip_tracker_track_batch_stage.argtypes = [vp, C.c_int, c_i, c_f, c_d, c_d, C.c_int, c_d]
    L.dmvio_hip_tracker_track_batch_launch.argtypes = [vp]
    L.dmvio_hip_tracker_track_batch_fetch_begin.argtypes = [vp]
    L.dmvio_hip_tracker_track_batch_fetch.argtypes = [vp, c_d, c_d, c_d, c_d, c_d, c_d, c_i, c_i]
    L.dmvio_hip_make_track_hypotheses.argtypes = [c_d, c_d, c_d, c_d, C.c_int]
    L.dmvio_hip_tracker_track_new_coarse.argtypes = [vp, C.c_int, C.c_float, C.c_int, c_d, c_d, c_d, C.c_double, c_d, c_d, c_d, c_i, c_i, c_i]
    L.dmvio_hip_tracker_last_ticks.argtypes = [vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    L.dmvio_hip_tracker_set_single_frame_mode.argtypes = [vp, C.c_int]
    L.dmvio_hip_tracker_last_work.argtypes = [vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    L.dmvio_hip_tracker_last_launch.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    c_u8 = C.POINTER(C.c_ubyte)
    c_u8 = C.POINTER(C.c_ubyte)
    L.dmvio_hip_initializer_create.restype = vp
    L.dmvio_hip_initializer_create.argtypes = [vp, C.c_int]
    L.dmvio_hip_initializer_destroy.argtypes = [vp]
    L.dmvio_hip_initializer_set_points.argtypes = [vp, C.c_int, c_f, c_f, c_f, c_u8, c_f, c_f]
    L.dmvio_hip_initializer_calc_res_and_gs.argtypes = [vp, C.c_int, C.c_int, C.c_int, c_d, c_f, c_d, c_d, c_f, C.c_float, C.c_float, C.c_float, C.c_double, C.c_double,
                                                        c_f, c_f, c_f, c_f, c_f, c_f, c_u8, c_f, c_f, c_f]
    L.dmvio_hip_undistorter_create.restype = vp
    L.dmvio_hip_undistorter_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, c_f, c_f, c_f, c_f]
    L.dmvio_hip_undistorter_destroy.argtypes = [vp]
    L.dmvio_hip_frame_upload_raw.argtypes = [vp, vp, C.c_int, C.c_void_p, C.c_float, c_f]
    L.dmvio_hip_frames_from_raw_device_batch.argtypes = [vp, vp, C.c_int, c_i, vp, C.c_size_t, C.c_float]
    L.dmvio_hip_immature_create.restype = vp
    L.dmvio_hip_immature_create.argtypes = [vp, C.c_int]
    L.dmvio_hip_immature_destroy.argtypes = [vp]
    L.dmvio_hip_immature_clear.argtypes = [vp]
    L.dmvio_hip_immature_count.argtypes = [vp]
    L.dmvio_hip_immature_add_points.argtypes = [vp, C.c_int, C.c_int, C.c_int, c_i, c_i]
    L.dmvio_hip_immature_get_static.argtypes = [vp, c_f, c_f, c_i, c_f, c_f, c_f, c_f]
    L.dmvio_hip_immature_get_state.argtypes = [vp, c_f, c_f, c_f, c_f, c_f, c_i]
    L.dmvio_hip_immature_set_state.argtypes = [vp, c_f, c_f, c_f, c_i]
    L.dmvio_hip_immature_trace.argtypes = [vp, C.c_int, C.c_int, c_f, c_f, c_f]
    L.dmvio_hip_immature_optimize.argtypes = [vp, C.c_int, c_i, c_d, c_d, c_f, c_d, C.c_char_p, C.c_int, c_i, c_f, c_i]
    L.dmvio_hip_trace_new_coarse.argtypes = [vp, C.c_int, c_d, c_d, C.c_float, C.c_int, c_d, c_d, c_f, c_d, c_i]
    L.dmvio_hip_ba_set_frame_state.argtypes = [vp, C.c_int, c_d]
    L.dmvio_hip_ba_marginalize_frame.argtypes = [vp, C.c_int, c_d, c_d]
    L.dmvio_hip_ba_get_marg_prior.argtypes = [vp, c_d, c_d]
    L.dmvio_hip_ba_marginalize_points.argtypes = [vp, C.c_char_p, C.POINTER(C.c_ubyte), c_d, c_d, c_i, C.c_int]
    L.dmvio_hip_ba_set_stream.argtypes = [vp, vp]
    L.dmvio_hip_ba_create.restype = vp
    L.dmvio_hip_ba_create.argtypes = [vp]
    L.dmvio_hip_ba_destroy.argtypes = [vp]
    L.dmvio_hip_ba_set_window.argtypes = [vp, C.c_int, c_i, c_d, c_d, c_f, c_i, c_d]
    L.dmvio_hip_ba_set_marg_prior.argtypes = [vp, c_d, c_d]
    L.dmvio_hip_ba_set_graph.argtypes = [vp, C.c_int, c_i, c_f, c_f, c_f, c_f, c_f, c_u8, C.c_int, c_i, c_i]
    L.dmvio_hip_ba_activate_all.argtypes = [vp]
    L.dmvio_hip_ba_linearize.argtypes = [vp, C.c_int, c_d]
    L.dmvio_hip_ba_apply.argtypes = [vp]
    L.dmvio_hip_ba_get_res_state.argtypes = [vp, c_u8, c_f, c_f, c_u8, c_f]
    L.dmvio_hip_ba_get_jacobians.argtypes = [vp, c_f]
    L.dmvio_hip_ba_get_frame_energy_th.argtypes = [vp, c_f]
    L.dmvio_hip_ba_accumulate.argtypes = [vp, c_d, c_d, c_d, c_d, c_i]
    L.dmvio_hip_ba_get_point_acc.argtypes = [vp, c_f, c_f, c_f, c_f, c_f]
    L.dmvio_hip_ba_solve.argtypes = [vp, C.c_int, C.c_double, c_d]
    L.dmvio_hip_ba_resubstitute.argtypes = [vp, c_d]
    L.dmvio_hip_ba_get_points.argtypes = [vp, c_f, c_f]
    L.dmvio_hip_ba_get_frame.argtypes = [vp, C.c_int, c_d, c_d, c_d]
    L.dmvio_hip_ba_get_calib.argtypes = [vp, c_d]
    L.dmvio_hip_ba_gn_iteration.argtypes = [vp, C.c_int, c_d, c_d, c_i]
    L.dmvio_hip_ba_optimize.argtypes = [vp, C.c_int, c_f, c_d, c_i, c_d]
    L.dmvio_hip_ba_backup.argtypes = [vp]
    L.dmvio_hip_ba_solve_system.argtypes = [vp, C.c_int, C.c_double, c_d, c_d, c_d, c_d, c_d]
    L.dmvio_hip_ba_step.argtypes = [vp, C.c_float, c_f]
    L.dmvio_hip_ba_restore.argtypes = [vp]
    L.dmvio_hip_ba_linearize_local.argtypes = [vp, C.c_int, c_d, c_f, c_i]
    L.dmvio_hip_ba_set_new_frame_energy_th.argtypes = [vp, C.c_float]
    L.dmvio_hip_ba_energy_terms.argtypes = [vp, c_d, c_d]
    L.dmvio_hip_ba_set_comm.argtypes = [vp, vp, C.c_int, C.c_int]
    L.dmvio_hip_ba_set_comm_callbacks.argtypes = [vp, C.POINTER(CommCallbacks), C.c_int, C.c_int]
    L.dmvio_hip_ba_optimize_vio.argtypes = [vp, C.c_int, C.POINTER(BACallbacks), C.POINTER(BAVioOptions), C.POINTER(C.c_float), c_d, c_i, c_d]
    L.dmvio_hip_ba_solve_ldlt.argtypes = [C.c_int, c_d, c_d, c_d]
    L.dmvio_hip_ba_get_point_hessian.argtypes = [vp, c_f]
    L.dmvio_hip_comm_unique_id.argtypes = [c_u8]
    L.dmvio_hip_comm_init_rank.argtypes = [vp, c_u8, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.dmvio_hip_comm_destroy.argtypes = [vp]


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


def _err(L):
    return (L.dmvio_hip_last_error() or b"").decode()


def _f(a):
    return a.ctypes.data_as(c_f)


def _d(a):
    return a.ctypes.data_as(c_d)


def _i(a):
    return a.ctypes.data_as(c_i)


def write_result_txt(path, timestamps, camToWorld7, pose_valid=None, tracking_ref=None, camToTrackingRef7=None, firstPose7=(0, 0, 0, 0, 0, 0, 1.0)):
    """FullSystem::printResult (FullSystem.cpp:256-298) through the library: host-only, no device needed."""
    L = load_library()
    ts = np.ascontiguousarray(timestamps, dtype=np.float64); P = np.ascontiguousarray(camToWorld7, dtype=np.float64).reshape(-1, 7)
    pv = None if pose_valid is None else np.ascontiguousarray(pose_valid, dtype=np.uint8)
    tr = None if tracking_ref is None else np.ascontiguousarray(tracking_ref, dtype=np.int32)
    cr = None if camToTrackingRef7 is None else np.ascontiguousarray(camToTrackingRef7, dtype=np.float64)
    fp = np.ascontiguousarray(firstPose7, dtype=np.float64)
    ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    _chk(L, L.dmvio_hip_write_result_txt(str(path).encode(), len(ts), ptr(ts), ptr(P), ptr(pv), ptr(tr), ptr(cr), ptr(fp)), "write_result_txt")


def make_track_hypotheses(slast_c2w, sprelast_c2w, lastF_c2w):
    """lastF_2_fh_tries of FullSystem::trackNewCoarse (host-side pose algebra of the library)."""
    L = load_library()
    out = np.zeros((31, 7))
    n = L.dmvio_hip_make_track_hypotheses(_d(np.ascontiguousarray(slast_c2w, dtype=np.float64)), _d(np.ascontiguousarray(sprelast_c2w, dtype=np.float64)),
                                          _d(np.ascontiguousarray(lastF_c2w, dtype=np.float64)), _d(out), 31)
    return out[:n]


def coarse_update_visual(H, b, extrapFac, lam, pose7_cur, settings=None):
    """dmvio_hip_coarse_update_visual — the host implementation of the visual-only LM step (CoarseTracker.cpp:639-682), no handle needed -> (pose7_new, incA, incB, incNorm).
    settings = (huberTH, coarseCutoffTH, affineOptModeA, affineOptModeB) or None for the reference's defaults."""
    L = load_library()
    fn = L.dmvio_hip_coarse_update_visual
    fn.argtypes = [c_f, c_d, c_d, C.c_float, C.c_float, c_d, c_d, c_d, c_d, c_d]; fn.restype = C.c_int
    st = None if settings is None else _f(np.ascontiguousarray(settings, dtype=np.float32))
    pn = np.zeros(7); ia = np.zeros(1); ib = np.zeros(1); nn = np.zeros(1)
    _chk(L, fn(st, _d(np.ascontiguousarray(H, dtype=np.float64).reshape(-1)), _d(np.ascontiguousarray(b, dtype=np.float64)), extrapFac, lam,
               _d(np.ascontiguousarray(pose7_cur, dtype=np.float64)), _d(pn), _d(ia), _d(ib), _d(nn)), "coarse_update_visual")
    return pn, float(ia[0]), float(ib[0]), float(nn[0])


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

    def frames_attach_device_batch(self, slots, dev_ptr, stride_bytes):
        slots = np.ascontiguousarray(slots, dtype=np.int32)
        _chk(self.L, self.L.dmvio_hip_frames_attach_device_batch(self.p, len(slots), _i(slots), C.c_void_p(dev_ptr), stride_bytes), "frames_attach_device_batch")

    def selftest_divide(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.float32); b = np.ascontiguousarray(b, dtype=np.float32)
        qs = np.zeros_like(a); qi = np.zeros_like(a)
        _chk(self.L, self.L.dmvio_hip_selftest_divide(self.p, a.size, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                                                      qs.ctypes.data_as(C.c_void_p), qi.ctypes.data_as(C.c_void_p)), "selftest_divide")
        return qs, qi

    def set_build_stream(self, stream_ptr):
        """dmvio_hip_set_build_stream: the batched pyramid builds on a stream of their own (0 / None: the context's stream); the caller orders the streams with events."""
        fn = self.L.dmvio_hip_set_build_stream; fn.argtypes = [C.c_void_p, C.c_void_p]; fn.restype = C.c_int
        _chk(self.L, fn(self.p, C.c_void_p(stream_ptr or 0)), "set_build_stream")

    def frame_is_clean(self, slot):
        self.L.dmvio_hip_frame_is_clean.argtypes = [C.c_void_p, C.c_int]; self.L.dmvio_hip_frame_is_clean.restype = C.c_int
        return bool(_chk(self.L, self.L.dmvio_hip_frame_is_clean(self.p, int(slot)), "frame_is_clean"))

    def frame_mark_unclean(self, slot):
        _chk(self.L, self.L.dmvio_hip_frame_mark_unclean(self.p, slot), "frame_mark_unclean")

    def frame_download(self, slot, lvl):
        out = np.zeros(((self.h >> lvl), (self.w >> lvl), 3), dtype=np.float32)
        _chk(self.L, self.L.dmvio_hip_frame_download(self.p, slot, lvl, _f(out)), "frame_download")
        return out

    def abs_squared_grad(self, slot, n_levels=3, B=None):
        """FrameHessian::absSquaredGrad[0..n_levels-1] of a resident frame (the pixel selector's input); B = CalibHessian::B (256 floats) or None."""
        outs = [np.zeros(((self.h >> l), (self.w >> l)), dtype=np.float32) for l in range(n_levels)]
        ptrs = (C.POINTER(C.c_float) * n_levels)(*[_f(o) for o in outs])
        Bp = None
        if B is not None:
            Bf = np.ascontiguousarray(B, dtype=np.float32)
            if Bf.size != 256:
                raise ValueError("B must hold 256 floats")
            Bp = _f(Bf)
        _chk(self.L, self.L.dmvio_hip_frame_abs_squared_grad(self.p, slot, n_levels, Bp, ptrs), "frame_abs_squared_grad")
        return outs


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

    def get_idepth_map(self, lvl):
        """CoarseTracker::idepth[lvl], weightSums[lvl] after makeCoarseDepthL0 (what debugPlotIDepthMap reads)."""
        n = (self.ctx.w >> lvl) * (self.ctx.h >> lvl)
        a = np.zeros(n, dtype=np.float32); b = np.zeros(n, dtype=np.float32)
        _chk(self.L, self.L.dmvio_hip_tracker_get_idepth_map(self.p, lvl, _f(a), _f(b)), "get_idepth_map")
        return a, b

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

    def set_launch_shape(self, eval_blocks=0, lm_threads=0, lm_waves=0, lm_cluster=0):
        """Measurement knobs (0 = the library's own choice): dmvio_hip_tracker_set_launch_shape."""
        fn = self.L.dmvio_hip_tracker_set_launch_shape; fn.argtypes = [C.c_void_p] + [C.c_int] * 4; fn.restype = C.c_int
        _chk(self.L, fn(self.p, int(eval_blocks), int(lm_threads), int(lm_waves), int(lm_cluster)), "tracker_set_launch_shape")

    def set_batch_kernel(self, mode):
        """0 = four wavefronts per problem (default), 1 = two problems per workgroup (dmvio_hip_tracker_set_batch_kernel)"""
        fn = self.L.dmvio_hip_tracker_set_batch_kernel; fn.argtypes = [C.c_void_p, C.c_int]; fn.restype = C.c_int
        _chk(self.L, fn(self.p, int(mode)), "tracker_set_batch_kernel")

    def set_template_order(self, row_major):
        """storage order of the template from the next setCoarseTrackingRef on (dmvio_hip_tracker_set_template_order)"""
        fn = self.L.dmvio_hip_tracker_set_template_order; fn.argtypes = [C.c_void_p, C.c_int]; fn.restype = C.c_int
        _chk(self.L, fn(self.p, 1 if row_major else 0), "tracker_set_template_order")

    def set_eval_server(self, on=True):
        fn = self.L.dmvio_hip_tracker_set_eval_server; fn.argtypes = [C.c_void_p, C.c_int]; fn.restype = C.c_int
        _chk(self.L, fn(self.p, 1 if on else 0), "tracker_set_eval_server")

    def set_single_frame_mode(self, host_lm=True):
        """One alignment problem per call: host LM against the evaluation server (default) or the device-resident LM."""
        _chk(self.L, self.L.dmvio_hip_tracker_set_single_frame_mode(self.p, 1 if host_lm else 0), "tracker_set_single_frame_mode")

    def trackNewestCoarseVIO(self, new_slot, pose7, aff, coarsestLvl=None, minResForAbort=None, new_exposure=1.0, update=None, accept=None, visual=None):
        """trackNewestCoarse with the LM step handed to the host (the reference's setting_useIMU branch, CoarseTracker.cpp:612-637).
        update(H[8,8], b[8], extrapFac, lambda, pose7_cur, aff_cur) -> (pose7_new, incA, incB, incNorm) plays computeCoarseUpdate; None = the
        library's visual-only step.  accept() / visual(H, b, good) play acceptCoarseUpdate / addVisualToCoarseGraph."""
        pose = np.array(pose7, dtype=np.float64); a = np.array(aff, dtype=np.float64)
        if coarsestLvl is None:
            coarsestLvl = self.ctx.levels - 1
        mr = np.full(5, np.nan) if minResForAbort is None else np.ascontiguousarray(minResForAbort, dtype=np.float64)
        UPD = C.CFUNCTYPE(C.c_int, C.c_void_p, c_d, c_d, C.c_float, C.c_float, c_d, c_d, c_d, c_d, c_d, c_d)
        ACC = C.CFUNCTYPE(None, C.c_void_p)
        VIS = C.CFUNCTYPE(None, C.c_void_p, c_d, c_d, C.c_int)

        class CB(C.Structure):
            _fields_ = [("user", C.c_void_p), ("update", UPD), ("accept", ACC), ("visual", VIS)]

        def _upd(user, H, b, extrapFac, lam, pcur, acur, pnew, incA, incB, incNorm):
            try:
                p, ia, ib, nrm = update(np.array(H[:64]).reshape(8, 8), np.array(b[:8]), extrapFac, lam, np.array(pcur[:7]), np.array(acur[:2]))
                for i in range(7):
                    pnew[i] = float(p[i])
                incA[0] = float(ia); incB[0] = float(ib); incNorm[0] = float(nrm)
                return 0
            except Exception:   # an exception must not cross the C frame
                import traceback; traceback.print_exc()
                return 1

        def _vis(user, H, b, good):
            visual(np.array(H[:64]).reshape(8, 8), np.array(b[:8]), bool(good))

        cb = CB(None, UPD(_upd) if update else UPD(), ACC(lambda user: accept()) if accept else ACC(), VIS(_vis) if visual else VIS())
        lr = np.zeros(5); fl = np.zeros(3); H = np.zeros(64); b = np.zeros(8); good = C.c_int(0); ne = C.c_int(0)
        fn = self.L.dmvio_hip_tracker_track_vio
        fn.argtypes = [C.c_void_p, C.c_int, C.c_float, c_d, c_d, C.c_int, c_d, C.POINTER(CB), c_d, c_d, c_d, c_d, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        fn.restype = C.c_int
        _chk(self.L, fn(self.p, new_slot, new_exposure, _d(pose), _d(a), coarsestLvl, _d(mr), C.byref(cb), _d(lr), _d(fl), _d(H), _d(b), C.byref(good), C.byref(ne)),
             "track_vio")
        return dict(good=bool(good.value), pose7=pose, aff=a, lastResiduals=lr, flow=fl, H=H.reshape(8, 8), b=b, n_evals=ne.value)

    def coarse_update_visual(self, H, b, extrapFac, lam, pose7_cur, settings=None):
        """The library's host implementation of the visual-only LM step (CoarseTracker.cpp:639-682) -> (pose7_new, incA, incB, incNorm)."""
        fn = self.L.dmvio_hip_coarse_update_visual
        fn.argtypes = [c_f, c_d, c_d, C.c_float, C.c_float, c_d, c_d, c_d, c_d, c_d]; fn.restype = C.c_int
        st = None if settings is None else _f(np.ascontiguousarray(settings, dtype=np.float32))
        pn = np.zeros(7); ia = np.zeros(1); ib = np.zeros(1); nn = np.zeros(1)
        _chk(self.L, fn(st, _d(np.ascontiguousarray(H, dtype=np.float64).reshape(-1)), _d(np.ascontiguousarray(b, dtype=np.float64)), extrapFac, lam,
                        _d(np.ascontiguousarray(pose7_cur, dtype=np.float64)), _d(pn), _d(ia), _d(ib), _d(nn)), "coarse_update_visual")
        return pn, ia[0], ib[0], nn[0]

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

    def fetch_begin(self):
        _chk(self.L, self.L.dmvio_hip_tracker_track_batch_fetch_begin(self.p), "fetch_begin")

    def fetch(self):
        B = self._B
        poses = np.zeros((B, 7)); affs = np.zeros((B, 2))
        lr = np.zeros((B, 5)); fl = np.zeros((B, 3)); H = np.zeros((B, 64)); b = np.zeros((B, 8))
        good = np.zeros(B, dtype=np.int32); its = np.zeros(B, dtype=np.int32)
        _chk(self.L, self.L.dmvio_hip_tracker_track_batch_fetch(self.p, _d(poses), _d(affs), _d(lr), _d(fl), _d(H), _d(b), _i(good), _i(its)), "fetch")
        return dict(good=good, pose7=poses, aff=affs, lastResiduals=lr, flow=fl, H=H.reshape(B, 8, 8), b=b, iterations=its)

    def trackNewCoarse(self, new_slot, tries7, aff_last=(0.0, 0.0), lastCoarseRMSE=None, reTrackThreshold=1.5, new_exposure=1.0):
        """The try loop of FullSystem::trackNewCoarse over the hypothesis list tries7 [n,7]."""
        tries = np.ascontiguousarray(tries7, dtype=np.float64).reshape(-1, 7)
        rm = np.full(5, 100.0) if lastCoarseRMSE is None else np.array(lastCoarseRMSE, dtype=np.float64)
        pose = np.zeros(7); aff = np.zeros(2); flow = np.zeros(3); w = C.c_int(0); used = C.c_int(0); good = C.c_int(0)
        al = np.array(aff_last, dtype=np.float64)
        _chk(self.L, self.L.dmvio_hip_tracker_track_new_coarse(self.p, new_slot, new_exposure, len(tries), _d(tries), _d(al), _d(rm), reTrackThreshold,
                                                               _d(pose), _d(aff), _d(flow), C.byref(w), C.byref(used), C.byref(good)), "trackNewCoarse")
        return dict(winner=w.value, pose7=pose, aff=aff, achievedRes=rm, flow=flow, tries_used=used.value, good=bool(good.value))

    # ---- hypothesis-parallel trackNewCoarse over several GPUs (include/dmvio_hip.h)
    def set_comm(self, comm, rank, world):
        """comm: an RcclCommunicator (or a raw ncclComm_t address); None detaches."""
        fn = self.L.dmvio_hip_tracker_set_comm; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        _chk(self.L, fn(self.p, getattr(comm, "p", comm), int(rank), int(world)), "tracker_set_comm")
        self._comm_keep = comm

    def debug_split_single_rank(self, on):
        fn = self.L.dmvio_hip_tracker_debug_split_single_rank; fn.argtypes = [C.c_void_p, C.c_int]; fn.restype = C.c_int
        _chk(self.L, fn(self.p, 1 if on else 0), "tracker_debug_split_single_rank")

    def set_comm_allreduce(self, allreduce, rank, world):
        """allreduce(numpy float64 array) sums the array over all ranks IN PLACE (any transport: torch.distributed / gloo, MPI, threads); None detaches."""
        fn = self.L.dmvio_hip_tracker_set_comm_callbacks; fn.argtypes = [C.c_void_p, C.POINTER(CommCallbacks), C.c_int, C.c_int]
        if allreduce is None:
            _chk(self.L, fn(self.p, None, 0, 0), "tracker_set_comm_callbacks"); self._comm_keep = None; return

        def thunk(_user, buf, count):
            try:
                allreduce(np.ctypeslib.as_array(buf, shape=(count,)))
                return 0
            except Exception:       # never unwind through the C frames
                return 1
        cb = CommCallbacks(None, _ALLREDUCE_F64(thunk), _ALLGATHER(lambda *a: 1))
        self._comm_keep = cb
        _chk(self.L, fn(self.p, C.byref(cb), int(rank), int(world)), "tracker_set_comm_callbacks")

    def last_ticks(self):
        a = C.c_longlong(0); b = C.c_longlong(0)
        _chk(self.L, self.L.dmvio_hip_tracker_last_ticks(self.p, C.byref(a), C.byref(b)), "last_ticks")
        return a.value, b.value

    def last_launch(self):
        a = C.c_int(0); b = C.c_int(0)
        _chk(self.L, self.L.dmvio_hip_tracker_last_launch(self.p, C.byref(a), C.byref(b)), "last_launch")
        return a.value, b.value

    def last_work(self):
        a = C.c_longlong(0); b = C.c_longlong(0)
        _chk(self.L, self.L.dmvio_hip_tracker_last_work(self.p, C.byref(a), C.byref(b)), "last_work")
        return a.value, b.value


def set_raw_batch_layout(ctx, tiled):
    """What UndistorterHip.from_raw_device_batch writes as level 0: 8x4 tiles (True) or row-major (False, the default) (dmvio_hip_set_raw_batch_layout)."""
    fn = ctx.L.dmvio_hip_set_raw_batch_layout; fn.argtypes = [C.c_void_p, C.c_int]; fn.restype = C.c_int
    _chk(ctx.L, fn(ctx.p, 1 if tiled else 0), "set_raw_batch_layout")


def set_raw_batch_kernel(ctx, variant):
    """Kernel behind UndistorterHip.from_raw_device_batch: 1 = a 4 x 8 pixel block per thread, levels in registers (default where the geometry allows), 0 = the LDS-tile build
    (dmvio_hip_set_raw_batch_kernel)."""
    fn = ctx.L.dmvio_hip_set_raw_batch_kernel; fn.argtypes = [C.c_void_p, C.c_int]; fn.restype = C.c_int
    _chk(ctx.L, fn(ctx.p, int(variant)), "set_raw_batch_kernel")


def frame_level0_is_tiled(ctx, slot):
    fn = ctx.L.dmvio_hip_frame_level0_is_tiled; fn.argtypes = [C.c_void_p, C.c_int]; fn.restype = C.c_int
    return bool(_chk(ctx.L, fn(ctx.p, int(slot)), "frame_level0_is_tiled"))


class UndistorterHip:
    """Raw camera image -> PhotometricUndistorter::processFrame + Undistort::undistort on the device (upload path)."""

    def __init__(self, ctx, wOrg, hOrg, bits=8, G=None, vignetteMapInv=None, remapX=None, remapY=None):
        self.ctx, self.L = ctx, ctx.L
        self._keep = [None if a is None else np.ascontiguousarray(a, dtype=np.float32) for a in (G, vignetteMapInv, remapX, remapY)]
        ptr = [None if a is None else _f(a) for a in self._keep]
        p = self.L.dmvio_hip_undistorter_create(ctx.p, wOrg, hOrg, bits, ptr[0], ptr[1], ptr[2], ptr[3])
        if not p:
            raise HipLibraryError("dmvio_hip_undistorter_create: %s" % _err(self.L))
        self.p = C.c_void_p(p)
        self.dtype = np.uint8 if bits == 8 else np.uint16

    def close(self):
        if getattr(self, "p", None):
            self.L.dmvio_hip_undistorter_destroy(self.p); self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, slot, raw, factor=1.0, want_image=True):
        raw = np.ascontiguousarray(raw, dtype=self.dtype)
        out = np.zeros((self.ctx.h, self.ctx.w), np.float32) if want_image else None
        _chk(self.L, self.L.dmvio_hip_frame_upload_raw(self.ctx.p, self.p, slot, raw.ctypes.data_as(C.c_void_p), factor, None if out is None else _f(out)),
             "frame_upload_raw")
        return out

    def from_raw_device_batch(self, slots, raw_dev_ptr, stride_bytes, factor=1.0):
        """B raw images resident in device memory -> undistorted level 0 + pyramids of `slots` in one launch (asynchronous on the ctx stream)."""
        slots = np.ascontiguousarray(slots, dtype=np.int32)
        _chk(self.L, self.L.dmvio_hip_frames_from_raw_device_batch(self.ctx.p, self.p, len(slots), _i(slots), C.c_void_p(raw_dev_ptr), stride_bytes, factor),
             "frames_from_raw_device_batch")


class CoarseInitializerHip:
    """Mirror of CoarseInitializer's hot function calcResAndGS (CoarseInitializer.cpp:331-624) for one pyramid level's point set."""

    def __init__(self, ctx, capacity=32768):
        self.ctx, self.L = ctx, ctx.L
        p = self.L.dmvio_hip_initializer_create(ctx.p, capacity)
        if not p:
            raise HipLibraryError("dmvio_hip_initializer_create: %s" % _err(self.L))
        self.p = C.c_void_p(p)
        self.n = 0

    def close(self):
        if getattr(self, "p", None):
            self.L.dmvio_hip_initializer_destroy(self.p); self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_points(self, pts):
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        u, v, iR, en, oth = f32(pts["u"]), f32(pts["v"]), f32(pts["iR"]), f32(pts["energy"]), f32(pts["outlierTH"])
        good = np.ascontiguousarray(pts["isGood"], dtype=np.uint8)
        _chk(self.L, self.L.dmvio_hip_initializer_set_points(self.p, len(u), _f(u), _f(v), _f(iR), good.ctypes.data_as(C.POINTER(C.c_ubyte)), _f(en), _f(oth)),
             "initializer_set_points")
        self.n = len(u)

    def calcResAndGS(self, lvl, first_slot, new_slot, Ki9, fxfycxcy_lvl, refToNew7, aff_ab, idepth_new, alphaW=150 * 150, alphaK=2.5 * 2.5, couplingWeight=1.0,
                     priorY=0.0, priorX=0.0):
        n = self.n
        o = dict(H=np.zeros((8, 8), np.float32), b=np.zeros(8, np.float32), Hsc=np.zeros((8, 8), np.float32), bsc=np.zeros(8, np.float32), res3=np.zeros(3, np.float32),
                 energy_new=np.zeros((n, 2), np.float32), isGood_new=np.zeros(n, np.uint8), maxstep=np.zeros(n, np.float32), lastHessian_new=np.zeros(n, np.float32),
                 JbBuffer_new=np.zeros((n, 10), np.float32))
        c_u8 = C.POINTER(C.c_ubyte)
        idn = np.ascontiguousarray(idepth_new, dtype=np.float32)
        K = np.ascontiguousarray(fxfycxcy_lvl, dtype=np.float32)
        _chk(self.L, self.L.dmvio_hip_initializer_calc_res_and_gs(self.p, lvl, first_slot, new_slot, _d(np.ascontiguousarray(Ki9, dtype=np.float64)), _f(K),
                                                                  _d(np.ascontiguousarray(refToNew7, dtype=np.float64)), _d(np.array(aff_ab, dtype=np.float64)), _f(idn),
                                                                  alphaW, alphaK, couplingWeight, priorY, priorX, _f(o["H"]), _f(o["b"]), _f(o["Hsc"]), _f(o["bsc"]),
                                                                  _f(o["res3"]), _f(o["energy_new"]), o["isGood_new"].ctypes.data_as(c_u8), _f(o["maxstep"]),
                                                                  _f(o["lastHessian_new"]), _f(o["JbBuffer_new"])), "initializer_calc_res_and_gs")
        return o


class ImmaturePointsHip:
    """Mirror of the immature-point path: ImmaturePoint construction, ImmaturePoint::traceOn, FullSystem::traceNewCoarse."""

    def __init__(self, ctx, capacity=16384):
        self.ctx, self.L = ctx, ctx.L
        p = self.L.dmvio_hip_immature_create(ctx.p, capacity)
        if not p:
            raise HipLibraryError("dmvio_hip_immature_create: %s" % _err(self.L))
        self.p = C.c_void_p(p)

    def close(self):
        if getattr(self, "p", None):
            self.L.dmvio_hip_immature_destroy(self.p); self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def n(self):
        return self.L.dmvio_hip_immature_count(self.p)

    def clear(self):
        _chk(self.L, self.L.dmvio_hip_immature_clear(self.p), "immature_clear")

    def add_points(self, host_tag, host_slot, u, v):
        u = np.ascontiguousarray(u, dtype=np.int32); v = np.ascontiguousarray(v, dtype=np.int32)
        r = self.L.dmvio_hip_immature_add_points(self.p, host_tag, host_slot, len(u), _i(u), _i(v))
        if r < 0:
            raise HipLibraryError("immature_add_points: %s" % _err(self.L))
        return r

    def get_static(self):
        n = self.n
        o = dict(u=np.zeros(n, np.float32), v=np.zeros(n, np.float32), host=np.zeros(n, np.int32), color=np.zeros((n, 8), np.float32),
                 weights=np.zeros((n, 8), np.float32), gradH=np.zeros((n, 4), np.float32), energyTH=np.zeros(n, np.float32))
        _chk(self.L, self.L.dmvio_hip_immature_get_static(self.p, _f(o["u"]), _f(o["v"]), _i(o["host"]), _f(o["color"]), _f(o["weights"]), _f(o["gradH"]),
                                                           _f(o["energyTH"])), "immature_get_static")
        return o

    def get_state(self):
        n = self.n
        o = dict(idepth_min=np.zeros(n, np.float32), idepth_max=np.zeros(n, np.float32), quality=np.zeros(n, np.float32), lastTraceUV=np.zeros((n, 2), np.float32),
                 lastTracePixelInterval=np.zeros(n, np.float32), lastTraceStatus=np.zeros(n, np.int32))
        _chk(self.L, self.L.dmvio_hip_immature_get_state(self.p, _f(o["idepth_min"]), _f(o["idepth_max"]), _f(o["quality"]), _f(o["lastTraceUV"]),
                                                          _f(o["lastTracePixelInterval"]), _i(o["lastTraceStatus"])), "immature_get_state")
        return o

    def set_state(self, idepth_min, idepth_max, quality, status):
        a = [np.ascontiguousarray(x, dtype=np.float32) for x in (idepth_min, idepth_max, quality)]
        st = np.ascontiguousarray(status, dtype=np.int32)
        _chk(self.L, self.L.dmvio_hip_immature_set_state(self.p, _f(a[0]), _f(a[1]), _f(a[2]), _i(st)), "immature_set_state")

    def trace(self, new_slot, KRKi, Kt, aff):
        KRKi = np.ascontiguousarray(KRKi, dtype=np.float32).reshape(-1, 9); Kt = np.ascontiguousarray(Kt, dtype=np.float32).reshape(-1, 3)
        aff = np.ascontiguousarray(aff, dtype=np.float32).reshape(-1, 2)
        _chk(self.L, self.L.dmvio_hip_immature_trace(self.p, new_slot, len(KRKi), _f(KRKi), _f(Kt), _f(aff)), "immature_trace")

    def optimize(self, frame_slots, w2c7, fxfycxcy, aff=None, exposure=None, select=None, min_obs=1):
        """FullSystem::optimizeImmaturePoint for the selected points; returns (result, idepth, res_state[n, F])."""
        slots = np.ascontiguousarray(frame_slots, dtype=np.int32); F = len(slots)
        w2c7 = np.ascontiguousarray(w2c7, dtype=np.float64).reshape(F, 7)
        a = np.zeros((F, 2)) if aff is None else np.ascontiguousarray(aff, dtype=np.float64)
        e = np.ones(F, np.float32) if exposure is None else np.ascontiguousarray(exposure, dtype=np.float32)
        n = self.n
        result = np.zeros(n, np.int32); idepth = np.zeros(n, np.float32); res_state = np.zeros((n, F), np.int32)
        sel = None if select is None else np.ascontiguousarray(select, dtype=np.uint8).tobytes()
        _chk(self.L, self.L.dmvio_hip_immature_optimize(self.p, F, _i(slots), _d(w2c7), _d(a), _f(e), _d(np.ascontiguousarray(fxfycxcy, dtype=np.float64)), sel, min_obs,
                                                         _i(result), _f(idepth), _i(res_state)), "immature_optimize")
        return result, idepth, res_state

    def traceNewCoarse(self, new_slot, new_w2c7, host_c2w7, fxfycxcy, new_aff=(0.0, 0.0), new_exposure=1.0, host_aff=None, host_exposure=None):
        host_c2w7 = np.ascontiguousarray(host_c2w7, dtype=np.float64).reshape(-1, 7)
        H = len(host_c2w7)
        ha = np.zeros((H, 2)) if host_aff is None else np.ascontiguousarray(host_aff, dtype=np.float64)
        he = np.ones(H, np.float32) if host_exposure is None else np.ascontiguousarray(host_exposure, dtype=np.float32)
        counts = np.zeros(6, np.int32)
        _chk(self.L, self.L.dmvio_hip_trace_new_coarse(self.p, new_slot, _d(np.ascontiguousarray(new_w2c7, dtype=np.float64)), _d(np.array(new_aff, dtype=np.float64)),
                                                        new_exposure, H, _d(host_c2w7), _d(ha), _f(he), _d(np.ascontiguousarray(fxfycxcy, dtype=np.float64)), _i(counts)),
             "trace_new_coarse")
        return dict(zip(("good", "oob", "outlier", "skipped", "badcondition", "uninitialized"), counts.tolist()))


class BundleAdjusterBatch:
    """dmvio_hip_ba_optimize_batch: FullSystem::optimize for W windows (BundleAdjusterHip objects of one context) per launch sequence, on the device-resident loop."""

    def __init__(self, ctx, max_windows):
        self.ctx = ctx; self.L = ctx.L
        L = self.L
        L.dmvio_hip_ba_batch_create.restype = C.c_void_p; L.dmvio_hip_ba_batch_create.argtypes = [C.c_void_p, C.c_int]
        L.dmvio_hip_ba_batch_destroy.argtypes = [C.c_void_p]; L.dmvio_hip_ba_batch_destroy.restype = None
        L.dmvio_hip_ba_optimize_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]; L.dmvio_hip_ba_optimize_batch.restype = C.c_int
        L.dmvio_hip_ba_batch_set_exact_backsub.argtypes = [C.c_void_p, C.c_int]; L.dmvio_hip_ba_batch_last_ms.argtypes = [C.c_void_p, C.c_void_p]
        p = L.dmvio_hip_ba_batch_create(ctx.p, int(max_windows))
        if not p:
            raise HipLibraryError("ba_batch_create: " + _err(L))
        self.p = C.c_void_p(p)

    def close(self):
        if getattr(self, "p", None):
            self.L.dmvio_hip_ba_batch_destroy(self.p); self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_exact_backsub(self, on=True):
        _chk(self.L, self.L.dmvio_hip_ba_batch_set_exact_backsub(self.p, 1 if on else 0), "ba_batch_set_exact_backsub")

    def optimize(self, windows, its=6):
        W = len(windows)
        hs = (C.c_void_p * W)(*[w.p for w in windows])
        rm = np.zeros(W, np.float32); fe = np.zeros(W); it = np.zeros(W, np.int32); tr = np.zeros((W, 64, 4))
        _chk(self.L, self.L.dmvio_hip_ba_optimize_batch(self.p, W, hs, its, rm.ctypes.data, fe.ctypes.data, it.ctypes.data, tr.ctypes.data), "ba_optimize_batch")
        return [dict(rmse=float(rm[k]), finalEnergy=float(fe[k]), iterations=int(it[k]), trace=tr[k, :it[k] + 1].copy()) for k in range(W)]

    def last_ms(self):
        ms = np.zeros(3, np.float32)
        _chk(self.L, self.L.dmvio_hip_ba_batch_last_ms(self.p, ms.ctypes.data), "ba_batch_last_ms")
        return float(ms[0]), float(ms[1]), float(ms[2])

    def set_profile(self, on=True):
        fn = self.L.dmvio_hip_ba_batch_set_profile; fn.argtypes = [C.c_void_p, C.c_int]; fn.restype = C.c_int
        _chk(self.L, fn(self.p, 1 if on else 0), "ba_batch_set_profile")

    def set_streams(self, streams=0):
        """0: automatic (up to three groups of windows on their own streams from 4 windows on); k >= 1: at most k groups"""
        fn = self.L.dmvio_hip_ba_batch_set_streams; fn.argtypes = [C.c_void_p, C.c_int]; fn.restype = C.c_int
        _chk(self.L, fn(self.p, int(streams)), "ba_batch_set_streams")

    def set_linearize_lanes(self, lanes=1):
        """1: one lane per residual from 4 windows on (k_ba_linearize_b1, default); 8: the eight-lane kernel for every batch size"""
        fn = self.L.dmvio_hip_ba_batch_set_linearize_lanes; fn.argtypes = [C.c_void_p, C.c_int]; fn.restype = C.c_int
        _chk(self.L, fn(self.p, int(lanes)), "ba_batch_set_linearize_lanes")

    def last_solve_ticks(self):
        """in-kernel timeline of window 0's last k_ba_solve, microseconds since the kernel started (12 phase boundaries)"""
        t = np.zeros(12, np.int32)
        fn = self.L.dmvio_hip_ba_batch_last_solve_ticks; fn.argtypes = [C.c_void_p, C.c_void_p]; fn.restype = C.c_int
        _chk(self.L, fn(self.p, t.ctypes.data), "ba_batch_last_solve_ticks")
        return t / 100.0

    def last_host_us(self):
        """host clock at the phase boundaries of the last call (dmvio_hip_ba_batch_last_host_us)"""
        o = (C.c_double * 8)()
        fn = self.L.dmvio_hip_ba_batch_last_host_us; fn.argtypes = [C.c_void_p, C.POINTER(C.c_double)]; fn.restype = C.c_int
        _chk(self.L, fn(self.p, o), "ba_batch_last_host_us")
        return [float(x) for x in o]

    def last_pivot_branch(self, w=0):
        """how window w's last solve found Eigen's pivot order: 0 ranks (distinct |diagonal|), 1 ties replayed, 2 NaN"""
        b = C.c_int(-1)
        fn = self.L.dmvio_hip_ba_batch_last_pivot_branch; fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p]; fn.restype = C.c_int
        _chk(self.L, fn(self.p, int(w), C.byref(b)), "ba_batch_last_pivot_branch")
        return b.value


def debug_solve(ctx, H, b, exact_backsub=True):
    """dmvio_hip_ba_debug_solve: the device-resident loop's 68x68 solve (k_ba_solve's scaling, pivot order, LDL^T, substitutions) for a given system.
    Returns (x, perm, branch, zero)."""
    Hc = np.ascontiguousarray(H, dtype=np.float64); bc = np.ascontiguousarray(b, dtype=np.float64); n = len(bc)
    x = np.zeros(n); perm = np.zeros(n, np.int32); br = C.c_int(-1); z = C.c_int(-1)
    fn = ctx.L.dmvio_hip_ba_debug_solve
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]; fn.restype = C.c_int
    _chk(ctx.L, fn(ctx.p, n, Hc.ctypes.data, bc.ctypes.data, 1 if exact_backsub else 0, x.ctypes.data, perm.ctypes.data, C.byref(br), C.byref(z)), "ba_debug_solve")
    return x, perm, br.value, z.value


def host_solve_ldlt(L, H, b):
    """dmvio_hip_ba_solve_ldlt (host): BAHost::ldltSolveTransposed behind the reference's diagonal pre-scaling"""
    Hc = np.ascontiguousarray(H, dtype=np.float64); bc = np.ascontiguousarray(b, dtype=np.float64); x = np.zeros(len(bc))
    _chk(L, L.dmvio_hip_ba_solve_ldlt(len(bc), _d(Hc), _d(bc), _d(x)), "ba_solve_ldlt")
    return x


class RcclCommunicator:
    """ncclComm_t created through the library's wrappers (dmvio_hip_comm_*): rank `rank` of `world` on the context's device.
    unique_id(): 128 bytes from one rank, to be handed to all ranks (e.g. torch.distributed broadcast) before construction."""

    @staticmethod
    def unique_id(L):
        buf = (C.c_ubyte * 128)()
        _chk(L, L.dmvio_hip_comm_unique_id(buf), "comm_unique_id")
        return bytes(buf)

    def __init__(self, ctx, unique_id, rank, world):
        self.L = ctx.L
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        out = C.c_void_p()
        _chk(self.L, self.L.dmvio_hip_comm_init_rank(ctx.p, buf, int(rank), int(world), C.byref(out)), "comm_init_rank")
        self.p = out
        self.rank, self.world = rank, world

    def info(self):
        """(ncclCommCount, ncclCommUserRank): what RCCL itself reports for this communicator."""
        n = C.c_int(0); r = C.c_int(-1)
        fn = self.L.dmvio_hip_comm_info; fn.argtypes = [C.c_void_p, c_i, c_i]
        _chk(self.L, fn(self.p, C.byref(n), C.byref(r)), "comm_info")
        return n.value, r.value

    def close(self):
        if getattr(self, "p", None):
            self.L.dmvio_hip_comm_destroy(self.p)
            self.p = None


class WindowGraph:
    """dmvio_hip_graph: the host-side mirror of the window's point / residual graph with EnergyFunctional's own mutators (insertFrame / insertPoint / insertResidual /
    dropResidual / removePoint / marginalizeFrame, EnergyFunctional.cpp:435-518, 641-646, 766-782).  Host only: needs the library, no device."""

    def __init__(self, L=None):
        self.L = L or load_library()
        vp, ci, cf = C.c_void_p, C.c_int, C.c_float
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
        self.L.dmvio_hip_graph_create.restype = vp
        for name, args in (("destroy", [vp]), ("clear", [vp]), ("insert_frame", [vp]), ("remove_frame", [vp, ci]), ("insert_point", [vp, ci, cf, cf, cf, fp, fp, ci]),
                           ("remove_point", [vp, ci, ci]), ("insert_residual", [vp, ci, ci, ci]), ("drop_residual", [vp, ci, ci, ci]), ("set_idepth", [vp, ci, ci, cf]),
                           ("set_idepths", [vp, ci, fp]), ("counts", [vp, ip, ip, ip]), ("frame_points", [vp, ci]), ("point_residuals", [vp, ci, ci]),
                           ("export", [vp, ip, fp, fp, fp, fp, fp, C.POINTER(C.c_ubyte), ip, ip]), ("set_residual_linearized", [vp, ci, ci, ci, fp, fp]),
                           ("linearized_count", [vp]), ("export_linearized", [vp, C.POINTER(C.c_ubyte), fp, fp])):
            fn = getattr(self.L, "dmvio_hip_graph_" + name); fn.argtypes = args; fn.restype = None if name == "destroy" else ci
        self.p = C.c_void_p(self.L.dmvio_hip_graph_create())
        if not self.p:
            raise HipLibraryError("dmvio_hip_graph_create failed")

    def __del__(self):
        if getattr(self, "p", None):
            self.L.dmvio_hip_graph_destroy(self.p); self.p = None

    def _c(self, name, *a):
        return _chk(self.L, getattr(self.L, "dmvio_hip_graph_" + name)(self.p, *a), "graph_" + name)

    def clear(self): self._c("clear")
    def insert_frame(self): return self._c("insert_frame")
    def remove_frame(self, idx): self._c("remove_frame", int(idx))

    def insert_point(self, host, u, v, idepth, color8, weights8, prior=False):
        c = np.ascontiguousarray(color8, dtype=np.float32); w = np.ascontiguousarray(weights8, dtype=np.float32)
        return self._c("insert_point", int(host), float(u), float(v), float(idepth), _f(c), _f(w), 1 if prior else 0)

    def remove_point(self, host, idx): self._c("remove_point", int(host), int(idx))
    def insert_residual(self, host, idx, target): return self._c("insert_residual", int(host), int(idx), int(target))
    def drop_residual(self, host, idx, k): self._c("drop_residual", int(host), int(idx), int(k))
    def set_idepth(self, host, idx, idepth): self._c("set_idepth", int(host), int(idx), float(idepth))

    def set_idepths(self, idepth):
        a = np.ascontiguousarray(idepth, dtype=np.float32)
        self._c("set_idepths", len(a), _f(a))

    def set_residual_linearized(self, host, idx, k, J74=None, res_toZeroF=None):
        """EFResidual::isLinearized = true with its frozen Jacobian (74) and res_toZeroF (8); J74 None: not linearised any more"""
        if J74 is None:
            self._c("set_residual_linearized", int(host), int(idx), int(k), None, None); return
        J = np.ascontiguousarray(J74, dtype=np.float32).ravel(); r = np.ascontiguousarray(res_toZeroF, dtype=np.float32).ravel()
        assert J.size == 74 and r.size == 8
        self._c("set_residual_linearized", int(host), int(idx), int(k), _f(J), _f(r))

    def linearized_count(self): return self._c("linearized_count")

    def export_linearized(self):
        _, _, R = self.counts()
        fl = np.zeros(R, np.uint8); J = np.zeros((R, 74), np.float32); r = np.zeros((R, 8), np.float32)
        self._c("export_linearized", fl.ctypes.data_as(C.POINTER(C.c_ubyte)), _f(J), _f(r))
        return fl, J, r

    def counts(self):
        F, N, R = C.c_int(0), C.c_int(0), C.c_int(0)
        self._c("counts", C.byref(F), C.byref(N), C.byref(R))
        return F.value, N.value, R.value

    def frame_points(self, host): return self._c("frame_points", int(host))
    def point_residuals(self, host, idx): return self._c("point_residuals", int(host), int(idx))

    def export(self):
        """The flat arrays of BundleAdjusterHip.set_graph, in makeIDX order."""
        _, N, R = self.counts()
        o = dict(host=np.zeros(N, np.int32), u=np.zeros(N, np.float32), v=np.zeros(N, np.float32), idepth=np.zeros(N, np.float32), color=np.zeros((N, 8), np.float32),
                 weights=np.zeros((N, 8), np.float32), hasDepthPrior=np.zeros(N, np.uint8), res_point=np.zeros(R, np.int32), res_target=np.zeros(R, np.int32))
        self._c("export", _i(o["host"]), _f(o["u"]), _f(o["v"]), _f(o["idepth"]), _f(o["color"]), _f(o["weights"]), o["hasDepthPrior"].ctypes.data_as(C.POINTER(C.c_ubyte)),
                _i(o["res_point"]), _i(o["res_target"]))
        return o

    @classmethod
    def from_case(cls, case, L=None):
        """The graph of a synth.ba_case dictionary, built through the mutators."""
        g = cls(L)
        for _ in range(case["n_frames"]):
            g.insert_frame()
        hp = case.get("hasDepthPrior")
        idx = []
        for p in range(len(case["u"])):
            idx.append(g.insert_point(case["host"][p], case["u"][p], case["v"][p], case["idepth0"][p], case["color"][p], case["weights"][p], bool(hp[p]) if hp is not None else False))
        for r in range(len(case["res_point"])):
            p = case["res_point"][r]
            g.insert_residual(case["host"][p], idx[p], case["res_target"][r])
        return g


class BundleAdjusterHip:
    """The sliding window FullSystem::optimize works on, over the C ABI (include/dmvio_hip.h, "sliding-window BA")."""

    def __init__(self, ctx, accumulators=None, keep_jacobians=False):
        """accumulators: partial accumulators per bucket (1 = the reference's single-threaded order, bit for bit; None = library default, 4);
        keep_jacobians: materialise the 74-float RawResidualJacobian per residual for jacobians()."""
        self.ctx = ctx
        self.L = ctx.L
        p = self.L.dmvio_hip_ba_create(ctx.p)
        if not p:
            raise HipLibraryError("ba_create: " + (self.L.dmvio_hip_last_error() or b"").decode())
        self.p = C.c_void_p(p)
        if accumulators is not None:
            fn = self.L.dmvio_hip_ba_set_accumulators; fn.argtypes = [C.c_void_p, C.c_int]; fn.restype = C.c_int
            _chk(self.L, fn(self.p, int(accumulators)), "ba_set_accumulators")
        if keep_jacobians:
            fn = self.L.dmvio_hip_ba_keep_jacobians; fn.argtypes = [C.c_void_p, C.c_int]; fn.restype = C.c_int
            _chk(self.L, fn(self.p, 1), "ba_keep_jacobians")

    def close(self):
        if getattr(self, "p", None):
            self.L.dmvio_hip_ba_destroy(self.p)
            self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_case(self, case, slots, poses=None, idepth=None):
        """Window + graph from a dm-vio_amd.synth.ba_case dictionary (frames must already be uploaded into `slots`)."""
        F = case["n_frames"]
        poses = np.ascontiguousarray(case["poses0"] if poses is None else poses, dtype=np.float64).reshape(F, 7)
        idepth = np.ascontiguousarray(case["idepth0"] if idepth is None else idepth, dtype=np.float32)
        aff = np.zeros((F, 2)) if case.get("aff") is None else case["aff"]
        expo = np.ones(F, dtype=np.float32) if case.get("exposure") is None else case["exposure"]
        fids = np.arange(F, dtype=np.int32) if case.get("frameIDs") is None else case["frameIDs"]
        self.set_window(slots, poses, aff, expo, fids, case["K4"])
        self.set_graph(case["host"], case["u"], case["v"], idepth, case["color"], case["weights"], case.get("hasDepthPrior"), case["res_point"], case["res_target"])

    def set_residual_flags(self, is_linearized):
        fl = np.ascontiguousarray(is_linearized, dtype=np.uint8)
        fn = self.L.dmvio_hip_ba_set_residual_flags; fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p]; fn.restype = C.c_int
        _chk(self.L, fn(self.p, len(fl), fl.ctypes.data), "ba_set_residual_flags")

    def set_linearized_residuals(self, is_linearized, J74, res_toZeroF):
        """Residuals that arrive linearised (dmvio_hip_ba_set_linearized_residuals); returns the number of linearised residuals of the graph"""
        fl = np.ascontiguousarray(is_linearized, dtype=np.uint8); J = np.ascontiguousarray(J74, dtype=np.float32); r = np.ascontiguousarray(res_toZeroF, dtype=np.float32)
        assert J.size == 74 * len(fl) and r.size == 8 * len(fl)
        fn = self.L.dmvio_hip_ba_set_linearized_residuals; fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]; fn.restype = C.c_int
        n = C.c_int(0)
        _chk(self.L, fn(self.p, len(fl), fl.ctypes.data, J.ctypes.data, r.ctypes.data, C.byref(n)), "ba_set_linearized_residuals")
        return n.value

    def linearized_residuals(self):
        """(flags, J74, res_toZeroF) of the graph's residuals as they stand (dmvio_hip_ba_get_linearized_residuals)"""
        R = self.R
        fl = np.zeros(R, np.uint8); J = np.zeros((R, 74), np.float32); r = np.zeros((R, 8), np.float32)
        fn = self.L.dmvio_hip_ba_get_linearized_residuals; fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]; fn.restype = C.c_int
        _chk(self.L, fn(self.p, R, fl.ctypes.data, J.ctypes.data, r.ctypes.data), "ba_get_linearized_residuals")
        return fl, J, r

    def fix_linearization(self, res_mask):
        """EFResidual::fixLinearizationF for the active residuals with res_mask != 0 on the resident graph (dmvio_hip_ba_fix_linearization): they leave activeResiduals and
        enter every later system / energy through accumulateLF_MT (addPoint<1>) / calcLEnergyPt.  Needs BundleAdjusterHip(ctx, keep_jacobians=True).  Returns the
        number of linearised residuals of the graph."""
        m = np.ascontiguousarray(res_mask, dtype=np.uint8)
        fn = self.L.dmvio_hip_ba_fix_linearization; fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int)]; fn.restype = C.c_int
        n = C.c_int(0)
        _chk(self.L, fn(self.p, len(m), m.ctypes.data, C.byref(n)), "ba_fix_linearization")
        return n.value

    def lf_system(self):
        """accumulateLF_MT's [HL, bL] (linearised residuals + priors) for the state of the last accumulation"""
        n = self.n
        HL = np.zeros((n, n)); bL = np.zeros(n)
        fn = self.L.dmvio_hip_ba_get_lf_system; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]; fn.restype = C.c_int
        _chk(self.L, fn(self.p, HL.ctypes.data, bL.ctypes.data), "ba_get_lf_system")
        return HL, bL

    def set_stream(self, stream_ptr):
        _chk(self.L, self.L.dmvio_hip_ba_set_stream(self.p, C.c_void_p(stream_ptr)), "ba_set_stream")

    def set_window(self, slots, poses7_w2c, aff_ab, exposures, frameIDs, K4):
        self.F = len(slots); self.n = 4 + 8 * self.F
        slots = np.ascontiguousarray(slots, dtype=np.int32); poses = np.ascontiguousarray(poses7_w2c, dtype=np.float64)
        aff = np.ascontiguousarray(aff_ab, dtype=np.float64); ex = np.ascontiguousarray(exposures, dtype=np.float32)
        ids = np.ascontiguousarray(frameIDs, dtype=np.int32); K = np.ascontiguousarray(K4, dtype=np.float64)
        _chk(self.L, self.L.dmvio_hip_ba_set_window(self.p, self.F, _i(slots), _d(poses), _d(aff), _f(ex), _i(ids), _d(K)), "ba_set_window")

    def set_marg_prior(self, HM, bM):
        HM = np.ascontiguousarray(HM, dtype=np.float64); bM = np.ascontiguousarray(bM, dtype=np.float64)
        _chk(self.L, self.L.dmvio_hip_ba_set_marg_prior(self.p, _d(HM), _d(bM)), "ba_set_marg_prior")

    def marginalize_frame(self, k):
        n = self.n - 8
        H = np.zeros((n, n)); b = np.zeros(n)
        _chk(self.L, self.L.dmvio_hip_ba_marginalize_frame(self.p, k, _d(H), _d(b)), "ba_marginalize_frame")
        return H, b

    def get_marg_prior(self):
        n = self.n
        H = np.zeros((n, n)); b = np.zeros(n)
        _chk(self.L, self.L.dmvio_hip_ba_get_marg_prior(self.p, _d(H), _d(b)), "ba_get_marg_prior")
        return H, b

    def set_frame_state(self, k, state10):
        _chk(self.L, self.L.dmvio_hip_ba_set_frame_state(self.p, k, _d(np.ascontiguousarray(state10, dtype=np.float64))), "ba_set_frame_state")

    def set_frame_zero(self, k, state_zero10):
        fn = self.L.dmvio_hip_ba_set_frame_zero; fn.argtypes = [C.c_void_p, C.c_int, c_d]; fn.restype = C.c_int
        _chk(self.L, fn(self.p, k, _d(np.ascontiguousarray(state_zero10, dtype=np.float64))), "ba_set_frame_zero")

    def set_frame_energy_th(self, th):
        fn = self.L.dmvio_hip_ba_set_frame_energy_th; fn.argtypes = [C.c_void_p, c_f]; fn.restype = C.c_int
        _chk(self.L, fn(self.p, _f(np.ascontiguousarray(th, dtype=np.float32))), "ba_set_frame_energy_th")

    def set_calib_values(self, value4, value_zero4):
        fn = self.L.dmvio_hip_ba_set_calib_values; fn.argtypes = [C.c_void_p, c_d, c_d]; fn.restype = C.c_int
        _chk(self.L, fn(self.p, _d(np.ascontiguousarray(value4, dtype=np.float64)), _d(np.ascontiguousarray(value_zero4, dtype=np.float64))), "ba_set_calib_values")

    def marginalize_points(self, candidates, update_prior=False):
        """flagPointsForRemoval's relinearisation + EnergyFunctional::marginalizePointsF: (decision[N], Hadd, badd, resInM)."""
        n = self.n
        cand = np.ascontiguousarray(candidates, dtype=np.uint8)
        dec = np.zeros(self.N, np.uint8); H = np.zeros((n, n)); b = np.zeros(n); r = C.c_int(0)
        _chk(self.L, self.L.dmvio_hip_ba_marginalize_points(self.p, cand.tobytes(), dec.ctypes.data_as(C.POINTER(C.c_ubyte)), _d(H), _d(b), C.byref(r),
                                                             1 if update_prior else 0), "ba_marginalize_points")
        return dec, H, b, r.value

    def set_graph(self, host, u, v, idepth, color, weights, hasDepthPrior, res_point, res_target):
        self.N = len(u); self.R = len(res_point)
        a = [np.ascontiguousarray(host, dtype=np.int32), np.ascontiguousarray(u, dtype=np.float32), np.ascontiguousarray(v, dtype=np.float32),
             np.ascontiguousarray(idepth, dtype=np.float32), np.ascontiguousarray(color, dtype=np.float32), np.ascontiguousarray(weights, dtype=np.float32)]
        hp = None if hasDepthPrior is None else np.ascontiguousarray(hasDepthPrior, dtype=np.uint8)
        rp = np.ascontiguousarray(res_point, dtype=np.int32); rt = np.ascontiguousarray(res_target, dtype=np.int32)
        self._n_newest = int((rt == self.F - 1).sum())
        u8 = C.POINTER(C.c_ubyte)
        _chk(self.L, self.L.dmvio_hip_ba_set_graph(self.p, self.N, _i(a[0]), _f(a[1]), _f(a[2]), _f(a[3]), _f(a[4]), _f(a[5]),
                                                   None if hp is None else hp.ctypes.data_as(u8), self.R, _i(rp), _i(rt)), "ba_set_graph")

    def set_graph_from(self, graph):
        """dmvio_hip_ba_set_graph_from: the device arrays from a resident WindowGraph (after set_window)."""
        fn = self.L.dmvio_hip_ba_set_graph_from; fn.argtypes = [C.c_void_p, C.c_void_p]; fn.restype = C.c_int
        _chk(self.L, fn(self.p, graph.p), "ba_set_graph_from")
        _, self.N, self.R = graph.counts()
        self._n_newest = int((graph.export()["res_target"] == self.F - 1).sum())

    def activate_all(self):
        _chk(self.L, self.L.dmvio_hip_ba_activate_all(self.p), "ba_activate_all")

    def linearize_all(self, fix=False):
        e = C.c_double(0)
        _chk(self.L, self.L.dmvio_hip_ba_linearize(self.p, 1 if fix else 0, C.byref(e)), "ba_linearize")
        return e.value

    def apply_res(self):
        _chk(self.L, self.L.dmvio_hip_ba_apply(self.p), "ba_apply")

    def res_state(self):
        R = self.R; u8 = C.POINTER(C.c_ubyte)
        ns = np.zeros(R, dtype=np.uint8); ne = np.zeros(R, dtype=np.float32); nw = np.zeros(R, dtype=np.float32)
        ia = np.zeros(R, dtype=np.uint8); cp = np.zeros((R, 3), dtype=np.float32)
        _chk(self.L, self.L.dmvio_hip_ba_get_res_state(self.p, ns.ctypes.data_as(u8), _f(ne), _f(nw), ia.ctypes.data_as(u8), _f(cp)), "ba_get_res_state")
        return dict(newState=ns, newEnergy=ne, newEnergyWO=nw, isActive=ia, center=cp)

    def jacobians(self):
        J = np.zeros((self.R, 74), dtype=np.float32)
        _chk(self.L, self.L.dmvio_hip_ba_get_jacobians(self.p, _f(J)), "ba_get_jacobians")
        return J

    def frame_energy_th(self):
        o = np.zeros(self.F, dtype=np.float32)
        _chk(self.L, self.L.dmvio_hip_ba_get_frame_energy_th(self.p, _f(o)), "ba_get_frame_energy_th"); return o

    def accumulate(self):
        n = self.n
        HA = np.zeros((n, n)); bA = np.zeros(n); Hsc = np.zeros((n, n)); bsc = np.zeros(n); r = C.c_int(0)
        _chk(self.L, self.L.dmvio_hip_ba_accumulate(self.p, _d(HA), _d(bA), _d(Hsc), _d(bsc), C.byref(r)), "ba_accumulate")
        return dict(HA=HA, bA=bA, Hsc=Hsc, bsc=bsc, resInA=r.value)

    def point_acc(self):
        N = self.N
        o = [np.zeros(N, dtype=np.float32), np.zeros(N, dtype=np.float32), np.zeros((N, 4), dtype=np.float32), np.zeros(N, dtype=np.float32), np.zeros(N, dtype=np.float32)]
        _chk(self.L, self.L.dmvio_hip_ba_get_point_acc(self.p, *[_f(a) for a in o]), "ba_get_point_acc")
        return dict(Hdd=o[0], bd=o[1], Hcd=o[2], HdiF=o[3], bdSumF=o[4])

    def solve(self, iteration, lam):
        x = np.zeros(self.n)
        _chk(self.L, self.L.dmvio_hip_ba_solve(self.p, iteration, lam, _d(x)), "ba_solve"); return x

    def resubstitute(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        _chk(self.L, self.L.dmvio_hip_ba_resubstitute(self.p, _d(x)), "ba_resubstitute")

    def point_state(self):
        a = np.zeros(self.N, dtype=np.float32); b = np.zeros(self.N, dtype=np.float32)
        _chk(self.L, self.L.dmvio_hip_ba_get_points(self.p, _f(a), _f(b)), "ba_get_points"); return a, b

    def frame_pose(self, k):
        p = np.zeros(7); a = np.zeros(2); s = np.zeros(10)
        _chk(self.L, self.L.dmvio_hip_ba_get_frame(self.p, k, _d(p), _d(a), _d(s)), "ba_get_frame"); return p, a, s

    # ---- sharded-iteration building blocks
    def backup(self):
        _chk(self.L, self.L.dmvio_hip_ba_backup(self.p), "ba_backup")

    def solve_system(self, iteration, lam, HA, bA, Hsc, bsc):
        x = np.zeros(self.n)
        a = [np.ascontiguousarray(v, dtype=np.float64) for v in (HA, bA, Hsc, bsc)]
        _chk(self.L, self.L.dmvio_hip_ba_solve_system(self.p, iteration, lam, _d(a[0]), _d(a[1]), _d(a[2]), _d(a[3]), _d(x)), "ba_solve_system")
        return x

    def step(self, stepfac=1.0):
        s6 = np.zeros(6, dtype=np.float32)
        _chk(self.L, self.L.dmvio_hip_ba_step(self.p, stepfac, _f(s6)), "ba_step"); return s6

    def restore(self):
        _chk(self.L, self.L.dmvio_hip_ba_restore(self.p), "ba_restore")

    def count_newest_residuals(self):
        """Residuals of this graph that target the newest keyframe (the inputs of setNewFrameEnergyTH)."""
        return self._n_newest

    def linearize_local(self, fix=False):
        e = C.c_double(0); n = C.c_int(0); buf = np.zeros(self.R, dtype=np.float32)
        _chk(self.L, self.L.dmvio_hip_ba_linearize_local(self.p, 1 if fix else 0, C.byref(e), _f(buf), C.byref(n)), "ba_linearize_local")
        return e.value, buf[:n.value].copy()

    def set_new_frame_energy_th(self, th):
        _chk(self.L, self.L.dmvio_hip_ba_set_new_frame_energy_th(self.p, float(th)), "ba_set_new_frame_energy_th")

    def energy_terms(self):
        a = C.c_double(0); b = C.c_double(0)
        _chk(self.L, self.L.dmvio_hip_ba_energy_terms(self.p, C.byref(a), C.byref(b)), "ba_energy_terms"); return a.value, b.value

    def gn_iteration(self, iteration, lam, lastE):
        l = C.c_double(lam); e = np.array(lastE, dtype=np.float64); acc = C.c_int(0)
        _chk(self.L, self.L.dmvio_hip_ba_gn_iteration(self.p, iteration, C.byref(l), _d(e), C.byref(acc)), "ba_gn_iteration")
        return bool(acc.value), l.value, e

    # ---- one window over several GPUs: this handle holds the rank's points; the library runs the exchanges itself (include/dmvio_hip.h)
    def set_comm(self, comm, rank, world):
        """comm: an RcclCommunicator (or a raw ncclComm_t address); None detaches."""
        addr = getattr(comm, "p", comm)
        _chk(self.L, self.L.dmvio_hip_ba_set_comm(self.p, addr, int(rank), int(world)), "ba_set_comm")
        self._comm_keep = comm

    def set_comm_torch(self, dist, group=None):
        """The same protocol over torch.distributed CPU collectives (gloo), staged through host memory: for tests and hosts without RCCL peers."""
        import torch
        world, rank = dist.get_world_size(group), dist.get_rank(group)

        def allreduce(_user, buf, count):
            try:
                a = np.ctypeslib.as_array(buf, shape=(count,))
                t = torch.from_numpy(a)
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                return 0
            except Exception:       # never unwind through the C frames
                return 1

        def allgather(_user, src, dst, nbytes):
            try:
                a = np.ctypeslib.as_array(C.cast(src, C.POINTER(C.c_ubyte)), shape=(nbytes,))
                o = np.ctypeslib.as_array(C.cast(dst, C.POINTER(C.c_ubyte)), shape=(nbytes * world,))
                dist.all_gather_into_tensor(torch.from_numpy(o), torch.from_numpy(a.copy()), group=group)
                return 0
            except Exception:
                return 1

        cb = CommCallbacks(None, _ALLREDUCE_F64(allreduce), _ALLGATHER(allgather))
        self._comm_keep = cb      # the C side stores the function pointers: keep the thunks alive
        _chk(self.L, self.L.dmvio_hip_ba_set_comm_callbacks(self.p, C.byref(cb), rank, world), "ba_set_comm_callbacks")

    def profile_chain(self, reps=20):
        """Mean microseconds of the five kernels of an accepted GN iteration (linearise, per-point sums, accumulate, stitch, gather), HIP events on the handle's stream."""
        us = np.zeros(5, dtype=np.float32)
        fn = self.L.dmvio_hip_ba_profile_chain; fn.argtypes = [C.c_void_p, C.c_int, c_f]
        _chk(self.L, fn(self.p, int(reps), _f(us)), "ba_profile_chain")
        return dict(zip(("k_ba_linearize", "k_ba_point_sums", "k_ba_accumulate", "k_ba_stitch", "k_ba_stitch_gather"), [float(x) for x in us]))

    def point_hessian(self):
        o = np.zeros(self.N, dtype=np.float32)
        _chk(self.L, self.L.dmvio_hip_ba_get_point_hessian(self.p, _f(o)), "ba_get_point_hessian"); return o

    def solve_ldlt(self, HPassed, b):
        """The reference's non-GTSAM solve (diagonal pre-scaling + pivoted LDL^T), host-only."""
        H = np.ascontiguousarray(HPassed, dtype=np.float64); bb = np.ascontiguousarray(b, dtype=np.float64); x = np.zeros(len(bb))
        _chk(self.L, self.L.dmvio_hip_ba_solve_ldlt(len(bb), _d(H), _d(bb), _d(x)), "ba_solve_ldlt"); return x

    def optimize_vio_own_solver(self, its=6, HMForGTSAM=None, bMForGTSAM=None):
        """dmvio_hip_ba_optimize_vio with dmvio_hip_ba_hook_ldlt (the library's own solve, a C function) as computeBAUpdate: the hand-off path without Python in the loop."""
        n = self.n
        cb = BACallbacks()
        cb.computeBAUpdate = C.cast(self.L.dmvio_hip_ba_hook_ldlt, _BA_COMPUTE)
        opt = BAVioOptions(1, 0, -1, -1, None, None)
        keep = []
        if HMForGTSAM is not None:
            HMg = np.ascontiguousarray(HMForGTSAM, dtype=np.float64).reshape(n, n); bMg = np.ascontiguousarray(bMForGTSAM, dtype=np.float64).reshape(n)
            keep = [HMg, bMg]; opt.HMForGTSAM = _d(HMg); opt.bMForGTSAM = _d(bMg)
        rm = C.c_float(0); fe = C.c_double(0); it = C.c_int(0); tr = np.zeros((64, 4))
        _chk(self.L, self.L.dmvio_hip_ba_optimize_vio(self.p, its, C.byref(cb), C.byref(opt), C.byref(rm), C.byref(fe), C.byref(it), _d(tr)), "ba_optimize_vio")
        return dict(rmse=rm.value, finalEnergy=fe.value, iterations=it.value, trace=tr[:it.value + 1])

    def optimize_vio(self, its, hooks, trackingWasGood=True, updateDuring=False, minOptIterations=-1, resInA_at_entry=-1, HMForGTSAM=None, bMForGTSAM=None):
        """FullSystem::optimize with the reference's default (GTSAM) solver branch.  `hooks`: an object with the methods of dmvio::BAGTSAMIntegration the loop calls —
        computeBAUpdate(HPassed, b, lam, HNoLambda, frames, calib) -> x (required), and optionally acceptBAUpdate(E), getBAEnergy(useNew), updateBAValues(frames, calib),
        updateDynamicWeight(E, rmse, good), canBreak(), postOptimization(frames, calib); frames = list of dicts (frameID, PRE_worldToCam, evalPT, state, state_zero)."""
        n = self.n
        err = []

        def views(F, fr):
            return [dict(frameID=fr[k].frameID, index=fr[k].index, PRE_worldToCam=np.array(fr[k].PRE_worldToCam7[:]), evalPT=np.array(fr[k].worldToCam_evalPT7[:]),
                         state=np.array(fr[k].state10[:]), state_zero=np.array(fr[k].state_zero10[:])) for k in range(F)]

        def compute(_u, nn, HP, bP, lam, HN, F, fr, calib, xo):
            try:       # never unwind through the C frames
                x = hooks.computeBAUpdate(np.ctypeslib.as_array(HP, shape=(nn, nn)).copy(), np.ctypeslib.as_array(bP, shape=(nn,)).copy(), lam,
                                          np.ctypeslib.as_array(HN, shape=(nn, nn)).copy(), views(F, fr), np.ctypeslib.as_array(calib, shape=(4,)).copy())
                np.ctypeslib.as_array(xo, shape=(nn,))[:] = x
                return 0
            except Exception as e:
                err.append(e); return 1

        def guard(fn, default=None):
            def g(*a):
                try:
                    return fn(*a)
                except Exception as e:
                    err.append(e); return default
            return g
        has = lambda name: getattr(hooks, name, None) is not None
        cb = BACallbacks()
        cb.user = None
        cb.computeBAUpdate = _BA_COMPUTE(compute)
        if has("acceptBAUpdate"): cb.acceptBAUpdate = _BA_ACCEPT(guard(lambda _u, e: hooks.acceptBAUpdate(e)))
        if has("getBAEnergy"): cb.getBAEnergy = _BA_ENERGY(guard(lambda _u, new: float(hooks.getBAEnergy(bool(new))), 0.0))
        if has("updateBAValues"): cb.updateBAValues = _BA_VALUES(guard(lambda _u, F, fr, c: hooks.updateBAValues(views(F, fr), np.ctypeslib.as_array(c, shape=(4,)).copy())))
        if has("updateDynamicWeight"): cb.updateDynamicWeight = _BA_WEIGHT(guard(lambda _u, e, r, g: float(hooks.updateDynamicWeight(e, r, bool(g))), 1.0))
        if has("canBreak"): cb.canBreak = _BA_BREAK(guard(lambda _u: 1 if hooks.canBreak() else 0, 0))
        if has("postOptimization"): cb.postOptimization = _BA_VALUES(guard(lambda _u, F, fr, c: hooks.postOptimization(views(F, fr), np.ctypeslib.as_array(c, shape=(4,)).copy())))
        opt = BAVioOptions(1 if trackingWasGood else 0, 1 if updateDuring else 0, int(minOptIterations), int(resInA_at_entry), None, None)
        keep = []
        if HMForGTSAM is not None:
            HMg = np.ascontiguousarray(HMForGTSAM, dtype=np.float64).reshape(n, n); bMg = np.ascontiguousarray(bMForGTSAM, dtype=np.float64).reshape(n)
            keep = [HMg, bMg]; opt.HMForGTSAM = _d(HMg); opt.bMForGTSAM = _d(bMg)
        rm = C.c_float(0); fe = C.c_double(0); it = C.c_int(0); tr = np.zeros((64, 4))
        rc = self.L.dmvio_hip_ba_optimize_vio(self.p, its, C.byref(cb), C.byref(opt), C.byref(rm), C.byref(fe), C.byref(it), _d(tr))
        if err:
            raise err[0]
        _chk(self.L, rc, "ba_optimize_vio")
        return dict(rmse=rm.value, finalEnergy=fe.value, iterations=it.value, trace=tr[:it.value + 1])

    def optimize(self, its=6):
        rm = C.c_float(0); fe = C.c_double(0); it = C.c_int(0); tr = np.zeros((64, 4))
        _chk(self.L, self.L.dmvio_hip_ba_optimize(self.p, its, C.byref(rm), C.byref(fe), C.byref(it), _d(tr)), "ba_optimize")
        return dict(rmse=rm.value, finalEnergy=fe.value, iterations=it.value, trace=tr[:it.value + 1])

    def comm_timing(self, on=True):
        fn = self.L.dmvio_hip_ba_comm_timing; fn.argtypes = [C.c_void_p, C.c_int]; fn.restype = C.c_int
        _chk(self.L, fn(self.p, 1 if on else 0), "ba_comm_timing")

    def comm_times(self):
        """mean us of [all-reduce, all-gather] of the sharded iteration (HIP events on the BA stream) and how many of each were issued"""
        us = (C.c_double * 2)(); nn = (C.c_long * 2)()
        fn = self.L.dmvio_hip_ba_comm_times; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]; fn.restype = C.c_int
        _chk(self.L, fn(self.p, us, nn), "ba_comm_times")
        return dict(allreduce_us=float(us[0]), allgather_us=float(us[1]), allreduces=int(nn[0]), allgathers=int(nn[1]))

    def set_device_loop(self, on=True):
        """dmvio_hip_ba_optimize through the device-resident Gauss-Newton loop (solve, frame step, accept test on the device; a batch of one window)."""
        fn = self.L.dmvio_hip_ba_set_device_loop; fn.argtypes = [C.c_void_p, C.c_int]; fn.restype = C.c_int
        _chk(self.L, fn(self.p, 1 if on else 0), "ba_set_device_loop")

    def last_x(self):
        x = np.zeros(self.n)
        fn = self.L.dmvio_hip_ba_get_last_x; fn.argtypes = [C.c_void_p, C.c_void_p]; fn.restype = C.c_int
        _chk(self.L, fn(self.p, x.ctypes.data), "ba_get_last_x")
        return x
