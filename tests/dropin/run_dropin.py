"""Runs the REFERENCE'S OWN FullSystem (oracle/_ref/libref.so: FullSystem.cpp, FullSystemOptimize.cpp, ... compiled unmodified) over a synthetic sequence with
oracle/_ref/libdropin_hip.so (tests/dropin/dmvio_hip_adapter.cpp) loaded in front of it, and writes what came out to an .npz:

    python tests/dropin/run_dropin.py --mode cpu|hip|plain --out run.npz [--w 256 --h 192 --frames 62 --step 1.6 --density 300 --accumulators 0]

  plain: libref.so alone (no adapter in the process);
  cpu:   the adapter loaded and interposed, switched off — its five members forward to the reference's own definitions;
  hip:   makeImages / setCoarseTrackingRef / trackNewestCoarse / traceNewCoarse / optimize run on libdmvio_hip.so (needs the MI355X).
Test infrastructure (tests/test_dropin_*.py, bench.py's drop_in leg); one mode per process because symbol interposition is decided at load time."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402

DROPIN = os.path.join(ROOT, "oracle", "_ref", "libdropin_hip.so")


def main():
    # The reference reads heap memory it never wrote: PixelSelector::select (PixelSelector2.cpp:418) looks at rows 0 and h-1 of FrameHessian::absSquaredGrad, which makeImages
    # allocates with new[] and fills for rows 1..h-2 only (HessianBlocks.cpp:134, 169-189) — found with MemorySanitizer on the sources compiled by oracle/Makefile.ref.  What a
    # run selects therefore depends on what the allocator hands out.  glibc's MALLOC_PERTURB_=255 makes every fresh allocation zero-filled: the same memory image in every mode.
    if os.environ.get("MALLOC_PERTURB_") != "255":
        os.environ["MALLOC_PERTURB_"] = "255"
        os.execv(sys.executable, [sys.executable] + sys.argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["plain", "cpu", "hip"], required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--w", type=int, default=256); ap.add_argument("--h", type=int, default=192)
    ap.add_argument("--frames", type=int, default=62); ap.add_argument("--step", type=float, default=1.6)
    ap.add_argument("--density", type=int, default=300); ap.add_argument("--accumulators", type=int, default=0)
    ap.add_argument("--result-txt", default=None)
    ap.add_argument("--shadow", action="store_true", help="mode hip: the reference's own members run the pipeline, libdmvio_hip.so runs every call beside them from the same "
                                                         "inputs; the deviations per call are recorded (no drift through flipped discrete decisions)")
    ap.add_argument("--brightness", action="store_true", help="exposure times that change from frame to frame (handed to addActiveFrame, images scaled accordingly) plus an affine "
                                                             "brightness drift the exposure does not explain: the AffLight / exposure path of tracker, tracer and BA on live data")
    ap.add_argument("--window", action="store_true", help="instead of the whole FullSystem: one window through the reference's members one by one (makeImages, traceNewCoarse, "
                                                         "optimize, setCoarseTrackingRef + trackNewestCoarse) — single-threaded code only, deterministic")
    ap.add_argument("--cache", default=None, help="directory that keeps the rendered sequence between runs (rendering 512x512 frames costs more than tracking them)")
    ap.add_argument("--init", choices=["ref", "seq", "hip"], default="ref",
                    help="CoarseInitializer::calcResAndGS: the reference's own (multi-threaded: run-to-run noise), the oracle's single-threaded restatement (deterministic "
                         "CPU baseline), or libdmvio_hip.so (mode hip only)")
    ap.add_argument("--realtime", type=float, default=0.0, metavar="FPS",
                    help="FullSystem(linearizeOperation = false): frames arrive at FPS on this thread (tracking), the reference's own mapping thread makes the keyframes — "
                         "tracking and mapping overlap like in a live system; nothing is recorded, the run is not reproducible bit for bit")
    ap.add_argument("--resident", type=int, default=1, choices=[0, 1, 2],
                    help="mode hip: how FullSystem::optimize hands the window over — 0 the pointer graph flattened every keyframe, 1 the resident window graph (the forwarded "
                         "EnergyFunctional mutators keep it in step), 2 both, compared element for element")
    ap.add_argument("--real-marg", type=int, default=1, choices=[0, 1],
                    help="mode hip, resident window: EnergyFunctional::marginalizePointsF accumulates on the device from the window optimize left there (1) or stays the reference's own (0)")
    ap.add_argument("--batch-tries", type=int, default=1, choices=[0, 1],
                    help="mode hip: FullSystem::trackNewCoarse hands its try loop to dmvio_hip_tracker_track_new_coarse (1: try 0 alone, the other motion hypotheses as one device "
                         "batch, the sequential rule replayed) or stays the reference's own loop with one trackNewestCoarse per try (0)")
    ap.add_argument("--jump", default=None, metavar="K:N[,K:N...]",
                    help="forced relocalisation: N frames are dropped from the sequence in front of frame K — the motion-model guess of that frame is off by N frame steps, so "
                         "trackNewCoarse's walk goes past try 0")
    ap.add_argument("--vio", action="store_true",
                    help="the reference's DEFAULT configuration: setting_useIMU / setting_useGTSAMIntegration on, a live stand-in for the (absent) IMU / GTSAM side behind the "
                         "facade's hooks (oracle/ref_glue.cpp: VioStandIn) — trackNewestCoarse takes its computeCoarseUpdate branch once the stand-in declares the IMU "
                         "initialised, optimize its computeBAUpdate / getBAEnergy / acceptBAUpdate / canBreak branch from the first keyframe on; mode hip: through "
                         "dmvio_hip_tracker_track_vio / dmvio_hip_ba_optimize_vio with the facade's members as callbacks")
    ap.add_argument("--vio-init-after", type=int, default=3, help="--vio: keyframe optimisations after which the stand-in declares the IMU initialised")
    ap.add_argument("--mt", action="store_true", help="settings.cpp multiThreading = true (the reference's default: linearizeAll, applyRes, the accumulators on 6 workers)")
    ap.add_argument("--scopes", action="store_true", help="inclusive wall time per profiler label of the reference (util/TimeMeasurement scopes) for this run; switches the "
                                                           "event recording of the run off, so wall_s is the pipeline alone")
    a = ap.parse_args()
    pkg = graft.load_package()
    import dmvio_amd.synth as synth
    D = None
    if a.mode != "plain":
        if a.mode == "hip":
            pkg.load_library()                       # torch first (one HIP runtime per process), then libdmvio_hip.so
        D = C.CDLL(DROPIN, mode=C.RTLD_GLOBAL)       # BEFORE libref.so is looked at: its definitions of the five members come first in the lookup order
        D.dropin_enable.argtypes = [C.c_int] * 5
        D.dropin_attach.argtypes = [C.c_void_p]
        D.dropin_set_initializer.argtypes = [C.c_int, C.c_char_p]
        D.dropin_set_shadow.argtypes = [C.c_int]; D.dropin_get_shadow.argtypes = [C.POINTER(C.c_double)]
        D.dropin_get_stats.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_long)]
        D.dropin_failures.argtypes = [C.c_char_p, C.c_int]; D.dropin_failures.restype = C.c_long
    import ref_py as R
    import replay
    if a.window:
        return window_run(a, D, R, synth)
    cache = os.path.join(a.cache, "seq_%dx%d_%d_%g.npz" % (a.w, a.h, a.frames, a.step)) if a.cache else None
    if cache and os.path.exists(cache):
        z = np.load(cache); K4, imgs, poses_true = z["K4"], list(z["imgs"]), list(z["poses"])
    else:
        K4, imgs, poses_true = replay.make_sequence(synth, a.w, a.h, a.frames, a.step)
        if cache:
            os.makedirs(a.cache, exist_ok=True)
            np.savez(cache + ".tmp.npz", K4=np.asarray(K4), imgs=np.asarray(imgs, np.float32), poses=np.asarray(poses_true)); os.replace(cache + ".tmp.npz", cache)
    if a.jump:
        drop = set()
        for item in a.jump.split(","):
            k, n = (int(x) for x in item.split(":"))
            drop.update(range(k, k + n))
        imgs = [im for i, im in enumerate(imgs) if i not in drop]; poses_true = [p for i, p in enumerate(poses_true) if i not in drop]
    if D is not None and D.dropin_enable(1 if a.mode == "hip" else 0, 0, a.w, a.h, a.accumulators) != 0:
        raise SystemExit("dropin_enable failed")
    if D is not None:
        D.dropin_set_resident.argtypes = [C.c_int]; D.dropin_set_resident(a.resident)
        D.dropin_set_real_marginalization.argtypes = [C.c_int]; D.dropin_set_real_marginalization(a.real_marg)
        D.dropin_set_batch_tries.argtypes = [C.c_int]; D.dropin_set_batch_tries(a.batch_tries)
    # the reference draws from the C library's rand() (PixelSelector's random pattern, CoarseInitializer's point selection): the same sequence in every mode, whatever the
    # static initialisers of the libraries loaded so far have consumed
    C.CDLL(None).srand(1)
    if a.shadow:
        if a.mode != "hip":
            raise SystemExit("--shadow needs --mode hip")
        D.dropin_set_shadow(1)
    if a.init != "ref":
        if D is None or (a.init == "hip" and a.mode != "hip"):
            raise SystemExit("--init %s needs the adapter (mode cpu / hip)" % a.init)
        graft.load_oracle().lib()     # builds oracle/_build/liboracle.so when missing
        if D.dropin_set_initializer({"seq": 1, "hip": 2}[a.init], os.path.join(ROOT, "oracle", "_build", "liboracle.so").encode()) != 0:
            raise SystemExit("dropin_set_initializer failed")
    if a.realtime > 0:
        R.lib().ref_set_linearize_operation(0)
    if a.vio:
        R.lib().ref_set_live_vio.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double]
        R.lib().ref_set_live_vio(1, a.vio_init_after, 0.02, 1e3)
    S = R.System(a.w, a.h, K4, point_density=a.density)
    if a.vio and D is not None:
        # shadow mode: both sides of a call go through the same stateful stand-in; the adapter saves / restores it around the device's call
        L = R.lib()
        D.dropin_set_vio_state_hooks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        D.dropin_set_vio_state_hooks(S.p, C.cast(L.ref_system_vio_save, C.c_void_p), C.cast(L.ref_system_vio_restore, C.c_void_p))
    if D is not None:
        R.lib().ref_system_fullsystem.restype = C.c_void_p; R.lib().ref_system_fullsystem.argtypes = [C.c_void_p]
        D.dropin_attach(R.lib().ref_system_fullsystem(S.p))
    if a.mt:
        R.set_multithreading(True)
    if a.scopes:
        R.lib().ref_system_scope_timing.argtypes = [C.c_void_p, C.c_int, C.c_int]
        R.lib().ref_system_scope_timing(S.p, 1, 0)
    expo = [1.0] * len(imgs)
    if a.brightness:
        expo = [float(1.0 + 0.2 * np.sin(0.13 * k)) for k in range(len(imgs))]
        imgs = [np.float32(expo[k] * (1.0 + 0.03 * np.sin(0.31 * k))) * img + np.float32(4.0 * np.sin(0.2 * k)) for k, img in enumerate(imgs)]
    t0 = time.perf_counter()
    if a.realtime > 0:
        R.lib().ref_system_unmapped.argtypes = [C.c_void_p]; R.lib().ref_system_finish.argtypes = [C.c_void_p]
        status = []; busy = 0.0
        for k, img in enumerate(imgs):
            due = t0 + k / a.realtime
            while time.perf_counter() < due:
                time.sleep(2e-4)
            t1 = time.perf_counter()
            status.append(S.add_frame(img, exposure=expo[k]))
            busy += time.perf_counter() - t1
        t_end = time.perf_counter() + 5.0
        while R.lib().ref_system_unmapped(S.p) > 0 and time.perf_counter() < t_end:      # let the mapping thread take what is queued, as a live system would
            time.sleep(1e-3)
        time.sleep(0.05)
        R.lib().ref_system_finish(S.p)                                                    # FullSystem::blockUntilMappingIsFinished
        wall = busy                                                                       # seconds inside addActiveFrame (the tracking thread's share of the frame period)
    else:
        status = [S.add_frame(img, exposure=expo[k]) for k, img in enumerate(imgs)]
        wall = time.perf_counter() - t0
    tr = S.trajectory()
    ev = S.events()
    opt = [e for e in ev if e["kind"] == "opt_out"]
    opt_in = [e for e in ev if e["kind"] == "opt_in"]
    # what the initialiser handed over (its calcResAndGS always runs on NUM_THREADS workers that take 50-point chunks as they come, CoarseInitializer.cpp:507 /
    # util/IndexThreadReduce.h:83-87, whatever `multiThreading` says: the reference's own result varies in the last bits from run to run, and with it every discrete
    # decision downstream).  Runs with the same signature started from the same initialisation; everything after it is single-threaded and deterministic.
    import hashlib
    sig = hashlib.md5(np.ascontiguousarray(opt_in[0]["idepth"]).tobytes() + np.ascontiguousarray(opt_in[0]["frames"][1]["evalPT"]).tobytes()).hexdigest() if opt_in else ""
    out = dict(init_signature=np.array([sig]), camToWorld=tr["camToWorld"], valid=tr["valid"], keyframeId=tr["keyframeId"], trackingRef=tr["trackingRef"], aff=tr["aff"],
               poses_true=np.array(poses_true), wall_s=np.array([wall]), initialized=np.array([s["initialized"] for s in status]), lost=np.array([s["isLost"] for s in status]),
               window=np.array([s["window"] for s in status]), opt_rmse=np.array([e["rmse"] for e in opt]), opt_resInA=np.array([e["resInA"] for e in opt]),
               opt_F=np.array([e["F"] for e in opt_in]), opt_N=np.array([e["N"] for e in opt_in]), opt_R=np.array([e["R"] for e in opt_in]),
               n_tracks=np.array([sum(1 for e in ev if e["kind"] == "track_out")]))
    if a.vio:
        vc = (C.c_double * 16)()
        R.lib().ref_system_vio_counters.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        R.lib().ref_system_vio_counters(S.p, vc)
        # addIMUData, computeCoarseUpdate, acceptCoarseUpdate, addVisualToCoarseGraph, updateBAValues, computeBAUpdate, getBAEnergy, acceptBAUpdate, updateDynamicWeight, canBreak,
        # postOptimization, addMarginalizedPointsBA; sum of coarse |inc|, sum of max |x| of the BA updates, last factor energy, coarseInitialized
        out["vio_counters"] = np.array(list(vc))
        print("vio stand-in:", [int(x) for x in vc[:12]], ["%.6g" % x for x in vc[12:]])
        if D is not None:
            dv = (C.c_long * 4)(); D.dropin_get_vio.argtypes = [C.POINTER(C.c_long)]; D.dropin_get_vio(dv)
            out["vio_adapter"] = np.array(list(dv))      # track_vio calls with computeCoarseUpdate / with the visual step, optimize_vio calls, facade members called from callbacks
    if D is not None:
        sec = (C.c_double * 7)(); calls = (C.c_long * 7)()
        D.dropin_get_stats(sec, calls)
        if a.shadow:
            sh = (C.c_double * 16)(); D.dropin_get_shadow(sh)
            out["shadow"] = np.array(list(sh))
            act = (C.c_long * 2)(); D.dropin_get_shadow_activation.argtypes = [C.POINTER(C.c_long)]; D.dropin_get_shadow_activation(act)
            out["shadow_activation"] = np.array(list(act))
            mg = (C.c_double * 6)(); D.dropin_get_shadow_marginalization.argtypes = [C.POINTER(C.c_double)]; D.dropin_get_shadow_marginalization(mg)
            out["shadow_marginalization"] = np.array(list(mg))
            mg2 = (C.c_double * 3)(); D.dropin_get_shadow_marginalization2.argtypes = [C.POINTER(C.c_double)]; D.dropin_get_shadow_marginalization2(mg2)
            out["shadow_marginalization2"] = np.array(list(mg2))
        out["stat_seconds"] = np.array(list(sec)); out["stat_calls"] = np.array(list(calls))
        tn = (C.c_double * 12)(); D.dropin_get_track_new_coarse.argtypes = [C.POINTER(C.c_double)]; D.dropin_get_track_new_coarse(tn)
        # trackNewCoarse calls, served by the batched try loop, of those past try 0, tries walked; shadow: compared, past try 0 (reference), tries differ, verdict differs, most tries
        # in one call; max deviation of the winning pose (m), aff a, achieved level-0 residual (relative)
        out["track_new_coarse"] = np.array(list(tn))
        sp = (C.c_double * 3)(); D.dropin_get_optimize_split.argtypes = [C.POINTER(C.c_double)]; D.dropin_get_optimize_split(sp)
        out["optimize_split_seconds"] = np.array(list(sp))      # flatten + upload, dmvio_hip_ba_optimize, write-back
        us = (C.c_double * 6)(); D.dropin_get_upload_split.argtypes = [C.POINTER(C.c_double)]; D.dropin_get_upload_split(us)
        out["upload_split_seconds"] = np.array(list(us))        # frame tables, graph walk, set_window, set_graph(_from), frame states / thresholds / calibration, marginalisation prior
        ws = (C.c_double * 5)(); D.dropin_get_writeback_split.argtypes = [C.POINTER(C.c_double)]; D.dropin_get_writeback_split(ws)
        out["writeback_split_seconds"] = np.array(list(ws))     # calibration + keyframe states + adjoints / precalc, downloads, per-point pass, removals, tail
        rm = (C.c_long * 2)(); D.dropin_get_real_marginalization.argtypes = [C.POINTER(C.c_long)]; D.dropin_get_real_marginalization(rm)
        out["real_marginalization"] = np.array(list(rm))        # marginalizePointsF calls whose accumulation ran on the device, points marginalised there
        rs = (C.c_long * 4)(); D.dropin_get_resident.argtypes = [C.POINTER(C.c_long)]; D.dropin_get_resident(rs)
        out["resident"] = np.array(list(rs))                    # forwarded EnergyFunctional mutations, resyncs, keyframes verified against the flattened graph, mismatches
        msg = C.create_string_buffer(512)
        out["failures"] = np.array([D.dropin_failures(msg, 512)])
        if out["failures"][0]:
            print("adapter failures:", msg.value.decode(errors="replace"))
    if a.result_txt:
        S.print_result(a.result_txt)
    if a.scopes:
        buf = C.create_string_buffer(1 << 16)
        R.lib().ref_system_scope_totals.argtypes = [C.c_char_p, C.c_int]
        R.lib().ref_system_scope_totals(buf, len(buf))
        rows = [ln.rsplit(" ", 2) for ln in buf.value.decode().splitlines() if ln.strip()]
        rows.sort(key=lambda r: -float(r[1]))
        out["scope_labels"] = np.array([r[0] for r in rows]); out["scope_seconds"] = np.array([float(r[1]) for r in rows]); out["scope_calls"] = np.array([int(r[2]) for r in rows])
        for r in rows:
            print("scope %-40s %9.4f s %6d calls" % (r[0], float(r[1]), int(r[2])))
    np.savez(a.out, **out)
    print("signature", sig)
    print("%s: %d frames in %.2f s, %d keyframe optimisations, initialised %s, lost %s" % (a.mode, len(imgs), wall, len(opt), status[-1]["initialized"], status[-1]["isLost"]))


def window_run(a, D, R, synth):
    """The members the adapter replaces, called one by one on a synthetic window through oracle/ref_py.py (no initialiser: nothing multi-threaded)."""
    if D is not None and D.dropin_enable(1 if a.mode == "hip" else 0, 0, 256, 192, a.accumulators) != 0:
        raise SystemExit("dropin_enable failed")
    if D is not None:
        D.dropin_set_resident.argtypes = [C.c_int]; D.dropin_set_resident(a.resident)
    case = synth.ba_case(256, 192, n_frames=4, n_points=150, hosts_share=(60, 50, 40, 0), seed=7)
    W = R.BAWindow(case)                                          # FrameHessian::makeImages per keyframe
    rng = np.random.RandomState(3)
    u, v = synth.select_points(case["imgs"][0], 200, rng, min_grad=8.0)
    u = np.clip(u.astype(np.int32), 8, 256 - 9); v = np.clip(v.astype(np.int32), 8, 192 - 9)
    for host in range(3):
        W.immature_add(host, u, v)
    W.trace_new_coarse(3)                                         # FullSystem::traceNewCoarse
    imm = W.immature_get(0)
    ro = W.optimize_quiet(6)                                      # FullSystem::optimize
    poses = np.stack([W.frame_pose(k)[0] for k in range(4)]); idepth = W.point_state()[0]
    tc = synth.tracking_case(256, 256, n_ref=400, n_frames=1)
    T = R.Tracker(256, 256, tc["K4"])
    T.set_ref(tc["ref_img"], tc["u"], tc["v"], tc["idepth"], tc["hdiF"])      # CoarseTracker::setCoarseTrackingRef
    T.set_new(tc["frames"][0]["img"])
    tr = T.track(np.array([0, 0, 0, 0, 0, 0, 1.0]), [0.0, 0.0])               # CoarseTracker::trackNewestCoarse
    out = dict(rmse=np.array([ro]), poses=poses, idepth=idepth, imm_min=np.asarray(imm["idepth_min"]), imm_max=np.asarray(imm["idepth_max"]),
               imm_status=np.asarray(imm["lastTraceStatus"]), track_pose=np.asarray(tr["pose7"]), track_res=np.asarray(tr["lastResiduals"]))
    if D is not None:
        sec = (C.c_double * 7)(); calls = (C.c_long * 7)()
        D.dropin_get_stats(sec, calls)
        out["stat_calls"] = np.array(list(calls))
        out["failures"] = np.array([D.dropin_failures(None, 0)])
    np.savez(a.out, **out)
    print("%s window: rmse %.6f" % (a.mode, ro))


if __name__ == "__main__":
    main()
