#!/usr/bin/env python
"""bench.py — tracked frames/s of the photometric direct-alignment hot path on N MI355X.

One "step" = one batch of B independent new frames aligned against the current reference keyframe:
per frame the image pyramid is built on the device from the resident irradiance image (FrameHessian::makeImages:
the image itself serves as level 0 — the library keeps intensity planes only and rebuilds the gradient channels at the
taps — the coarser levels are reduced from it) and the full 4-level CoarseTracker::trackNewestCoarse LM loop runs
device-resident (calcRes + calcGSSSE fused).  Inputs are resident in HBM before the timed region.

Workload (BASELINE.json configs[1]): synthetic 512x512 4-level pyramid, ~2000 reference points
(TUM-VI rectified intrinsics), plane-world renderer, seed 20250204.

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank aligns its own batch against its
own replica of the reference (weak scaling, no data-path collective — coarse tracking of independent frames
does not shard); timing is barrier + synchronize bracketed and MAX-reduced.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
SETTLE_STEPS = 40              # untimed steps between the warm-up and the timed region (see main)
BYTES_PER_POINT_EVAL = 64      # 16 B template record + 4 taps x 12 B


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("DMVIO_BENCH_BATCH", "4096")),
                    help="frames per step per GPU (4096 = four rounds of the 1024 resident workgroup slots: the launch's tail — one slow frame per slot at 1024 — is amortised)")
    ap.add_argument("--points", type=int, default=2000, help="reference points (active points of the window)")
    ap.add_argument("--distinct", type=int, default=0, help="distinct rendered frames (0 = one per batch slot: every frame of the batch is its own render at its own pose)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the batch-size sweep (256 ... 4096 frames per step)")
    ap.add_argument("--no-pcie", action="store_true", help="skip the leg that uploads the raw frames from pinned host memory inside the step")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc child run that measures the HBM traffic of k_track_lm")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU budget of the cpu_baseline leg (rank 0, N=1 only)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-pyramid", action="store_true", help="exclude makeImages from the step (track only)")
    ap.add_argument("--build-overlap", action="store_true", help="TIME the variant that builds the pyramids of batch k+1 on a stream of their own while batch k is tracked (two "
                                                                  "slot sets) instead of the build on the tracking stream")
    ap.add_argument("--build-overlap-leg", action="store_true", help="measure that variant AFTER the timed region (same K steps) and report it beside the headline "
                                                                      "(pipeline.ms_per_step_build_on_its_own_stream); off by default: its launches would otherwise sit in the same "
                                                                      "rows of a rocprofv3 kernel summary as the timed region's, stretched by the concurrency")
    ap.add_argument("--template-order", type=int, default=-1, help="measurement: storage order of the tracker's template (0 tiles, 1 row-major; -1 = the library's default)")
    ap.add_argument("--batch-kernel", type=int, default=-1, help="measurement: dmvio_hip_tracker_set_batch_kernel (-1 = the library's default)")
    ap.add_argument("--no-ba", action="store_true", help="skip the bundle-adjustment leg (BA GN-iterations/s)")
    ap.add_argument("--no-dropin", action="store_true", help="skip the drop_in leg (the reference's own FullSystem all-CPU vs with its hot-path members on libdmvio_hip.so)")
    ap.add_argument("--dropin-frames", type=int, default=100)
    ap.add_argument("--no-concurrent", action="store_true", help="skip ba.concurrent_windows (several windows in flight on one GPU)")
    ap.add_argument("--ba-points", type=int, default=2000)
    ap.add_argument("--ba-iters", type=int, default=300, help="timed GN iterations of the BA leg")
    return ap.parse_args()


_REAL_STDOUT = None


def emit(line):
    """The ONE line of the contract, on the process's real stdout."""
    if _REAL_STDOUT is None:
        print(line, flush=True)
    else:
        os.write(_REAL_STDOUT, (line + "\n").encode())


def self_launch(args):
    """`python bench.py --gpus N` without a launcher (the driver's plain command form): start N ranks of this very command under torch.distributed.run — one process per
    GPU, rendezvous on 127.0.0.1 — and hand their exit code on.  Rank 0 of the children prints the one JSON line (the children inherit this process's stdout).  Fewer than
    N visible devices is an error, never a silent 1-GPU number (DMVIO_BENCH_SHARE_DEVICE=1, the one-GPU test hook, lets the ranks share devices)."""
    import socket
    import subprocess
    if not os.environ.get("DMVIO_BENCH_SHARE_DEVICE"):
        import torch
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus:
            sys.stderr.write("bench: --gpus %d asked for, %d device(s) visible: refusing to report a %d-GPU number as %d GPUs\n" % (args.gpus, n_dev, n_dev, args.gpus))
            raise SystemExit(2)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env["DMVIO_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("bench: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1 and not args.traffic_child:
            self_launch(args)
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit("bench: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (args.gpus, os.environ["WORLD_SIZE"]))
    # stdout carries exactly one JSON line.  The CPU baselines run the reference's own compiled sources (oracle/_ref/libref.so), which print from C++ (PixelSelector's block
    # sizes, "destroyed ThreadReduce" from static destructors at exit, ...): file descriptor 1 is pointed at stderr for the whole run, the JSON line goes to a saved copy of it
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # test hooks for single-GPU boxes: DMVIO_BENCH_BACKEND=gloo + DMVIO_BENCH_SHARE_DEVICE=1 run the N>1 code path with all ranks on GPU 0
    backend = os.environ.get("DMVIO_BENCH_BACKEND", "nccl")
    if os.environ.get("DMVIO_BENCH_SHARE_DEVICE"):
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")

    pkg = graft.load_package()
    import dmvio_amd.synth as synth

    w = h = args.size
    B = args.batch
    # ---------------- synthetic inputs (deterministic; rank-dependent jitter so ranks do not share data): every frame of the batch is its own render
    # of the plane world at its own pose (rendered on the device with torch: plumbing)
    if args.distinct <= 0:
        args.distinct = B
    case = synth.tracking_case(w, h, n_ref=args.points, seed=synth.SEED + 1000 * rank, n_frames=1, xi_jitter=0.35)
    rngx = np.random.RandomState(synth.SEED + 1000 * rank + 2)
    xi0 = case["frames"][0]["xi"]
    frames_meta = []
    for k in range(args.distinct):
        xi = xi0 if k == 0 else xi0 * (1.0 + 0.35 * rngx.standard_normal(6))
        R, t = synth.se3_exp(xi)
        frames_meta.append(dict(R=R, t=t, pose7=synth.pose7(R, t), xi=xi))
    overlap_build = args.build_overlap and not args.no_pyramid
    overlap_leg = (args.build_overlap or args.build_overlap_leg) and not args.no_pyramid
    # slot 0: reference keyframe, 1..B: batch (slot set 0), [B+1..2B: slot set 1 of the double-buffered pipeline variant,] then 8 slots for the BA window of the overlap leg
    args.ba_slot0 = (2 * B if overlap_leg else B) + 1
    ctx = pkg.Context(w, h, n_slots=args.ba_slot0 + 8, device=local_rank)
    stream = torch.cuda.Stream(device=dev)
    ctx.set_stream(stream.cuda_stream)
    trk = pkg.CoarseTrackerHip(ctx)
    if args.template_order >= 0:
        trk.set_template_order(args.template_order)
    if args.batch_kernel >= 0:
        trk.set_batch_kernel(args.batch_kernel)
    trk.makeK(case["K4"])
    ctx.frame_upload(0, case["ref_img"])
    trk.setCoarseTrackingRef(0, case["u"], case["v"], case["idepth"], case["hdiF"])
    pc_n = [trk.pc_n(l) for l in range(ctx.levels)]
    # resident raw irradiance images of the batch (device memory via torch: plumbing only)
    raw = torch.empty((B, h, w), dtype=torch.float32, device=dev)   # attached in place: level 0 of each pyramid IS this resident image
    raw_distinct = synth.render_batch_torch(case["world"], case["K4"], [f["R"] for f in frames_meta], [f["t"] for f in frames_meta], w, h, dev)
    idx = torch.arange(B, device=dev) % args.distinct
    raw.copy_(raw_distinct[idx])
    n_host = min(args.distinct, 64)                                   # host copies for the CPU baseline and the single-stream legs
    host_frames = raw_distinct[:n_host].cpu().numpy()
    for k in range(n_host):
        frames_meta[k]["img"] = host_frames[k]
    case["frames"] = frames_meta
    if args.distinct < B:
        del raw_distinct
    torch.cuda.synchronize(dev)
    slots = np.arange(1, B + 1, dtype=np.int32)
    rng = np.random.RandomState(99 + rank)
    # initial guesses: identity + small perturbation (the constant-motion guess of trackNewCoarse is near the truth)
    poses0 = np.zeros((B, 7)); poses0[:, 6] = 1.0
    for i in range(B):
        R, t = synth.se3_exp(rng.normal(0, 0.002, 6))
        poses0[i] = synth.pose7(R, t)
    affs0 = np.zeros((B, 2))
    raw_ptr = raw.data_ptr()
    frame_bytes = w * h * 4

    ctx.frames_attach_device_batch(slots, raw_ptr, frame_bytes)   # --no-pyramid runs track against these

    def step(fetch=True):
        if not args.no_pyramid:
            ctx.frames_attach_device_batch(slots, raw_ptr, frame_bytes)
        trk.stage(slots, poses0, affs0)
        trk.launch()
        return trk.fetch() if fetch else None

    # Steady-state form of the same step: the download of batch k-1 is queued behind the pyramid build of batch k, batch k is staged
    # and launched, and only then does the host block on (and unpack) the results of batch k-1 — the device never idles while the
    # host unpacks.  Work per step is unchanged: one pyramid build, one launch, one result download + unpack.
    def step_pipelined(have_prev):
        if not args.no_pyramid:
            ctx.frames_attach_device_batch(slots, raw_ptr, frame_bytes)
        if have_prev:
            trk.fetch_begin()
        trk.stage(slots, poses0, affs0)
        trk.launch()
        return trk.fetch() if have_prev else None

    # The pyramid build on a stream of its own (dmvio_hip_set_build_stream), double-buffered over two slot sets: while batch k is tracked (k_track_lm: bound by its L1 miss
    # path and VALU issue) the pyramids of batch k+1 are built beside it (k_build_pyramids_reg: bound by HBM).  Work per step is unchanged — one build, one launch, one
    # result download + unpack —, only the build of the NEXT batch no longer waits for the tracking of this one.  Events order the two streams: tracking of batch k behind
    # its build, the build into a slot set behind the tracking that last read it.
    slot_sets = [slots, np.arange(B + 1, 2 * B + 1, dtype=np.int32)]
    if overlap_leg:
        bstream = torch.cuda.Stream(device=dev)   # default priority: a high-priority build stream was measured slower (4.45-4.50 vs 4.33-4.37 ms per step)
        build_done = [torch.cuda.Event(), torch.cuda.Event()]
        track_done = [torch.cuda.Event(), torch.cuda.Event()]
        pipe = dict(k=0)

        def overlap_prologue():
            ctx.set_build_stream(bstream.cuda_stream)
            ctx.frames_attach_device_batch(slot_sets[0], raw_ptr, frame_bytes)
            build_done[0].record(bstream)
            pipe["k"] = 0

        def step_overlap(have_prev):
            cur = pipe["k"] & 1; nxt = cur ^ 1
            stream.wait_event(build_done[cur])                  # tracking of batch k behind the build of its slot set
            if have_prev:
                trk.fetch_begin()
            trk.stage(slot_sets[cur], poses0, affs0)
            trk.launch()
            track_done[cur].record(stream)
            bstream.wait_event(track_done[nxt])                 # the other slot set was last read by the tracking of batch k-1 (no-op before the first recording)
            ctx.frames_attach_device_batch(slot_sets[nxt], raw_ptr, frame_bytes)   # batch k+1, on the build stream
            build_done[nxt].record(bstream)
            pipe["k"] += 1
            return trk.fetch() if have_prev else None

        def overlap_epilogue():
            torch.cuda.synchronize(dev)
            ctx.set_build_stream(0)

    # ---------------- warmup + correctness guard (poses must reach the ground truth)
    res = None
    for _ in range(max(args.warmup, 1)):
        res = step()
    truth = np.stack([frames_meta[i % args.distinct]["pose7"] for i in range(B)])
    terr = np.linalg.norm(res["pose7"][:, :3] - truth[:, :3], axis=1)
    n_evals, n_point_evals = trk.last_work()
    if not (res["good"].all() and terr.max() < 5e-3):
        raise SystemExit("bench: tracking did not converge (good=%d/%d, max err %.3g m)" % (res["good"].sum(), B, terr.max()))

    # ---------------- timed region: EXACTLY K steps, barrier + synchronize on both sides
    def barrier():
        if dist is not None:
            dist.barrier()
    torch.cuda.synchronize(dev); barrier(); torch.cuda.synchronize(dev)
    # untimed: the device settles (clocks, first touches of the freshly allocated pyramids and result buffers: the first ~40 steps after start-up run up to
    # 2.5x slower), then the pipeline is filled (the results of that launch are unpacked by the first timed step)
    for _ in range(SETTLE_STEPS):
        step()
    timed_step = step_pipelined
    if overlap_build:
        overlap_prologue()                               # the build of the first batch: the pipeline's fill, like the first launch below
        timed_step = step_overlap
    timed_step(False)
    torch.cuda.synchronize(dev); barrier(); torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    step_marks = []
    for _ in range(args.steps):
        res_pipe = timed_step(True)
        step_marks.append(time.perf_counter())          # the step returns once the PREVIOUS launch's results are unpacked: per-step wall time of the steady-state pipeline
    torch.cuda.synchronize(dev); barrier(); torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    step_ms = 1e3 * np.diff(np.array([t0] + step_marks))
    if not res_pipe["good"].all():
        raise SystemExit("bench: a pipelined step lost tracking")
    trk.fetch()                                         # drain the last launch (outside the timed region: K launches, K unpacks inside)
    # the other pipeline variant, after the timed region (the same K steps, the same work per step): reported beside the headline, not in it
    other_ms = None
    if overlap_build:
        terr_pipe = np.linalg.norm(res_pipe["pose7"][:, :3] - truth[:, :3], axis=1)
        if terr_pipe.max() >= 5e-3:
            raise SystemExit("bench: the overlapped pipeline tracked a stale slot set (max err %.3g m)" % terr_pipe.max())
        overlap_epilogue()
        ctx.frames_attach_device_batch(slots, raw_ptr, frame_bytes)
        step_pipelined(False)
        torch.cuda.synchronize(dev)
        t0s = time.perf_counter()
        for _ in range(args.steps):
            step_pipelined(True)
        torch.cuda.synchronize(dev)
        other_ms = 1e3 * (time.perf_counter() - t0s) / args.steps
        trk.fetch()
    elif overlap_leg:
        overlap_prologue()
        step_overlap(False)
        torch.cuda.synchronize(dev)
        t0s = time.perf_counter()
        for _ in range(args.steps):
            r_o = step_overlap(True)
        torch.cuda.synchronize(dev)
        other_ms = 1e3 * (time.perf_counter() - t0s) / args.steps
        if not r_o["good"].all() or np.linalg.norm(r_o["pose7"][:, :3] - truth[:, :3], axis=1).max() >= 5e-3:
            raise SystemExit("bench: the overlapped pipeline lost tracking or tracked a stale slot set")
        trk.fetch()
        overlap_epilogue()
        ctx.frames_attach_device_batch(slots, raw_ptr, frame_bytes)
    rank_ms = [1e3 * elapsed / args.steps]
    if dist is not None:
        tall = [torch.zeros(1, dtype=torch.float64, device=coll_dev) for _ in range(world)]
        dist.all_gather(tall, torch.tensor([elapsed], dtype=torch.float64, device=coll_dev))
        rank_ms = [1e3 * float(t.item()) / args.steps for t in tall]
        elapsed = max(float(t.item()) for t in tall)       # MAX over ranks
    ms_per_step = 1e3 * elapsed / args.steps
    frames_per_s = world * B * args.steps / elapsed

    # ---------------- kernel-level roofline of the dominant kernel (k_track_lm), HIP events on ITS stream
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    reps = max(3, min(args.steps, 10))
    trk.stage(slots, poses0, affs0)
    torch.cuda.synchronize(dev)
    kms = []
    for _ in range(reps):
        ev0.record(stream); trk.launch(); ev1.record(stream)
        ev1.synchronize()
        kms.append(ev0.elapsed_time(ev1))
    trk.fetch()
    n_evals, n_point_evals = trk.last_work()
    tk_step, tk_eval = trk.last_ticks()
    k_ms = float(np.mean(kms))
    alg_bytes = BYTES_PER_POINT_EVAL * n_point_evals
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    roofline = dict(bound="hbm", kernel="k_track_lm", achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 5), traffic=None,
                    kernel_ms=round(k_ms, 4), algorithmic_bytes_per_launch=int(alg_bytes),
                    point_evals_per_launch=int(n_point_evals), evals_per_launch=int(n_evals),
                    in_kernel_us_per_problem=dict(lm_control=round(tk_step / 100.0 / B, 2), evaluation=round(tk_eval / 100.0 / B, 2)))
    # HBM traffic of k_track_lm: measured now, by a rocprofv3 --kernel-trace --pmc FETCH_SIZE child run of this very workload (rank 0, N = 1), calibrated on
    # k_build_pyramids of the same run, whose read volume is known exactly (MI355X_MICROARCH.md: gfx950 tallies wide reads at half their size)
    tr = None
    if rank == 0 and world == 1 and not args.no_traffic and not args.traffic_child:
        tr = measure_traffic(args, B, w, h)
        if tr is not None:
            roofline["traffic"] = tr["traffic"]
            roofline["traffic_source"] = tr["source"]
            roofline["frac_hbm_counter"] = round(tr["traffic"] / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
    pm = None
    if not args.no_pyramid:
        pms = []
        for _ in range(reps):
            ev0.record(stream); ctx.frames_attach_device_batch(slots, raw_ptr, frame_bytes); ev1.record(stream)
            ev1.synchronize(); pms.append(ev0.elapsed_time(ev1))
        pm = float(np.mean(pms))
        pyr_bytes = B * (4 * w * h + sum(4 * (w >> l) * (h >> l) for l in range(1, ctx.levels)))  # read the resident image + write the coarser levels (level 0 is the image itself)
        roofline["pyramid_kernel_ms"] = round(pm, 4)
        roofline["pyramid_GBps"] = round(pyr_bytes / (pm * 1e-3) / 1e9, 1)
        # the whole step against the same peak: algorithmic bytes of tracking + the build's traffic over the step time of the timed region
        roofline["step"] = dict(bytes=int(alg_bytes + pyr_bytes), ms=round(ms_per_step, 4), achieved=round((alg_bytes + pyr_bytes) / (ms_per_step * 1e-3) / 1e9, 1),
                                frac=round((alg_bytes + pyr_bytes) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5))

    # ---------------- CPU baseline: the oracle (port of the reference's SSE path), 1 thread, bounded sample
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        O = graft.load_oracle()
        T = O.Tracker(w, h)
        T.make_k(case["K4"])
        T.set_ref(O.make_images(case["ref_img"], w, h)[0], case["u"], case["v"], case["idepth"], case["hdiF"])
        n_done = 0
        t_cpu0 = time.perf_counter()
        while True:
            i = n_done % B
            tA = time.perf_counter()
            dIn = O.make_images(host_frames[i % n_host], w, h)[0]
            T.set_new(dIn)
            o = T.track(poses0[i], affs0[i])
            n_done += 1
            if time.perf_counter() - t_cpu0 > args.cpu_seconds:
                break
        t_cpu = time.perf_counter() - t_cpu0
        cpu = dict(value=round(n_done / t_cpu, 2), unit="frames/s", cores=1, kind="port",
                   sample="%d frames of the same batch (makeImages + trackNewestCoarse), oracle -O3 -msse2, 1 thread "
                          "(the reference tracks single-threaded), %.1f s on %s" % (n_done, t_cpu, _cpu_name()))
        # the reference's OWN sources, compiled by oracle/Makefile.ref (oracle/_ref/libref.so), when that library travelled with the tree: same frames, same calls
        try:
            Rf = graft.load_reference()
            if Rf.available():
                RT = Rf.Tracker(w, h, case["K4"])
                RT.set_ref(case["ref_img"], case["u"], case["v"], case["idepth"], case["hdiF"])
                n_ref = 0; t_r0 = time.perf_counter(); per_frame = []
                while time.perf_counter() - t_r0 < max(2.0, args.cpu_seconds / 2):
                    i = n_ref % B
                    tA = time.perf_counter()
                    RT.set_new(host_frames[i % n_host])
                    RT.track(poses0[i], affs0[i])
                    per_frame.append(time.perf_counter() - tA)
                    n_ref += 1
                t_r = time.perf_counter() - t_r0
                q = np.percentile(np.array(per_frame) * 1e3, [10, 50, 90])
                cpu["ms_per_frame_p10_p50_p90"] = [round(float(x), 4) for x in q]
                cpu["value_port"] = cpu["value"]
                cpu["value"] = round(n_ref / t_r, 2); cpu["kind"] = "reference"
                cpu["sample"] = ("%d frames of the same batch through the reference's own FrameHessian::makeImages + CoarseTracker::trackNewestCoarse (its sources compiled -O3 -msse2 by "
                                 "oracle/Makefile.ref against stand-in Eigen / Sophus headers), 1 thread (the reference tracks single-threaded), %.1f s on %s; `value_port` = the "
                                 "oracle restatement on %d frames in %.1f s" % (n_ref, t_r, _cpu_name(), n_done, t_cpu))
        except Exception as ex:      # the port's figure stands
            sys.stderr.write("bench: reference build not timed (%s: %s)\n" % (type(ex).__name__, ex))

    out = {
        "metric": "tracked frames/sec (512x512, CoarseTracker direct image alignment, 4 pyramid levels)",
        "value": round(frames_per_s, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "ms_per_step_ranks": {"min": round(min(rank_ms), 4), "max": round(max(rank_ms), 4)},
        # the K timed steps one by one (this rank): the spread behind the mean — without the first one, which only waits for the batch the warm-up left in flight
        "ms_per_step_p10_p50_p90": [round(float(x), 4) for x in np.percentile(step_ms[1:] if len(step_ms) > 1 else step_ms, [10, 50, 90])],
        # (the first timed step only waits for the batch the warm-up left in flight: it is not a step's duration and stays out of min / max)
        "ms_per_step_min_max": [round(float(step_ms[1:].min()), 4), round(float(step_ms[1:].max()), 4)] if len(step_ms) > 1 else [round(float(step_ms.min()), 4), round(float(step_ms.max()), 4)],
        "launcher": ("bench.py --gpus N (self-launched torch.distributed.run)" if os.environ.get("DMVIO_BENCH_SELF_LAUNCHED") else
                     ("torch.distributed.run" if world > 1 else "single process")),
        "config": {"workload": "synthetic %dx%d plane-world, %d-level pyramid, %d reference points (pc_n=%s), batch of %d new frames per GPU "
                               "per step (%d distinct renders, each at its own pose), makeImages%s (level 0 = the resident image, attached in place; levels 1.. built) + trackNewestCoarse (useimu=0 LM) per frame; steady-state pipeline: the host unpacks the results of "
                               "batch k-1 while batch k runs%s"
                               % (w, h, ctx.levels, args.points, pc_n, B, args.distinct, " excluded" if args.no_pyramid else "",
                                  ", and the pyramids of batch k+1 are built on a second stream meanwhile (two slot sets)" if overlap_build else ""),
                   "frames_per_step_per_gpu": B, "points": args.points, "parallelism": "replicas x%d (independent frames)" % world},
        "roofline": roofline,
        "cpu_baseline": cpu,
        "settle_steps": SETTLE_STEPS,
        "pipeline": dict(build_overlaps_tracking=bool(overlap_build),
                         ms_per_step_build_on_the_tracking_stream=round(ms_per_step, 4) if not overlap_build else (None if other_ms is None else round(other_ms, 4)),
                         ms_per_step_build_on_its_own_stream=round(ms_per_step, 4) if overlap_build else (None if other_ms is None else round(other_ms, 4)),
                         what="per step one pyramid build (B frames), one k_track_lm launch, one result download + unpack in every variant; `value` times the build on the "
                              "tracking stream (every kernel alone on the device: the durations the roofline and the rocprofv3 statistics quote); on its own stream "
                              "(dmvio_hip_set_build_stream, two slot sets, events between the streams; --build-overlap / --build-overlap-leg) the build of batch k+1 runs "
                              "beside the tracking of batch k — measured between 1.5 % slower and 5.8 % faster over five boxes (profiles/r04_pyramid_build.md), because the "
                              "register file is full of k_track_lm waves"),
        "lm_iterations_mean": float(np.mean(res["iterations"])),
        "max_pose_err_m": float(terr.max()),
    }

    # the legs below are secondary: if one of them hangs at N > 1 (a rank lost in a collective), the headline line is still printed
    def _watchdog(signum, frame):
        if rank == 0:
            out["ba"] = dict(error="secondary legs timed out")
            emit(json.dumps(out))
        os._exit(0)
    if world > 1:
        import signal
        signal.signal(signal.SIGALRM, _watchdog)
        signal.alarm(240)

    # ---------------- PCIe-inclusive leg (rank 0, N = 1): the same step with the raw frames coming from pinned host memory inside the timed region
    pcie_out = None
    if rank == 0 and world == 1 and not args.no_pcie and not args.traffic_child:
        pinned = torch.empty((B, h, w), dtype=torch.float32, pin_memory=True)
        pinned.copy_(raw)
        torch.cuda.synchronize(dev)
        n_p = max(5, min(args.steps, 20))

        def step_pcie():
            with torch.cuda.stream(stream):
                raw.copy_(pinned, non_blocking=True)
            return step()
        step_pcie()
        torch.cuda.synchronize(dev)
        t0p = time.perf_counter()
        for _ in range(n_p):
            rp = step_pcie()
        torch.cuda.synchronize(dev)
        tp = (time.perf_counter() - t0p) / n_p
        ev0.record(stream)
        with torch.cuda.stream(stream):
            raw.copy_(pinned, non_blocking=True)
        ev1.record(stream); ev1.synchronize()
        h2d_ms = ev0.elapsed_time(ev1)
        pcie_out = dict(value=round(B / tp, 1), unit="frames/s", ms_per_step=round(1e3 * tp, 4), steps=n_p, h2d_ms=round(h2d_ms, 3),
                        h2d_GBps=round(B * frame_bytes / (h2d_ms * 1e-3) / 1e9, 2), good=bool(rp["good"].all()),
                        note="per step: %d x %d B fp32 frames copied from pinned host memory to HBM, then makeImages + trackNewestCoarse + result fetch, not pipelined" % (B, frame_bytes))
        del pinned
        # the same leg with the frames entering as the camera delivers them: 8-bit raw images, photometric undistortion (passthrough geometry, factor 1) fused
        # into the pyramid build (dmvio_hip_frames_from_raw_device_batch) — a quarter of the bytes over PCIe
        try:
            und = pkg.UndistorterHip(ctx, w, h, 8)
            raw8 = torch.clamp(torch.round(raw), 0, 255).to(torch.uint8)
            pinned8 = torch.empty((B, h, w), dtype=torch.uint8, pin_memory=True)
            pinned8.copy_(raw8)
            torch.cuda.synchronize(dev)

            def step_pcie8():
                with torch.cuda.stream(stream):
                    raw8.copy_(pinned8, non_blocking=True)
                und.from_raw_device_batch(slots, raw8.data_ptr(), w * h)
                trk.stage(slots, poses0, affs0)
                trk.launch()
                return trk.fetch()
            step_pcie8()
            torch.cuda.synchronize(dev)
            t0p = time.perf_counter()
            for _ in range(n_p):
                rp8 = step_pcie8()
            torch.cuda.synchronize(dev)
            tp8 = (time.perf_counter() - t0p) / n_p
            ev0.record(stream); und.from_raw_device_batch(slots, raw8.data_ptr(), w * h); ev1.record(stream); ev1.synchronize()
            k8_ms = ev0.elapsed_time(ev1)
            terr8 = np.linalg.norm(rp8["pose7"][:, :3] - truth[:, :3], axis=1)
            # pipelined form: a copy stream brings batch k+1 into the other of two device buffers while batch k is undistorted, built and tracked
            copy_stream = torch.cuda.Stream(device=dev)
            bufs = [raw8, torch.empty_like(raw8)]
            ev_copied = [torch.cuda.Event(), torch.cuda.Event()]
            ev_built = [torch.cuda.Event(), torch.cuda.Event()]
            built_once = [False, False]

            def enqueue_copy(k):
                with torch.cuda.stream(copy_stream):
                    if built_once[k % 2]:
                        copy_stream.wait_event(ev_built[k % 2])
                    bufs[k % 2].copy_(pinned8, non_blocking=True)
                    ev_copied[k % 2].record(copy_stream)

            def step_pipe8(k):
                enqueue_copy(k + 1)
                stream.wait_event(ev_copied[k % 2])
                und.from_raw_device_batch(slots, bufs[k % 2].data_ptr(), w * h)
                ev_built[k % 2].record(stream); built_once[k % 2] = True
                trk.stage(slots, poses0, affs0)
                trk.launch()
                return trk.fetch()
            enqueue_copy(0)
            step_pipe8(0)
            torch.cuda.synchronize(dev)
            t0p = time.perf_counter()
            for k in range(1, n_p + 1):
                rq8 = step_pipe8(k)
            torch.cuda.synchronize(dev)
            tq8 = (time.perf_counter() - t0p) / n_p
            pcie_out["raw_u8"] = dict(value=round(B / tq8, 1), unit="frames/s", ms_per_step=round(1e3 * tq8, 4), steps=n_p, good=bool(rp8["good"].all() and rq8["good"].all()),
                                      value_not_pipelined=round(B / tp8, 1), ms_per_step_not_pipelined=round(1e3 * tp8, 4),
                                      max_pose_err_m=float(terr8.max()), undistort_pyramid_kernel_ms=round(k8_ms, 4),
                                      undistort_pyramid_GBps=round(B * (w * h + sum(4 * (w >> l) * (h >> l) for l in range(ctx.levels))) / (k8_ms * 1e-3) / 1e9, 1),
                                      note="frames cross PCIe as %d-byte 8-bit raw images; undistortion + makeImages fused in one launch (dmvio_hip_frames_from_raw_device_batch), then trackNewestCoarse "
                                           "+ result fetch; `value`: the copy of batch k+1 on a second stream overlaps batch k (two device buffers), `value_not_pipelined`: copy, build, track in sequence" % (w * h))
            # the same frames RESIDENT in HBM as 8-bit raw images (no copy in the step): undistortion + pyramid build + tracking, with level 0 written row-major (the default)
            # and in 8x4 tiles (2.4 instead of 4.3 lines per tap, but twelve dword loads with their own tile addresses per tap: measured slower in k_track_lm);
            # kernel times by HIP events on the stream
            try:
                res = {}
                for name, tiled in (("tiled_8x4", True), ("row_major", False)):
                    pkg.set_raw_batch_layout(ctx, tiled)
                    und.from_raw_device_batch(slots, raw8.data_ptr(), w * h)
                    trk.stage(slots, poses0, affs0); trk.launch(); r0 = trk.fetch()          # warm
                    kb, kt = [], []
                    for _ in range(5):
                        ev0.record(stream); und.from_raw_device_batch(slots, raw8.data_ptr(), w * h); ev1.record(stream); ev1.synchronize(); kb.append(ev0.elapsed_time(ev1))
                        trk.stage(slots, poses0, affs0)
                        ev0.record(stream); trk.launch(); ev1.record(stream); ev1.synchronize(); kt.append(ev0.elapsed_time(ev1))
                        r0 = trk.fetch()
                    torch.cuda.synchronize(dev)
                    t0r = time.perf_counter()
                    for _ in range(n_p):
                        und.from_raw_device_batch(slots, raw8.data_ptr(), w * h)
                        trk.stage(slots, poses0, affs0); trk.launch(); r0 = trk.fetch()
                    torch.cuda.synchronize(dev)
                    tr_ = (time.perf_counter() - t0r) / n_p
                    _, npe = trk.last_work()
                    res[name] = dict(k_track_lm_ms=round(float(np.mean(kt)), 4), k_build_pyramids_raw_ms=round(float(np.mean(kb)), 4), ms_per_step=round(1e3 * tr_, 4),
                                     value=round(B / tr_, 1), good=bool(r0["good"].all()), max_pose_err_m=float(np.linalg.norm(r0["pose7"][:, :3] - truth[:, :3], axis=1).max()),
                                     k_track_lm_roofline_frac=round(BYTES_PER_POINT_EVAL * npe / (float(np.mean(kt)) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                     build_GBps=round(B * (w * h + sum(4 * (w >> l) * (h >> l) for l in range(ctx.levels))) / (float(np.mean(kb)) * 1e-3) / 1e9, 1))
                pkg.set_raw_batch_layout(ctx, False)
                res["k_track_lm_speedup_tiled"] = round(res["row_major"]["k_track_lm_ms"] / res["tiled_8x4"]["k_track_lm_ms"], 4)
                res["what"] = ("%d 8-bit raw frames resident in HBM per step: dmvio_hip_frames_from_raw_device_batch (undistortion + all pyramid levels) + trackNewestCoarse, not "
                               "pipelined; level 0 in 8x4-pixel tiles vs row-major — identical results bit for bit (tests/test_io_gpu.py), other addresses" % B)
                pcie_out["raw_u8"]["resident"] = res
            except Exception as ex:
                pcie_out["raw_u8"]["resident"] = dict(error="%s: %s" % (type(ex).__name__, ex))
            del bufs, copy_stream
            und.close(); del pinned8, raw8
        except Exception as ex:
            pcie_out["raw_u8"] = dict(error="%s: %s" % (type(ex).__name__, ex))
        ctx.frames_attach_device_batch(slots, raw_ptr, frame_bytes)

    # ---------------- batch-size sweep (rank 0, N = 1): frames per step from a quarter of the resident workgroup slots to four times their number
    sweep_out = None
    if rank == 0 and world == 1 and not args.no_sweep and not args.traffic_child:
        sweep_out = bench_sweep(args, pkg, torch, dev, stream, case, raw, frames_meta, poses0, w, h)

    # ---------------- BA leg: Gauss-Newton iterations / s of the 8-keyframe sliding-window photometric BA (rank 0 window per rank)
    ba_out = None
    if not args.no_ba:
        try:
            ba_out = bench_ba(args, pkg, synth, ctx_device=local_rank, rank=rank, world=world, dist=dist, dev=dev, coll_dev=coll_dev, torch=torch,
                              cpu=(rank == 0 and world == 1 and not args.no_cpu))
        except Exception as ex:      # the secondary leg must not take the headline line down with it
            if world == 1:
                raise
            ba_out = dict(error="%s: %s" % (type(ex).__name__, ex))

    # ---------------- trace leg (rank 0, N = 1): FullSystem::traceNewCoarse over 7 hosts x 1500 immature points
    trace_out = None
    if rank == 0 and world == 1 and not args.no_ba and not args.traffic_child:
        trace_out = bench_trace(args, pkg, synth, ctx, torch, stream, case, cpu=not args.no_cpu)

    # ---------------- overlap leg (rank 0, N = 1): tracking thread + mapping thread on their own HIP streams (BASELINE config 5)
    overlap_out = None
    if rank == 0 and world == 1 and not args.no_ba and not args.traffic_child:
        overlap_out = bench_overlap(args, pkg, synth, ctx, trk, slots, poses0, affs0, B, w, h)

    # ---------------- live leg (rank 0, N = 1): ONE camera stream, frame after frame — the latency-bound regime of the reference's tracking thread
    live_out = None
    if rank == 0 and world == 1 and not args.no_ba and not args.traffic_child:
        live_out = bench_live(args, pkg, synth, ctx, trk, raw, case, w, h)

    # ---------------- VIO hand-off leg (rank 0, N = 1): the reference's default branch — every LM step computed on the host
    vio_out = None
    if rank == 0 and world == 1 and not args.no_ba and not args.traffic_child:
        vio_out = bench_vio(args, pkg, ctx, trk, raw, case, w, h)

    # ---------------- drop-in leg (rank 0, N = 1): the reference's own FullSystem::addActiveFrame over a synthetic sequence, all-CPU and HIP-backed
    dropin_out = None
    if rank == 0 and world == 1 and not args.no_ba and not args.no_cpu and not args.no_dropin and not args.traffic_child:
        try:
            dropin_out = bench_dropin(args, w, h)
        except Exception as ex:
            dropin_out = dict(error="%s: %s" % (type(ex).__name__, ex))

    if rank == 0 and tr is not None and tr.get("ba_linearize") and ba_out and isinstance(ba_out.get("roofline"), dict):
        t = tr["ba_linearize"]; rb = ba_out["roofline"]
        rb["traffic"] = t["traffic"]
        rb["traffic_source"] = ("the same rocprofv3 --pmc FETCH_SIZE child run as roofline.traffic_source: %d k_ba_linearize dispatches, FETCH_SIZE %.1f KiB each, x %.3f = %.2fx the "
                                "algorithmic bytes (a 128-byte line per tap row for 16 useful bytes); FETCH_SIZE counts what L2 requests from the fabric, the 256 MB Infinity Cache behind "
                                "it holds the window's eight images between launches" % (t["dispatches"], t["fetch_kib"], t["factor"], t["traffic"] / max(rb["algorithmic_bytes_per_launch"], 1)))
        rb["frac_hbm_counter"] = round(t["traffic"] / (rb["kernel_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 5)
    if rank == 0:
        if world > 1:
            # ranks that really exchanged over RCCL: ncclCommCount of the library's own communicator (the sharded window ran on it); the launcher's process group otherwise
            out["rccl_ranks"] = int((ba_out or {}).get("rccl_ranks", world if backend == "nccl" else 0))
            out["process_group"] = backend
        if isinstance(ba_out, dict) and "value" in ba_out:
            # the mapping half of BASELINE.json's metric inside the objects a reader of the first level keeps (roofline / cpu_baseline / config) and as a top-level scalar
            rb = ba_out.get("roofline") if isinstance(ba_out.get("roofline"), dict) else {}
            bw = ba_out.get("batched_windows") if isinstance(ba_out.get("batched_windows"), dict) else {}
            w64 = next((r for r in bw.get("sweep", []) if r.get("windows") == 64), {})
            w16 = next((r for r in bw.get("sweep", []) if r.get("windows") == 16), {})
            out["ba_value"] = ba_out["value"]; out["ba_unit"] = "GN-iters/s"
            out["ba_value_batched_w16"] = w16.get("value"); out["ba_value_batched_w64"] = w64.get("value")
            out["roofline"]["ba"] = dict(kernel=rb.get("kernel"), frac=rb.get("frac"), kernel_us=rb.get("kernel_us"), achieved=rb.get("achieved"), unit="GB/s",
                                         iteration_us=(rb.get("iteration") or {}).get("wall_us"), kernels_us=(rb.get("iteration") or {}).get("kernels_us"),
                                         kernel_batched="k_ba_linearize_b1", frac_batched_w64=w64.get("k_ba_linearize_b_frac"), kernel_us_batched_w64=w64.get("k_ba_linearize_b_us"),
                                         frac_batched_w16=w16.get("k_ba_linearize_b_frac"),
                                         what="ba.roofline (one window: k_ba_linearize, 464 B per residual) and ba.batched_windows (W windows per launch: k_ba_linearize_b1)")
            cb = ba_out.get("cpu_baseline") if isinstance(ba_out.get("cpu_baseline"), dict) else None
            if cb and isinstance(out.get("cpu_baseline"), dict):
                out["cpu_baseline"]["ba"] = dict(value=cb.get("value"), unit=cb.get("unit"), cores=cb.get("cores"), kind=cb.get("kind"))
            wd = ba_out.get("window") or {}
            out["config"]["ba_workload"] = ("sliding-window photometric bundle adjustment, %sx%s synthetic plane-world, %s keyframes, %s points, %s residuals; ba_value = accepted "
                                            "Gauss-Newton iterations / s of FullSystem::optimize(6) on fresh windows (one window per call); ba_value_batched_wN = the same with N "
                                            "windows per dmvio_hip_ba_optimize_batch call" % (w, h, wd.get("frames"), wd.get("points"), wd.get("residuals")))
        out.update(ba=ba_out, trace=trace_out, drop_in=dropin_out, overlap=overlap_out, live=live_out, vio_handoff=vio_out, pcie=pcie_out, batch_sweep=sweep_out)
        emit(json.dumps(out))
    if world > 1:
        signal.alarm(0)
    if dist is not None:
        dist.destroy_process_group()


def measure_traffic(args, B, w, h):
    """HBM bytes per k_track_lm launch from a `rocprofv3 --kernel-trace --pmc FETCH_SIZE` child run of this workload (counters only, no other tracing)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    d = tempfile.mkdtemp(prefix="dmvio_pmc_", dir="/tmp")
    try:
        env = dict(os.environ); env["TMPDIR"] = "/tmp"
        cmd = [exe, "--kernel-trace", "--pmc", "FETCH_SIZE", "-d", d, "-o", "c", "--", sys.executable, os.path.abspath(__file__), "--traffic-child", "--no-cpu"] + \
              (["--no-ba"] if args.no_ba else ["--ba-iters", "60", "--ba-points", str(args.ba_points)]) + ["--no-sweep", "--no-pcie", "--no-traffic", "--no-concurrent", "--steps", "3", "--warmup", "1", "--batch", str(B), "--points", str(args.points), "--size", str(args.size),
               "--distinct", str(args.distinct)]
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
        dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
        if not dbs:
            return None
        db = sqlite3.connect(dbs[0])
        rows = db.execute("select kernel_name, avg(value), max(value), count(*) from counters_collection where counter_name = 'FETCH_SIZE' group by kernel_name").fetchall()
        lm = [x for x in rows if "k_track_lm" in x[0]]
        pyr = [x for x in rows if "k_build_pyramids" in x[0] and "_raw" not in x[0]]
        if not lm or not pyr:
            return None
        pyr = [max(pyr, key=lambda x: x[2])]     # the full-batch build among the pyramid kernels of the run (single uploads go through the LDS-tile kernel, the batch through the register build)
        # calibration: the full-batch pyramid build reads B raw images exactly once (its largest dispatch); FETCH_SIZE is reported in KiB
        factor = (B * w * h * 4) / (pyr[0][2] * 1024.0)
        traffic = int(lm[0][1] * 1024.0 * factor)
        ba_lin = [x for x in rows if "k_ba_linearize" in x[0]]
        ba = None
        if ba_lin:     # the BA leg of the same child run: every k_ba_linearize dispatch reads the same window (taps, point and residual tables)
            ba = dict(traffic=int(ba_lin[0][1] * 1024.0 * factor), dispatches=int(ba_lin[0][3]), fetch_kib=float(ba_lin[0][1]), factor=float(factor))
        return dict(traffic=traffic, ba_linearize=ba, source="rocprofv3 --kernel-trace --pmc FETCH_SIZE child run of this workload: %d k_track_lm dispatches, FETCH_SIZE %.1f KiB each, x %.3f "
                                            "(calibrated in the same run on %s, which reads %d B per launch and reports %.1f KiB)"
                                            % (lm[0][3], lm[0][1], factor, pyr[0][0].split("(")[0].replace("void dmv::", ""), B * w * h * 4, pyr[0][2]))
    except Exception as ex:      # a diagnostic must not take the bench down
        sys.stderr.write("bench: traffic measurement skipped (%s: %s)\n" % (type(ex).__name__, ex))
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)


def bench_sweep(args, pkg, torch, dev, stream, case, raw, frames_meta, poses0, w, h):
    """makeImages + trackNewestCoarse for batches of 256 ... 4096 frames (same renders, replicated beyond the distinct ones): where the device fills up
    (1024 = one 256-thread workgroup in each of the 4 slots of all 256 CUs), what a half-filled second wave costs (1536) and the steady state beyond."""
    B0 = raw.shape[0]
    sizes = [256, 512, 1024, 1536, 2048, 4096, 8192]
    Bmax = max(sizes)
    ctx2 = pkg.Context(w, h, n_slots=Bmax + 1, device=dev.index)
    ctx2.set_stream(stream.cuda_stream)
    trk2 = pkg.CoarseTrackerHip(ctx2)
    trk2.makeK(case["K4"])
    ctx2.frame_upload(0, case["ref_img"])
    trk2.setCoarseTrackingRef(0, case["u"], case["v"], case["idepth"], case["hdiF"])
    if B0 >= Bmax:
        big = raw
    else:
        big = torch.empty((Bmax, h, w), dtype=torch.float32, device=dev)
        idx = torch.arange(Bmax, device=dev) % B0
        for c0 in range(0, Bmax, 512):
            big[c0:c0 + 512] = raw[idx[c0:c0 + 512]]
    torch.cuda.synchronize(dev)
    p0 = poses0[np.arange(Bmax) % B0]
    a0 = np.zeros((Bmax, 2))
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    frame_bytes = w * h * 4
    rows = []
    for Bs in sizes:
        sl = np.arange(1, Bs + 1, dtype=np.int32)

        def one(fetch_prev):
            ctx2.frames_attach_device_batch(sl, big.data_ptr(), frame_bytes)
            if fetch_prev:
                trk2.fetch_begin()
            trk2.stage(sl, p0[:Bs], a0[:Bs]); trk2.launch()
            return trk2.fetch() if fetch_prev else None
        one(False); trk2.fetch(); one(False)
        torch.cuda.synchronize(dev)
        n = max(5, min(args.steps, 50))
        t0 = time.perf_counter()
        for _ in range(n):
            r = one(True)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / n
        trk2.fetch()
        trk2.stage(sl, p0[:Bs], a0[:Bs])
        kms = []
        for _ in range(5):
            ev0.record(stream); trk2.launch(); ev1.record(stream); ev1.synchronize(); kms.append(ev0.elapsed_time(ev1))
        trk2.fetch()
        rows.append(dict(frames_per_step=Bs, value=round(Bs / dt, 1), ms_per_step=round(1e3 * dt, 4), track_kernel_ms=round(float(np.mean(kms)), 4),
                         track_kernel_us_per_frame=round(1e3 * float(np.mean(kms)) / Bs, 4), good=bool(r["good"].all())))
    trk2.close() if hasattr(trk2, "close") else None
    ctx2.close()
    return dict(unit="frames/s", distinct_frames=int(B0), resident_workgroup_slots=1024, points=rows)


def bench_overlap(args, pkg, synth, ctx, trk, slots, poses0, affs0, B, w, h):
    """Tracking batches (context stream) and BA iterations (BA handle's own stream) issued from two host threads: wall time of both
    together vs one after the other — the reference's tracking / mapping threads on HIP streams."""
    import threading
    bcase = synth.ba_case(w, h, n_frames=8, n_points=args.ba_points, seed=synth.SEED)
    bslots = list(range(args.ba_slot0, args.ba_slot0 + 8))
    for k in range(8):
        ctx.frame_upload(bslots[k], bcase["imgs"][k])
    ba = pkg.BundleAdjusterHip(ctx)
    ba.set_case(bcase, bslots)
    # a saturating tracking batch leaves no CU free for the (tiny) BA kernels; the live-tracking regime is a few frames per call
    Bt = min(B, 64)
    n_t, n_m = 150, 240

    done = {}

    def T(n):
        for _ in range(n):
            trk.stage(slots[:Bt], poses0[:Bt], affs0[:Bt]); trk.launch(); trk.fetch()
        done["t"] = time.perf_counter()

    def M(n):
        ba.activate_all(); e = ba.linearize_all(False); ba.apply_res()
        lam, lastE = 1e-5, [e, 0.0, 0.0]
        for it in range(n):
            _, lam, lastE = ba.gn_iteration(it % 6, lam, lastE)
        done["m"] = time.perf_counter()

    T(2); M(20)
    t0 = time.perf_counter(); T(n_t); t_t = time.perf_counter() - t0
    t0 = time.perf_counter(); M(n_m); t_m = time.perf_counter() - t0
    a = threading.Thread(target=T, args=(n_t,)); b = threading.Thread(target=M, args=(n_m,))
    t0 = time.perf_counter(); a.start(); b.start(); a.join(); b.join(); t_p = time.perf_counter() - t0
    ba.close()
    return dict(tracking_calls=n_t, frames_per_call=Bt, ba_iterations=n_m, tracking_alone_ms=round(1e3 * t_t, 2), ba_alone_ms=round(1e3 * t_m, 2),
                sequential_ms=round(1e3 * (t_t + t_m), 2), overlapped_ms=round(1e3 * t_p, 2),
                overlapped_tracking_done_ms=round(1e3 * (done["t"] - t0), 2), overlapped_ba_done_ms=round(1e3 * (done["m"] - t0), 2),
                note="tracking on the context stream, bundle adjustment on the BA handle's stream, two host threads (no pyramid builds in this leg)")


def bench_live(args, pkg, synth, ctx, trk, raw, case, w, h):
    """One camera stream: per frame makeImages (attached in place) -> trackNewestCoarse (one alignment problem, cluster mode) ->
    traceNewCoarse over the window's immature points (7 hosts x 1500), each step waiting for the previous one's result as the
    reference's tracking thread does (FullSystem::addActiveFrame).  Host wall time per frame."""
    n_per_host, hosts = 1500, 7
    rng = np.random.RandomState(6)
    u, v = synth.select_points(case["ref_img"], n_per_host, rng, min_grad=8.0)
    u = np.clip(u.astype(np.int32), 8, w - 9); v = np.clip(v.astype(np.int32), 8, h - 9)
    imm = pkg.ImmaturePointsHip(ctx, capacity=n_per_host * hosts)
    for tag in range(hosts):
        imm.add_points(tag, 0, u, v)
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0]); idents = np.tile(ident, (hosts, 1))
    n = imm.n
    st0 = (np.zeros(n, np.float32), np.full(n, np.nan, np.float32), np.full(n, 10000.0, np.float32), np.full(n, 5, np.int32))
    frame_bytes = w * h * 4
    base = raw.data_ptr()
    pose = ident.copy()
    parts = np.zeros(3)
    frames = 200
    imm.set_state(*st0)
    t_all = 0.0
    for k in range(frames + 10):
        if k == 10:
            parts[:] = 0; t_all = 0.0
        slot = 1 + (k % args.distinct)
        t0 = time.perf_counter()
        ctx.frames_attach_device_batch([slot], base + (slot - 1) * frame_bytes, frame_bytes)
        t1 = time.perf_counter()
        r = trk.track_batch([slot], [pose], [(0.0, 0.0)])
        t2 = time.perf_counter()
        imm.traceNewCoarse(slot, r["pose7"][0], idents, case["K4"])
        t3 = time.perf_counter()
        parts += (t1 - t0, t2 - t1, t3 - t2); t_all += t3 - t0
        if k % 20 == 19:
            imm.set_state(*st0)          # keep the traces doing first-trace work (outside the timed spans would be cleaner; it is 1 call in 20)
    return dict(metric="frames/s of one live stream (makeImages + trackNewestCoarse + traceNewCoarse per frame, each waiting for the previous result)",
                value=round(frames / t_all, 1), unit="frames/s", ms_per_frame=round(1e3 * t_all / frames, 4),
                ms_make_images=round(1e3 * parts[0] / frames, 4), ms_track=round(1e3 * parts[1] / frames, 4), ms_trace=round(1e3 * parts[2] / frames, 4),
                immature_points=n)


def bench_vio(args, pkg, ctx, trk, raw, case, w, h):
    """One frame at a time through dmvio_hip_tracker_track_vio (the reference's setting_useIMU branch, CoarseTracker.cpp:612-637: one fused
    evaluation launch per LM iteration, the step computed on the host — here the library's visual-only step, so the work equals the
    device-resident LM's) next to dmvio_hip_tracker_track on the same frames.  Host wall time per frame."""
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    frame_bytes = w * h * 4
    base = raw.data_ptr()
    nd = min(args.distinct, 64)
    for slot in range(1, nd + 1):
        ctx.frames_attach_device_batch([slot], base + (slot - 1) * frame_bytes, frame_bytes)
    frames = 200
    out = {}
    for name in ("device_lm", "handoff"):
        trk.set_single_frame_mode(host_lm=(name != "device_lm"))      # "device_lm": the device-resident LM in cluster mode, what round 1 ran for a single frame
        t_all = 0.0; evals = 0; err = 0.0
        for k in range(frames + 10):
            if k == 10:
                t_all = 0.0; evals = 0
            slot = 1 + (k % nd)
            t0 = time.perf_counter()
            if name == "device_lm":
                r = trk.trackNewestCoarse(slot, ident, (0.0, 0.0)); ne = r["iterations"] + ctx.levels
            else:
                r = trk.trackNewestCoarseVIO(slot, ident, (0.0, 0.0)); ne = r["n_evals"]
            t_all += time.perf_counter() - t0; evals += ne
            err = max(err, float(np.linalg.norm(r["pose7"][:3] - case["frames"][(slot - 1) % nd]["pose7"][:3])))
        out[name] = dict(ms_per_frame=round(1e3 * t_all / frames, 4), evals_per_frame=round(evals / frames, 2), us_per_eval=round(1e6 * t_all / max(evals, 1), 2),
                         max_pose_err_m=err)
    trk.set_single_frame_mode(host_lm=True)
    out["ratio_handoff_to_device_lm"] = round(out["handoff"]["ms_per_frame"] / out["device_lm"]["ms_per_frame"], 3)
    out["note"] = ("handoff = dmvio_hip_tracker_track_vio with the library's visual-only LM step as computeCoarseUpdate: ONE kernel launch per frame (the evaluation server), "
                   "every LM evaluation a request through host-coherent memory, its sums polled from there; a GTSAM-backed computeCoarseUpdate adds its own host time per "
                   "iteration.  device_lm = the device-resident LM (cluster mode).  dmvio_hip_tracker_track of a single frame takes the host-LM path by default.")
    return out


def bench_trace(args, pkg, synth, ctx, torch, stream, case, cpu):
    """ImmaturePoint::traceOn over the immature points of a 7-host window against one new frame (FullSystem::traceNewCoarse):
    points / s on the GPU (kernel, HIP events) and for the oracle on one host core."""
    w, h = case["w"], case["h"]
    rng = np.random.RandomState(5)
    n_per_host, hosts = 1500, 7
    u, v = synth.select_points(case["ref_img"], n_per_host, rng, min_grad=8.0)
    u = np.clip(u.astype(np.int32), 8, w - 9); v = np.clip(v.astype(np.int32), 8, h - 9)
    imm = pkg.ImmaturePointsHip(ctx, capacity=n_per_host * hosts)
    ctx.frame_upload(0, case["ref_img"]); ctx.frame_upload(1, case["frames"][0]["img"])
    for tag in range(hosts):
        imm.add_points(tag, 0, u, v)
    n = imm.n
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    st0 = (np.zeros(n, np.float32), np.full(n, np.nan, np.float32), np.full(n, 10000.0, np.float32), np.full(n, 5, np.int32))
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(12):
        imm.set_state(*st0)
        ev0.record(stream); imm.traceNewCoarse(1, case["frames"][0]["pose7"], np.tile(ident, (hosts, 1)), case["K4"]); ev1.record(stream)
        ev1.synchronize(); ts.append(ev0.elapsed_time(ev1))
    k_ms = float(np.median(ts[2:]))
    st = imm.get_state()["lastTraceStatus"]
    out = dict(metric="immature points traced / s (ImmaturePoint::traceOn, first trace with unbounded depth interval)", value=round(n / (k_ms * 1e-3), 1),
               unit="points/s", points=n, hosts=hosts, ms_per_call=round(k_ms, 4), good_fraction=round(float((st == 0).mean()), 3))
    if cpu:
        O = graft.load_oracle()
        KRKi, Kt, aff = O.trace_precalc(case["frames"][0]["pose7"], ident, case["K4"])
        dIh = O.make_images(case["ref_img"], w, h)[0][0]; dIn = O.make_images(case["frames"][0]["img"], w, h)[0][0]
        P = O.ImmaturePoints(dIh, w, h, u, v)
        t0 = time.perf_counter(); reps = 0
        while time.perf_counter() - t0 < 2.0:
            P.idepth_min[:] = 0; P.idepth_max[:] = np.nan; P.quality[:] = 10000; P.lastTraceStatus[:] = 5
            P.trace_on(dIn, KRKi, Kt, aff); reps += 1
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = dict(value=round(reps * n_per_host / dt, 1), unit="points/s", cores=1, kind="port",
                                   sample="%d x %d points, oracle traceOn, 1 thread (the reference traces single-threaded)" % (reps, n_per_host))
        # the reference's OWN FullSystem::traceNewCoarse (ImmaturePoint.cpp compiled unmodified into oracle/_ref/libref.so): a window of 7 hosts + the new frame, the same
        # 1500 pixels per host, every point in its constructor's state (first trace, unbounded interval)
        try:
            Rf = graft.load_reference()
            if Rf.available():
                one = np.zeros(1)
                wcase = dict(K4=case["K4"], w=w, h=h, n_frames=hosts + 1, imgs=[case["ref_img"]] * hosts + [case["frames"][0]["img"]],
                             poses0=[ident] * hosts + [case["frames"][0]["pose7"]], host=np.zeros(1, np.int32), u=np.array([u[0]], np.float32), v=np.array([v[0]], np.float32),
                             idepth0=np.ones(1, np.float32), color=np.zeros((1, 8), np.float32), weights=np.ones((1, 8), np.float32), res_point=np.zeros(1, np.int32),
                             res_target=np.ones(1, np.int32))
                WR = Rf.BAWindow(wcase, use_case_color=False)
                for tag in range(hosts):
                    WR.immature_add(tag, u, v)
                t0 = time.perf_counter(); reps_r = 0; spent = 0.0
                while spent < 2.0:
                    WR.immature_reset()
                    tA = time.perf_counter(); WR.trace_new_coarse(hosts); spent += time.perf_counter() - tA; reps_r += 1
                port = out["cpu_baseline"]
                out["cpu_baseline"] = dict(value=round(reps_r * n_per_host * hosts / spent, 1), unit="points/s", cores=1, kind="reference", value_port=port["value"],
                                           sample="%d x FullSystem::traceNewCoarse over %d hosts x %d immature points (the reference's own ImmaturePoint::traceOn, compiled -O3 -msse2 by "
                                                  "oracle/Makefile.ref), 1 thread (the reference traces single-threaded), %.1f s on %s" % (reps_r, hosts, n_per_host, spent, _cpu_name()))
        except Exception as ex:
            sys.stderr.write("bench: reference trace not timed (%s: %s)\n" % (type(ex).__name__, ex))
    imm.close()
    return out


def bench_ba_concurrent(args, pkg, ctx, case, F, bytes_iter, bytes_lin, torch, dev, M=32):
    """What ONE GPU sustains with several windows in flight — the BA analogue of the tracker's 4096-frame batch.  A single window is a chain of dependent sub-10-us
    launches on a 256-CU device (latency-bound: `ba.value`); K host threads, each optimising its own fresh windows through its own dmvio_hip_ba handles (own HIP stream, own
    lock, own pinned result block), overlap those chains.  Every thread owns M windows, all set up BEFORE the timed region (set_graph is per-keyframe set-up, not iteration
    work); timed: every thread runs dmvio_hip_ba_optimize(6) on its M windows one after the other.  value = accepted iterations of all threads / wall time."""
    import threading
    n_cpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    levels = [k for k in (1, 2, 4, 8, 16) if k <= max(1, n_cpu - 1)]       # the waits are polls of host-coherent memory: one core per thread
    slots = list(range(F))
    pool = []
    rows = []
    best = None
    try:
        for K in levels:
            while len(pool) < K * M:
                pool.append(pkg.BundleAdjusterHip(ctx))
            for h in pool[:K * M]:
                h.set_case(case, slots)                                      # fresh window: every step of optimize(6) is accepted until convergence
            torch.cuda.synchronize(dev)
            accepted = [0] * K
            errors = []
            start = threading.Barrier(K + 1)

            def work(t):
                try:
                    start.wait()
                    n = 0
                    for h in pool[t * M:(t + 1) * M]:
                        n += int(h.optimize(6)["trace"][1:, 3].sum())
                    accepted[t] = n
                except Exception as ex:   # reported, never swallowed
                    errors.append("%s: %s" % (type(ex).__name__, ex))
            th = [threading.Thread(target=work, args=(t,)) for t in range(K)]
            for x in th:
                x.start()
            start.wait()
            t0 = time.perf_counter()
            for x in th:
                x.join()
            wall = time.perf_counter() - t0
            if errors:
                return dict(error=errors[0], threads=K)
            n_acc = sum(accepted)
            val = n_acc / wall
            rows.append(dict(threads=K, windows=K * M, accepted_iterations=n_acc, wall_ms=round(1e3 * wall, 3), value=round(val, 1),
                             us_per_iteration_per_window=round(1e6 * wall * K / max(n_acc, 1), 2)))
            if best is None or val > best["value"]:
                best = rows[-1]
    finally:
        for h in pool:
            h.close()
    ach = bytes_iter * best["value"] / 1e9
    return dict(unit="GN-iters/s", value=best["value"], at_threads=best["threads"], sweep=rows, host_cores_available=n_cpu,
                roofline=dict(bound="hbm", achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 5),
                              k_ba_linearize_achieved=round(bytes_lin * best["value"] / 1e9, 2),
                              what="algorithmic bytes of an accepted iteration (and of its k_ba_linearize launch alone) x accepted iterations per second at saturation"),
                what="K host threads x %d fresh windows each (own handle, own stream), dmvio_hip_ba_optimize(6) per window, all windows set up before the timed region; "
                     "accepted Gauss-Newton iterations of all threads / wall time.  `ba.value` is the K = 1 latency figure of the same call" % M)


def bench_ba_batched(args, pkg, ctx, case, F, bytes_iter, bytes_lin, torch, dev):
    """W windows per launch sequence on the device-resident Gauss-Newton loop (dmvio_hip_ba_optimize_batch): the 68x68 solve, the frame step and the accept test run on the
    device, every kernel of the chain takes its window from blockIdx.y — the BA analogue of the tracker's batch.  All windows are set up before the timed region; timed: ONE
    dmvio_hip_ba_optimize_batch(W windows, 6 iterations) call (host wall clock) and, inside it, the loop by HIP events on the batch's stream.  value = accepted iterations of all
    windows / wall time.  W = 1 is the device loop's latency figure for a single window (`ba.value` is the host-driven loop's)."""
    slots = list(range(F))
    rows = []
    pool = []
    pool_default = []
    Wmax = 64
    B = pkg.BundleAdjusterBatch(ctx, Wmax)
    default_order = None
    try:
        for W in (1, 4, 16, 64):
            while len(pool) < W:
                # the reference's single-threaded accumulation order (dmvio_hip_ba_set_accumulators(1)): bit-exact AND, in a grid that fills the device, the cheaper one — a
                # quarter of the workgroups, each four times as long, less per-workgroup overhead (the latency argument for 4 partial accumulators only holds for one window)
                pool.append(pkg.BundleAdjusterHip(ctx, accumulators=1))
                pool_default.append(pkg.BundleAdjusterHip(ctx))            # the library's default: 4 partial accumulators per bucket
            row = dict(windows=W)
            for order, hs in (("default", pool_default[:W]), ("single_threaded", pool[:W])):
                walls, loops, lins = [], [], []
                n_acc = 0
                for rep in range(6 if order == "default" else 4):
                    for h in hs:
                        h.set_case(case, slots)
                    torch.cuda.synchronize(dev)
                    B.set_profile(order == "default" and rep >= 4)   # the last two repetitions carry the events around one stepped linearisation (not used for the wall figure)
                    t0 = time.perf_counter(); rs = B.optimize(hs, 6); wall = time.perf_counter() - t0
                    ms = B.last_ms()
                    if order == "default" and rep == 3:
                        row["host_phase_us"] = [round(x, 1) for x in B.last_host_us()]
                    if order == "default" and rep >= 4:
                        lins.append(ms[2])
                    elif rep >= 1:
                        walls.append(wall); loops.append(ms[0] + ms[1])
                    n_acc = sum(int(r["trace"][1:, 3].sum()) for r in rs)
                wall = float(np.median(walls)); loop_ms = float(np.median(loops))
                if order == "default":
                    lin_us = 1e3 * float(np.median(lins))
                    row.update(accepted_iterations=n_acc, wall_ms=round(1e3 * wall, 4), device_ms=round(loop_ms, 4), value=round(n_acc / wall, 1),
                               us_per_iteration_per_window=round(1e6 * wall / max(n_acc, 1) * W, 2), us_per_accepted_iteration=round(1e6 * wall / max(n_acc, 1), 3),
                               k_ba_linearize_b_us=round(lin_us, 2), k_ba_linearize_b_GBs=round(W * bytes_lin / (lin_us * 1e-6) / 1e9, 1),
                               k_ba_linearize_b_frac=round(W * bytes_lin / (lin_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4))
                else:
                    row.update(value_single_threaded_order=round(n_acc / wall, 1), wall_ms_single_threaded_order=round(1e3 * wall, 4))
            rows.append(row)
        default_order = dict(windows=rows[-1]["windows"], value=rows[-1]["value"])
    finally:
        for h in pool + pool_default:
            h.close()
        B.close()
    best = max(rows, key=lambda r: r["value"])
    ach = bytes_iter * best["value"] / 1e9
    return dict(unit="GN-iters/s", value=best["value"], at_windows=best["windows"], sweep=rows, accumulators_per_bucket=4, default_accumulation_order=default_order,
                value_single_threaded_order=max(r["value_single_threaded_order"] for r in rows),
                single_window=dict(optimize6_ms=rows[0]["wall_ms"], us_per_accepted_iteration=rows[0]["us_per_accepted_iteration"],
                                   what="dmvio_hip_ba_optimize_batch of ONE window = dmvio_hip_ba_optimize with dmvio_hip_ba_set_device_loop(1): the whole loop enqueued up front, two host waits per call"),
                roofline=dict(bound="hbm", kernel="k_ba_linearize_b1" if best["windows"] >= 4 else "k_ba_linearize_b", achieved=best["k_ba_linearize_b_GBs"], peak=HBM_PEAK_GBS, unit="GB/s", frac=best["k_ba_linearize_b_frac"],
                              kernel_us=best["k_ba_linearize_b_us"], algorithmic_bytes_per_launch=int(best["windows"] * bytes_lin),
                              iteration=dict(achieved=round(ach, 2), frac=round(ach / HBM_PEAK_GBS, 5)),
                              what="the stepped linearisation of ALL windows of a call (one launch, HIP events on the batch's stream; the profiled repetitions run as one group on one "
                                   "stream, alone on the device; from 4 windows on the one-lane-per-residual kernel k_ba_linearize_b1): 464 B per residual x residuals of all windows / "
                                   "its duration; `iteration`: all algorithmic bytes of an accepted iteration x accepted iterations per second"),
                what="W fresh windows (own handles in the library's DEFAULT accumulation order, 4 partial accumulators per bucket; set up before the timed region), ONE "
                     "dmvio_hip_ba_optimize_batch(6) call: accepted Gauss-Newton iterations of all windows / its wall time (`value`, every row of `sweep`); device_ms = the same "
                     "call by HIP events (loop + final fix-linearisation); value_single_threaded_order (every row): the same with handles in the reference's single-threaded "
                     "accumulation order (dmvio_hip_ba_set_accumulators(1): the bit-exact replay)")


def bench_ba(args, pkg, synth, ctx_device, rank, world, dist, dev, coll_dev, torch, cpu):
    """GN iterations / s of FullSystem::optimize's loop body (solveSystemF + doStepFromBackup + linearizeAll + energies + applyRes)
    on an 8-keyframe, ~2000-point, ~12k-residual window."""
    import dmvio_amd.sharding as sh
    w = h = args.size
    # N == 1: the rank optimises the whole window.  N > 1: ONE window, its points sharded by host keyframe over the ranks
    # (north_star: "one keyframe per GPU"); the exchanges of the sharded iteration run behind the C ABI (dmvio_hip_ba_set_comm: RCCL
    # all-reduce of the packed system in HBM + all-gather of the decision records, on the BA handle's stream) — strong scaling.
    case_full = synth.ba_case(w, h, n_frames=8, n_points=args.ba_points, seed=synth.SEED)
    parts = sh.partition_points_by_host(case_full["host"], world)
    case = sh.shard_case(case_full, parts[rank]) if world > 1 else case_full
    F = case["n_frames"]
    ctx = pkg.Context(w, h, n_slots=F, device=ctx_device)
    for k in range(F):
        ctx.frame_upload(k, case["imgs"][k])
    ba = pkg.BundleAdjusterHip(ctx)
    ba.set_case(case, list(range(F)))
    optimize_ms = None
    handoff_ms = None
    chain_us = None
    if world == 1:
        # correctness guard: the full optimize must decrease the energy; timed on fresh windows (every step accepted until convergence)
        r = ba.optimize(6)
        if not (r["trace"][-1, 0] < 0.7 * r["trace"][0, 0]):
            raise SystemExit("bench: BA did not converge")
        n_acc_fresh = int(r["trace"][1:, 3].sum())
        ts = []
        for _ in range(10):
            ba.set_case(case, list(range(F)))
            t0 = time.perf_counter(); ba.optimize(6); ts.append(time.perf_counter() - t0)
        optimize_ms = 1e3 * float(np.median(ts))
        # the same call with every solve handed to a hook (the reference's default GTSAM branch, dmvio_hip_ba_optimize_vio); the hook is the library's own LDLT as a C
        # function, so the figure is the price of the hand-off itself (system assembled for computeBAUpdate, frame views, the extra hooks' call sites)
        ts = []
        for _ in range(10):
            ba.set_case(case, list(range(F)))
            t0 = time.perf_counter(); rv = ba.optimize_vio_own_solver(6); ts.append(time.perf_counter() - t0)
        handoff_ms = 1e3 * float(np.median(ts))
        if not np.array_equal(rv["trace"], r["trace"]):
            raise SystemExit("bench: the hook path did not reproduce dmvio_hip_ba_optimize")
        ba.set_case(case, list(range(F)))
        ba.activate_all(); ba.linearize_all(False); ba.apply_res(); ba.accumulate()
        chain_us = ba.profile_chain(30)
        ba.set_case(case, list(range(F)))
    replicas = None
    batched_replicas = None
    comm = None
    if world > 1:
        # reference point for the sharded figure: every rank optimising its OWN whole window (independent windows, no exchange)
        ba_full = pkg.BundleAdjusterHip(ctx)
        ba_full.set_case(case_full, list(range(F)))
        ba_full.activate_all(); e_f = ba_full.linearize_all(False); ba_full.apply_res()
        lam_f, lastE_f = 1e-5, [e_f, 0.0, 0.0]
        for it in range(12):
            _, lam_f, lastE_f = ba_full.gn_iteration(it % 6, lam_f, lastE_f)
        torch.cuda.synchronize(dev); dist.barrier()
        t0 = time.perf_counter()
        for it in range(args.ba_iters):
            _, lam_f, lastE_f = ba_full.gn_iteration(it % 6, lam_f, lastE_f)
        torch.cuda.synchronize(dev)
        tr = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tr, op=dist.ReduceOp.MAX)
        replicas = world * args.ba_iters / float(tr.item())
        ba_full.close()
        # ... and the batched device-resident loop on W = 16 windows of its own per rank (dmvio_hip_ba_optimize_batch): the fast path's weak-scaling figure
        try:
            Wb = 16
            Bb = pkg.BundleAdjusterBatch(ctx, Wb)
            poolb = [pkg.BundleAdjusterHip(ctx) for _ in range(Wb)]
            walls_b = []
            n_acc_b = 0
            for rep in range(4):
                for hb in poolb:
                    hb.set_case(case_full, list(range(F)))
                torch.cuda.synchronize(dev); dist.barrier()
                t0 = time.perf_counter(); rsb = Bb.optimize(poolb, 6); torch.cuda.synchronize(dev)
                tb = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=coll_dev)
                dist.all_reduce(tb, op=dist.ReduceOp.MAX)
                if rep >= 1:
                    walls_b.append(float(tb.item()))
                nb = torch.tensor([float(sum(int(r["trace"][1:, 3].sum()) for r in rsb))], dtype=torch.float64, device=coll_dev)
                dist.all_reduce(nb, op=dist.ReduceOp.SUM)
                n_acc_b = float(nb.item())
            batched_replicas = dict(windows_per_rank=Wb, value=round(n_acc_b / float(np.median(walls_b)), 1), value_per_rank=round(n_acc_b / float(np.median(walls_b)) / world, 1),
                                    what="every rank: ONE dmvio_hip_ba_optimize_batch(16 fresh windows of its own, 6 iterations) call, all ranks started together; accepted iterations of "
                                         "all ranks / the slowest rank's wall time (weak scaling of the batched device-resident loop; default accumulation order)")
            for hb in poolb:
                hb.close()
            Bb.close()
        except Exception as ex:
            batched_replicas = dict(error="%s: %s" % (type(ex).__name__, ex))
        if coll_dev.type != "cpu":
            # the library's own RCCL communicator: the unique id travels over the process group that launched us
            uid = torch.zeros(128, dtype=torch.uint8, device=coll_dev)
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(pkg.RcclCommunicator.unique_id(ctx.L)), dtype=torch.uint8))
            dist.broadcast(uid, 0)
            torch.cuda.synchronize(dev)
            comm = pkg.RcclCommunicator(ctx, bytes(uid.cpu().numpy().tobytes()), rank, world)
            ba.set_comm(comm, rank, world)
            ba.comm_timing(True)
            transport = "RCCL (ncclAllReduce fp64 sum + ncclAllGather on the BA stream)"
        else:
            ba.set_comm_torch(dist)    # single-GPU test hook (gloo, all ranks on one device): RCCL refuses two ranks per device
            transport = "host-staged callbacks over gloo (test hook)"
    ba.activate_all(); e0 = ba.linearize_all(False); ba.apply_res()
    lam, lastE = 1e-5, [e0, 0.0, 0.0]
    for it in range(12):  # warmup (first touches of the freshly allocated window buffers)
        _, lam, lastE = ba.gn_iteration(it % 6, lam, lastE)
    if not (lastE[0] < 0.8 * e0):
        raise SystemExit("bench: BA did not reduce the energy (%g -> %g)" % (e0, lastE[0]))
    torch.cuda.synchronize(dev)
    if comm is not None:
        ba.comm_timing(True)            # from here: the collectives of the timed iterations
    if dist is not None:
        dist.barrier()
    n_it = args.ba_iters
    t0 = time.perf_counter()
    done = 0
    while done < n_it:   # keep iterating on the same window (accepted or rejected, an iteration does the same work) — like the CPU leg below
        acc, lam, lastE = ba.gn_iteration(done % 6, lam, lastE)
        done += 1
    torch.cuda.synchronize(dev)
    elapsed_calls = time.perf_counter() - t0
    # the same iterations inside the library's own FullSystem::optimize loop (the reference's call surface): no foreign-function call per iteration, and after a
    # rejected step the host goes on to the next solve while the restored state is still being relinearised (its energy stays on the device for the accept test)
    per_call = 50
    n_calls = max(1, n_it // per_call)
    ba.optimize(per_call)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    n_acc = 0
    for _ in range(n_calls):
        n_acc += int(ba.optimize(per_call)["trace"][1:, 3].sum())
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    done = n_calls * per_call
    case = case_full
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    conv_value = done / elapsed
    Rn = int(len(case["res_point"])); Nn = int(len(case["u"]))
    npairs = int(sum(int(c) ** 2 for c in np.bincount(case["res_point"], minlength=Nn)))
    # algorithmic bytes of ONE accepted iteration: linearise 464 B read + 208 B record written per residual; per-point sums 180 B per residual;
    # accumulation 204 B per (host,target) member (= residual) + 68 B per Schur member (= residual pair of a point); the 68x68 system itself is noise next to these
    bytes_lin = Rn * 464
    bytes_iter = Rn * (464 + 208) + Rn * 180 + Rn * 204 + npairs * 68
    out = dict(metric="BA GN-iterations/sec (8-KF window)", unit="GN-iters/s",
               scaling="strong" if world > 1 else None, shard_points=[int(len(p)) for p in parts],
               window=dict(frames=F, points=Nn, residuals=Rn, schur_members=npairs),
               value_converged_loop=round(conv_value, 1), ms_per_iter_converged_loop=round(1e3 * elapsed / done, 4),
               value_per_call_api=round(n_it / elapsed_calls, 1),
               loop="`value`: ACCEPTED Gauss-Newton iterations on fresh windows = 6 / wall time of dmvio_hip_ba_optimize(6) (what FullSystem::optimize does per keyframe; the "
                    "initial linearisation + accumulation and the final fix-linearisation of the call are charged to its iterations); `value_converged_loop`: %d x "
                    "dmvio_hip_ba_optimize(%d) on the converged window, where most steps are tried and REJECTED (a rejected iteration skips accumulation and stitching; this was "
                    "`value` up to round 2); `value_per_call_api`: one dmvio_hip_ba_gn_iteration call per iteration from the harness" % (n_calls, per_call),
               algorithmic_bytes_per_iter=int(bytes_iter),
               accumulation="4 partial accumulators per bucket (the structure of the reference's multi-threaded accumulation); "
                            "`value_single_threaded_order` replays the reference's single-threaded summation order bit for bit",
               note="N=1: whole window on the GPU; N>1: ONE window, points sharded by host keyframe, per iteration one all-reduce of the packed 68x68 systems + one all-gather per "
                    "linearisation, both inside the library (latency-bound strong scaling of a <0.1 ms iteration); `independent_windows_value` = every GPU optimising its own window "
                    "(weak scaling) — the figure that answers north_star's '>= 3.5x at 8 GPUs'")
    if world > 1:
        out["transport"] = transport
        out["value"] = round(conv_value, 1)
        out["value_definition"] = "N > 1: the sharded window is timed on the converged loop (value_converged_loop); fresh-window timing is an N = 1 figure"
        if comm is not None:
            nr, rr = comm.info()
            out["rccl_ranks"] = int(nr)          # ncclCommCount of the communicator the sharded iteration ran on
            out["rccl_rank_of_reporter"] = int(rr)
            ct = ba.comm_times()                 # HIP events around the two collectives of the sharded iteration on this rank's BA stream
            out["allreduce_us"] = round(ct["allreduce_us"], 2); out["allgather_us"] = round(ct["allgather_us"], 2)
            out["collectives_issued"] = dict(allreduce=ct["allreduces"], allgather=ct["allgathers"])
            out["collectives_what"] = ("mean duration of the sharded iteration's ncclAllReduce (packed 68x68 systems, %d doubles) and ncclAllGather (decision records) on the BA "
                                       "stream, the first 64 of each in the timed loop; a sharded accepted iteration pays one of each plus one more all-gather per rejected step"
                                       % (2 * ((4 + 8 * F) ** 2 + (4 + 8 * F)) + 1))
    out["accepted_in_converged_loop"] = "%d of %d" % (n_acc, done)
    if optimize_ms is not None:
        out["value"] = round(6.0 / (optimize_ms * 1e-3), 1)
        out["ms_per_iter"] = round(optimize_ms / 6.0, 4)
        out["optimize6_ms"] = round(optimize_ms, 4)      # FullSystem::optimize(6) on a fresh window: initial linearisation + 6 iterations + the final fix-linearisation
        out["accepted_in_fresh_window"] = "%d of 6" % n_acc_fresh
        out["gtsam_handoff"] = dict(optimize6_ms=round(handoff_ms, 4), ratio_to_builtin=round(handoff_ms / optimize_ms, 3),
                                    what="dmvio_hip_ba_optimize_vio(6) on the same fresh windows, computeBAUpdate = dmvio_hip_ba_hook_ldlt (the library's own solve behind the "
                                         "hook: identical trace, checked); the reference's default branch (setting_useGTSAMIntegration) costs this much more than the built-in loop")
    if chain_us is not None:
        lin_us = chain_us["k_ba_linearize"]
        chain_total = float(sum(chain_us.values()))
        out["roofline"] = dict(bound="hbm", kernel="k_ba_linearize", achieved=round(bytes_lin / (lin_us * 1e-6) / 1e9, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                               frac=round(bytes_lin / (lin_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5), traffic=None, kernel_us=round(lin_us, 2),
                               algorithmic_bytes_per_launch=int(bytes_lin),
                               chain_us={k: round(v, 2) for k, v in chain_us.items()},
                               iteration=dict(achieved=round(bytes_iter / (optimize_ms / 6.0 * 1e-3) / 1e9, 2), frac=round(bytes_iter / (optimize_ms / 6.0 * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                              kernels_us=round(chain_total, 2), wall_us=round(1e3 * optimize_ms / 6.0, 2),
                                              what="all algorithmic bytes of an accepted iteration / its wall time inside optimize(6); kernels_us = the five kernels of the chain, "
                                                   "HIP events on the BA stream (dmvio_hip_ba_profile_chain)"),
                               note="the window's working set (12.8k residuals x ~1 KB) lives in L2 / Infinity Cache: the iteration is bound by launch + dependent-chain latency, "
                                    "not by HBM — the fraction says how far from the HBM roofline a latency-bound path sits")
    if replicas is not None:
        out["independent_windows_value"] = round(replicas, 1)
        out["independent_windows_value_per_rank"] = round(replicas / world, 1)
    if batched_replicas is not None:
        out["independent_batched_windows"] = batched_replicas
    if world == 1 and not getattr(args, "no_concurrent", False):
        out["concurrent_windows"] = bench_ba_concurrent(args, pkg, ctx, case, F, bytes_iter, bytes_lin, torch, dev)
    if world == 1:
        try:
            out["batched_windows"] = bench_ba_batched(args, pkg, ctx, case, F, bytes_iter, bytes_lin, torch, dev)
        except Exception as ex:   # reported in the line, never swallowed
            out["batched_windows"] = dict(error="%s: %s" % (type(ex).__name__, ex))
    if world == 1:
        ba1 = pkg.BundleAdjusterHip(ctx, accumulators=1)
        ba1.set_case(case, list(range(F)))
        ts = []
        for _ in range(6):
            ba1.set_case(case, list(range(F)))
            t0 = time.perf_counter(); ba1.optimize(6); ts.append(time.perf_counter() - t0)
        out["value_single_threaded_order"] = round(6.0 / float(np.median(ts)), 1)
        ba1.close()
    if cpu:
        O = graft.load_oracle()
        res = {}
        for threads in (1, 6):
            ts = []
            t_leg = time.perf_counter()
            while len(ts) < 2 or (time.perf_counter() - t_leg < max(1.5, args.cpu_seconds / 6) and len(ts) < 20):
                W = O.BAWindow(case, threads=threads)
                t0 = time.perf_counter(); W.optimize(6); ts.append(time.perf_counter() - t0)
            res[threads] = 6.0 / float(np.median(ts))
        best = 6 if res[6] >= res[1] else 1
        out["cpu_baseline"] = dict(value=round(res[best], 2), unit="GN-iters/s", cores=best, kind="port", value_6workers=round(res[6], 2), value_1thread=round(res[1], 2),
                                   sample="oracle optimize(6) on fresh copies of the same window (6 / wall time): 6 workers (NUM_THREADS 6, persistent pool for linearizeAll + "
                                          "accumulation + resubstitution) and single-threaded, on %s" % _cpu_name())
        # the reference's OWN FullSystem::optimize (oracle/_ref/libref.so: FullSystemOptimize.cpp, EnergyFunctional.cpp, Accumulated*Hessian.cpp compiled unmodified), the same
        # fresh window, multiThreading on (NUM_THREADS 6, what dmvio_dataset runs) and off
        try:
            Rf = graft.load_reference()
            if Rf.available():
                rres = {}
                for mt in (1, 0):
                    ts = []
                    t_leg = time.perf_counter()
                    while len(ts) < 2 or (time.perf_counter() - t_leg < max(2.0, args.cpu_seconds / 4) and len(ts) < 12):
                        WR = Rf.BAWindow(case)              # (building the pointer graph from flat arrays is not timed)
                        Rf.set_multithreading(mt)
                        t0 = time.perf_counter(); WR.optimize_quiet(6); ts.append(time.perf_counter() - t0)
                        del WR
                    rres[mt] = (6.0 / float(np.median(ts)), len(ts))
                Rf.set_multithreading(0)
                bestr = 1 if rres[1][0] >= rres[0][0] else 0
                port = out["cpu_baseline"]
                out["cpu_baseline"] = dict(value=round(rres[bestr][0], 2), unit="GN-iters/s", cores=6 if bestr else 1, kind="reference",
                                           value_multithreaded=round(rres[1][0], 2), value_single_threaded=round(rres[0][0], 2), value_port=port["value"], port=port,
                                           sample="the reference's own FullSystem::optimize(6) (its sources compiled -O3 -msse2 by oracle/Makefile.ref against stand-in Eigen / Sophus "
                                                  "headers) on %d + %d fresh copies of the same window, multiThreading on (NUM_THREADS 6) and off: 6 / median wall time, on %s"
                                                  % (rres[1][1], rres[0][1], _cpu_name()))
        except Exception as ex:
            sys.stderr.write("bench: reference BA not timed (%s: %s)\n" % (type(ex).__name__, ex))
    ba.close()
    if comm is not None:
        comm.close()
    ctx.close()
    return out


def bench_dropin(args, w, h):
    """the reference's OWN FullSystem (oracle/_ref/libref.so: its sources compiled unmodified) run over one synthetic sequence in child processes —
    all-CPU, and with the seven hot-path members (FrameHessian::makeImages, CoarseTracker::setCoarseTrackingRef / trackNewestCoarse, FullSystem::traceNewCoarse /
    activatePointsMT_Reductor / optimize, CoarseInitializer::calcResAndGS) re-defined on top of libdmvio_hip.so by the compiled INTEGRATION.md adapter
    (oracle/_ref/libdropin_hip.so, tests/dropin/).  Wall clock = the loop of addActiveFrame calls alone (sequence rendered before, recording of the run switched off)."""
    import subprocess
    import tempfile
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    if not (os.path.exists(os.path.join(ref_dir, "libref.so")) and os.path.exists(os.path.join(ref_dir, "libdropin_hip.so"))):
        return dict(error="oracle/_ref/libref.so / libdropin_hip.so not built (they are built where /root/reference exists and travel with the tree)")
    tmp = tempfile.mkdtemp(prefix="dmvio_dropin_")
    seq = ["--w", str(w), "--h", str(h), "--frames", str(args.dropin_frames), "--step", "1.6", "--density", "2000", "--cache", tmp, "--scopes"]

    def run(name, *extra):
        outp = os.path.join(tmp, name + ".npz")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dropin", "run_dropin.py"), "--out", outp] + seq + list(extra), capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            raise RuntimeError("run_dropin %s failed: %s" % (name, (r.stdout + r.stderr)[-400:]))
        return np.load(outp)

    def scopes(z, names):
        t = dict(zip(map(str, z["scope_labels"]), z["scope_seconds"]))
        return {k: round(float(t.get(k, 0.0)), 4) for k in names}
    cpu_mt = run("cpu_mt", "--mode", "cpu", "--init", "ref", "--mt")     # the reference as it ships: multiThreading = true (6 workers), its own initialiser
    cpu_st = run("cpu_st", "--mode", "cpu", "--init", "seq")             # deterministic single-threaded baseline (the trajectory the HIP-backed run is compared with)
    hip = run("hip", "--mode", "hip", "--init", "hip")
    hip_mt = run("hip_mt", "--mode", "hip", "--init", "hip", "--mt")     # what the reference keeps doing itself on its 6 workers, like cpu_mt: the like-for-like wall clock
    # the reference's DEFAULT configuration (setting_useIMU / setting_useGTSAMIntegration) with the live stand-in for the absent IMU / GTSAM side behind the facade on both sides
    vio = None
    try:
        vcpu = run("vio_cpu", "--mode", "cpu", "--init", "seq", "--vio")
        vhip = run("vio_hip", "--mode", "hip", "--init", "hip", "--vio")
        vv = (vcpu["valid"] != 0) & (vhip["valid"] != 0)
        vd = vcpu["camToWorld"][vv, :3] - vhip["camToWorld"][vv, :3]
        nk = max(int(vhip["stat_calls"][4]), 1)
        vio = dict(what="the same run with setting_useIMU = setting_useGTSAMIntegration = true: trackNewestCoarse through dmvio_hip_tracker_track_vio (IMUIntegration::computeCoarseUpdate "
                        "/ acceptCoarseUpdate / addVisualToCoarseGraph as callbacks), optimize through dmvio_hip_ba_optimize_vio (the seven BAGTSAMIntegration members as callbacks); "
                        "behind the facade on BOTH sides the same stand-in (oracle/ref_glue.cpp: VioStandIn)",
                   all_cpu_single_threaded_s=round(float(vcpu["wall_s"][0]), 3), hip_backed_s=round(float(vhip["wall_s"][0]), 3),
                   traj_rmse_m=float(np.sqrt((vd ** 2).sum(1).mean())), traj_max_m=float(np.abs(vd).max()),
                   adapter_calls=dict(zip(["track_vio_computeCoarseUpdate", "track_vio_visual_step", "optimize_vio", "facade_members_called"], [int(x) for x in vhip["vio_adapter"]])),
                   facade_calls_all_cpu=[int(x) for x in vcpu["vio_counters"][:11]], facade_calls_hip_backed=[int(x) for x in vhip["vio_counters"][:11]],
                   adapter_ms_per_keyframe=dict(zip(["hand_over", "dmvio_hip_ba_optimize_vio", "write_back"], [round(1e3 * float(x) / nk, 4) for x in vhip["optimize_split_seconds"]])),
                   adapter_failures=int(vhip["failures"][0]), lost=bool(vhip["lost"][-1]))
    except Exception as e:   # noqa: BLE001
        vio = dict(error=str(e)[-300:])
    v = (cpu_st["valid"] != 0) & (hip["valid"] != 0)
    d = cpu_st["camToWorld"][v, :3] - hip["camToWorld"][v, :3]
    v2 = (cpu_st["valid"] != 0) & (cpu_mt["valid"] != 0)
    d2 = cpu_st["camToWorld"][v2, :3] - cpu_mt["camToWorld"][v2, :3]
    names = ["InitializerOtherFrames", "initObjectsAndMakeImage", "fullCoarseTracking", "traceNewCoarse", "FullSystemOptimize", "makeKeyframe", "makeNewTraces", "activatePointsMT",
             "marginalizeAndRemovePoints"]
    n_init = int(np.argmax(hip["initialized"] != 0)) + 1 if (hip["initialized"] != 0).any() else 0

    def steady(z):   # seconds per frame once the initialiser has handed over: everything outside the initialiser's scopes
        t = dict(zip(map(str, z["scope_labels"]), z["scope_seconds"]))
        n = len(z["valid"]) - n_init
        return (float(z["wall_s"][0]) - t.get("InitializerOtherFrames", 0.0) - t.get("InitializerFirstFrame", 0.0)) / max(n, 1)
    return dict(what="the reference's own FullSystem::addActiveFrame over %d synthetic %dx%d frames (~2000 active points, 7-keyframe window, visual-only, linearizeOperation): "
                     "all-CPU vs its seven hot-path members on libdmvio_hip.so through the compiled adapter (tests/dropin/dmvio_hip_adapter.cpp, ELF interposition in front of "
                     "oracle/_ref/libref.so)" % (len(hip["valid"]), w, h),
                frames=int(len(hip["valid"])), keyframe_optimisations=int(hip["stat_calls"][4]),
                all_cpu_s=round(float(cpu_mt["wall_s"][0]), 3), all_cpu_single_threaded_s=round(float(cpu_st["wall_s"][0]), 3), hip_backed_s=round(float(hip["wall_s"][0]), 3),
                hip_backed_default_threading_s=round(float(hip_mt["wall_s"][0]), 3),
                speedup_vs_reference_default=round(float(cpu_mt["wall_s"][0]) / float(hip["wall_s"][0]), 2),
                speedup_vs_reference_default_same_threading=round(float(cpu_mt["wall_s"][0]) / float(hip_mt["wall_s"][0]), 2),
                adapter_ms_per_keyframe=dict(zip(["hand_over", "dmvio_hip_ba_optimize", "write_back"], [round(1e3 * float(x) / max(int(hip["stat_calls"][4]), 1), 4) for x in hip["optimize_split_seconds"]])),
                adapter_ms_per_keyframe_default_threading=dict(zip(["hand_over", "dmvio_hip_ba_optimize", "write_back"],
                                                                   [round(1e3 * float(x) / max(int(hip_mt["stat_calls"][4]), 1), 4) for x in hip_mt["optimize_split_seconds"]])),
                window_graph=dict(zip(["forwarded_mutations", "resyncs"], [int(hip["resident"][0]), int(hip["resident"][1])])),
                marginalizePointsF_on_device=dict(zip(["calls", "points"], [int(x) for x in hip["real_marginalization"]])),
                ms_per_frame_after_initialisation=dict(all_cpu=round(1e3 * steady(cpu_mt), 3), all_cpu_single_threaded=round(1e3 * steady(cpu_st), 3), hip_backed=round(1e3 * steady(hip), 3)),
                traj_rmse_m=float(np.sqrt((d ** 2).sum(1).mean())), traj_max_m=float(np.abs(d).max()),
                reference_own_spread_rmse_m=float(np.sqrt((d2 ** 2).sum(1).mean())),
                adapter_failures=int(hip["failures"][0]), lost=bool(hip["lost"][-1]),
                seconds_in_replaced_members=dict(zip(["makeImages", "setCoarseTrackingRef", "trackNewestCoarse", "traceNewCoarse", "optimize", "activatePoints", "calcResAndGS"],
                                                     [round(float(x), 4) for x in hip["stat_seconds"]])),
                scopes_all_cpu=scopes(cpu_mt, names), scopes_hip_backed=scopes(hip, names), vio=vio,
                track_new_coarse=dict(zip(["calls", "served_by_the_batched_try_loop", "past_try_0", "tries_walked"], [int(x) for x in hip["track_new_coarse"][:4]])),
                note="all_cpu_s: multiThreading = true and the reference's own thread-pooled initialiser (settings.cpp defaults), %d host threads available; all_cpu_single_threaded_s: "
                     "multiThreading = false with the oracle's sequential CoarseInitializer::calcResAndGS (bit-reproducible: the trajectory baseline); reference_own_spread = "
                     "those two all-CPU runs against each other; scopes = inclusive seconds under the reference's own util/TimeMeasurement labels" % (os.cpu_count() or 0))


def _cpu_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip() + " (%d hw threads)" % os.cpu_count()
    except Exception:
        pass
    return "unknown cpu"


if __name__ == "__main__":
    main()
