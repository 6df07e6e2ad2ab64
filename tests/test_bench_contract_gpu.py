"""The driver's contract with bench.py: ONE JSON line on stdout, BASELINE.json's metric on the named workload, whole-job value consistent with its own step time, a
roofline object whose fraction follows from its own numbers and a cpu_baseline object timed on this box (reduced sizes here; the default run is the judged one)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract(gpu_required):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--batch", "256", "--no-sweep", "--no-traffic",
           "--cpu-seconds", "2", "--ba-iters", "50", "--dropin-frames", "60"]
    p = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=400)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "frames/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["frames_per_step_per_gpu"] == 256
    assert abs(d["value"] - 256 / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]                 # whole-job frames/s == frames per step / step time
    q = d["ms_per_step_p10_p50_p90"]
    assert len(q) == 3 and q[0] <= q[1] <= q[2] and d["ms_per_step_min_max"][0] <= q[0] and q[2] <= d["ms_per_step_min_max"][1]      # the K timed steps one by one
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["kernel"] == "k_track_lm"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-3 * r["achieved"]
    assert r["algorithmic_bytes_per_launch"] == 64 * r["point_evals_per_launch"]                 # 64 B per template-point evaluation
    assert r["kernel_ms"] < d["ms_per_step"] and r["traffic"] is None                           # --no-traffic: not measured, not invented
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] == 1 and c["unit"] == "frames/s" and c["value"] > 0 and "frames" in c["sample"]
    assert d["value"] > 20 * c["value"]
    ba = d["ba"]
    assert ba["value"] > 0 and ba["unit"] == "GN-iters/s" and ba["cpu_baseline"]["value"] > 0
    # the BA half of the metric: accepted iterations on fresh windows, its own roofline object, the reference's own compiled optimize as the CPU baseline
    assert abs(ba["value"] - 6.0 / (ba["optimize6_ms"] * 1e-3)) < 1e-3 * ba["value"] and ba["accepted_in_fresh_window"].startswith(("5", "6"))
    assert ba["value_converged_loop"] > 0 and ba["gtsam_handoff"]["ratio_to_builtin"] < 1.5
    rb = ba["roofline"]
    assert rb["bound"] == "hbm" and rb["kernel"] == "k_ba_linearize" and rb["peak"] == 8000.0 and abs(rb["frac"] - rb["achieved"] / rb["peak"]) < 1e-4
    assert rb["algorithmic_bytes_per_launch"] == 464 * ba["window"]["residuals"]
    assert abs(rb["achieved"] - rb["algorithmic_bytes_per_launch"] / (rb["kernel_us"] * 1e-6) / 1e9) < 2e-3 * rb["achieved"]
    assert set(rb["chain_us"]) == {"k_ba_linearize", "k_ba_point_sums", "k_ba_accumulate", "k_ba_stitch", "k_ba_stitch_gather"} and rb["iteration"]["kernels_us"] < rb["iteration"]["wall_us"] * 1.2
    # W windows per launch sequence on the device-resident loop: the sweep, its own roofline object for the batched linearisation
    bw = ba["batched_windows"]
    assert "error" not in bw, bw
    assert [r_["windows"] for r_ in bw["sweep"]] == [1, 4, 16, 64] and bw["value"] == max(r_["value"] for r_ in bw["sweep"]) and bw["value"] > ba["value"]
    assert bw["roofline"]["kernel"] in ("k_ba_linearize_b", "k_ba_linearize_b1") and abs(bw["roofline"]["frac"] - bw["roofline"]["achieved"] / 8000.0) < 1e-3
    assert bw["roofline"]["algorithmic_bytes_per_launch"] == bw["at_windows"] * 464 * ba["window"]["residuals"]
    cb = ba["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["unit"] == "GN-iters/s"
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref.so")):
        assert cb["kind"] == "reference" and cb["cores"] in (1, 6) and d["trace"]["cpu_baseline"]["kind"] == "reference"
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libdropin_hip.so")):
        # the drop_in leg: the reference's own FullSystem all-CPU vs with its hot-path members on the library
        di = d["drop_in"]
        assert "error" not in di, di
        assert di["frames"] == 60 and di["keyframe_optimisations"] >= 3 and di["adapter_failures"] == 0 and not di["lost"]
        assert di["traj_rmse_m"] < 1e-3 and 0 < di["hip_backed_s"] < di["all_cpu_s"] <= di["all_cpu_single_threaded_s"] * 1.2
        assert di["ms_per_frame_after_initialisation"]["hip_backed"] < di["ms_per_frame_after_initialisation"]["all_cpu"]
        # the reference's DEFAULT configuration (VIO) through the same adapter, and the batched try loop of trackNewCoarse
        assert "error" not in di["vio"], di["vio"]
        assert di["vio"]["adapter_failures"] == 0 and not di["vio"]["lost"] and di["vio"]["traj_rmse_m"] < 1.5e-3 and di["vio"]["adapter_calls"]["optimize_vio"] >= 3
        assert di["track_new_coarse"]["served_by_the_batched_try_loop"] > 0
    assert d["pcie"]["good"] and d["pcie"]["raw_u8"]["good"] and d["pcie"]["raw_u8"]["value"] > d["pcie"]["value"]
    assert d["max_pose_err_m"] < 5e-3


@pytest.mark.parametrize("launcher", ["torch.distributed.run", "plain"])
def test_bench_two_ranks_share_the_device(gpu_required, launcher):
    """The N > 1 code path of bench.py in BOTH forms the driver uses — under torch.distributed.run (one process per rank) and as plain `python bench.py --gpus 2`, which starts
    its own ranks — on a one-GPU box: both ranks on device 0, collectives over gloo (bench.py's test hooks — RCCL refuses two ranks per device).  One JSON line from rank 0,
    whole-job value = N x frames per step / max-over-ranks step time, the BA leg sharded over the two ranks through the library's callback transport."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, DMVIO_BENCH_BACKEND="gloo", DMVIO_BENCH_SHARE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    bench = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "256", "--ba-iters", "50"]
    if launcher == "plain":
        cmd = [sys.executable] + bench
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port)] + bench
    p = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=400, env=env)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3
    assert ("self-launched" in d["launcher"]) == (launcher == "plain") and d["rccl_ranks"] == 0 and d["process_group"] == "gloo"     # gloo hook: no rank exchanged over RCCL
    assert d["ms_per_step_ranks"]["min"] <= d["ms_per_step_ranks"]["max"] == d["ms_per_step"]
    assert abs(d["value"] - 2 * 256 / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    assert d["config"]["frames_per_step_per_gpu"] == 256 and "x2" in d["config"]["parallelism"]
    ba = d["ba"]
    assert "error" not in ba and ba["value"] > 0 and len(ba["shard_points"]) == 2 and sum(ba["shard_points"]) == ba["window"]["points"]
    assert ba["independent_windows_value"] > 0 and "gloo" in ba["transport"]
    assert d["max_pose_err_m"] < 5e-3


def test_bench_refuses_more_gpus_than_visible(gpu_required):
    """`--gpus N` with fewer than N devices visible exits non-zero with a message and prints no line (never a 1-GPU number labelled N)."""
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "DMVIO_BENCH_SHARE_DEVICE")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=200, env=env)
    assert p.returncode != 0 and not p.stdout.strip() and b"device(s) visible" in p.stderr
