"""One BA window over several ranks with the exchanges behind the C ABI (dmvio_hip_ba_set_comm / _set_comm_callbacks)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _run(pkg, case, ctx, setup=None, its=6, before_close=None):
    ba = pkg.BundleAdjusterHip(ctx)
    ba.set_case(case, list(range(case["n_frames"])))
    if setup:
        setup(ba)
    out = ba.optimize(its)
    poses = np.stack([ba.frame_pose(k)[0] for k in range(case["n_frames"])])
    idepth = ba.point_state()[0]
    if before_close:
        before_close(ba)
    ba.close()
    return out, poses, idepth


def test_rccl_world1_runs_the_sharded_path_bit_identically(pkg, synth, gpu_required):
    """An RCCL communicator of one rank: all-reduce and all-gather degenerate to copies, so optimize through the sharded code path (records packed,
    gathered, decided by k_ba_decide_global; system reduced in HBM and published) must reproduce the single-GPU path bit for bit."""
    case = synth.ba_case(512, 512, n_frames=8, n_points=1500, seed=21)
    ctx = pkg.Context(case["w"], case["h"], n_slots=8)
    for k in range(8):
        ctx.frame_upload(k, case["imgs"][k])
    ref, rposes, rid = _run(pkg, case, ctx)
    comm = pkg.RcclCommunicator(ctx, pkg.RcclCommunicator.unique_id(ctx.L), 0, 1)
    times = {}

    def attach(ba):
        ba.set_comm(comm, 0, 1); ba.comm_timing(True)
    try:
        out, poses, idepth = _run(pkg, case, ctx, attach, before_close=lambda ba: times.update(ba.comm_times()))
    finally:
        comm.close()
    # the collectives of the sharded iteration as HIP events on the BA stream (what the N > 1 bench line reports as ba.allreduce_us / ba.allgather_us): one all-reduce per
    # accumulation, one all-gather per linearisation
    print("RCCL world 1: all-reduce %.1f us x %d, all-gather %.1f us x %d" % (times["allreduce_us"], times["allreduces"], times["allgather_us"], times["allgathers"]))
    acc = out["trace"][1:out["iterations"] + 1, 3]
    # one system per state that gets solved (the initial one + every accepted step but the last iteration's); one decision exchange per linearisation (initial, one per
    # iteration, one more per rejected step, the final fix-linearisation)
    assert times["allreduces"] == 1 + int(acc[:-1].sum()) and times["allgathers"] == 2 + len(acc) + int((acc == 0).sum())
    assert 0 < times["allreduce_us"] < 5e4 and 0 < times["allgather_us"] < 5e4
    assert out["iterations"] == ref["iterations"] and out["rmse"] == ref["rmse"] and out["finalEnergy"] == ref["finalEnergy"]
    assert np.array_equal(out["trace"], ref["trace"])
    assert np.array_equal(poses, rposes) and np.array_equal(idepth, rid)


def test_set_comm_validates_rank_and_world(pkg, synth, gpu_required):
    ctx = pkg.Context(64, 64, n_slots=2)
    ba = pkg.BundleAdjusterHip(ctx)
    comm = pkg.RcclCommunicator(ctx, pkg.RcclCommunicator.unique_id(ctx.L), 0, 1)
    try:
        with pytest.raises(pkg.HipLibraryError):
            ba.set_comm(comm, 1, 2)          # does not match the communicator
        with pytest.raises(pkg.HipLibraryError):
            ba.set_comm(comm, 3, 1)
        ba.set_comm(comm, 0, 1)
        ba.set_comm(None, 0, 0)              # detach
    finally:
        ba.close(); comm.close()


@pytest.mark.parametrize("lin", [False, True, "import"], ids=["plain", "residuals-kept-linearised", "residuals-arrive-linearised"])
def test_sharded_optimize_world2_on_a_shared_device(gpu_required, lin):
    """Two processes on one GPU, each with its share of the points: identical decisions and frame states on both ranks, the unsharded result within
    the rounding of a different summation order (tests/dist_worker_gpu.py).  Second case: after two iterations a third of the residuals is kept linearised
    (dmvio_hip_ba_fix_linearization on every rank, for its own residuals) and four more iterations run over that graph — accumulateLF_MT's system, the A and
    Schur views and calcLEnergyPt's term are each all-reduced."""
    port = _free_port()
    env = dict(os.environ); env["MASTER_ADDR"] = "127.0.0.1"
    if lin:
        env["SHARD_LIN"] = "import" if lin == "import" else "1"   # import: the rows of an unsharded window handed to the ranks (dmvio_hip_ba_set_linearized_residuals, collective)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker_gpu.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("OK") == 2, r.stdout
