"""Real RCCL at world size > 1 — runs only on a box with at least two GPUs (`-m multigpu`, also collected by `-m gpu`): the sharded BA iteration
(ncclAllReduce of the packed system + ncclAllGather of the decision records, both inside the library on the BA stream) and the hypothesis-parallel trackNewCoarse
(ncclAllReduce of the per-try records) over one process per GPU.  On single-GPU boxes the same code paths are covered with the gloo / thread transports and an RCCL
communicator of one rank (tests/test_sharded_ba_gpu.py, tests/test_tracker_gpu.py)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.multigpu]


def _n_devices():
    try:
        sys.path.insert(0, ROOT)
        import __graft_entry__ as graft
        return graft.load_package().load_library().dmvio_hip_device_count()
    except Exception:
        return 0


@pytest.mark.skipif(_n_devices() < 2, reason="needs >= 2 GPUs (RCCL refuses two ranks on one device)")
@pytest.mark.parametrize("world", [2, 4, 8])
def test_rccl_world_n(world):
    if _n_devices() < world:
        pytest.skip("only %d GPUs" % _n_devices())
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ); env["MASTER_ADDR"] = "127.0.0.1"; env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "dist_worker_rccl.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-4000:]
    assert r.stdout.count("OK ba") == 1 and r.stdout.count("OK track") == 1, r.stdout
