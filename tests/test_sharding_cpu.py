"""CPU tests of the N>1 path: world-size-2 gloo run of the sharded-BA protocol + unit tests of the partition / packing helpers."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_partition_by_host_covers_and_balances(pkg):
    import dmvio_amd.sharding as sh
    host = np.repeat(np.arange(8), [400, 350, 300, 300, 250, 250, 150, 0])
    for world in (1, 2, 4, 8):
        parts = sh.partition_points_by_host(host, world)
        assert len(parts) == world
        assert np.array_equal(np.sort(np.concatenate(parts)), np.arange(len(host)))
        sizes = np.array([len(p) for p in parts])
        assert sizes.max() <= 1.25 * len(host) / world + 1
    # whole keyframes stay together when that is balanced enough
    parts = sh.partition_points_by_host(host, 2, max_imbalance=1.25)
    assert all(len(np.unique(host[p])) >= 1 for p in parts)
    hs = [set(host[p].tolist()) for p in parts]
    assert not (hs[0] & hs[1])


def test_pack_unpack_roundtrip(pkg):
    import dmvio_amd.sharding as sh
    n = 36
    rng = np.random.RandomState(0)
    HA, Hsc = rng.normal(size=(n, n)), rng.normal(size=(n, n)); bA, bsc = rng.normal(size=n), rng.normal(size=n)
    out = sh.unpack_system(sh.pack_system(HA, bA, Hsc, bsc, 12.5, 777), n)
    assert np.array_equal(out[0], HA) and np.array_equal(out[1], bA) and np.array_equal(out[2], Hsc) and np.array_equal(out[3], bsc)
    assert out[4] == 12.5 and out[5] == 777


def test_new_frame_energy_th_matches_oracle(pkg, oracle, synth):
    import dmvio_amd.sharding as sh
    case = synth.ba_case(w=320, h=256, n_frames=4, n_points=240, hosts_share=(100, 80, 60, 0), seed=777)
    W = oracle.BAWindow(case)
    W.activate_all(); W.linearize_all(False)
    st = W.res_state()
    e = st["newEnergyWO"][(case["res_target"] == 3) & (st["newEnergyWO"] >= 0)]
    assert sh.new_frame_energy_th(e) == W.frame_energy_th()[3]


def test_sharded_ba_protocol_gloo_world2():
    """2 processes, gloo backend: all-reduced shard systems == the full window's system; gathered threshold identical."""
    port = _free_port()
    env = dict(os.environ); env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("OK") == 2, r.stdout
