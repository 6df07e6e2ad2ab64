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


def _partition_numpy(host, world, max_imbalance=1.25):
    """Independent statement of the policy the C entry point is held against."""
    host = np.asarray(host)
    n = len(host)
    if world <= 1:
        return [np.arange(n)], 0
    counts = np.bincount(host) if n else np.zeros(0, dtype=np.int64)
    load = np.zeros(world, dtype=np.int64)
    owner = np.zeros(len(counts), dtype=np.int64)
    for kf in np.argsort(-counts, kind="stable"):
        r = int(np.argmin(load)); owner[kf] = r; load[r] += counts[kf]
    if n and load.max() > max_imbalance * max(n / world, 1.0):
        bounds = [(n * r) // world for r in range(world + 1)]
        return [np.arange(bounds[r], bounds[r + 1]) for r in range(world)], 1
    return [np.nonzero(owner[host] == r)[0] for r in range(world)], 0


def test_partition_entry_point_equals_the_stated_policy(pkg):
    """dmvio_hip_ba_partition_points (C ABI, host-only: runs without a device) against the numpy statement of the policy: random windows, every world size up to 9,
    several imbalance bounds, both outcomes (by keyframe / equal ranges); error returns for bad arguments."""
    import ctypes
    import dmvio_amd.sharding as sh
    L = pkg.load_library()
    L.dmvio_hip_ba_partition_points.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
    L.dmvio_hip_ba_partition_points.restype = ctypes.c_int
    rng = np.random.RandomState(3)
    seen = set()
    for trial in range(300):
        F = int(rng.randint(1, 13))
        n = int(rng.choice([0, 1, 7, 500, 2000, 4001]))
        weights = rng.rand(F) ** 2 + (trial % 3 == 0) * 0.5
        host = np.sort(rng.choice(F, size=n, p=weights / weights.sum())).astype(np.int32)
        if trial % 5 == 0:
            host = rng.permutation(host).astype(np.int32)            # points need not be grouped by host
        for world in (1, 2, 3, 4, 8, 9):
            for imb in (1.0, 1.25, 2.0):
                ref, kind = _partition_numpy(host, world, imb)
                owner = np.full(n, -1, dtype=np.int32)
                r = L.dmvio_hip_ba_partition_points(host.ctypes.data, n, world, imb, owner.ctypes.data)
                assert r == kind, (trial, world, imb, r, kind)
                for q in range(world):
                    assert np.array_equal(np.nonzero(owner == q)[0], ref[q]), (trial, world, imb, q)
                assert all(np.array_equal(a, b) for a, b in zip(sh.partition_points_by_host(host, world, imb), ref))
                seen.add((kind, world > 1))
    assert {(0, True), (1, True), (0, False)} <= seen
    host = np.array([0, 1, -1], dtype=np.int32); owner = np.zeros(3, dtype=np.int32)
    assert L.dmvio_hip_ba_partition_points(host.ctypes.data, 3, 2, 1.25, owner.ctypes.data) < 0
    assert L.dmvio_hip_ba_partition_points(host.ctypes.data, 3, 0, 1.25, owner.ctypes.data) < 0
    assert L.dmvio_hip_ba_partition_points(None, 3, 2, 1.25, owner.ctypes.data) < 0
    host = np.repeat(np.arange(8), [400, 350, 300, 300, 250, 250, 150, 0]).astype(np.int32); owner = np.zeros(len(host), dtype=np.int32)
    assert L.dmvio_hip_ba_partition_points(host.ctypes.data, len(host), 2, 0.0, owner.ctypes.data) == 0        # max_imbalance <= 0: the default 1.25
    assert L.dmvio_hip_ba_partition_points(host.ctypes.data, len(host), 8, 0.0, owner.ctypes.data) == 1        # one keyframe per GPU at 8 GPUs: 400 > 1.25 x 250 -> equal ranges


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
