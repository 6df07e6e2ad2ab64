"""Multi-GPU sharding of the sliding-window BA.

Units = points (with all their residuals): GPU g owns the points hosted in "its" keyframes, every GPU keeps all F
level-0 images and the precalc/adjoint tables.  Per-point quantities (Hdd, bd, Hcd, HdiF, all JpJdF pairs) never leave
a GPU; the ONLY exchange of a Gauss-Newton iteration is the sum of the packed dense system

    [ H_A (n*n) | b_A (n) | H_sc (n*n) | b_sc (n) | energy | resInA ]      n = 4 + 8F  (75 KB for F = 8)

— one small all-reduce (RCCL over xGMI on GPUs, gloo in the CPU tests) — plus ONE fixed-width all-gather per linearisation carrying
[local energy | count | newest-frame residual energies] for the accept test and setNewFrameEnergyTH (FullSystemOptimize.cpp:96-149).
Every rank then solves the identical reduced system.

This module is harness-level plumbing (numpy + torch.distributed); the arithmetic lives behind the C ABI.
"""
import numpy as np


def partition_points_by_host(host, world, max_imbalance=1.25):
    """Point indices owned by every rank — the library's policy (dmvio_hip_ba_partition_points, include/dmvio_hip.h: whole keyframes per rank, largest first, greedy; equal
    contiguous point ranges when the keyframe split is more unbalanced than max_imbalance).  This is only a caller: a C++ host gets the same split from the same entry point."""
    import ctypes
    from . import load_library, HipLibraryError
    host = np.ascontiguousarray(host, dtype=np.int32)
    n = len(host)
    owner = np.zeros(n, dtype=np.int32)
    L = load_library()
    L.dmvio_hip_ba_partition_points.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
    L.dmvio_hip_ba_partition_points.restype = ctypes.c_int
    r = L.dmvio_hip_ba_partition_points(host.ctypes.data, n, int(world), float(max_imbalance), owner.ctypes.data)
    if r < 0:
        L.dmvio_hip_last_error.restype = ctypes.c_char_p
        raise HipLibraryError("dmvio_hip_ba_partition_points: %s" % (L.dmvio_hip_last_error() or b"").decode())
    return [np.nonzero(owner == q)[0] for q in range(world)]


def shard_case(case, idx):
    """Sub-window of a synth.ba_case holding only the points idx (sorted) and their residuals; frames are unchanged."""
    idx = np.asarray(idx)
    remap = -np.ones(len(case["u"]), dtype=np.int64)
    remap[idx] = np.arange(len(idx))
    keep = remap[case["res_point"]] >= 0
    out = dict(case)
    for k in ("host", "u", "v", "idepth_true", "idepth0", "color", "weights"):
        out[k] = case[k][idx]
    out["res_point"] = remap[case["res_point"][keep]].astype(np.int32)
    out["res_target"] = case["res_target"][keep]
    return out


def pack_system(HA, bA, Hsc, bsc, energy, res_in_a):
    return np.concatenate([np.ravel(HA), np.ravel(bA), np.ravel(Hsc), np.ravel(bsc), [float(energy), float(res_in_a)]]).astype(np.float64)


def unpack_system(buf, n):
    o = 0
    HA = buf[o:o + n * n].reshape(n, n); o += n * n
    bA = buf[o:o + n]; o += n
    Hsc = buf[o:o + n * n].reshape(n, n); o += n * n
    bsc = buf[o:o + n]; o += n
    return HA, bA, Hsc, bsc, float(buf[o]), int(round(buf[o + 1]))


def new_frame_energy_th(energies, th_n=0.7, fac_median=1.5, const_weight=0.5, overall=1.0):
    """FullSystem::setNewFrameEnergyTH (FullSystemOptimize.cpp:96-149) on the gathered newest-frame energies."""
    e = np.asarray(energies, dtype=np.float32)
    if e.size == 0:
        return np.float32(12 * 12 * 8)
    nth = int(np.float32(th_n) * e.size)
    v = np.sqrt(np.partition(e, nth)[nth]).astype(np.float32)
    th = np.float32(v * np.float32(fac_median))
    th = np.float32(26.0) * np.float32(const_weight) + th * np.float32(1 - const_weight)
    th = np.float32(th * th)
    return np.float32(th * np.float32(overall * overall))


class Collective:
    """Thin wrapper over torch.distributed (nccl = RCCL on ROCm, or gloo on CPU); world == 1 degenerates to identity.
    Buffers are persistent (no per-call allocation); with a device the exchange goes HBM -> RCCL -> HBM."""

    def __init__(self, dist=None, device=None):
        self.dist = dist
        self.device = device
        self.world = dist.get_world_size() if dist is not None else 1
        self._bufs = {}

    def _buf(self, key, n, dtype):
        import torch
        b = self._bufs.get(key)
        if b is None or b[0].numel() != n:
            host = torch.zeros(n, dtype=dtype)
            if self.device is not None:
                host = host.pin_memory()
            dev = torch.zeros(n, dtype=dtype, device=self.device) if self.device is not None else host
            b = (host, dev)
            self._bufs[key] = b
        return b

    def allreduce_sum(self, arr):
        arr = np.asarray(arr, dtype=np.float64)
        if self.world == 1:
            return arr
        import torch
        host, dev = self._buf("ar", arr.size, torch.float64)
        host.numpy()[:] = arr.ravel()
        if dev is not host:
            dev.copy_(host, non_blocking=True)
        self.dist.all_reduce(dev, op=self.dist.ReduceOp.SUM)
        if dev is not host:
            host.copy_(dev)
        return host.numpy().copy()

    def allgather_fixed(self, arr, width):
        """All ranks contribute exactly `width` float64 values; returns [world, width]."""
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        if self.world == 1:
            return arr.reshape(1, width)
        import torch
        host, dev = self._buf("ag_in", width, torch.float64)
        ohost, odev = self._buf("ag_out", width * self.world, torch.float64)
        host.numpy()[:] = arr
        if dev is not host:
            dev.copy_(host, non_blocking=True)
        self.dist.all_gather_into_tensor(odev, dev)
        if odev is not ohost:
            ohost.copy_(odev)
        return ohost.numpy().reshape(self.world, width).copy()

    def allgather_var(self, arr):
        """Concatenate variable-length float32 arrays of all ranks (rank order)."""
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        if self.world == 1:
            return arr
        n = int(self.allreduce_max(arr.size))
        pad = np.zeros(n + 1, np.float64); pad[0] = arr.size; pad[1:1 + arr.size] = arr
        g = self.allgather_fixed(pad, n + 1)
        return np.concatenate([row[1:1 + int(row[0])] for row in g]).astype(np.float32)

    def allreduce_max(self, v):
        if self.world == 1:
            return v
        import torch
        t = torch.tensor([float(v)], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


class ShardedBA:
    """FullSystem::optimize's GN loop over a window whose points are sharded across ranks.  `ba` is this rank's
    BundleAdjusterHip holding its shard (all frames, its points); `coll` a Collective.
    Exchanges per Gauss-Newton iteration: ONE all-reduce of the packed system and ONE fixed-width all-gather per linearisation
    ([local energy | count | newest-frame residual energies], width fixed at begin())."""

    def __init__(self, ba, coll):
        self.ba, self.coll = ba, coll
        self.lam = 1e-5
        self.lastE = None
        self.width = None

    def _linearize(self, fix=False):
        e_local, nf = self.ba.linearize_local(fix)
        if self.width is None:      # residuals that target the newest keyframe: a property of the graph, agreed on once
            self.width = 2 + int(self.coll.allreduce_max(self.ba.count_newest_residuals()))
        buf = np.zeros(self.width, np.float64)          # float64 carries the local energy exactly; the float32 energies fit losslessly
        buf[0] = e_local; buf[1] = len(nf); buf[2:2 + len(nf)] = nf
        g = self.coll.allgather_fixed(buf, self.width)
        e = float(np.sum(g[:, 0]))                       # rank order: deterministic
        allnf = np.concatenate([row[2:2 + int(row[1])] for row in g]).astype(np.float32)
        self.ba.set_new_frame_energy_th(new_frame_energy_th(allnf))
        return e

    def begin(self):
        self.ba.activate_all()
        e = self._linearize(False)
        el, em = self.ba.energy_terms()
        self.ba.apply_res()
        self.lastE = [e, el, em]
        self.lam = 1e-5
        return self.lastE

    def iteration(self, it):
        ba = self.ba
        ba.backup()
        a = ba.accumulate()
        buf = self.coll.allreduce_sum(pack_system(a["HA"], a["bA"], a["Hsc"], a["bsc"], 0.0, a["resInA"]))
        HA, bA, Hsc, bsc, _, self.res_in_a = unpack_system(buf, ba.n)
        ba.solve_system(it, self.lam, HA, bA, Hsc, bsc)
        ba.step(1.0)
        e = self._linearize(False)
        el, em = ba.energy_terms()
        accept = e + el + em < sum(self.lastE)
        if accept:
            ba.apply_res()
            self.lastE = [e, el, em]
            self.lam = max(self.lam * 0.25, 1e-5)
        else:
            ba.restore()
            e = self._linearize(False)
            el, em = ba.energy_terms()
            self.lastE = [e, el, em]
            self.lam *= 1e2
        return accept
