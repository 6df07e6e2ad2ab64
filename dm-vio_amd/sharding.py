"""Multi-GPU sharding of the sliding-window BA.

Units = points (with all their residuals): GPU g owns the points hosted in "its" keyframes, every GPU keeps all F
level-0 images and the precalc/adjoint tables.  Per-point quantities (Hdd, bd, Hcd, HdiF, all JpJdF pairs) never leave
a GPU; the ONLY exchange of a Gauss-Newton iteration is the sum of the packed dense system

    [ H_A (n*n) | b_A (n) | H_sc (n*n) | b_sc (n) | energy | resInA ]      n = 4 + 8F  (75 KB for F = 8)

— one small all-reduce (RCCL over xGMI on GPUs, gloo in the CPU tests) — plus an all-gather of the newest-frame residual
energies for setNewFrameEnergyTH (FullSystemOptimize.cpp:96-149).  Every rank then solves the identical reduced system.

This module is harness-level plumbing (numpy + torch.distributed); the arithmetic lives behind the C ABI.
"""
import numpy as np


def partition_points_by_host(host, world, max_imbalance=1.25):
    """Point indices owned by every rank: whole keyframes per rank (largest first, greedy); falls back to equal contiguous
    point ranges when the keyframe split is more unbalanced than max_imbalance (newest keyframe hosts no points, old ones few)."""
    host = np.asarray(host)
    n = len(host)
    if world <= 1:
        return [np.arange(n)]
    counts = np.bincount(host)
    order = np.argsort(-counts, kind="stable")
    load = np.zeros(world, dtype=np.int64)
    owner = np.zeros(len(counts), dtype=np.int64)
    for kf in order:
        r = int(np.argmin(load))
        owner[kf] = r
        load[r] += counts[kf]
    if load.max() > max_imbalance * max(n / world, 1.0):
        bounds = [(n * r) // world for r in range(world + 1)]
        return [np.arange(bounds[r], bounds[r + 1]) for r in range(world)]
    return [np.nonzero(owner[host] == r)[0] for r in range(world)]


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
    """Thin wrapper over torch.distributed (nccl = RCCL on ROCm, or gloo on CPU); world == 1 degenerates to identity."""

    def __init__(self, dist=None, device=None):
        self.dist = dist
        self.device = device
        self.world = dist.get_world_size() if dist is not None else 1

    def allreduce_sum(self, arr):
        if self.world == 1:
            return np.asarray(arr, dtype=np.float64)
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64))
        if self.device is not None:
            t = t.to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def allgather_var(self, arr):
        """Concatenate variable-length float32 arrays of all ranks (rank order)."""
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        if self.world == 1:
            return arr
        import torch
        n = torch.tensor([arr.size], dtype=torch.int64, device=self.device)
        sizes = [torch.zeros_like(n) for _ in range(self.world)]
        self.dist.all_gather(sizes, n)
        sizes = [int(s.item()) for s in sizes]
        m = max(max(sizes), 1)
        pad = torch.zeros(m, dtype=torch.float32, device=self.device)
        pad[:arr.size] = torch.from_numpy(arr).to(pad.device)
        outs = [torch.zeros_like(pad) for _ in range(self.world)]
        self.dist.all_gather(outs, pad)
        return np.concatenate([o[:s].cpu().numpy() for o, s in zip(outs, sizes)])


class ShardedBA:
    """FullSystem::optimize's GN loop over a window whose points are sharded across ranks.  `ba` is this rank's
    BundleAdjusterHip holding its shard (all frames, its points); `coll` a Collective."""

    def __init__(self, ba, coll):
        self.ba, self.coll = ba, coll
        self.lam = 1e-5
        self.lastE = None

    def _linearize(self, fix=False):
        e_local, nf = self.ba.linearize_local(fix)
        e = float(self.coll.allreduce_sum(np.array([e_local]))[0])
        self.ba.set_new_frame_energy_th(new_frame_energy_th(self.coll.allgather_var(nf)))
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
