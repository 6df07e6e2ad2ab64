"""Worker of tests/test_multigpu.py: one process per GPU (torchrun), the library's own RCCL communicator (unique id broadcast over the launcher's process group).
(1) one BA window, points sharded by host keyframe: optimize(6) through dmvio_hip_ba_set_comm — every rank identical, the unsharded window within summation-order rounding;
(2) the tries of trackNewCoarse split over the ranks through dmvio_hip_tracker_set_comm — every rank identical, the unsplit answer within cluster-size rounding."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402
from dist_worker_gpu import run_window  # noqa: E402


def same_on_all_ranks(blob, world):
    t = torch.from_numpy(np.ascontiguousarray(blob, dtype=np.float64).copy()).cuda()
    allb = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(allb, t)
    for o in allb[1:]:
        assert torch.equal(o.view(torch.int64), allb[0].view(torch.int64)), "ranks disagree"


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    P = graft.load_package()
    import dmvio_amd.synth as synth
    import dmvio_amd.sharding as sh
    O = graft.load_oracle()
    probe = P.Context(64, 64, n_slots=1, device=local)
    uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
    if rank == 0:
        uid.copy_(torch.frombuffer(bytearray(P.RcclCommunicator.unique_id(probe.L)), dtype=torch.uint8))
    dist.broadcast(uid, 0); torch.cuda.synchronize()
    comm = P.RcclCommunicator(probe, bytes(uid.cpu().numpy().tobytes()), rank, world)
    assert comm.info() == (world, rank)
    # ---- (1) sharded BA
    case = synth.ba_case(512, 512, n_frames=8, n_points=1500, seed=21)
    parts = sh.partition_points_by_host(case["host"], world, max_imbalance=1.0)
    mine = sh.shard_case(case, parts[rank])
    F = case["n_frames"]
    ctx = P.Context(case["w"], case["h"], n_slots=F, device=local)
    for k in range(F):
        ctx.frame_upload(k, case["imgs"][k])
    ba = P.BundleAdjusterHip(ctx); ba.set_case(mine, list(range(F))); ba.set_comm(comm, rank, world)
    out = ba.optimize(6)
    poses = np.stack([ba.frame_pose(k)[0] for k in range(F)])
    same_on_all_ranks(np.concatenate([out["trace"].ravel(), poses.ravel(), [out["rmse"], out["finalEnergy"]]]), world)
    ba.close()
    if rank == 0:
        full = P.BundleAdjusterHip(ctx); full.set_case(case, list(range(F))); fo = full.optimize(6)
        fposes = np.stack([full.frame_pose(k)[0] for k in range(F)])
        assert np.array_equal(out["trace"][:, 3], fo["trace"][:, 3]) and abs(out["rmse"] - fo["rmse"]) <= 1e-4 * fo["rmse"] and np.abs(poses - fposes).max() < 1e-5
        print("OK ba world %d rmse %.6f (unsharded %.6f)" % (world, out["rmse"], fo["rmse"]), flush=True)
    # ---- (2) hypothesis-parallel trackNewCoarse
    w = h = 256
    tc = synth.tracking_case(w, h, n_ref=600, n_frames=1)
    f = tc["frames"][0]
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    bad = O.se3_exp(np.array([0.0, 0, 0, 0.0, 0.35, 0.0]))
    tries = np.concatenate([bad[None], O.se3_mul(bad, bad)[None], P.make_track_hypotheses(O.se3_exp(-0.5 * f["xi"]), ident, ident)])
    tctx = P.Context(w, h, n_slots=2, device=local)
    trk = P.CoarseTrackerHip(tctx); trk.makeK(tc["K4"])
    tctx.frame_upload(0, tc["ref_img"]); tctx.frame_upload(1, f["img"])
    trk.setCoarseTrackingRef(0, tc["u"], tc["v"], tc["idepth"], tc["hdiF"])
    single = trk.trackNewCoarse(1, tries, lastCoarseRMSE=np.full(5, 100.0), reTrackThreshold=0.2)
    trk.set_comm(comm, rank, world)
    split = trk.trackNewCoarse(1, tries, lastCoarseRMSE=np.full(5, 100.0), reTrackThreshold=0.2)
    same_on_all_ranks(np.concatenate([split["pose7"], split["aff"], np.nan_to_num(split["achievedRes"]), [split["winner"], split["tries_used"]]]), world)
    assert split["winner"] == single["winner"] and split["tries_used"] == single["tries_used"] and np.max(np.abs(split["pose7"] - single["pose7"])) < 1e-5
    if rank == 0:
        print("OK track world %d winner %d after %d tries" % (world, split["winner"], split["tries_used"]), flush=True)
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
