"""Worker of tests/test_sharded_ba_gpu.py: two processes (gloo rendezvous) share ONE GPU; each holds its share of a window's points in its own
BundleAdjusterHip and calls the library's optimize — the exchanges of the sharded iteration run behind the C ABI
(dmvio_hip_ba_set_comm_callbacks: RCCL refuses two ranks on one device, so the transport here is gloo; the RCCL transport of the same code path is
covered with world size 1 in test_sharded_ba_gpu.py and runs at world size N in bench.py)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def run_window(P, case, slots_case, comm_setup=None, its=6, lin_mask=None, lin_rows=None):
    F = case["n_frames"]
    ctx = P.Context(case["w"], case["h"], n_slots=F)
    for k in range(F):
        ctx.frame_upload(k, slots_case["imgs"][k])
    ba = P.BundleAdjusterHip(ctx, keep_jacobians=lin_mask is not None)
    ba.set_case(case, list(range(F)))
    if comm_setup:
        comm_setup(ba)
    if lin_mask is not None:
        # EFResidual::fixLinearizationF after two iterations, on the states they left (a collective: every rank passes the flags of its own residuals) — the remaining
        # iterations run over a graph whose L system, linearised energy and A / Schur views are each the sum of the ranks' parts
        first = ba.optimize(2)
        if lin_rows is None:
            n_lin = ba.fix_linearization(lin_mask)
        else:
            # the same residuals handed over with their frozen Jacobians / res_toZeroF (dmvio_hip_ba_set_linearized_residuals: collective on a sharded window as well)
            n_lin = ba.set_linearized_residuals(lin_rows[0], lin_rows[1], lin_rows[2])
            assert n_lin == int(lin_rows[0].sum())
        assert 0 < n_lin <= int(lin_mask.sum())
        out_rows = ba.linearized_residuals()
        out = ba.optimize(its)
        out["trace"] = np.concatenate([first["trace"], out["trace"]])
        out["iterations"] += first["iterations"]
        out["EL"] = ba.energy_terms()[0]
        out["n_lin"] = n_lin
        out["rows"] = out_rows
    else:
        out = ba.optimize(its)
    poses = np.stack([ba.frame_pose(k)[0] for k in range(F)])
    aff = np.stack([ba.frame_pose(k)[1] for k in range(F)])
    idepth = ba.point_state()[0]
    ba.close()
    return out, poses, aff, idepth


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    P = graft.load_package()
    import dmvio_amd.synth as synth
    import dmvio_amd.sharding as sh
    case = synth.ba_case(512, 512, n_frames=8, n_points=1500, seed=21)
    modes = [("by-keyframe", 10.0), ("equal-ranges", 1.0)]
    if os.environ.get("SHARD_MODES"):
        modes = [dict(modes)[m] and (m, dict(modes)[m]) for m in os.environ["SHARD_MODES"].split(",")]
    lin = bool(os.environ.get("SHARD_LIN"))
    full_mask = (np.arange(len(case["res_point"])) % 3 == 0).astype(np.uint8) if lin else None
    for mode, imbalance in modes:
        parts = sh.partition_points_by_host(case["host"], world, max_imbalance=imbalance)
        mine = sh.shard_case(case, parts[rank])
        my_mask = None
        if lin:
            member = np.zeros(len(case["u"]), dtype=bool); member[np.asarray(parts[rank])] = True
            my_mask = np.ascontiguousarray(full_mask[member[case["res_point"]]])      # shard_case keeps a point's residuals in their order
            assert len(my_mask) == len(mine["res_point"])
        my_rows = None
        if os.environ.get("SHARD_LIN") == "import":
            # the rows come from an unsharded window at (to rounding) the same state; every rank takes those of its own residuals
            pre, _, _, _ = run_window(P, case, case, its=1, lin_mask=full_mask)
            keep = member[case["res_point"]]
            my_rows = tuple(np.ascontiguousarray(x[keep]) for x in pre["rows"])
            full_rows = pre["rows"]
        out, poses, aff, idepth = run_window(P, mine, case, lambda ba: ba.set_comm_torch(dist), its=4 if lin else 6, lin_mask=my_mask, lin_rows=my_rows)
        # every rank took the same decisions and holds the same frame states, bit for bit
        blob = torch.from_numpy(np.concatenate([out["trace"].ravel(), poses.ravel(), aff.ravel(), [out["rmse"], out["finalEnergy"], out.get("EL", 0.0)]]).copy())
        allb = [torch.zeros_like(blob) for _ in range(world)]
        dist.all_gather(allb, blob)
        for o in allb[1:]:
            assert torch.equal(o.view(torch.int64), allb[0].view(torch.int64)), "ranks disagree"
        if os.environ.get("SHARD_DEBUG") and rank == 0:
            print("sharded trace", mode, "\n", out["trace"], flush=True)
        if rank == 0 and not os.environ.get("SHARD_NOFULL"):
            full, fposes, faff, fid = run_window(P, case, case, its=4 if lin else 6, lin_mask=full_mask, lin_rows=full_rows if os.environ.get("SHARD_LIN") == "import" else None)
            if lin:
                assert abs(out["EL"] - full["EL"]) <= 1e-4 * max(1.0, abs(full["EL"])), (out["EL"], full["EL"])
            if os.environ.get("SHARD_DEBUG"):
                print("sharded trace\n", out["trace"], "\nfull trace\n", full["trace"], flush=True)
            assert out["iterations"] == full["iterations"]
            assert np.array_equal(out["trace"][:, 3], full["trace"][:, 3]), (out["trace"][:, 3], full["trace"][:, 3])     # same accept sequence
            # the split changes the fp32 summation order of the buckets; six GN iterations carry that rounding into the energies at the 1e-5 level
            assert np.allclose(out["trace"][:, 0], full["trace"][:, 0], rtol=2e-4), np.abs(out["trace"][:, 0] / full["trace"][:, 0] - 1).max()
            assert abs(out["rmse"] - full["rmse"]) <= 1e-4 * full["rmse"]
            assert np.abs(poses - fposes).max() < 1e-5 and np.abs(aff - faff).max() < 1e-4
            assert np.abs(idepth - fid[parts[0]]).max() < 1e-4
            print("OK", mode, [len(p) for p in parts], "rmse", out["rmse"], full["rmse"], "pose diff", np.abs(poses - fposes).max())
        dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
