"""Worker of tests/test_sharding_cpu.py: run under torch.distributed.run with the gloo backend (world size 2, CPU only).
Checks the multi-GPU BA protocol of dm-vio_amd/sharding.py with the CPU oracle standing in for the per-rank accumulation."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    graft.load_package()
    import dmvio_amd.synth as synth
    import dmvio_amd.sharding as sh
    O = graft.load_oracle()
    coll = sh.Collective(dist, None)
    case = synth.ba_case(w=320, h=256, n_frames=4, n_points=240, hosts_share=(100, 80, 60, 0), seed=777)
    for mode, imbalance in (("by-keyframe", 10.0), ("equal-ranges", 1.0)):
        parts = sh.partition_points_by_host(case["host"], world, max_imbalance=imbalance)
        allidx = np.sort(np.concatenate(parts))
        assert np.array_equal(allidx, np.arange(len(case["u"]))), "every point owned exactly once"
        mine = sh.shard_case(case, parts[rank])
        W = O.BAWindow(mine)
        W.activate_all()
        e_local = W.linearize_all(False)
        nf_local = W.res_state()["newEnergyWO"][(mine["res_target"] == case["n_frames"] - 1)]
        nf_local = nf_local[nf_local >= 0]
        W.apply_res()
        a = W.accumulate()
        buf = coll.allreduce_sum(sh.pack_system(a["HA"], a["bA"], a["Hsc"], a["bsc"], e_local, a["resInA"]))
        HA, bA, Hsc, bsc, e_sum, res = sh.unpack_system(buf, W.n)
        th = sh.new_frame_energy_th(coll.allgather_var(nf_local))
        if rank == 0:
            Wf = O.BAWindow(case)
            Wf.activate_all()
            e_full = Wf.linearize_all(False)
            Wf.apply_res()
            f = Wf.accumulate()
            sc = np.sqrt(np.outer(np.diag(f["HA"]) + 1e-9, np.diag(f["HA"]) + 1e-9))
            tol = 1e-11 if mode == "by-keyframe" else 2e-6   # whole buckets stay on one rank vs a host split over two ranks
            assert np.max(np.abs(HA - f["HA"]) / sc) < tol, (mode, np.max(np.abs(HA - f["HA"]) / sc))
            # frame blocks of H_sc are per-host buckets too; its calibration block (accHcc / accbc) is ONE accumulator over all points,
            # so splitting the points changes its fp32 summation order
            assert np.max(np.abs(Hsc - f["Hsc"])[4:, 4:] / sc[4:, 4:]) < tol
            assert np.max(np.abs(Hsc - f["Hsc"]) / sc) < 2e-6
            assert np.allclose(bA, f["bA"], rtol=1e-5, atol=1e-6 * np.abs(f["bA"]).max())
            assert np.allclose(bsc, f["bsc"], rtol=1e-5, atol=1e-6 * np.abs(f["bsc"]).max())
            assert res == f["resInA"] and abs(e_sum - e_full) < 1e-9 * e_full
            assert th == Wf.frame_energy_th()[case["n_frames"] - 1], (th, Wf.frame_energy_th())
            print("OK", mode, "rank sizes", [len(p) for p in parts])
        dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
