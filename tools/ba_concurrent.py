#!/usr/bin/env python
"""Several windows in flight on one GPU (bench.py's ba.concurrent_windows leg on its own): K host threads x M fresh windows each.
Usage: python tools/ba_concurrent.py [--m 12] [--points 2000]   (GPU_MAX_HW_QUEUES=<n> in the environment changes the number of hardware queues the streams map to)"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import __graft_entry__ as graft  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=12)
    ap.add_argument("--points", type=int, default=2000)
    ap.add_argument("--size", type=int, default=512)
    a = ap.parse_args()
    import torch
    pkg = graft.load_package()
    import dmvio_amd.synth as synth
    dev = torch.device("cuda", 0)
    case = synth.ba_case(a.size, a.size, n_frames=8, n_points=a.points, seed=synth.SEED)
    F = case["n_frames"]
    ctx = pkg.Context(a.size, a.size, n_slots=F, device=0)
    for k in range(F):
        ctx.frame_upload(k, case["imgs"][k])
    Rn = len(case["res_point"])
    out = bench.bench_ba_concurrent(a, pkg, ctx, case, F, Rn * 1056, Rn * 464, torch, dev, M=a.m)
    print(json.dumps(dict(hw_queues=os.environ.get("GPU_MAX_HW_QUEUES", "default"), m=a.m, value=out.get("value"), at=out.get("at_threads"),
                          sweep=[(r["threads"], r["value"]) for r in out.get("sweep", [])], error=out.get("error"))))


if __name__ == "__main__":
    main()
