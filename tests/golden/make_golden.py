#!/usr/bin/env python
"""Generates tests/golden/tracker_small.npz: a small tracking case (inputs) + float64 expectations computed by the
INDEPENDENT NumPy restatement tests/np_ref.py (not by the oracle, not by the HIP path).

The reference has no golden vectors for this path and cannot run here; these fixtures pin the oracle
against an independent implementation of the same published formulas.  Only the template (pc_*) comes from the oracle
itself — np_ref takes it as input — and pc_n is recorded to detect drifts of makeCoarseDepthL0.

usage (from the repo root):  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402
import np_ref  # noqa: E402

graft.load_package()
import dmvio_amd.synth as synth  # noqa: E402

O = graft.load_oracle()
w, h = 320, 256
case = synth.tracking_case(w, h, n_ref=400, seed=4242, xi_true=(0.012, -0.008, 0.015, 0.004, -0.005, 0.003))
T = O.Tracker(w, h); T.make_k(case["K4"])
dIr, _ = O.make_images(case["ref_img"], w, h); dIn, _ = O.make_images(case["frames"][0]["img"], w, h)
T.set_ref(dIr, case["u"], case["v"], case["idepth"], case["hdiF"]); T.set_new(dIn)
pose = case["frames"][0]["pose7"]; aff = np.array([0.01, -1.0])
E, n, H, b = [], [], [], []
for lvl in range(T.levels):
    r = np_ref.calc_res_gs(case["K4"], lvl, T.get_pc(lvl), np_ref.make_images(case["frames"][0]["img"])[lvl], pose, aff, 20.0)
    E.append(r["E"]); n.append(r["n"]); H.append(r["H"]); b.append(r["b"])
np.savez_compressed(os.path.join(HERE, "tracker_small.npz"), w=w, h=h, K4=case["K4"], ref_img=case["ref_img"], new_img=case["frames"][0]["img"],
                    u=case["u"], v=case["v"], idepth=case["idepth"], hdiF=case["hdiF"], pose7=pose, aff=aff,
                    pc_n=np.array([T.pc_n(l) for l in range(T.levels)]), E=np.array(E), n=np.array(n), H=np.array(H), b=np.array(b))
print("wrote tracker_small.npz", [T.pc_n(l) for l in range(T.levels)], E)
