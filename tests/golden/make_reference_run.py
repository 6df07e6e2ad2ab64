"""Generates tests/golden/reference_run_256x192.npz: the inputs and outputs of every trackNewCoarse / optimize / setCoarseTrackingRef call
the REFERENCE'S OWN FullSystem (oracle/_ref/libref.so, built from /root/reference by oracle/Makefile.ref) made on a synthetic 62-frame
256x192 sequence.  The images are not stored: tests/replay.make_sequence re-renders them from the seed.

    python tests/golden/make_reference_run.py        (in the container that holds /root/reference)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402

W, H, FRAMES, STEP, DENSITY = 256, 192, 62, 1.6, 300


def main():
    graft.load_package()
    import dmvio_amd.synth as synth
    import ref_py as R
    import replay
    run = replay.run_reference(R, synth, W, H, FRAMES, step=STEP, point_density=DENSITY)
    assert run["status"][-1]["initialized"] and not run["status"][-1]["isLost"]
    flat = {"meta": np.array([W, H, FRAMES, DENSITY], np.int32), "step": np.array([STEP]), "n_events": np.array([len(run["events"])], np.int32)}
    for k, e in enumerate(run["events"]):
        for name, v in e.items():
            if name == "kind":
                flat["e%d__kind" % k] = np.array([["setref", "track_in", "track_out", "opt_in", "opt_out"].index(v)], np.int32)
            elif name == "frames":
                for fk, f in enumerate(v):
                    for fn_, fv in f.items():
                        flat["e%d__frames__%d__%s" % (k, fk, fn_)] = np.asarray(fv)
            else:
                flat["e%d__%s" % (k, name)] = np.asarray(v)
    tr = run["trajectory"]
    for name, v in tr.items():
        flat["traj__" + name] = np.asarray(v)
    out = os.path.join(ROOT, "tests", "golden", "reference_run_256x192.npz")
    np.savez_compressed(out, **flat)
    kinds = [e["kind"] for e in run["events"]]
    print("wrote %s: %d events (%d tracks, %d optimisations), %.0f KiB" % (out, len(kinds), kinds.count("track_out"), kinds.count("opt_out"), os.path.getsize(out) / 1024))


if __name__ == "__main__":
    main()
