"""The device code's memory instructions by address space (tools/isa_check.py): no kernel reaches global memory through a generic pointer.  hipcc cross-compiles gfx950 here;
no GPU needed."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# what is knowingly left: the single-window linearisation picks its pair table with a select between the kernel-argument table and the device table (five loads of 14 floats,
# a latency-bound 28 us kernel); k_ref_write stores one value through a pointer it reads from a device table
FLAT_ALLOWED = {"k_ba_linearize": 5, "k_ref_write": 1}
# register spills: the LM control step (outside the evaluation loops), the eight-lane batched linearisation, and one 64-bit value of the batched
# accumulation at its 80-register budget (six waves per SIMD: measured faster than five without the spill)
SCRATCH_ALLOWED = {"k_track_lm": 200, "k_ba_linearize_b": 8, "k_ba_accumulate_b": 8}


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not installed")
def test_no_generic_pointer_reaches_global_memory():
    import isa_check
    r = isa_check.run()
    assert len(r) > 40 and any(k.startswith("capi:k_track_lm") for k in r) and any("k_ba_accumulate_b" in k for k in r)
    for k, c in r.items():
        name = k.split(":")[1]
        flat = c["flat_load"] + c["flat_store"] + c["flat_atomic"]
        allowed = max([v for kk, v in FLAT_ALLOWED.items() if name.startswith(kk) and not name.startswith(kk + "_b")] or [0])
        assert flat <= allowed, "%s: %d flat accesses (a pointer read from memory? take it through gl() / dmvUniformGlobal())" % (k, flat)
        scratch = c["scratch_load"] + c["scratch_store"]
        budget = max([v for kk, v in SCRATCH_ALLOWED.items() if name.startswith(kk)] or [0])
        assert scratch <= budget, "%s: %d scratch accesses (a dynamically indexed local structure? see baRec())" % (k, scratch)
    # the hot kernels by name: global accesses only
    for hot in ("k_ba_accumulate_b", "k_ba_linearize_b1", "k_ba_solve", "k_ba_stitch_b", "k_ba_stitch_gather_b", "k_ba_post_decide_b", "k_ba_resubstitute_b"):
        ks = [k for k in r if k.split(":")[1].startswith(hot)]
        assert ks, hot
        for k in ks:
            assert r[k]["flat_load"] + r[k]["flat_store"] == 0, (k, r[k])
            if hot != "k_ba_accumulate_b":
                assert r[k]["scratch_load"] + r[k]["scratch_store"] == 0, (k, r[k])
    for k in [k for k in r if k.split(":")[1].startswith("k_track_lm")]:
        assert r[k]["flat_load"] + r[k]["flat_store"] == 0, (k, r[k])
