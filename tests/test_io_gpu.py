"""GPU parity of the raw-image upload path: photometric + geometric undistortion on the device is bit-identical to the oracle and the
pyramids built from it equal those of the float upload path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bits", [8, 16])
def test_upload_raw_bit_exact(pkg, oracle, gpu_required, bits):
    from test_io_cpu import io_case
    c = io_case(seed=5 + bits, wOrg=640, hOrg=480, w=512, h=384, bits=bits)
    ctx = pkg.Context(c["w"], c["h"], n_slots=2)
    und = pkg.UndistorterHip(ctx, c["wOrg"], c["hOrg"], bits, c["G"], c["vig"], c["rx"], c["ry"])
    img = und.upload(0, c["raw"])
    ref = oracle.undistort(c["raw"], c["G"], c["vig"], c["rx"], c["ry"], c["w"], c["h"])
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    ctx.frame_upload(1, ref)
    for lvl in range(ctx.levels):
        assert np.array_equal(ctx.frame_download(0, lvl).view(np.uint32), ctx.frame_download(1, lvl).view(np.uint32))
    # without photometric calibration (factor) and without geometric maps (passthrough)
    c2 = io_case(seed=9, wOrg=c["w"], hOrg=c["h"], bits=bits)
    und2 = pkg.UndistorterHip(ctx, c["w"], c["h"], bits)
    img2 = und2.upload(0, c2["raw"], factor=0.25)
    assert np.array_equal(img2, oracle.undistort(c2["raw"], None, None, None, None, c["w"], c["h"], factor=0.25))
    with pytest.raises(pkg.HipLibraryError):
        pkg.UndistorterHip(ctx, c["wOrg"], c["hOrg"], bits)      # passthrough with a different raw size
    # a remap entry whose bilinear taps would leave the raw image is refused when the table is handed over (the kernel does not bounds-check per tap)
    for bad_x, bad_y in ((c["wOrg"] - 1.0, 3.0), (5.0, c["hOrg"] - 1.0), (5.0, -0.5), (float("nan"), 2.0)):
        rx, ry = c["rx"].copy(), c["ry"].copy()
        rx.flat[17] = bad_x; ry.flat[17] = bad_y
        with pytest.raises(pkg.HipLibraryError):
            pkg.UndistorterHip(ctx, c["wOrg"], c["hOrg"], bits, c["G"], c["vig"], rx, ry)
