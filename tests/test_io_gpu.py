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


@pytest.mark.parametrize("bits,size", [(8, (640, 480, 512, 384)), (16, (640, 480, 512, 384)), (8, (300, 210, 250, 170))])
def test_raw_device_batch_equals_single_uploads(pkg, oracle, gpu_required, bits, size):
    """The fused batched build (raw images resident on the device -> undistortion + every pyramid level in one launch) gives the bits of the
    per-frame upload path for every frame and level, with geometric maps and in passthrough; sizes that are no multiple of the 128x32 tile too."""
    import torch
    from test_io_cpu import io_case
    wOrg, hOrg, w, h = size
    B = 5
    cases = [io_case(seed=40 + 3 * i + bits, wOrg=wOrg, hOrg=hOrg, w=w, h=h, bits=bits) for i in range(B)]
    c = cases[0]
    ctx = pkg.Context(w, h, n_slots=2 * B)
    und = pkg.UndistorterHip(ctx, wOrg, hOrg, bits, c["G"], c["vig"], c["rx"], c["ry"])
    raws = np.stack([k["raw"].reshape(hOrg, wOrg) for k in cases]).astype(und.dtype)
    stride = raws[0].nbytes + 64                          # frames need not be packed
    host = np.zeros((B, stride), np.uint8)
    host[:, :raws[0].nbytes] = raws.reshape(B, -1).view(np.uint8)
    dev = torch.from_numpy(host).to("cuda:0")
    torch.cuda.synchronize()
    und.from_raw_device_batch(list(range(B, 2 * B)), dev.data_ptr(), stride)
    ctx.synchronize()
    for i in range(B):
        ref = oracle.undistort(cases[i]["raw"], c["G"], c["vig"], c["rx"], c["ry"], w, h)
        und.upload(i, cases[i]["raw"], want_image=False)
        for lvl in range(ctx.levels):
            a, b = ctx.frame_download(B + i, lvl), ctx.frame_download(i, lvl)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (i, lvl)
        assert np.array_equal(ctx.frame_download(B + i, 0)[..., 0].reshape(h, w).view(np.uint32), ref.reshape(h, w).view(np.uint32))
    # passthrough with a factor
    ctx2 = pkg.Context(wOrg, hOrg, n_slots=2 * B)
    und2 = pkg.UndistorterHip(ctx2, wOrg, hOrg, bits)
    und2.from_raw_device_batch(list(range(B, 2 * B)), dev.data_ptr(), stride, factor=0.5)
    ctx2.synchronize()
    for i in range(B):
        und2.upload(i, cases[i]["raw"], factor=0.5, want_image=False)
        for lvl in range(ctx2.levels):
            assert np.array_equal(ctx2.frame_download(B + i, lvl).view(np.uint32), ctx2.frame_download(i, lvl).view(np.uint32))
    with pytest.raises(pkg.HipLibraryError):
        und.from_raw_device_batch([0, 2 * B], dev.data_ptr(), stride)            # slot out of range
    with pytest.raises(pkg.HipLibraryError):
        und.from_raw_device_batch([0, 1], dev.data_ptr(), raws[0].nbytes - 2)    # frames would overlap
