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


@pytest.mark.parametrize("bits,size,calib", [(8, (512, 512), False), (8, (640, 480), True), (16, (256, 192), True), (8, (128, 64), False)])
def test_raw_batch_register_build_equals_the_lds_tile_build(pkg, gpu_required, bits, size, calib):
    """dmvio_hip_frames_from_raw_device_batch has two kernels: the wave-autonomous build (a 4 x 8 pixel block per thread, every level formed in registers, no barrier; the
    default for pyramids of <= 4 levels on images whose sides are multiples of 8) and the general LDS-tile build.  Every level of every frame is the same, bit for bit, in both
    level-0 layouts (8x4 tiles / row-major), with and without photometric tables, for frames whose base is not 4-pixel aligned (the unaligned load path), and at a size whose
    pyramid has fewer than four levels."""
    import torch
    from test_io_cpu import io_case
    w, h = size
    B = 7
    cases = [io_case(seed=70 + i + bits, wOrg=w, hOrg=h, bits=bits) for i in range(B)]
    ctx = pkg.Context(w, h, n_slots=4 * B)
    und = pkg.UndistorterHip(ctx, w, h, bits, cases[0]["G"] if calib else None, cases[0]["vig"] if calib else None)
    raws = np.stack([k["raw"].reshape(h, w) for k in cases]).astype(und.dtype)
    for pad in (64, 64 + raws.itemsize):                    # frame bases aligned to 4 pixels / not aligned
        stride = raws[0].nbytes + pad
        host = np.zeros((B, stride), np.uint8)
        host[:, :raws[0].nbytes] = raws.reshape(B, -1).view(np.uint8)
        dev = torch.from_numpy(host).to("cuda:0"); torch.cuda.synchronize()
        k = 0
        for variant in (0, 1):
            for tiled in (False, True):
                pkg.set_raw_batch_kernel(ctx, variant); pkg.set_raw_batch_layout(ctx, tiled)
                und.from_raw_device_batch(list(range(k * B, (k + 1) * B)), dev.data_ptr(), stride, factor=0.5)
                k += 1
        ctx.synchronize()
        if w % 8 == 0 and h % 4 == 0:
            assert pkg.frame_level0_is_tiled(ctx, B) and pkg.frame_level0_is_tiled(ctx, 3 * B) and not pkg.frame_level0_is_tiled(ctx, 2 * B)
        for i in range(B):
            for lvl in range(ctx.levels):
                ref = ctx.frame_download(i, lvl).view(np.uint32)
                for kk in (1, 2, 3):
                    assert np.array_equal(ctx.frame_download(kk * B + i, lvl).view(np.uint32), ref), (pad, i, lvl, kk)
    pkg.set_raw_batch_kernel(ctx, 1); pkg.set_raw_batch_layout(ctx, False)


@pytest.mark.parametrize("size", [(512, 512), (640, 480), (128, 64)])
def test_fp32_register_build_equals_the_lds_tile_build(pkg, gpu_required, size):
    """The same two kernels behind the fp32 entry points: single upload, batch copied from device memory, batch attached in place (level 0 = the caller's image) — with frame
    bases that are 16-byte aligned and that are not.  Every level the same bits; a NaN pixel withdraws the slot's clean stamp in both."""
    import torch
    w, h = size
    B = 5
    rng = np.random.RandomState(3)
    imgs = (rng.rand(B, h * w) * 255).astype(np.float32)
    ctx = pkg.Context(w, h, n_slots=6 * B + 2)
    for pad in (0, 1):                                       # floats between frames: bases aligned to 16 bytes / to 4
        host = np.zeros((B, h * w + 4 + pad), np.float32); host[:, :h * w] = imgs
        dev = torch.from_numpy(host).to("cuda:0"); torch.cuda.synchronize()
        k = 0
        for variant in (0, 1):
            pkg.set_raw_batch_kernel(ctx, variant)
            ctx.frames_from_device_batch(list(range(k * B, (k + 1) * B)), dev.data_ptr(), host.shape[1] * 4); k += 1
            ctx.frames_attach_device_batch(list(range(k * B, (k + 1) * B)), dev.data_ptr(), host.shape[1] * 4); k += 1
            for i in range(B):
                ctx.frame_upload(k * B + i, imgs[i])
            k += 1
        ctx.synchronize()
        for i in range(B):
            for lvl in range(ctx.levels):
                ref = ctx.frame_download(i, lvl).view(np.uint32)
                for kk in range(1, 6):
                    assert np.array_equal(ctx.frame_download(kk * B + i, lvl).view(np.uint32), ref), (pad, i, lvl, kk)
    bad = imgs[0].copy(); bad[h * w // 2 + 3] = np.nan
    for variant in (0, 1):
        pkg.set_raw_batch_kernel(ctx, variant)
        ctx.frame_upload(6 * B + variant, bad)
    ctx.synchronize()
    assert not ctx.frame_is_clean(6 * B) and not ctx.frame_is_clean(6 * B + 1) and ctx.frame_is_clean(0)
    pkg.set_raw_batch_kernel(ctx, 1)


def test_batched_builds_on_a_stream_of_their_own(pkg, synth, gpu_required):
    """dmvio_hip_set_build_stream: the batched pyramid builds enqueued on a second stream, ordered against the tracking stream by the caller's events (a build behind the
    consumers of the slots it rewrites, a consumer behind the build of its slots) — two slot sets used alternately over several rounds with changing images, every round's
    tracking results equal to those of the same images built on the context's stream, bit for bit; the slot lists stay cached on the device (no re-upload per round)."""
    import torch
    w, h = 256, 192
    case = synth.tracking_case(w, h, n_ref=700, n_frames=4, xi_jitter=0.3)
    B = 48
    ctx = pkg.Context(w, h, n_slots=3 * B + 1)
    stream = torch.cuda.Stream(); ctx.set_stream(stream.cuda_stream)
    ctx.frame_upload(0, case["ref_img"])
    trk = pkg.CoarseTrackerHip(ctx); trk.makeK(case["K4"])
    trk.setCoarseTrackingRef(0, case["u"], case["v"], case["idepth"], case["hdiF"])
    base = torch.from_numpy(np.stack([case["frames"][i % 4]["img"].reshape(-1) for i in range(B)]).astype(np.float32)).to("cuda:0")
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    sets = [list(range(1, B + 1)), list(range(B + 1, 2 * B + 1))]
    ref_slots = list(range(2 * B + 1, 3 * B + 1))
    bstream = torch.cuda.Stream()
    built = [torch.cuda.Event(), torch.cuda.Event()]; tracked = [torch.cuda.Event(), torch.cuda.Event()]
    for rnd in range(5):
        imgs = torch.roll(base, shifts=rnd, dims=0).contiguous()              # another assignment of images to slots every round
        torch.cuda.synchronize()
        ctx.set_build_stream(0)
        ctx.frames_attach_device_batch(ref_slots, imgs.data_ptr(), w * h * 4)     # the reference: built on the context's stream
        want = trk.track_batch(ref_slots, [ident] * B, [(0.0, 0.0)] * B)
        ctx.set_build_stream(bstream.cuda_stream)
        cur = rnd & 1
        bstream.wait_event(tracked[cur])                                       # the last consumer of this slot set (two rounds ago)
        ctx.frames_attach_device_batch(sets[cur], imgs.data_ptr(), w * h * 4)
        built[cur].record(bstream)
        stream.wait_event(built[cur])
        got = trk.track_batch(sets[cur], [ident] * B, [(0.0, 0.0)] * B)
        tracked[cur].record(stream)
        assert got["good"].all()
        for k in ("pose7", "aff", "lastResiduals", "H", "b", "iterations"):
            assert np.array_equal(got[k], want[k], equal_nan=True), (rnd, k)
    ctx.set_build_stream(0)
    ctx.synchronize()


@pytest.mark.parametrize("B,launch", [(6, None), (200, (1, 512)), (600, (1, 256))])
def test_tiled_level0_of_the_raw_batch_build_tracks_bit_for_bit(pkg, synth, gpu_required, B, launch):
    """dmvio_hip_frames_from_raw_device_batch can store level 0 in 8x4-pixel tiles (dmvio_hip_set_raw_batch_layout; it writes level 0 anyway): the coarse tracker's batch kernel gathers the same twelve values per
    tap from other addresses, so every result — pose, affine, residuals, flow indicators, H, b, iteration count — equals the row-major layout's BIT FOR BIT, in cluster mode
    (B = 6), with 512-thread (B = 200) and 256-thread workgroups (B = 600); every other consumer converts such a slot back on first use (download, reference template,
    single-frame tracking, a window of the optimiser)."""
    import torch
    w, h = 256, 192
    case = synth.tracking_case(w, h, n_ref=700, n_frames=4, xi_jitter=0.3)
    ctx = pkg.Context(w, h, n_slots=2 * B + 1)
    ctx.frame_upload(0, case["ref_img"])
    trk = pkg.CoarseTrackerHip(ctx); trk.makeK(case["K4"])
    trk.setCoarseTrackingRef(0, case["u"], case["v"], case["idepth"], case["hdiF"])
    und = pkg.UndistorterHip(ctx, w, h, 8)                                   # passthrough geometry, no photometric calibration: image = factor * raw
    raws = np.stack([np.clip(np.rint(case["frames"][i % 4]["img"]), 0, 255).astype(np.uint8) for i in range(B)])
    dev = torch.from_numpy(raws.reshape(B, -1)).to("cuda:0"); torch.cuda.synchronize()
    tiled_slots, plain_slots = list(range(1, B + 1)), list(range(B + 1, 2 * B + 1))
    und.from_raw_device_batch(plain_slots, dev.data_ptr(), w * h)                              # row-major is the default
    pkg.set_raw_batch_layout(ctx, True)
    und.from_raw_device_batch(tiled_slots, dev.data_ptr(), w * h)
    pkg.set_raw_batch_layout(ctx, False)
    ctx.synchronize()
    assert all(pkg.frame_level0_is_tiled(ctx, s) for s in tiled_slots) and not any(pkg.frame_level0_is_tiled(ctx, s) for s in plain_slots)
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    rt = trk.track_batch(tiled_slots, [ident] * B, [(0.0, 0.0)] * B)
    if launch:
        assert trk.last_launch() == launch
    assert all(pkg.frame_level0_is_tiled(ctx, s) for s in tiled_slots)                         # the batch kernel read the tiles in place
    rp = trk.track_batch(plain_slots, [ident] * B, [(0.0, 0.0)] * B)
    assert rt["good"].all() and rp["good"].all()
    for k in ("pose7", "aff", "lastResiduals", "flow", "H", "b", "iterations"):
        assert np.array_equal(rt[k], rp[k], equal_nan=True), k
    # mixed batch: tiled and row-major slots side by side in one launch
    mixed = [tiled_slots[i] if i % 2 else plain_slots[i] for i in range(B)]
    rm = trk.track_batch(mixed, [ident] * B, [(0.0, 0.0)] * B)
    assert all(np.array_equal(rm[k], rp[k], equal_nan=True) for k in ("pose7", "lastResiduals", "H", "b"))
    # every other consumer converts the slot back: download (all levels equal the row-major build's), single-frame tracking, the reference template, a BA window
    for lvl in range(ctx.levels):
        assert np.array_equal(ctx.frame_download(tiled_slots[0], lvl).view(np.uint32), ctx.frame_download(plain_slots[0], lvl).view(np.uint32))
    assert not pkg.frame_level0_is_tiled(ctx, tiled_slots[0]) and pkg.frame_level0_is_tiled(ctx, tiled_slots[1])
    a = trk.trackNewestCoarse(tiled_slots[1], ident, [0.0, 0.0]); b = trk.trackNewestCoarse(plain_slots[1], ident, [0.0, 0.0])
    assert not pkg.frame_level0_is_tiled(ctx, tiled_slots[1]) and np.array_equal(np.asarray(a["pose7"]), np.asarray(b["pose7"]))
    trk2 = pkg.CoarseTrackerHip(ctx); trk2.makeK(case["K4"])
    trk2.setCoarseTrackingRef(tiled_slots[2], case["u"], case["v"], case["idepth"], case["hdiF"])
    trk3 = pkg.CoarseTrackerHip(ctx); trk3.makeK(case["K4"])
    trk3.setCoarseTrackingRef(plain_slots[2], case["u"], case["v"], case["idepth"], case["hdiF"])
    assert not pkg.frame_level0_is_tiled(ctx, tiled_slots[2])
    for lvl in range(ctx.levels):
        assert np.array_equal(np.stack(trk2.get_pc(lvl)).view(np.uint32), np.stack(trk3.get_pc(lvl)).view(np.uint32))
    # a rebuilt slot is row-major again
    ctx.frame_upload(tiled_slots[3], case["ref_img"])
    assert not pkg.frame_level0_is_tiled(ctx, tiled_slots[3])


def test_reference_generated_remap_tables_are_accepted(pkg, oracle, gpu_required, tmp_path):
    """The tables Undistort::getUndistorterForFile builds itself (Undistort.cpp:266-384, 900-942: entries with 0 < x < wOrg-1, 0 < y < hOrg-1, everything else -1) — a `crop`
    rectification whose border pixels map into the last raw row — are taken by the HIP undistorter and give the reference's own image bit for bit.  Entries inside
    (wOrg-2, wOrg-1): their taps (int)x, (int)x+1 <= wOrg-1 are in bounds."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import ref_py as R
    if not R.available():
        pytest.skip("oracle/_ref/libref.so not built and /root/reference absent")
    wOrg, hOrg, w, h = 376, 288, 320, 240
    rng = np.random.default_rng(17)
    cam = tmp_path / "camera.txt"
    cam.write_text("RadTan 0.535 0.669 0.493 0.505 -0.28 0.07 0.0002 -0.0003\n%d %d\ncrop\n%d %d\n" % (wOrg, hOrg, w, h))
    g = np.cumsum(rng.uniform(0.5, 1.5, 256)) ** 1.1
    gam = tmp_path / "pcalib.txt"
    gam.write_text(" ".join("%.9g" % x for x in g) + "\n")
    yy, xx = np.mgrid[0:hOrg, 0:wOrg]
    vig = (65535 * (1 - 0.4 * (((xx - wOrg / 2) / wOrg) ** 2 + ((yy - hOrg / 2) / hOrg) ** 2))).astype(np.uint16)
    U = R.Undistorter(cam, gam, vig)
    rx, ry = np.array(U.remapX, np.float32).copy(), np.array(U.remapY, np.float32).copy()
    assert ((ry > hOrg - 2) & (ry < hOrg - 1)).sum() > 10      # the reference's own table has entries in the last raw row's interval
    # make sure the open intervals next to the last column / row are populated (what a crop / full calibration produces at its borders)
    valid = np.nonzero(rx.reshape(-1) >= 0)[0]
    rx.flat[valid[5]] = wOrg - 1.5; ry.flat[valid[9]] = hOrg - 1.25; rx.flat[valid[11]] = np.nextafter(np.float32(wOrg - 1), np.float32(0)); ry.flat[valid[11]] = np.nextafter(np.float32(hOrg - 1), np.float32(0))
    raw = rng.integers(0, 256, (hOrg, wOrg)).astype(np.uint8)
    ctx = pkg.Context(w, h, n_slots=1)
    und = pkg.UndistorterHip(ctx, wOrg, hOrg, 8, U.G, U.vignetteMapInv, rx, ry)
    img = und.upload(0, raw)
    ref = oracle.undistort(raw, U.G, U.vignetteMapInv, rx, ry, w, h)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    # and the unmodified tables against the reference's own undistort
    und2 = pkg.UndistorterHip(ctx, wOrg, hOrg, 8, U.G, U.vignetteMapInv, U.remapX, U.remapY)
    ref_img, _ = U.undistort(raw, exposure=0.02)
    assert np.array_equal(und2.upload(0, raw).view(np.uint32), np.asarray(ref_img, np.float32).view(np.uint32))
