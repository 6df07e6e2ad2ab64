"""CPU test of the input-edge oracle: photometric LUT + vignette + bilinear geometric remap (Undistort.cpp:214-250, 386-481)."""
import numpy as np
import pytest


def io_case(seed=4, wOrg=160, hOrg=120, w=128, h=96, bits=8):
    rng = np.random.RandomState(seed)
    top = 255 if bits == 8 else 65535
    raw = rng.randint(0, top + 1, (hOrg, wOrg)).astype(np.uint8 if bits == 8 else np.uint16)
    G = (255.0 * (np.arange(top + 1) / top) ** 0.8).astype(np.float32)            # inverse response, monotone
    yy, xx = np.mgrid[0:hOrg, 0:wOrg]
    vig = (1.0 / (1.0 - 0.3 * (((xx - wOrg / 2) / wOrg) ** 2 + ((yy - hOrg / 2) / hOrg) ** 2))).astype(np.float32)
    # radial-distortion style remap of the w x h output into the raw image; border pixels invalid (-1) like the reference's rounding guard
    v, u = np.mgrid[0:h, 0:w].astype(np.float64)
    xn, yn = (u - w / 2) / (0.9 * w), (v - h / 2) / (0.9 * w)
    r2 = xn * xn + yn * yn
    rx = ((xn * (1 + 0.15 * r2)) * (0.9 * wOrg) + wOrg / 2).astype(np.float32)
    ry = ((yn * (1 + 0.15 * r2)) * (0.9 * wOrg) + hOrg / 2).astype(np.float32)
    bad = (rx < 1) | (ry < 1) | (rx >= wOrg - 2) | (ry >= hOrg - 2)
    bad[::7, ::5] = True
    rx[bad] = -1; ry[bad] = -1
    return dict(raw=raw, G=G, vig=vig, rx=rx, ry=ry, w=w, h=h, wOrg=wOrg, hOrg=hOrg, bits=bits)


def test_undistort_matches_numpy(oracle):
    c = io_case()
    out = oracle.undistort(c["raw"], c["G"], c["vig"], c["rx"], c["ry"], c["w"], c["h"])
    data = (c["G"][c["raw"].astype(int)] * c["vig"]).astype(np.float64)
    valid = c["rx"] >= 0
    xi = np.floor(c["rx"]).astype(int); yi = np.floor(c["ry"]).astype(int)
    fx = c["rx"].astype(np.float64) - xi; fy = c["ry"].astype(np.float64) - yi
    xi[~valid] = 0; yi[~valid] = 0
    ref = (fx * fy * data[yi + 1, xi + 1] + (fy - fx * fy) * data[yi + 1, xi] + (fx - fx * fy) * data[yi, xi + 1] + (1 - fx - fy + fx * fy) * data[yi, xi])
    assert np.all(out[~valid] == 0)
    assert np.allclose(out[valid], ref[valid], rtol=1e-5, atol=1e-3)
    # no photometric calibration: factor * raw; passthrough without maps
    c2 = io_case(wOrg=128, hOrg=96)
    out2 = oracle.undistort(c2["raw"], None, None, None, None, 128, 96, factor=0.5)
    assert np.array_equal(out2, (0.5 * c2["raw"].astype(np.float32)))


def test_result_txt_writer_matches_oracle_and_matrix_algebra(pkg, oracle, tmp_path):
    """FullSystem::printResult through the library (host-only entry point) == the oracle's writer, byte for byte; the parsed poses equal
    firstPose^-1 * camToWorld (and camToWorld[ref] * camToTrackingRef for frames that are not keyframes) in float64 matrix algebra."""
    rng = np.random.RandomState(4)
    n = 40

    def rand_pose():
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        return np.concatenate([rng.normal(size=3), q])

    def mat(p):
        x, y, z, w = p[3:]
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        T = np.eye(4); T[:3, :3] = R; T[:3, 3] = p[:3]
        return T

    poses = np.stack([rand_pose() for _ in range(n)]); rel = np.stack([rand_pose() for _ in range(n)])
    ts = 1.4036365e9 + np.cumsum(rng.uniform(0.03, 0.07, n))
    valid = (rng.uniform(size=n) > 0.1).astype(np.uint8)
    ref = np.where(np.arange(n) % 4 == 0, -1, (np.arange(n) // 4) * 4).astype(np.int32)   # every 4th frame is a keyframe
    first = rand_pose()
    for kw in (dict(), dict(pose_valid=valid), dict(pose_valid=valid, tracking_ref=ref, camToTrackingRef7=rel)):
        a, b = tmp_path / "hip.txt", tmp_path / "orc.txt"
        pkg.write_result_txt(a, ts, poses, firstPose7=first, **kw)
        oracle.write_result_txt(b, ts, poses, firstPose7=first, **kw)
        assert a.read_bytes() == b.read_bytes()
        rows = np.loadtxt(a)
        keep = np.arange(n) if "pose_valid" not in kw else np.nonzero(valid)[0]
        assert rows.shape == (len(keep), 8) and np.allclose(rows[:, 0], ts[keep], rtol=1e-14, atol=0)   # 15 significant digits, as the reference
        for row, i in zip(rows, keep):
            c2w = mat(poses[i]) if ("tracking_ref" not in kw or ref[i] < 0) else mat(poses[ref[i]]) @ mat(rel[i])
            T = np.linalg.inv(mat(first)) @ c2w
            assert np.allclose(row[1:4], T[:3, 3], atol=1e-12)
            assert np.allclose(mat(row[1:])[:3, :3], T[:3, :3], atol=1e-12)
    assert np.loadtxt(tmp_path / "hip.txt").shape[0] == int(valid.sum())
    with pytest.raises(pkg.HipLibraryError):
        pkg.write_result_txt(tmp_path / "no_such_dir" / "x.txt", ts, poses)
