"""GPU parity of CoarseInitializer::calcResAndGS: per-point outputs bit-identical to the oracle; the two 9x9 reductions agree to
fp32 summation-order accuracy (the reference's own sums depend on its worker split)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lvl,alphaW", [(0, 150.0 * 150.0), (1, 150.0 * 150.0), (2, 0.0)])
def test_calc_res_and_gs_parity(pkg, oracle, synth, gpu_required, lvl, alphaW):
    from test_init_cpu import init_case
    c = init_case(synth, oracle, w=512, h=512, lvl=lvl, n=3000, seed=20 + lvl)
    ctx = pkg.Context(c["w"], c["h"], n_slots=2)
    ctx.frame_upload(0, c["img0"]); ctx.frame_upload(1, c["img1"])
    ini = pkg.CoarseInitializerHip(ctx)
    ini.set_points(c["pts"])
    dI0 = oracle.make_images(c["img0"], c["w"], c["h"])[0]; dI1 = oracle.make_images(c["img1"], c["w"], c["h"])[0]
    kw = dict(alphaW=alphaW, alphaK=2.5 * 2.5, couplingWeight=1.0, priorY=0.3, priorX=0.1)
    o = oracle.init_calc_res_and_gs(dI0[lvl], dI1[lvl], c["wl"], c["hl"], c["Ki"], c["K_lvl"], c["pose7"], c["aff"], c["pts"], c["idepth_new"], **kw)
    g = ini.calcResAndGS(lvl, 0, 1, c["Ki"], c["K_lvl"], c["pose7"], c["aff"], c["idepth_new"], **kw)
    assert np.array_equal(g["isGood_new"], o["isGood_new"])
    acc = o["isGood_new"].astype(bool); good_in = c["pts"]["isGood"].astype(bool)
    assert acc.sum() > 1500
    bits = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    assert np.array_equal(bits(g["energy_new"]), bits(o["energy_new"]))
    assert np.array_equal(bits(g["maxstep"]), bits(o["maxstep"]))
    assert np.array_equal(bits(g["lastHessian_new"][acc]), bits(o["lastHessian_new"][acc]))
    assert np.array_equal(bits(g["JbBuffer_new"][good_in]), bits(o["JbBuffer_new"][good_in]))
    assert g["res3"][2] == o["res3"][2] and g["res3"][1] == o["res3"][1]
    assert abs(g["res3"][0] - o["res3"][0]) <= 2e-5 * abs(o["res3"][0])
    for k in ("H", "Hsc"):
        sc = np.sqrt(np.outer(np.abs(np.diag(o[k])) + 1e-20, np.abs(np.diag(o[k])) + 1e-20))
        assert np.max(np.abs(g[k] - o[k]) / sc) < 5e-5, k
    for k, hk in (("b", "H"), ("bsc", "Hsc")):
        assert np.max(np.abs(g[k] - o[k]) / (np.abs(o[k]) + np.sqrt(np.abs(np.diag(o[hk])) * max(o["res3"][0], 1.0)))) < 5e-5, k
