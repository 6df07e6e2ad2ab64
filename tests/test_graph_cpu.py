"""The resident window graph (dmvio_hip_graph_*, csrc/capi_graph.hip) against a restatement of the reference's own mutators: EnergyFunctional::insertFrame / insertPoint /
insertResidual / dropResidual / removePoint / marginalizeFrame and makeIDX (src/dso/OptimizationBackend/EnergyFunctional.cpp:435-518, 641-646, 766-782, 997-1017) as
plain Python lists with the reference's statements (append; the LAST element takes a removed one's place; frames keep their order).  Random sequences of those calls — the
keyframe cycle of FullSystem::makeKeyFrame among them — must leave both with the same flat arrays, element for element.  Host only."""
import numpy as np
import pytest


class EF:
    """the reference's containers: frames -> points -> residualsAll (targets), mutated with the reference's statements"""

    def __init__(self):
        self.frames = []

    def insertFrame(self):
        self.frames.append([]); return len(self.frames) - 1

    def insertPoint(self, host, rec):
        self.frames[host].append(dict(rec=rec, res=[])); return len(self.frames[host]) - 1

    def insertResidual(self, host, i, target):
        self.frames[host][i]["res"].append(target); return len(self.frames[host][i]["res"]) - 1

    def dropResidual(self, host, i, k):
        res = self.frames[host][i]["res"]
        res[k] = res[-1]; res.pop()                       # p->residualsAll[r->idxInAll] = p->residualsAll.back(); pop_back()

    def removePoint(self, host, i):
        pts = self.frames[host]
        pts[i] = pts[-1]; pts.pop()                       # h->points[p->idxInPoints] = h->points.back(); pop_back()

    def marginalizeFrame(self, idx):
        assert not self.frames[idx]
        del self.frames[idx]                              # frames[i] = frames[i + 1] ... pop_back(): order kept
        for fr in self.frames:
            for p in fr:
                p["res"] = [(-1 if t == idx else (t - 1 if t > idx else t)) for t in p["res"]]

    def flat(self):
        host, rec, rp, rt = [], [], [], []
        for f, fr in enumerate(self.frames):
            for p in fr:
                for t in p["res"]:
                    rp.append(len(host)); rt.append(t)
                host.append(f); rec.append(p["rec"])
        return host, rec, rp, rt


def _rec(rng):
    return (np.float32(rng.uniform(4, 500)), np.float32(rng.uniform(4, 500)), np.float32(rng.uniform(0.1, 2)), rng.rand(8).astype(np.float32) * 255,
            rng.rand(8).astype(np.float32), bool(rng.rand() < 0.1))


def _same(g, ef):
    host, rec, rp, rt = ef.flat()
    o = g.export()
    F, N, R = g.counts()
    assert (F, N, R) == (len(ef.frames), len(host), len(rp))
    assert np.array_equal(o["host"], np.array(host, np.int32)) and np.array_equal(o["res_point"], np.array(rp, np.int32)) and np.array_equal(o["res_target"], np.array(rt, np.int32))
    if N:
        assert np.array_equal(o["u"], np.array([r[0] for r in rec], np.float32)) and np.array_equal(o["v"], np.array([r[1] for r in rec], np.float32))
        assert np.array_equal(o["idepth"], np.array([r[2] for r in rec], np.float32))
        assert np.array_equal(o["color"], np.stack([r[3] for r in rec])) and np.array_equal(o["weights"], np.stack([r[4] for r in rec]))
        assert np.array_equal(o["hasDepthPrior"], np.array([r[5] for r in rec], np.uint8))
    for f, fr in enumerate(ef.frames):
        assert g.frame_points(f) == len(fr)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_mutations_equal_the_references_containers(pkg, seed):
    rng = np.random.RandomState(seed)
    g = pkg.WindowGraph(); ef = EF()
    for _ in range(3):
        assert g.insert_frame() == ef.insertFrame()
    for step in range(3000):
        F = len(ef.frames)
        op = rng.randint(0, 100)
        if op < 30:                                                                           # insertPoint
            h = rng.randint(F); r = _rec(rng)
            assert g.insert_point(h, r[0], r[1], r[2], r[3], r[4], r[5]) == ef.insertPoint(h, r)
        elif op < 65:                                                                         # insertResidual
            h = rng.randint(F)
            if ef.frames[h] and F > 1:
                i = rng.randint(len(ef.frames[h])); have = ef.frames[h][i]["res"]
                cand = [t for t in range(F) if t != h and t not in have]
                if cand:
                    t = cand[rng.randint(len(cand))]
                    assert g.insert_residual(h, i, t) == ef.insertResidual(h, i, t)
        elif op < 80:                                                                         # dropResidual
            h = rng.randint(F)
            if ef.frames[h]:
                i = rng.randint(len(ef.frames[h]))
                if ef.frames[h][i]["res"]:
                    k = rng.randint(len(ef.frames[h][i]["res"]))
                    g.drop_residual(h, i, k); ef.dropResidual(h, i, k)
        elif op < 92:                                                                         # removePoint
            h = rng.randint(F)
            if ef.frames[h]:
                i = rng.randint(len(ef.frames[h]))
                g.remove_point(h, i); ef.removePoint(h, i)
        elif op < 96 and F < 10:                                                              # insertFrame
            assert g.insert_frame() == ef.insertFrame()
        elif F > 2:                                                                           # FullSystem::marginalizeFrame: the points of the frame first (flagPointsForRemoval ->
            idx = rng.randint(F)                                                              # removePoint), then ef->marginalizeFrame, then the residuals that targeted it
            while ef.frames[idx]:
                g.remove_point(idx, 0); ef.removePoint(idx, 0)
            g.remove_frame(idx); ef.marginalizeFrame(idx)
            for h, fr in enumerate(ef.frames):
                for i, p in enumerate(fr):
                    while -1 in p["res"]:
                        k = p["res"].index(-1)
                        g.drop_residual(h, i, k); ef.dropResidual(h, i, k)
        if step % 97 == 0:
            _same(g, ef)
    _same(g, ef)
    # values: one point, then all of them in flat order
    host, rec, _, _ = ef.flat()
    if host:
        g.set_idepth(host[0], 0, 0.75)
        assert g.export()["idepth"][ef.flat()[0].index(host[0])] == np.float32(0.75)
        g.set_idepths(np.arange(len(host), dtype=np.float32))
        assert np.array_equal(g.export()["idepth"], np.arange(len(host), dtype=np.float32))


@pytest.mark.parametrize("seed", [4, 5])
def test_linearised_records_travel_with_their_residuals(pkg, seed):
    """EFResidual::isLinearized / J / res_toZeroF (EnergyFunctionalStructs.h:63-87) are members of the residual object: they move with it when dropResidual fills a hole
    with the point's last residual, go when the residual or its point goes, and come out in makeIDX order (dmvio_hip_graph_set_residual_linearized / _export_linearized)."""
    rng = np.random.RandomState(seed)
    g = pkg.WindowGraph(); ef = EF()
    lin = {}                                   # id of the restatement's point dict -> list parallel to its "res": None or (J74, res_toZeroF)
    for _ in range(4):
        assert g.insert_frame() == ef.insertFrame()
    n_lin = 0

    def check():
        fl, J, r = g.export_linearized()
        want = [x for fr in ef.frames for p in fr for x in lin[id(p)]]
        assert len(want) == len(fl) and g.linearized_count() == sum(x is not None for x in want) == n_lin
        for ri, x in enumerate(want):
            if x is None:
                assert fl[ri] == 0 and not J[ri].any() and not r[ri].any()
            else:
                assert fl[ri] == 1 and np.array_equal(J[ri], x[0]) and np.array_equal(r[ri], x[1])

    for step in range(1500):
        F = len(ef.frames)
        op = rng.randint(0, 100)
        h = rng.randint(F)
        if op < 25:
            r = _rec(rng)
            i = ef.insertPoint(h, r); assert g.insert_point(h, r[0], r[1], r[2], r[3], r[4], r[5]) == i
            lin[id(ef.frames[h][i])] = []
        elif op < 55 and ef.frames[h]:
            i = rng.randint(len(ef.frames[h])); p = ef.frames[h][i]
            cand = [t for t in range(F) if t != h and t not in p["res"]]
            if cand:
                t = cand[rng.randint(len(cand))]
                assert g.insert_residual(h, i, t) == ef.insertResidual(h, i, t)
                lin[id(p)].append(None)
        elif op < 75 and ef.frames[h]:                                                        # fixLinearizationF / isLinearized = false on a random residual
            i = rng.randint(len(ef.frames[h])); p = ef.frames[h][i]
            if p["res"]:
                k = rng.randint(len(p["res"]))
                if rng.rand() < 0.75:
                    x = (rng.standard_normal(74).astype(np.float32), rng.standard_normal(8).astype(np.float32))
                    n_lin += lin[id(p)][k] is None
                    lin[id(p)][k] = x
                    g.set_residual_linearized(h, i, k, x[0], x[1])
                else:
                    n_lin -= lin[id(p)][k] is not None
                    lin[id(p)][k] = None
                    g.set_residual_linearized(h, i, k, None)
        elif op < 90 and ef.frames[h]:                                                        # dropResidual
            i = rng.randint(len(ef.frames[h])); p = ef.frames[h][i]
            if p["res"]:
                k = rng.randint(len(p["res"]))
                L = lin[id(p)]
                n_lin -= L[k] is not None
                L[k] = L[-1]; L.pop()
                g.drop_residual(h, i, k); ef.dropResidual(h, i, k)
        elif ef.frames[h]:                                                                    # removePoint
            i = rng.randint(len(ef.frames[h])); p = ef.frames[h][i]
            n_lin -= sum(x is not None for x in lin[id(p)])
            del lin[id(p)]
            g.remove_point(h, i); ef.removePoint(h, i)
        if step % 53 == 0:
            check(); _same(g, ef)
    check()
    r = _rec(rng)
    i = g.insert_point(0, r[0], r[1], r[2], r[3], r[4], r[5])
    with pytest.raises(pkg.HipLibraryError, match="residual index out of range"):
        g.set_residual_linearized(0, i, 0, np.zeros(74, np.float32), np.zeros(8, np.float32))
    g.clear()
    assert g.linearized_count() == 0


def test_errors_and_dangling_residuals(pkg):
    g = pkg.WindowGraph()
    with pytest.raises(pkg.HipLibraryError):
        g.insert_point(0, 1, 1, 1, np.zeros(8), np.zeros(8))                                   # no such frame
    a, b, c = g.insert_frame(), g.insert_frame(), g.insert_frame()
    p = g.insert_point(a, 10, 10, 1, np.zeros(8), np.ones(8))
    with pytest.raises(pkg.HipLibraryError):
        g.insert_residual(a, p, a)                                                             # target == host
    g.insert_residual(a, p, b); g.insert_residual(a, p, c)
    with pytest.raises(pkg.HipLibraryError):
        g.remove_frame(a)                                                                      # still hosts a point
    with pytest.raises(pkg.HipLibraryError):
        g.drop_residual(a, p, 2)
    g.remove_frame(b)                                                                          # the residual to b dangles, the one to c now targets frame 1
    o = g.export()
    assert list(o["res_target"]) == [-1, 1] and g.counts() == (2, 1, 2)
    g.drop_residual(a, p, 0)                                                                   # FullSystem::marginalizeFrame's own drop of it: the last residual takes index 0
    assert list(g.export()["res_target"]) == [1] and g.point_residuals(a, p) == 1
    with pytest.raises(pkg.HipLibraryError):
        g.set_idepths(np.zeros(5, np.float32))                                                 # wrong count
    g.export(); g.set_idepths(np.full(1, 0.4, np.float32))                                     # flat values are accepted for the structure that was flattened ...
    q = g.insert_point(a, 20, 20, 1, np.zeros(8), np.ones(8)); g.remove_point(a, q)
    with pytest.raises(pkg.HipLibraryError):
        g.set_idepths(np.full(1, 0.5, np.float32))                                             # ... and refused once it changed (same count, possibly another order)
    g.clear()
    assert g.counts() == (0, 0, 0)


def test_graph_of_a_case_equals_its_flat_arrays(pkg, synth):
    case = synth.ba_case(256, 192, n_frames=5, n_points=300, hosts_share=(90, 80, 70, 60, 0), seed=11)
    g = pkg.WindowGraph.from_case(case)
    o = g.export()
    # synth lists the points by host and the residuals by point already: the graph reproduces the arrays as given
    assert np.array_equal(o["host"], case["host"]) and np.array_equal(o["res_point"], case["res_point"]) and np.array_equal(o["res_target"], case["res_target"])
    assert np.array_equal(o["u"], case["u"].astype(np.float32)) and np.array_equal(o["color"], case["color"].astype(np.float32).reshape(-1, 8))
