"""Per-keyframe set-up cost of the BA window: dmvio_hip_ba_set_window + dmvio_hip_ba_set_graph (8 keyframes, 2000 points, 12.8k residuals), and optimize(6) beside it."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
P = g.load_package(); import dmvio_amd.synth as synth
case = synth.ba_case(512, 512, n_frames=8, n_points=2000)
ctx = P.Context(512, 512, n_slots=8)
for k in range(8): ctx.frame_upload(k, case["imgs"][k])
ba = P.BundleAdjusterHip(ctx)
ts, to = [], []
for _ in range(25):
    t0 = time.perf_counter(); ba.set_case(case, list(range(8))); t1 = time.perf_counter(); ba.optimize(6); t2 = time.perf_counter()
    ts.append(t1 - t0); to.append(t2 - t1)
print("set_window + set_graph: median %.3f ms (first %.3f ms); optimize(6): median %.3f ms" % (1e3 * np.median(ts[3:]), 1e3 * ts[0], 1e3 * np.median(to[3:])))

# the same window from the resident graph (dmvio_hip_ba_set_graph_from: stream-ordered, does not wait for its uploads); optimize(6) behind it absorbs what is left of them
g_ = P.WindowGraph.from_case(case)
F = case["n_frames"]
ts, to = [], []
for _ in range(25):
    t0 = time.perf_counter()
    ba.set_window(list(range(F)), case["poses0"], np.zeros((F, 2)), np.ones(F, np.float32), np.arange(F, dtype=np.int32), case["K4"]); ba.set_graph_from(g_)
    t1 = time.perf_counter(); ba.optimize(6); t2 = time.perf_counter()
    ts.append(t1 - t0); to.append(t2 - t1)
print("set_window + set_graph_from: median %.3f ms; optimize(6) behind it: median %.3f ms; together %.3f ms" % (1e3 * np.median(ts[3:]), 1e3 * np.median(to[3:]), 1e3 * np.median(np.array(ts[3:]) + np.array(to[3:]))))
