"""NN kernel steady-state check (not a pytest module): random 33-D descriptors at growing sizes."""
import os, sys
os.environ["QTR_NN_TRACE"] = "1"
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from quatro_amd import lib as ql
rng = np.random.default_rng(0)
for n, waves in ((9000, 0), (9000, 1024), (9000, 2048), (18000, 0), (36000, 0), (72000, 0)):
    os.environ["QTR_NN_WAVES"] = str(waves)
    h = ql.Handle(0, max_points=131072, max_voxels=131072, max_corr=16384)
    d1 = (rng.random((n, 33)) * 100).astype(np.float32)
    d2 = (rng.random((n - 100, 33)) * 100).astype(np.float32)
    x1 = np.zeros((n, 4), np.float32); x1[:, :3] = rng.random((n, 3)) * 50
    x2 = np.zeros((n - 100, 4), np.float32); x2[:, :3] = rng.random((n - 100, 3)) * 50
    ts = []
    for _ in range(4):
        try:
            h.match(x1, d1, x2, d2, ql.default_frontend_params(seed=1))
        except Exception as e:
            print("err", e); break
        ts.append(h.stage_times()["nn_kernel"])
    if ts:
        print("n", n, "waves", waves, "nn_kernel ms (2 launches) min", round(min(ts), 4), "TF/s",
              round(2 * 66.0 * n * (n - 100) / (min(ts) * 1e-3) / 1e12, 1), "recheck", h.debug_fetch(ql.DBG_MATCH_STATS, np.int32)[8:10], "clk/tile, longest WG life us, mean life us, WGs", (lambda d: [int(d[12]), int(d[13]) / 100.0, int(d[14]) / max(int(d[15]), 1) / 100.0, int(d[15])])(h.debug_fetch(ql.DBG_MATCH_STATS, np.uint32)), flush=True)
    h.close()
