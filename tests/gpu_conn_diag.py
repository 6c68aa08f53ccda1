"""Where the time of the connected configurations goes (run on the GPU box): stage times + solver state per config."""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
import torch
from quatro_amd import lib as ql, synth

dev = torch.device("cuda", 0)
h = ql.Handle(0, max_points=131072, max_voxels=65536, max_corr=32768)
prm = ql.demo_params()
res = ql.Result()
s, t, _ = synth.kitti64_pair_16k(0)
sd, td = torch.from_numpy(s).to(dev), torch.from_numpy(t).to(dev)
for name, kw in (("default", {}), ("mutual_nn", dict(use_tuple_test=0)), ("no_cross", dict(use_crosscheck=0, use_tuple_test=0))):
    fp = ql.default_frontend_params(seed=0, **kw)
    for rep in range(3):
        t0 = time.perf_counter()
        rc = h.register_pair_dev(sd.data_ptr(), sd.shape[0], td.data_ptr(), td.shape[0], fp, prm, res)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    st = h.stage_times()
    ss = h.debug_fetch(ql.DBG_SOLVER_STATE, np.int32)
    ms = h.debug_fetch(ql.DBG_MATCH_STATS, np.int32)
    print(name, "rc", rc, "L", res.n_corr, "clique", res.n_clique, "wall_ms %.3f" % (1e3 * el),
          {k: round(v, 3) for k, v in st.items()}, "solver_state", ss[:12].tolist(), "match_stats", ms[:13].tolist(), flush=True)
h.close()
