"""no_cross (use_crosscheck = 0, no tuple test: L ~ 20 k correspondences straight from the scans) pair by pair: wall time, stage
times, solver state.  usage: python tests/gpu_nocross_diag.py [pair ids...] (run on the GPU box; under rocprofv3 for kernels)."""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
import torch
from quatro_amd import lib as ql, synth

dev = torch.device("cuda", 0)
h = ql.Handle(0, max_points=131072, max_voxels=65536, max_corr=32768)
prm = ql.demo_params()
res = ql.Result()
ids = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3]
for pid in ids:
    s, t, _ = synth.kitti64_pair_16k(pid)
    sd, td = torch.from_numpy(s).to(dev), torch.from_numpy(t).to(dev)
    fp = ql.default_frontend_params(seed=pid, use_crosscheck=0, use_tuple_test=0)
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = h.register_pair_dev(sd.data_ptr(), sd.shape[0], td.data_ptr(), td.shape[0], fp, prm, res)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    st = h.stage_times()
    ss = h.debug_fetch(ql.DBG_SOLVER_STATE, np.int32)
    names = ["mc", "best_r", "pos", "done", "t0", "ub", "batch", "max_core", "n_edges2", "rounds"]
    state = dict(zip(names, ss[:10].tolist()))
    state.update(kcore_rounds=int(ss[10]), twice=int(ss[22]), core_floor=int(ss[29]), tainted=int(ss[30]), redo=int(ss[31]))
    print(f"pair {pid} rc {rc} L {res.n_corr} clique {res.n_clique} wall_ms {1e3 * el:.3f}", {k: round(v, 3) for k, v in st.items()},
          state, flush=True)
h.close()
