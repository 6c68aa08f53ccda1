"""Diagnostic (GPU box): N back-end solves at L correspondences — the thing rocprofv3 --kernel-trace is pointed at for the
solver-chain summaries under profiles/ (r3_solver5k_*, r3_dense_solver_*).
usage: python tests/gpu_solver_prof.py L [reps] [inlier_frac]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

from quatro_amd import lib as ql  # noqa: E402
from quatro_amd import synth  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
frac = float(sys.argv[3]) if len(sys.argv) > 3 else (0.05 if L <= 8192 else 0.02)
dev = torch.device("cuda", 0)
h = ql.Handle(0, max_points=65536, max_voxels=65536, max_corr=max(8192, L + 64))
prm = ql.demo_params()
res = ql.Result()
items = []
for sid in range(4):
    s, t, _, _ = synth.correspondences(L, frac, seed=sid if L <= 8192 else 7, noise=0.1)
    items.append((torch.from_numpy(s).to(dev), torch.from_numpy(t).to(dev)))
for s, t in items:
    h.solve_dev(s.data_ptr(), t.data_ptr(), L, prm, res)
torch.cuda.synchronize()
t0 = time.perf_counter()
acc = {}
for k in range(reps):
    s, t = items[k % len(items)]
    h.solve_dev(s.data_ptr(), t.data_ptr(), L, prm, res)
    for key, v in h.stage_times().items():
        acc[key] = acc.get(key, 0.0) + float(v)
torch.cuda.synchronize()
el = time.perf_counter() - t0
import numpy as np  # noqa: E402
stt = h.debug_fetch(ql.DBG_SOLVER_STATE, np.int32)
print("k_finalize phase clocks/16 (members+TIMs, GNC, rot-inliers+raw, COTE+rest):", [int(a) for a in stt[11:15]],
      "COTE steps:", [int(a) for a in stt[16:22]])
print(f"L={L} reps={reps} kcore_iters={int(stt[10])} clique_rounds={int(stt[9])} ms_per_solve={1e3 * el / reps:.4f} n_clique={res.n_clique} stage_ms=" +
      str({k: round(v / reps, 4) for k, v in acc.items() if v}))
