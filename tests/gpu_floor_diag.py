"""GPU box: what k_hcore_async's floor does on the bench's back-end inputs: floor, iterations of the slowest workgroup,
second runs, time per solve.   QTR_LIB=... python tests/gpu_floor_diag.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from quatro_amd import synth
from quatro_amd import lib as ql

h = ql.Handle(0, lib_path=os.environ.get("QTR_LIB"))
for L, frac, noise in ((5000, 0.05, 0.1), (5000, 0.02, 0.3), (3000, 0.1, 0.2), (8000, 0.05, 0.2)):
    src, tgt, T, inl = synth.correspondences(L, frac, 4, noise=noise)
    for _ in range(3):
        r = h.solve(src, tgt)
    st = h.debug_fetch(ql.DBG_SOLVER_STATE, np.int32)
    t0 = time.perf_counter()
    for _ in range(50):
        r = h.solve(src, tgt)
    dt = (time.perf_counter() - t0) / 50
    print("L", L, "frac", frac, "clique", len(r["clique"]), "max_core", r["max_core"], "floor", st[29], "tainted", st[30], "second run", st[22],
          "iters", st[10], "rounds", st[9], "ms/solve (host arrays)", round(1e3 * dt, 4), flush=True)
h.close()
