import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from quatro_amd import lib as ql, synth
h = ql.Handle(0, max_corr=24576)
for L, inl in ((2000, 0.1), (5000, 0.05), (8192, 0.05), (20000, 0.02)):
    src, tgt, _, _ = synth.correspondences(L, inl, seed=3, noise=0.05)
    r = h.solve(src, tgt)
    t0 = time.perf_counter()
    for _ in range(5):
        r = h.solve(src, tgt)
    dt = (time.perf_counter() - t0) / 5
    st = h.debug_fetch(ql.DBG_SOLVER_STATE, np.int32)
    print(os.environ.get("QTR_KCORE", "hcore"), "L", L, "ms", round(1e3 * dt, 3), "clique", len(r["clique"]), "max_core", st[7], "sweeps/rounds", st[10], "fallback", st[15])
