import sys, numpy as np
sys.path.insert(0, '.')
from quatro_amd import lib as ql, synth
h = ql.Handle(0)
for pid in range(3):
    s, t, _ = synth.kitti64_pair(pid)
    r = h.register_pair(s, t, ql.default_frontend_params(seed=pid))
    st = h.debug_fetch(ql.DBG_SOLVER_STATE, np.int32)
    print("pair", pid, "L", r["L"], "clique", r["clique"].size, "max_core", r["max_core"], "edges", r["n_edges"], "clique_rounds", st[9], "kcore_rounds", st[10], "finalize cycles/16: members,gnc,rot+raw,cote", st[11:15].tolist(), "kcore sweeps max,sum,active", st[22:25].tolist(), {k_: round(v_, 3) for k_, v_ in h.stage_times().items() if k_ in ("clique", "solve", "total")})
for L, frac in ((5000, 0.05), (2000, 0.1)):
    a, b, _, _ = synth.correspondences(L, frac, seed=4, noise=0.1)
    r = h.solve(a, b)
    st = h.debug_fetch(ql.DBG_SOLVER_STATE, np.int32)
    print("solver L", L, "clique", r["clique"].size, "edges", r["n_edges"], "clique_rounds", st[9], "kcore_rounds", st[10], "finalize cycles/16: members,gnc,rot+raw,cote", st[11:15].tolist(), "kcore sweeps max,sum,active", st[22:25].tolist(), {k_: round(v_, 3) for k_, v_ in h.stage_times().items() if k_ in ("clique", "solve", "total")})
