"""Reproducer of tests/gpu_fuzz.py seed 72's two mismatches: a 64-correspondence pair whose largest clique is an EDGE (ties between
edges), inside a batch group whose largest pair has 5000 / 9000 correspondences.  usage (GPU box): python tests/gpu_repro_batch_tie.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from quatro_amd import lib as ql

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
hb = ql.Handle(0, max_points=65536, max_voxels=32768, max_corr=12288, n_slots=8)
h1 = ql.Handle(0)
for f in ("batch_tie_case2176.npz", "batch_tie_case5740.npz"):
    d = np.load(os.path.join(G, f))
    n, i, nb = len(d["sizes"]), int(d["pair"]), float(d["noise_bound"])
    sets = [(d[f"src{j}"], d[f"tgt{j}"]) for j in range(n)]
    prm = ql.demo_params(noise_bound=nb)
    got = hb.register_batch([(None, None, 0, a, b) for a, b in sets], params=prm)
    sl = i  # (a group's pairs take the handle's slots in order)
    L = sets[i][0].shape[0]
    core = hb.debug_fetch(ql.DBG_CORE, np.int32, slot=sl)[:L]
    perm = hb.debug_fetch(ql.DBG_PERM, np.int32, slot=sl)[:L]
    st = hb.debug_fetch(ql.DBG_SOLVER_STATE, np.int32, slot=sl)
    print("  slot", sl, "core numbers (nonzero):", {int(v): int(c) for v, c in enumerate(core) if c}, "\n  perm (rank -> vertex):", perm.tolist(),
          "\n  state", st[:12].tolist(), "floor/tainted/redo", st[29:32].tolist(), flush=True)
    alone = hb.register_batch([(None, None, 0, sets[i][0], sets[i][1])], params=prm)
    single = h1.solve(sets[i][0], sets[i][1], params=prm)
    core1 = h1.debug_fetch(ql.DBG_CORE, np.int32)[:L]
    perm1 = h1.debug_fetch(ql.DBG_PERM, np.int32)[:L]
    print("  qtr_solve: core equal", np.array_equal(core, core1), "perm", perm1.tolist(), flush=True)
    print(f, "sizes", d["sizes"].tolist(), "| in the group:", got[i]["clique"], "| alone in a batch:", alone[0]["clique"], "| qtr_solve:",
          single["clique"], "| oracle:", d["want_clique"], flush=True)
