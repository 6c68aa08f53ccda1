"""prints k_finalize's phase clocks (SolverState.pad) for the bench pool pairs — run on the GPU box"""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from quatro_amd import lib as ql, synth
h = ql.Handle(0, max_points=131072, max_voxels=32768, max_corr=8192)
for pid in range(4):
    s, t, _ = synth.kitti64_pair_16k(pid)
    r = h.register_pair(s, t, ql.default_frontend_params(seed=pid))
    pad = h.debug_fetch(ql.DBG_SOLVER_STATE, np.int32)[10:32]
    print("pair", pid, "L", r["L"], "clique", len(r["clique"]), "gnc_iters", r.get("gnc_iters"), "clocks/16: members+TIMs", pad[1], "GNC", pad[2],
          "rot-inliers", pad[3], "COTE+final", pad[4], "COTE steps", [int(x) for x in pad[6:12]])
