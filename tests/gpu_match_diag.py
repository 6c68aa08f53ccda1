"""GPU box: how many rows the matcher's f16 filter leaves to the exact re-check on the bench pool.  python tests/gpu_match_diag.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from quatro_amd import lib as ql, synth
h = ql.Handle(0)
for pid in range(3):
    s, t, _ = synth.kitti64_pair_16k(pid)
    r = h.register_pair(s, t, ql.default_frontend_params(seed=pid))
    ms = h.debug_fetch(ql.DBG_MATCH_STATS, np.int32)
    print("pair", pid, "n", r["n_src"], r["n_tgt"], "L", r["L"], "stats", ms.tolist())
h.close()
