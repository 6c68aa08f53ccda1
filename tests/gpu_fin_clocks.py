"""Phase clocks of k_finalize on the bench pairs (run on the GPU box): python tests/gpu_fin_clocks.py"""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch  # noqa: F401,E402
from quatro_amd import lib as ql, synth  # noqa: E402
h = ql.Handle(0)
for pid in range(4):
    s, t, _ = synth.kitti64_pair(pid)
    for _ in range(3):
        r = h.register_pair(s, t, ql.default_frontend_params(seed=pid))
    st = h.debug_fetch(ql.DBG_SOLVER_STATE, np.int32)
    pad = st[10:32]
    us = lambda c: c * 16 / 2100.0  # clock64 ticks at ~2.1 GHz
    print(f"pair {pid}: L={r['L']} clique={r['clique'].size} | members+TIMs {us(pad[1]):.1f} us, GNC {us(pad[2]):.1f}, rot-inliers+raw {us(pad[3]):.1f}, "
          f"COTE+rest {us(pad[4]):.1f} | COTE steps {[round(us(x), 1) for x in pad[6:12]]} | stage {h.stage_times()['solve']:.4f} ms")
for L, frac in ((5000, 0.05), (8000, 0.05)):
    cs, ct, _, _ = synth.correspondences(L, frac, seed=4, noise=0.1)
    for _ in range(3):
        r = h.solve(cs, ct)
    st = h.debug_fetch(ql.DBG_SOLVER_STATE, np.int32)
    pad = st[10:32]
    print(f"solve L={L}: clique={r['clique'].size} rot inliers {r['rot_inliers'].size} | members+TIMs {us(pad[1]):.1f} us, GNC {us(pad[2]):.1f} ({r['gnc_iters']} iterations), "
          f"rot-inliers+raw {us(pad[3]):.1f}, COTE+rest {us(pad[4]):.1f} | COTE steps {[round(us(x), 1) for x in pad[6:12]]} | stage {h.stage_times()['solve']:.4f} ms")
