"""Where does k_recheck_filter spend a dense-mode launch?  Diagnostic, not a test (needs libquatro_hip_timing.so:
python tests/probe/nn_stamps.py --build).  Wave 0 of the first 32 workgroups stamps: entry, set-up done, hot sweep done
(it ends at the first list overflow on dense clouds), dense loop done, end — and adds up the clocks it spent inside drain()
(the exact flann::L2 evaluation of the listed pairs), the pairs drained, the drains, the tiles of its slice."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import quatro_amd.lib as ql  # noqa: E402

ql.LIB_PATH = os.path.join(ROOT, "quatro_amd", "libquatro_hip_timing.so")
import torch  # noqa: E402
from quatro_amd import synth  # noqa: E402

dev = torch.device("cuda", 0)
h = ql.Handle(0, max_points=65536, max_voxels=65536, max_corr=24576)
prm, res = ql.demo_params(), ql.Result()
a, b, _ = synth.dense_scene_pair(50000)
ad, bd = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
fp = ql.default_frontend_params(voxel_size=0.001, seed=1)
for _ in range(4):
    h.register_pair_dev(ad.data_ptr(), 50000, bd.data_ptr(), 50000, fp, prm, res)
torch.cuda.synchronize()
print("n %d / %d  L %d" % (res.n_src, res.n_tgt, res.n_corr))
lib = ql.load()
buf = np.zeros((12, 32, 8, 2), dtype=np.uint64)
assert lib.qtr_debug_stamps(C.c_void_p(buf.ctypes.data)) == 0
for name, kid in (("direction 1", 3), ("direction 2", 8)):
    g = buf[kid].astype(np.int64)
    clk, wall = g[:, :5, 0], g[:, :5, 1]
    tot = clk[:, 4] - clk[:, 0]
    live = tot > 0
    g, clk, wall, tot = g[live], clk[live], wall[live], tot[live]
    print(f"{name}: {live.sum()} workgroups stamped; listed rows {g[0, 7, 0]}, items {g[0, 7, 1]}, tiles per slice {np.median(g[:, 6, 1]):.0f}")
    print(f"  wall us entry -> end: median {np.median((wall[:, 4] - wall[:, 0]) / 100.0):.1f}, max {((wall[:, 4] - wall[:, 0]) / 100.0).max():.1f};"
          f" shader clock {np.median(tot / np.maximum((wall[:, 4] - wall[:, 0]) / 100.0, 1e-9)):.0f} MHz")
    for i, nm in enumerate(("set-up", "hot sweep (until the first overflow)", "dense loop", "final merge")):
        print(f"  {nm:38s} {np.median((clk[:, i + 1] - clk[:, i]) / tot):6.1%} of the item's clocks")
    print(f"  inside drain() (exact evaluation)      {np.median(g[:, 5, 0] / tot):6.1%}   — {np.median(g[:, 5, 1]):.0f} pairs in {np.median(g[:, 6, 0]):.0f} drains per wave,"
          f" {np.median(g[:, 5, 0] / np.maximum(g[:, 5, 1], 1) * 64):.0f} clocks per 64 pairs; sweep: {np.median((tot - g[:, 5, 0]) / np.maximum(g[:, 6, 1], 1)):.0f} clocks per tile")
