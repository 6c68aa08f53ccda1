"""How many correspondences does the matcher keep on synthetic scan pairs?  (CPU, oracle only; ~1 minute.)

BASELINE.json quotes its metric on "~5k correspondences".  This probe runs voxel grid -> FPFH -> mutual NN -> tuple
test through the oracle on variants of the synthetic scene and prints, per variant: voxel counts, mutual-NN matches
(and how many of them land within 0.6 m of the true location), and L after the tuple test.  Result (round 3): every
scene that is a second SCAN keeps 250-650; only a jittered copy of the same sweep (same sample points, 5 mm noise, no
re-voxelisation offset) reaches thousands.  The tuple test keeps a random match with probability ~3.5e-4 per trial
(three edge-length ratios within 5 %), i.e. L ~ 0.1 * mutual unless the matches are true — and FPFH on 0.3 m voxels is
too noisy for that (half a voxel of grid offset alone drops the true fraction from 97 % to 45 %).
usage: python tests/probe/synth_L_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as qo  # noqa: E402
from quatro_amd import synth  # noqa: E402

qo.set_threads(min(16, qo.max_threads()))


def match_counts(s, t, T, seed=0):
    vs, vt = qo.voxelize(s, 0.3), qo.voxelize(t, 0.3)
    ds, dt = qo.fpfh(vs, 0.5, 0.75)[2], qo.fpfh(vt, 0.5, 0.75)[2]
    cm = qo.match(vs, ds, vt, dt, True, False, 0.95, seed)
    c = qo.match(vs, ds, vt, dt, True, True, 0.95, seed)
    R, tt = T[:3, :3], T[:3, 3]

    def err(cc):
        return np.linalg.norm(vs[cc[:, 0], :3].astype(np.float64) @ R.T + tt - vt[cc[:, 1], :3], axis=1)
    return vs.shape[0], vt.shape[0], len(cm), int((err(cm) < 0.6).sum()), len(c), int((err(c) < 0.6).sum())


def run(name, pid=0, **kw):
    t0 = time.time()
    s, t, T = synth.kitti64_pair(pid, **kw)
    ns, nt, m, mt, L, Lt = match_counts(s, t, T, pid)
    print(f"{name:34s} n {ns:5d}/{nt:5d}  mutual {m:5d} (true {mt:5d})  L {L:5d} (true {Lt:5d})  {time.time() - t0:.1f}s", flush=True)


if __name__ == "__main__":
    K = synth.KITTI16K
    run("bench profile (KITTI16K)", **K)
    run("baseline <= 2 m", **{**K, "max_xy": 2.0})
    run("baseline <= 0.5 m", **{**K, "max_xy": 0.5})
    run("same pose (dz only)", **{**K, "max_xy": 0.0, "max_yaw": 0.0})
    run("solid crowns", **{**K, "leaf_p": 1.0})
    run("solid crowns, baseline <= 2 m", **{**K, "leaf_p": 1.0, "max_xy": 2.0})
    run("2000 solid clutter boxes", n_clutter=2000, clear_r=5.0)
    run("2000 clutter, <= 2 m, sigma 5 mm", n_clutter=2000, clear_r=5.0, max_xy=2.0, max_yaw=0.2, sigma=0.005)
    run("near facades (r >= 15 m), <= 2 m", n_far=300, far_r0=15.0, clear_r=5.0, max_xy=2.0)
    s, _, _ = synth.kitti64_pair(0, **K)
    rng = np.random.default_rng(1)
    eye = np.eye(4)
    for name, sig in (("COPY of the sweep, jitter 5 mm", 0.005), ("COPY of the sweep, jitter 2 cm", 0.02)):
        t2 = s.copy()
        t2[:, :3] += rng.normal(0, sig, (s.shape[0], 3)).astype(np.float32)
        ns, nt, m, mt, L, Lt = match_counts(s, t2, eye)
        print(f"{name:34s} n {ns:5d}/{nt:5d}  mutual {m:5d} (true {mt:5d})  L {L:5d} (true {Lt:5d})")
    t2 = s.copy()
    t2[:, :3] += np.float32(0.15)
    Tm = np.eye(4)
    Tm[:3, 3] = 0.15
    ns, nt, m, mt, L, Lt = match_counts(s, t2, Tm)
    print(f"{'COPY shifted by half a voxel':34s} n {ns:5d}/{nt:5d}  mutual {m:5d} (true {mt:5d})  L {L:5d} (true {Lt:5d})")
