"""k_graph_build on the GPU box: bit matrix against the oracle at awkward sizes, then the graph stage's time at the bench
sizes.  QTR_LIB=<a -DQTR_TEST_ENGINES build> with QTR_GRAPH=tiles times the tile-per-workgroup comparison kernel.
    python tests/gpu_graph_bench.py [check|time|both]"""
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
import torch

from oracle import oracle as qo
from quatro_amd import lib as ql, synth

mode = sys.argv[1] if len(sys.argv) > 1 else "both"
h = ql.Handle(0, max_points=65536, max_voxels=32768, max_corr=24576)
prm = ql.demo_params()
if mode in ("check", "both"):
    qo.set_threads(min(16, qo.max_threads()))
    bad = 0
    for L in (2, 3, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 511, 513, 1000, 1281, 2049, 5000, 7777, 8193, 9001):
        s, t, _, _ = synth.correspondences(max(L, 3), 0.1 if L > 20 else 1.0, seed=L, noise=0.1)
        s, t = s[:L].copy(), t[:L].copy()
        if L >= 300:  # a few coincident / nearly coincident points: zero-length TIMs and the band of the binary32 screen
            s[5] = s[4]
            t[5] = t[4]
            s[9, :3] = s[8, :3] + np.float32(1e-4)
        r = h.solve(s, t, prm)
        bm = h.debug_fetch(ql.DBG_GRAPH_BITMAP, np.uint64).reshape(L, (L + 63) // 64)
        ref = qo.build_graph(s, t)
        o = qo.solve(s, t)
        ok = np.array_equal(bm, ref) and np.array_equal(r["clique"], o["clique"]) and np.array_equal(r["T"], o["T"])
        bad += 0 if ok else 1
        print(f"L={L}: bitmap {'==' if np.array_equal(bm, ref) else '!='} oracle, edges {int(r['n_edges'])}, "
              f"clique {r['clique'].size} {'ok' if ok else 'MISMATCH'}", flush=True)
    # noise bounds small enough that the binary32 screen must not decide anything (margin = inf), and a mid one
    for nb_, L in ((0.002, 700), (0.004, 300), (0.02, 1500), (3.0, 900)):
        s, t, _, _ = synth.correspondences(L, 0.2, seed=L, noise=nb_ / 3)
        p2 = ql.demo_params(noise_bound=nb_)
        h.solve(s, t, p2)
        bm = h.debug_fetch(ql.DBG_GRAPH_BITMAP, np.uint64).reshape(L, (L + 63) // 64)
        ref = qo.build_graph(s, t, noise_bound=nb_)
        ok = np.array_equal(bm, ref)
        bad += 0 if ok else 1
        print(f"noise_bound={nb_} L={L}: bitmap {'==' if ok else '!='} oracle ({int(np.unpackbits(ref.view(np.uint8)).sum()) // 2} edges)", flush=True)
    print("graph check:", "0 mismatches" if bad == 0 else f"{bad} MISMATCHES", flush=True)
if mode in ("time", "both"):
    dev = torch.device("cuda", 0)
    res = ql.Result()
    for L, frac in ((1000, 0.1), (5000, 0.05), (20000, 0.02)):
        s, t, _, _ = synth.correspondences(L, frac, seed=7, noise=0.1)
        sd, td = torch.from_numpy(s).to(dev), torch.from_numpy(t).to(dev)
        for _ in range(3):
            h.solve_dev(sd.data_ptr(), td.data_ptr(), L, prm, res)
        g = []
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            h.solve_dev(sd.data_ptr(), td.data_ptr(), L, prm, res)
            g.append(h.stage_times()["graph"])
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print(f"L={L}: graph stage {1e3 * np.median(g):.1f} us (min {1e3 * min(g):.1f}), solve {1e3 * el / n:.3f} ms, "
              f"clique {res.n_clique}", flush=True)
h.close()
