"""Do the batched path's two lanes overlap whatever streams the process created before the handle?  (HIP multiplexes streams onto a few
hardware queues: profiles/r6_ab.txt section 19.)  usage (GPU box): python tests/gpu_lane_queues.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quatro_amd import lib as ql, synth

dev = torch.device("cuda", 0)
prm = ql.demo_params()
pool = []
for pid in range(4):
    s, t, _ = synth.kitti64_pair_16k(pid)
    cs, ct, _, _ = synth.correspondences(5000, 0.05, seed=pid, noise=0.1)
    pool.append({"src": torch.from_numpy(s).to(dev), "tgt": torch.from_numpy(t).to(dev), "fp": ql.default_frontend_params(seed=pid),
                 "cs": torch.from_numpy(cs).to(dev), "ct": torch.from_numpy(ct).to(dev)})
pairs = [pool[i % 4] for i in range(256)]
keep = []
for extra in (0, 1, 1, 1, 1, 2, 3):  # streams created (and left alive) before each batch handle: torch streams, two per spare slot
    for _ in range(extra):
        keep.append(torch.cuda.Stream())
    hb = ql.Handle(0, max_points=131072, max_voxels=32768, max_corr=8192, n_slots=32)
    hb.register_batch_dev(pairs[:64], prm, corr=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hb.register_batch_dev(pairs, prm, corr=True)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"streams alive before the handle: {len(keep):2d} torch  -> batch256 {256 / el:8.1f} pairs/s", flush=True)
    hb.close()
