"""Diagnostic (GPU box): a batch of composite pairs, then every slot's solver state — did k_hcore_async converge
(state[10] = its iterations + 1) or did the peeling workgroup take over (hundreds of rounds)?"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
import torch
from quatro_amd import lib as ql, synth

dev = torch.device("cuda", 0)
nslots = int(sys.argv[1]) if len(sys.argv) > 1 else 32
hb = ql.Handle(0, max_points=131072, max_voxels=32768, max_corr=8192, n_slots=nslots)
items = []
for k in range(4):
    s, t, _ = synth.kitti64_pair_16k(k)
    c = synth.correspondences(5000, 0.05, seed=k, noise=0.1)
    items.append({"src": torch.from_numpy(s).to(dev), "tgt": torch.from_numpy(t).to(dev), "fp": ql.default_frontend_params(seed=k),
                  "cs": torch.from_numpy(c[0]).to(dev), "ct": torch.from_numpy(c[1]).to(dev)})
torch.cuda.synchronize()
prm = ql.demo_params()
batch = [items[i % 4] for i in range(128)]
hb.register_batch_dev(batch[:64], prm, corr=True)
t0 = time.perf_counter()
res = hb.register_batch_dev(batch, prm, corr=True)
el = time.perf_counter() - t0
iters = [int(hb.debug_fetch(ql.DBG_SOLVER_STATE, np.int32, slot=sl)[10]) for sl in range(nslots)]
print(f"{128 / el:.1f} composite/s; k-core rounds per slot of the last chunk: {iters}")
hb.close()
