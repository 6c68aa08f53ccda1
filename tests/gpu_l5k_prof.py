"""Diagnostic (GPU box): the data-connected registrations at the metric's L ~ 5 k (bench.py connected_leg.l5k / l5k_dense18k),
N each, one qtr_register_pair call per registration — what rocprofv3 --kernel-trace is pointed at for
profiles/r6_l5k_kernel_stats.txt.   usage: python tests/gpu_l5k_prof.py [reps] [leaf]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

from quatro_amd import lib as ql  # noqa: E402
from quatro_amd import synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
leaf = float(sys.argv[2]) if len(sys.argv) > 2 else 0.07
dev = torch.device("cuda", 0)
h = ql.Handle(0, max_points=131072, max_voxels=65536, max_corr=8192)
prm = ql.demo_params()
res = ql.Result()
h.set_stage_events(False)
h.set_nn_event_stride(0)
cases = []
for pid in range(4):
    s, t, _ = synth.kitti64_pair_16k(pid)
    cases.append((f"l5k pool {pid}", torch.from_numpy(s).to(dev), torch.from_numpy(t).to(dev),
                  ql.default_frontend_params(voxel_size=leaf, use_tuple_test=0, seed=pid)))
a, b, _ = synth.dense_scene_pair(18000)
cases.append(("l5k_dense18k", torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev),
              ql.default_frontend_params(voxel_size=0.001, use_tuple_test=0, seed=1)))
for name, s, t, fp in cases:
    fn = lambda: h.register_pair_dev(s.data_ptr(), s.shape[0], t.data_ptr(), t.shape[0], fp, prm, res)  # noqa: E731
    fn()
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {1e3 * (time.perf_counter() - t0) / reps:.3f} ms per registration  n ({res.n_src}, {res.n_tgt}) n_corr {res.n_corr} "
          f"clique {res.n_clique} final {res.n_final} valid {res.valid}", flush=True)
