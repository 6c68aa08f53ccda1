"""Diagnostic (GPU box): N dense steps — BASELINE configs[4] as bench.py's dense_step_leg runs it, ONE qtr_register_pair_corr
call each (front end of two 50 000-point scans of synth.dense_scene_pair, no voxel down-sampling, + back end on 20 000
given correspondences) — and N data-connected dense registrations (qtr_register_pair, use_tuple_test = 0: the matcher's own
~14 k correspondences).  What rocprofv3 --kernel-trace is pointed at for profiles/r5_dense_step_kernel_stats.txt.
usage: python tests/gpu_dense_step_prof.py [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

from quatro_amd import lib as ql  # noqa: E402
from quatro_amd import synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda", 0)
if os.environ.get("QTR_DENSE_PREALLOC"):  # (bench.py's situation: another handle and a pool of scans allocated first)
    h0 = ql.Handle(0, max_points=131072, max_voxels=32768, max_corr=8192)
    pool0 = [torch.zeros(130000, 4, device=dev) for _ in range(8)]
    if os.environ.get("QTR_DENSE_PREALLOC") == "run":  # ... and used
        s0, t0_, _ = synth.kitti64_pair_16k(0)
        r0 = ql.Result()
        sd0, td0 = torch.from_numpy(s0).to(dev), torch.from_numpy(t0_).to(dev)
        for _ in range(50):
            h0.register_pair_dev(sd0.data_ptr(), sd0.shape[0], td0.data_ptr(), td0.shape[0], ql.default_frontend_params(seed=0),
                                 ql.demo_params(), r0)
h = ql.Handle(0, max_points=65536, max_voxels=65536, max_corr=24576)
prm = ql.demo_params()
res = ql.Result()
a, b, T = synth.dense_scene_pair(50000)
cs, ct, Tc, _ = synth.correspondences(20000, 0.02, seed=7, noise=0.1)
ad, bd = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
csd, ctd = torch.from_numpy(cs).to(dev), torch.from_numpy(ct).to(dev)
fp = ql.default_frontend_params(voxel_size=0.001, seed=1)
fpm = ql.default_frontend_params(voxel_size=0.001, use_tuple_test=0, seed=1)
h.set_stage_events(False)
h.set_nn_event_stride(0)
for name, fn in (("dense_step", lambda: h.register_pair_corr_dev(ad.data_ptr(), 50000, bd.data_ptr(), 50000, fp, csd.data_ptr(),
                                                                   ctd.data_ptr(), 20000, prm, res)),
                 ("dense_mutual", lambda: h.register_pair_dev(ad.data_ptr(), 50000, bd.data_ptr(), 50000, fpm, prm, res))):
    fn()
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {1e3 * (time.perf_counter() - t0) / reps:.3f} ms per registration  n_corr {res.n_corr} clique {res.n_clique} "
          f"final {res.n_final} valid {res.valid}", flush=True)
    if os.environ.get("QTR_DENSE_STAGES"):  # (one more call with the stage events on: where the time goes)
        h.set_stage_events(True)
        h.set_nn_event_stride(1)
        fn()
        fn()
        print("  stages", {k: round(v, 4) for k, v in h.stage_times().items() if isinstance(v, float)}, flush=True)
        h.set_stage_events(False)
        h.set_nn_event_stride(0)
h.close()
