"""NN kernel sensitivity sweep (not a pytest module): base-slice count vs k_nn_mfma time."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from quatro_amd import lib as ql, synth
from oracle import oracle as qo
qo.set_threads(16)
s, t, _ = synth.kitti64_pair(0)
vs, vt = qo.voxelize(s, 0.3), qo.voxelize(t, 0.3)
_, _, ds = qo.fpfh(vs, 0.5, 0.75)
_, _, dt = qo.fpfh(vt, 0.5, 0.75)
for waves in (512, 1024, 1536, 2048, 3072, 4096, 8192):
    os.environ["QTR_NN_WAVES"] = str(waves)
    h = ql.Handle(0)
    ts = []
    for _ in range(6):
        h.match(vs, ds, vt, dt, ql.default_frontend_params(seed=1))
        ts.append(h.stage_times()["nn_kernel"])
    print("target waves", waves, "nn_kernel ms (2 launches) min/median", round(min(ts), 4), round(float(np.median(ts)), 4),
          "TF/s", round(2 * 66.0 * vs.shape[0] * vt.shape[0] / (min(ts) * 1e-3) / 1e12, 1), flush=True)
    h.close()
