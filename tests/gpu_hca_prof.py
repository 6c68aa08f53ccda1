"""Diagnostic (GPU box): where k_hcore_async's time goes.  Needs a library built with -DQTR_HCA_PROF, in which every
workgroup accumulates 10 ns ticks per phase of its iterations:

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
        -fno-fast-math -Wno-unused-value -DQTR_HCA_PROF quatro_amd/csrc/unity.hip -ldl -o quatro_amd/libquatro_hip_prof.so
  QTR_LIB=$PWD/quatro_amd/libquatro_hip_prof.so python tests/gpu_hca_prof.py L [reps]

Prints the average over the workgroups and the slowest one: set-up, snapshots, the own rows (wave 0 alone, and the wait for
the other waves after it in lowering / idle iterations), the termination protocol, and the iteration counts."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

from quatro_amd import lib as ql  # noqa: E402
from quatro_amd import synth  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
frac = 0.05 if L <= 8192 else 0.02
dev = torch.device("cuda", 0)
h = ql.Handle(0, max_points=65536, max_voxels=65536, max_corr=max(8192, L + 64))
prm = ql.demo_params()
res = ql.Result()
s, t, _, _ = synth.correspondences(L, frac, seed=0 if L <= 8192 else 7, noise=0.1)
ds, dt = torch.from_numpy(s).to(dev), torch.from_numpy(t).to(dev)
lib = C.CDLL(ql.LIB_PATH)
names = ["setup", "snapshot", "rows_wave0", "protocol", "n_lowering", "n_idle", "wait_lowering", "wait_idle"]
acc = np.zeros((reps, 256, 8))
tot = 0.0
for r in range(reps + 2):
    t0 = time.perf_counter()
    h.solve_dev(ds.data_ptr(), dt.data_ptr(), L, prm, res)
    el = time.perf_counter() - t0
    if r < 2:
        continue
    tot += el
    buf = np.zeros(256 * 8, np.uint32)
    rc = lib.qtr_debug_hca_prof(buf.ctypes.data_as(C.c_void_p), 256)
    assert rc == 0, rc
    acc[r - 2] = buf.reshape(256, 8)
m = acc.mean(axis=0)  # per workgroup
busy = m[:, 4] + m[:, 5] > 0
stt = h.debug_fetch(ql.DBG_SOLVER_STATE, np.int32)
print(f"L={L} ms_per_solve={1e3 * tot / reps:.4f} kcore_iters={int(stt[10])} "
      f"clique={res.n_clique} workgroups={int(busy.sum())}")
for k, nm in enumerate(names):
    col = m[busy, k]
    if k in (0, 1, 2, 3, 6, 7):
        print(f"  {nm:12s} mean {col.mean() / 100:8.2f} us   max {col.max() / 100:8.2f} us")
    else:
        print(f"  {nm:12s} mean {col.mean():8.1f}      max {col.max():8.1f}")
for wg in (0, int(np.argmax(m[:, 1])), int(np.argmax(m[:, 2])), int(np.argmax(m[:, 3]))):
    print(f"  workgroup {wg:3d}: " + "  ".join(f"{nm} {m[wg, k] / (100 if k in (0, 1, 2, 3, 6, 7) else 1):.1f}" for k, nm in enumerate(names)))
