"""Where does a k_nn_f16 launch spend its time?  Diagnostic, not a test.

  (here)      python tests/probe/nn_stamps.py --build      # libquatro_hip_timing.so = the library with -DQTR_NN_TIMING
  (GPU box)   python tests/probe/nn_stamps.py              # a few registrations, then the stamps of the last one

Thread 0 of every workgroup stamps the shader clock and the 100 MHz wall clock at: kernel entry, after the plan, before
and after the hand-scheduled loop, after the partial records are written (match.hip, NN_STAMP)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
TLIB = os.path.join(ROOT, "quatro_amd", "libquatro_hip_timing.so")

if "--build" in sys.argv:
    csrc = os.path.join(ROOT, "quatro_amd", "csrc")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math",
                           "-Wno-unused-value", "-DQTR_NN_TIMING", os.path.join(csrc, "unity.hip"), "-ldl", "-o", TLIB])
    sys.exit(0)

import quatro_amd.lib as ql  # noqa: E402

ql.LIB_PATH = TLIB
import torch  # noqa: E402
from quatro_amd import synth  # noqa: E402

dev = torch.device("cuda", 0)
h = ql.Handle(0, max_points=131072, max_voxels=32768, max_corr=8192)
prm = ql.demo_params()
res = ql.Result()
s, t, _ = synth.kitti64_pair_16k(0)
sd, td = torch.from_numpy(s).to(dev), torch.from_numpy(t).to(dev)
fp = ql.default_frontend_params(seed=0)
for _ in range(8):
    rc = h.register_pair_dev(sd.data_ptr(), sd.shape[0], td.data_ptr(), td.shape[0], fp, prm, res, 0)
torch.cuda.synchronize()
print("n_src %d n_tgt %d L %d" % (res.n_src, res.n_tgt, res.n_corr))
lib = ql.load()
buf = np.zeros((2, 256, 12), dtype=np.uint64)
rc = lib.qtr_debug_nn_stamps(C.c_void_p(buf.ctypes.data))
assert rc == 0, rc
for d in range(2):
    b = buf[d].astype(np.int64)
    live = b[:, 10] > 0
    b = b[live]
    clk, wall = b[:, 0:10:2], b[:, 1:10:2]
    t0 = wall[:, 0].min()
    us = (wall - t0) / 100.0  # 100 MHz
    print("direction %d: %d workgroups with work, tiles per item %d..%d" % (d, live.sum(), b[:, 10].min(), b[:, 10].max()))
    names = ["entry", "after plan", "loop start", "loop end", "records written"]
    for i, nm in enumerate(names):
        print("  %-16s wall us since first entry: min %7.2f  median %7.2f  max %7.2f" % (nm, us[:, i].min(), np.median(us[:, i]), us[:, i].max()))
    dclk = clk[:, 3] - clk[:, 2]
    dwall = (wall[:, 3] - wall[:, 2]) / 100.0
    print("  loop: %.0f clocks/tile (median), %.2f us median, shader clock %.0f MHz" % (np.median(dclk / b[:, 10]), np.median(dwall), np.median(dclk / np.maximum(dwall, 1e-9))))
    for i in range(4):
        dc = clk[:, i + 1] - clk[:, i]
        print("  %-16s -> %-16s clocks: median %8.0f  max %8.0f" % (names[i], names[i + 1], np.median(dc), dc.max()))

# ---- the generic stamps (common.h QTR_STAMP): last launch of each instrumented kernel
KERNELS = ["desc_prep", "half_tables", "radix_scatter (raw cloud, middle pass)", "recheck_filter direction 0", "vox_centroids", "hit_compact",
           "cross_fused", "nn_finish", "recheck_filter direction 1", "spfh", "fpfh", "finalize"]
g = np.zeros((12, 32, 8, 2), dtype=np.uint64)
rc = lib.qtr_debug_stamps(C.c_void_p(g.ctypes.data))
assert rc == 0, rc
g = g.astype(np.int64)
for k, name in enumerate(KERNELS):
    pts = [p for p in range(8) if (g[k, :, p, 1] > 0).any()]
    if not pts:
        continue
    blocks = [b for b in range(32) if g[k, b, pts[0], 1] > 0]
    t0 = min(g[k, b, pts[0], 1] for b in blocks)
    print("%s: %d workgroups stamped" % (name, len(blocks)))
    for p in pts:
        us = np.array([(g[k, b, p, 1] - t0) / 100.0 for b in blocks if g[k, b, p, 1] > 0])
        print("  point %d: us since first entry  min %7.2f  median %7.2f  max %7.2f   (%d)" % (p, us.min(), np.median(us), us.max(), len(us)))
