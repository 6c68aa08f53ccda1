"""Dense-mode diagnostic (not a pytest module): python tests/gpu_dense_diag.py [scene|planes] — one dense registration with
stage events, the matcher's statistics (hit rows, rows sent to the exact re-check per direction) and the stage times.
QTR_LIB selects the library build."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: F401
from quatro_amd import lib as ql, synth

which = sys.argv[1] if len(sys.argv) > 1 else "scene"
a, b, T = synth.dense_scene_pair(50000) if which == "scene" else synth.dense_pair(50000)
h = ql.Handle(0, max_points=65536, max_voxels=65536, max_corr=24576)
fp = ql.default_frontend_params(voxel_size=0.001, seed=1)
for rep in range(3):
    r = h.register_pair(a, b, fp)
    st = h.stage_times()
    ms = h.debug_fetch(ql.DBG_MATCH_STATS, np.int32)
    print(os.path.basename(os.environ.get("QTR_LIB", "default")), which, "L", r["L"], "clique", r["clique"].size, "n_hit", int(ms[7]),
          "recheck rows", int(ms[8]), int(ms[9]), "| ms", {k: round(float(v), 3) for k, v in st.items() if k != "nn_launches"}, flush=True)
h.close()
