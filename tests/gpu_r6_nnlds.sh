#!/bin/bash
# usage (GPU box): tests/gpu_r6_nnlds.sh — k_nn_f16 with its base tiles staged through LDS (libquatro_hip_nnlds.so, built with
# -DQTR_NN_LDS_STAGE from gen_nn_f16_core.py --lds) against the register-direct loop of the product library: the nearest-neighbour
# launches' own event times on the headline loop and on 50 k-point clouds, NN-identity tests under the experiment library.
R=$GRAFT_REPO_ROOT; cd $R
for r in 1 2; do
for lib in libquatro_hip.so libquatro_hip_nnlds.so; do
  QTR_LIB=$R/quatro_amd/$lib timeout 300 python bench.py --steps 40 --cpu-seconds 0 --legs dense --nn-event-stride 1 > /tmp/b.json 2>/tmp/b.err
  python - $lib <<'PY'
import json, sys
try:
    d = json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
    rf, ds = d["roofline"], d["dense_step_leg"]
    dr = ds["roofline"]
    print(f"{sys.argv[1]:28s} step {d['ms_per_step']:.4f} ms | nn launch mean {1e3 * rf['mean_launch_ms']:.2f} us frac {rf['frac']:.4f} | dense step {ds['ms_per_step']:.3f} ms, nn dir1 {1e3 * dr['direction1']['launch_ms']:.1f} us frac {dr['direction1']['frac']:.3f}, dir2 {1e3 * dr['direction2']['launch_ms']:.1f} us frac {dr['direction2']['frac']:.3f}")
except Exception as e:
    print(sys.argv[1], "FAILED", e, open('/tmp/b.err').read()[-400:])
PY
done
done
QTR_LIB=$R/quatro_amd/libquatro_hip_nnlds.so timeout 600 python -m pytest tests -x -q -m gpu -k "nn_engines or adversarial or match_equals or pool_matches or feature_pair_matches or dense_mode_front" 2>&1 | tail -3
