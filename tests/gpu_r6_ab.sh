#!/bin/bash
# usage (GPU box): tests/gpu_r6_ab.sh LIB... — headline step + batched leg under each library (files under quatro_amd/), two
# rounds on one box; the last library then runs the GPU suite.  (Same-box A/B of a kernel change: keep the previous build
# as quatro_amd/libquatro_hip_prev.so.)
R=$GRAFT_REPO_ROOT; cd $R
for r in 1 2; do
for lib in "$@"; do
  QTR_LIB=$R/quatro_amd/$lib timeout 300 python bench.py --steps 40 --cpu-seconds 0 --legs batch,pair > /tmp/b.json 2>/tmp/b.err
  python - $lib <<'PY'
import json, sys
try:
    d = json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
    b = d["batch256_leg"]
    st = d.get("stage_ms", {})
    print(f"{sys.argv[1]:32s} step {d['ms_per_step']:.4f} ms ({d['value']:.0f}/s) pair {d['whole_pair_leg']['ms_per_step']:.4f} | batch256 {b['value']:7.1f}/s ident {b['identical_to_sequential']} scan {b['scan_pairs']['value']:7.1f} | stages {json.dumps({k: round(v, 4) for k, v in st.items() if isinstance(v, float)})[:200]}")
except Exception as e:
    print(sys.argv[1], "FAILED", e, open('/tmp/b.err').read()[-300:])
PY
done
done
