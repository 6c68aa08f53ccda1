#!/bin/bash
# usage (GPU box): tests/gpu_r6_ab2.sh LIB... — as gpu_r6_ab.sh, but 400 steps per region and the median / minimum of the bench's
# five regions (the 40-step form cannot see a 2 % change: its regions differ by +-10 us on one box), three rounds.
R=$GRAFT_REPO_ROOT; cd $R
for r in 1 2 3; do
for lib in "$@"; do
  QTR_LIB=$R/quatro_amd/$lib timeout 300 python bench.py --steps 400 --cpu-seconds 0 --legs batch,pair > /tmp/b.json 2>/tmp/b.err
  python - $lib <<'PY'
import json, sys
try:
    d = json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
    b = d["batch256_leg"]
    rr = sorted([d["ms_per_step"]] + d["repeat_regions"]["ms_per_step"])
    st = d.get("stage_ms", {})
    print(f"{sys.argv[1]:26s} step median {rr[2]:.4f} min {rr[0]:.4f} max {rr[-1]:.4f} ms | pair {d['whole_pair_leg']['ms_per_step']:.4f} | batch256 {b['value']:7.1f}/s ident {b['identical_to_sequential']} scan {b['scan_pairs']['value']:7.1f} | stages {json.dumps({k: round(v, 4) for k, v in st.items() if isinstance(v, float)})[:150]}")
except Exception as e:
    print(sys.argv[1], "FAILED", e, open('/tmp/b.err').read()[-300:])
PY
done
done
