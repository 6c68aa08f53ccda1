#!/bin/bash
# usage (GPU box): tests/gpu_r5_batchab.sh LIB... — the batched leg under each library, two rounds; batch tests under the last one
R=$GRAFT_REPO_ROOT; cd $R
for r in 1 2; do
for lib in "$@"; do
  QTR_LIB=$R/quatro_amd/$lib timeout 300 python bench.py --steps 20 --cpu-seconds 0 --legs batch > /tmp/b.json 2>/dev/null
  python - $lib <<'PY'
import json, sys
d = json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
b = d["batch256_leg"]
print(sys.argv[1], "batch256", round(b["value"], 1), "/s", round(b["ms_per_pair"], 4), "ms/pair identical", b["identical_to_sequential"], "| scan pairs", round(b["scan_pairs"]["value"], 1))
PY
done
done
last="${@: -1}"
QTR_LIB=$R/quatro_amd/$last timeout 600 python -m pytest tests -m gpu -q -x -k "batch" 2>&1 | tail -2
