#!/bin/bash
# usage (GPU box): tests/gpu_ab_batch.sh VAR v1 v2 ... — bench.py's batched leg (256 composite pairs) on the test-engine build
# under values of one environment variable, two rounds, same box
var=$1; shift
export QTR_LIB=$GRAFT_REPO_ROOT/quatro_amd/libquatro_hip_testengines.so
for r in 1 2; do
  for v in "$@"; do
    env $var=$v timeout 300 python $GRAFT_REPO_ROOT/bench.py --steps 10 --cpu-seconds 0 --legs batch > /tmp/abb.json 2>/dev/null
    python - "$var=$v" <<'PY'
import json, sys
d = json.loads(open('/tmp/abb.json').read().strip().splitlines()[-1])
b = d.get("batch256_leg", {})
print(sys.argv[1], "batch256", round(b.get("value", 0), 1), "reg/s", "| scan pairs", round(b.get("scan_pairs", {}).get("value", 0), 1))
PY
  done
done
