#!/bin/bash
# usage: tests/gpu_ab.sh VAR a b [reps] — A/B of bench.py (single stream) under an environment variable
var=$1; a=$2; b=$3; reps=${4:-2}
for r in $(seq $reps); do
  for v in $a $b; do
    env $var=$v timeout 200 python $GRAFT_REPO_ROOT/bench.py --steps 60 --cpu-seconds 0 --stream-slots 0 > /tmp/ab.json 2>/dev/null
    python - <<PY
import json
d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
print("$var=$v", round(d["value"],1), "reg/s  gpu total ms", d["stage_ms"]["total"])
PY
  done
done
