#!/bin/bash
# usage (GPU box): tests/gpu_ab.sh VAR a b [reps] — A/B of bench.py's headline step and whole-pair leg under an environment
# variable, alternating a / b on the SAME box (boxes differ by a few per cent).  QTR_LIB selects the library build:
#   QTR_LIB=$PWD/quatro_amd/libquatro_hip_testengines.so tests/gpu_ab.sh QTR_GRAPH tiles strips 3
var=$1; a=$2; b=$3; reps=${4:-2}
for r in $(seq $reps); do
  for v in $a $b; do
    env $var=$v timeout 200 python $GRAFT_REPO_ROOT/bench.py --steps 60 --cpu-seconds 0 --legs pair,solver5k > /tmp/ab.json 2>/dev/null
    python - <<PY
import json
d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
rr=d.get("repeat_regions",{})
print("$var=$v", round(d["value"],1), "reg/s", "ms_per_step", round(d["ms_per_step"],4), "repeat median", rr.get("median"),
      "| whole pair ms", round(d.get("whole_pair_leg",{}).get("ms_per_step",0),4), "| solver5k ms", round(d.get("solver_L5000_leg",{}).get("ms_per_solve",0),4),
      "| graph", d["stage_ms"].get("graph"))
PY
  done
done
