#!/bin/bash
# usage (GPU box): tests/gpu_ab_lib.sh LIB_A LIB_B [reps] — bench.py's headline step, whole-pair leg and solver leg under two
# builds of the library, alternating on the same box (boxes differ by a few per cent, runs by ~1.5 %)
A=$1; B=$2; reps=${3:-3}
for r in $(seq $reps); do
  for lib in $A $B; do
    QTR_LIB=$lib timeout 200 python $GRAFT_REPO_ROOT/bench.py --steps 60 --cpu-seconds 0 --legs pair,solver5k > /tmp/ab.json 2>/dev/null
    python - "$(basename $lib)" <<'PY'
import json, sys
d = json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
rr = d.get("repeat_regions", {})
print(sys.argv[1], round(d["value"], 1), "reg/s", "ms_per_step", round(d["ms_per_step"], 4), "repeat median", rr.get("median"),
      "| whole pair ms", round(d.get("whole_pair_leg", {}).get("ms_per_step", 0), 4), "| solver5k ms",
      round(d.get("solver_L5000_leg", {}).get("ms_per_solve", 0), 4))
PY
  done
done
