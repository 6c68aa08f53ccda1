#!/bin/bash
# usage (GPU box): tests/gpu_ab_nn.sh [reps] — k_nn_f16 A/B on one box: the library of HEAD~ (libquatro_hip_base.so), the
# current one, and the current test-engine build with one workgroup per compute unit (QTR_NN_WGS_PER_CU=1)
reps=${1:-2}
R=$GRAFT_REPO_ROOT
run() {  # label, env assignments...
  label=$1; shift
  env "$@" timeout 200 python $R/bench.py --steps 60 --cpu-seconds 0 --legs pair > /tmp/ab.json 2>/dev/null
  python - "$label" <<'PY'
import json, sys
d = json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[1], "ms_per_step", round(d["ms_per_step"], 4), "| nn launch us", round(1e3 * r["mean_launch_ms"], 2), "frac", round(r["frac"], 3),
      "| stages", {k: round(v, 4) for k, v in d.get("stage_ms", {}).items() if isinstance(v, (int, float))})
PY
}
for r in $(seq $reps); do
  run base QTR_LIB=$R/quatro_amd/libquatro_hip_base.so
  run new QTR_LIB=$R/quatro_amd/libquatro_hip.so
  run new_1wg QTR_LIB=$R/quatro_amd/libquatro_hip_testengines.so QTR_NN_WGS_PER_CU=1
done
