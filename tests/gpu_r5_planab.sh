#!/bin/bash
# usage (GPU box): tests/gpu_r5_planab.sh [reps] — the single-pair plan of k_nn_f16 (host / k_hit_compact, nn_plan_single)
# against the library of the commit before it (libquatro_hip_prev.so) on one box, and the runtime's kernel-argument
# placement (HIP_FORCE_DEV_KERNARG) under the current library
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
  run prev QTR_LIB=$R/quatro_amd/libquatro_hip_prev.so
  run new QTR_LIB=$R/quatro_amd/libquatro_hip.so
  run new_kernarg0 QTR_LIB=$R/quatro_amd/libquatro_hip.so HIP_FORCE_DEV_KERNARG=0
  run new_kernarg1 QTR_LIB=$R/quatro_amd/libquatro_hip.so HIP_FORCE_DEV_KERNARG=1
done
