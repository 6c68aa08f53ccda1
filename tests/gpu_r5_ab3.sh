#!/bin/bash
# usage (GPU box): tests/gpu_r5_ab3.sh OUTDIR LIB... — matcher tests, then the headline step under each library build, three rounds
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r5}; shift
mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "nn or match or f16 or adversarial or golden or register" > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
for r in 1 2 3; do
  for lib in "$@"; do
    QTR_LIB=$R/quatro_amd/$lib timeout 200 python $R/bench.py --steps 60 --cpu-seconds 0 --legs pair > /tmp/ab.json 2>/dev/null
    python - "$lib" <<'PY'
import json, sys
d = json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
r = d["roofline"]; st = d.get("stage_ms", {})
print(sys.argv[1], "ms_per_step", round(d["ms_per_step"], 4), "median", d["repeat_regions"]["median"], "| nn us", round(1e3 * r["mean_launch_ms"], 2), "frac", round(r["frac"], 3),
      "dir1/2", round(1e3 * st.get("nn_dir1", 0), 1), round(1e3 * st.get("nn_dir2", 0), 1), "match", round(st.get("match", 0), 4), "| whole pair", round(d.get("whole_pair_leg", {}).get("ms_per_step", 0), 4))
PY
  done
done 2>&1 | tee $O/ab3.txt
