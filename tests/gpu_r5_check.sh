#!/bin/bash
# usage (GPU box): tests/gpu_r5_check.sh OUTDIR [pytest -k expression] — the GPU suite, a same-box A/B of the previous library
# (libquatro_hip_base.so) against the current one, and one default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r5}; K=${2:-}
mkdir -p $O
cd $R
if [ -n "$K" ]; then
  timeout 900 python -m pytest tests -m gpu -q -x -k "$K" > $O/pytest.log 2>&1
else
  timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
fi
echo "pytest rc $?" >> $O/pytest.log
tail -25 $O/pytest.log
bash tests/gpu_ab_lib.sh $R/quatro_amd/libquatro_hip_base.so $R/quatro_amd/libquatro_hip.so 2 > $O/ab.txt 2>&1
cat $O/ab.txt
timeout 600 python bench.py --cpu-seconds 8 > $O/bench.json 2> $O/bench.err
echo "bench rc $?"
python - $O/bench.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("no bench line", e); sys.exit(0)
r = d.get("roofline", {})
print("value", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), "nn us", round(1e3 * r.get("mean_launch_ms", 0), 2), "frac", round(r.get("frac", 0), 3))
for k in ("whole_pair_leg", "batch256_leg", "solver_L5000_leg", "dense_step_leg", "dense_solver_leg", "dense_frontend_leg"):
    v = d.get(k)
    if v: print(k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in ("value", "ms_per_step", "ms_per_pair", "ms_per_solve", "record", "n_corr", "identical_to_sequential")}, "roof", {kk: round(vv, 3) if isinstance(vv, float) else vv for kk, vv in (v.get("roofline") or {}).items() if kk in ("frac", "mean_launch_ms", "direction1", "direction2")})
c = d.get("connected_leg", {})
for k, v in c.items():
    if isinstance(v, dict): print("connected", k, round(v["value"], 1), "/s", round(v["ms_per_registration"], 3), "ms", v["pairs"][0])
print("parity", d.get("parity_vs_oracle", {}).get("all_pool_pairs_ok"), "stage_ms", d.get("stage_ms"))
PY
