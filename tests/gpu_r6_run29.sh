R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6s
QTR_DENSE_STAGES=1 timeout 200 python tests/gpu_dense_step_prof.py 10 2>&1 | grep "ms per\|stages" | tee gpurun_out/r6s/dense_script_stages.txt
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --legs dense --no-pin > /tmp/b.json 2>/tmp/b.err
python - <<'PY' | tee -a gpurun_out/r6s/dense_script_stages.txt
import json
d = json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
l = d['dense_step_leg']
print('bench --no-pin dense_step ms', round(l['ms_per_step'], 4), {k: round(v, 4) for k, v in l.get('stage_ms', {}).items() if isinstance(v, float)})
PY
