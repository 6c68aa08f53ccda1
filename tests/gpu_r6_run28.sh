R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6s
for legs in dense "batch,dense" "pair,solver5k,batch,dense"; do
  python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --legs "$legs" > /tmp/b.json 2>/tmp/b.err
  python - "$legs" <<'PY'
import json, sys
d = json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
l = d['dense_step_leg']
print(sys.argv[1], '| dense_step ms', round(l['ms_per_step'], 4), 'stages', {k: round(v, 4) for k, v in l.get('stage_ms', {}).items() if isinstance(v, float)}, '| dense_solver', round(d['dense_solver_leg']['ms_per_solve'], 4))
PY
done 2>&1 | tee gpurun_out/r6s/dense_order.txt
timeout 200 python tests/gpu_dense_step_prof.py 10 2>&1 | grep "ms per" | tee -a gpurun_out/r6s/dense_order.txt
