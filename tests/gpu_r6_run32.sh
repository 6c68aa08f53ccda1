R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6t
for lib in libquatro_hip_prev.so libquatro_hip.so; do
for m in "" 1; do
QTR_LIB=$R/quatro_amd/$lib QTR_DENSE_PREALLOC=$m QTR_DENSE_STAGES=1 timeout 200 python tests/gpu_dense_step_prof.py 10 2>&1 | grep "ms per\|stages" | head -2 | sed "s/^/$lib prealloc=$m /"
done; done | cut -c1-260 | tee gpurun_out/r6t/dense_prio.txt
bash tests/gpu_r6_ab2.sh libquatro_hip_prev.so libquatro_hip.so > gpurun_out/r6t/ab_prio.txt 2>&1; cut -c1-300 gpurun_out/r6t/ab_prio.txt
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --legs dense,connected > /tmp/b.json 2>/tmp/b.err
python - <<'PY' | tee gpurun_out/r6t/bench_dense_prio.txt
import json
d = json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
print('bench dense_step ms', round(d['dense_step_leg']['ms_per_step'], 4))
c = d['connected_leg']
for k in ('mutual_nn', 'no_cross', 'dense', 'dense_mutual', 'l5k', 'l5k_dense18k'):
    if k in c: print(k, round(c[k]['ms_per_registration'], 3))
PY
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r6t/gpu_tests_full.txt 2>&1; grep -E "passed|failed|rror" gpurun_out/r6t/gpu_tests_full.txt | tail -3
