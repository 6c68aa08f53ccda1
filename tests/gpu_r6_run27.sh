R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6s
( time python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r6s/smoke.txt 2>&1; tail -5 gpurun_out/r6s/smoke.txt | cut -c1-300
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6s/bench_driver.json ) 2> gpurun_out/r6s/bench_driver.time; tail -4 gpurun_out/r6s/bench_driver.time
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r6s/bench_driver.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['repeat_regions']['ms_per_step'], d['config']['host_cpus'], 'cpp', d['cpp_driver_leg']['ms_per_step'], 'batch', d['batch256_leg']['value'])
print('parity', d['parity_vs_oracle']['all_pool_pairs_ok'], 'roofline', d['roofline']['frac'], d['roofline'].get('kernel_sha_now'), d['roofline'].get('traffic_kernel_sha'), d['roofline'].get('mfma_busy_kernel_sha'))
PY
timeout 700 python tests/gpu_fuzz.py 70 500 2>&1 | tail -2 | tee gpurun_out/r6s/fuzz70.txt
