R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6w
for s in 0 0.25 0 0.25 1.0; do
python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --legs "" --settle-seconds $s > /tmp/b.json 2>/tmp/b.err
python - $s <<'PY'
import json, sys
d = json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
print(f"settle {sys.argv[1]:5s} s: contract region {d['ms_per_step']:.4f} ms ({d['value']:.0f}/s)  repeats {d['repeat_regions']['ms_per_step']}  {d['config']['settle']}")
PY
done 2>&1 | tee gpurun_out/r6w/settle.txt
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6w/bench_driver.json ) 2> gpurun_out/r6w/bench_driver.time; tail -3 gpurun_out/r6w/bench_driver.time
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r6w/bench_driver.json').read().strip().splitlines()[-1])
print('driver-style:', d['value'], d['ms_per_step'], d['repeat_regions']['ms_per_step'], 'batch', d['batch256_leg']['value'], 'dense', d['dense_step_leg']['ms_per_step'], 'parity', d['parity_vs_oracle']['all_pool_pairs_ok'])
PY
