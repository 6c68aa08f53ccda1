R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6t
for q in default 2 8 16; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  for r in 1 2; do
  python bench.py --steps 200 --warmup 5 --cpu-seconds 0 --legs batch,rawbatch > /tmp/b.json 2>/tmp/b.err
  python - $q <<'PY'
import json, sys
d = json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
b = d['batch256_leg']
print(f"GPU_MAX_HW_QUEUES={sys.argv[1]:8s} step {d['ms_per_step']:.4f} | batch256 {b['value']:7.1f}/s scan {b['scan_pairs']['value']:7.1f} | raw sweeps {d['raw_batch_leg']['value']:7.1f}")
PY
  done
done 2>&1 | tee gpurun_out/r6t/hw_queues.txt
