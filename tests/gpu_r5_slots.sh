#!/bin/bash
# usage (GPU box): tests/gpu_r5_slots.sh — the batched leg by number of stream slots (two lanes of half as many pairs each)
R=$GRAFT_REPO_ROOT; cd $R
for slots in 32 48 64 96; do
  timeout 300 python bench.py --steps 20 --cpu-seconds 0 --legs batch --batch-slots $slots > /tmp/b.json 2>/dev/null
  python - $slots <<'PY'
import json, sys
d = json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
b = d["batch256_leg"]
print("slots", sys.argv[1], "batch256", round(b["value"], 1), "/s", round(b["ms_per_pair"], 4), "ms/pair identical", b["identical_to_sequential"], "| scan pairs", round(b["scan_pairs"]["value"], 1))
PY
done
