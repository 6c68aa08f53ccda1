#!/bin/bash
# usage (GPU box): tests/gpu_r5_scoutab.sh [reps] — the scout workgroup of k_hcore_async (large graphs of a single pair) against the
# library of the commit before it (libquatro_hip_prev.so) on one box: the dense legs and the connected path of bench.py
reps=${1:-2}
R=$GRAFT_REPO_ROOT
run() {
  label=$1; shift
  env "$@" timeout 400 python $R/bench.py --steps 20 --cpu-seconds 0 --legs dense,connected > /tmp/ab.json 2>/tmp/ab.err || tail -3 /tmp/ab.err
  python - "$label" <<'PY'
import json, sys
d = json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
out = [sys.argv[1], "step", round(d["ms_per_step"], 4)]
for k, v in d.items():
    if isinstance(v, dict) and ("dense" in k or "connected" in k):
        flat = {}
        def walk(p, x):
            if isinstance(x, dict):
                for a, b in x.items():
                    walk(p + "." + a if p else a, b)
            elif isinstance(x, (int, float)) and ("ms" in p or "value" in p or p.endswith("L") or "clique" in p):
                flat[p] = round(x, 4)
        walk("", v)
        out.append(k + " " + json.dumps(flat))
print(" | ".join(str(x) for x in out))
PY
}
for r in $(seq $reps); do
  run prev QTR_LIB=$R/quatro_amd/libquatro_hip_prev.so
  run new QTR_LIB=$R/quatro_amd/libquatro_hip.so
done
