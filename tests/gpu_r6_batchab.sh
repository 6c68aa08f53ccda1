#!/bin/bash
# usage (GPU box): tests/gpu_r6_batchab.sh — the batched leg (256 composite pairs, 32 slots) under the experiment knobs of the
# test build: lanes on disjoint compute-unit masks, more lanes.  Two rounds, same box.
R=$GRAFT_REPO_ROOT; cd $R
T=$R/quatro_amd/libquatro_hip_testengines.so
one() {  # label, env...
  label=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --cpu-seconds 0 --legs batch > /tmp/b.json 2>/tmp/b.err
  python - "$label" <<'PY'
import json, sys
try:
    d = json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
    b = d["batch256_leg"]
    print(f"{sys.argv[1]:34s} batch256 {b['value']:8.1f} /s  {b['ms_per_pair']:.4f} ms/pair identical {b['identical_to_sequential']} | scan pairs {b['scan_pairs']['value']:8.1f} | headline {d['ms_per_step']:.4f} ms")
except Exception as e:
    print(sys.argv[1], "FAILED", e, open('/tmp/b.err').read()[-400:])
PY
}
for r in 1 2; do
  one "product" QTR_X=0
  one "testbuild" QTR_LIB=$T
  one "cu-mask halves (2 lanes)" QTR_LIB=$T QTR_LANE_CU_MASK=halves
  one "cu-mask xcd (2 lanes)" QTR_LIB=$T QTR_LANE_CU_MASK=xcd
  one "3 lanes" QTR_LIB=$T QTR_BATCH_LANES=3
  one "3 lanes, cu-mask thirds" QTR_LIB=$T QTR_BATCH_LANES=3 QTR_LANE_CU_MASK=halves
  one "4 lanes" QTR_LIB=$T QTR_BATCH_LANES=4
  one "4 lanes, cu-mask quarters" QTR_LIB=$T QTR_BATCH_LANES=4 QTR_LANE_CU_MASK=halves
done
