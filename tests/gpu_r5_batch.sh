#!/bin/bash
# usage (GPU box): tests/gpu_r5_batch.sh OUTDIR — batched leg under the previous and the current library, stamps, batch kernel trace
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r5}
mkdir -p $O; cd $R
for lib in libquatro_hip_base.so libquatro_hip.so; do
  QTR_LIB=$R/quatro_amd/$lib timeout 300 python bench.py --steps 20 --cpu-seconds 0 --legs batch > /tmp/b.json 2>/dev/null
  python - $lib <<'PY'
import json, sys
d = json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
b = d["batch256_leg"]
print(sys.argv[1], "batch256", round(b["value"], 1), "/s", round(b["ms_per_pair"], 4), "ms/pair identical", b["identical_to_sequential"], "| scan pairs", round(b["scan_pairs"]["value"], 1), "| step ms", round(d["ms_per_step"], 4))
PY
done 2>&1 | tee $O/batch.txt
timeout 120 python tests/probe/nn_stamps.py > $O/stamps.txt 2>&1; grep -A12 "^direction\|^nn_finish\|^recheck" $O/stamps.txt | head -70
export TMPDIR=/tmp; cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats -d $O/prof_batch -o batch -- python $R/bench.py --steps 2 --warmup 1 --legs batch --cpu-seconds 0 > /dev/null 2>&1
cd $R
python profiles/summarize_rocpd.py $(ls $O/prof_batch/*.db | head -1) > $O/batch_kernel_stats.txt
rm -rf $O/prof_batch
head -16 $O/batch_kernel_stats.txt
