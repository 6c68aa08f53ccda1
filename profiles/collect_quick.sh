#!/bin/bash
# usage (GPU box): profiles/collect_quick.sh TAG — kernel traces of the headline loop and of the dense legs, summarised
TAG=${1:-q}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_seq -o seq -- python $R/bench.py --steps 40 --warmup 5 --legs "" --cpu-seconds 0 > /dev/null 2>&1
timeout 240 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_dense -o dense -- python $R/bench.py --steps 2 --warmup 1 --legs dense --cpu-seconds 0 > /dev/null 2>&1
cd $R
db() { ls gpurun_out/prof_${TAG}_$1/*.db | head -1; }
python profiles/summarize_rocpd.py $(db seq) > gpurun_out/$TAG/kernel_stats.txt
python profiles/timeline.py $(db seq) 30 > gpurun_out/$TAG/timeline.txt
python profiles/summarize_rocpd.py $(db dense) > gpurun_out/$TAG/dense_kernel_stats.txt
python profiles/timeline.py $(db dense) 12 > gpurun_out/$TAG/dense_timeline.txt
rm -rf gpurun_out/prof_${TAG}_seq gpurun_out/prof_${TAG}_dense
head -45 gpurun_out/$TAG/kernel_stats.txt
