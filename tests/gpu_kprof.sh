#!/bin/bash
# usage: tests/gpu_kprof.sh <tag> [pattern]   — kernel-trace of bench.py (20 steps), prints the per-kernel summary
tag=${1:-x}; pat=${2:-}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o $tag -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --stream-slots 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py $GRAFT_REPO_ROOT/gpurun_out/prof_$tag/${tag}_results.db 26 | grep -i "registrations\|$pat" | head -40
