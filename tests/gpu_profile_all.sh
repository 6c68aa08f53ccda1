#!/bin/bash
# usage: tests/gpu_profile_all.sh <tag> — bench JSON, kernel trace, and the two PMC passes (FETCH_SIZE / WRITE_SIZE) of bench.py
tag=${1:-x}
R=$GRAFT_REPO_ROOT
python $R/bench.py > $R/gpurun_out/bench_$tag.json 2> $R/gpurun_out/bench_$tag.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o $tag -- python $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --stream-slots 0 > $R/gpurun_out/prof_$tag.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch_$tag -o fetch -- python $R/bench.py --steps 8 --warmup 2 --cpu-seconds 0 --stream-slots 0 > $R/gpurun_out/pmc_fetch_$tag.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write_$tag -o write -- python $R/bench.py --steps 8 --warmup 2 --cpu-seconds 0 --stream-slots 0 > $R/gpurun_out/pmc_write_$tag.log 2>&1
ls $R/gpurun_out/prof_$tag $R/gpurun_out/pmc_fetch_$tag $R/gpurun_out/pmc_write_$tag
tail -c 600 $R/gpurun_out/bench_$tag.json
