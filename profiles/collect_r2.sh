#!/bin/bash
# Round-2 evidence, run on the GPU box from the repository root (gpurun -- 'bash profiles/collect_r2.sh TAG'):
#   gpurun_out/TAG_bench.json            the bench line (all legs)
#   gpurun_out/TAG_kernel_stats.txt      rocprofv3 --kernel-trace of the headline loop (sequential single pairs)
#   gpurun_out/TAG_batch_kernel_stats.txt  ... of the batched leg (256 pairs, groups of 16 per launch chain)
#   gpurun_out/TAG_pmc_nn.json           FETCH_SIZE / WRITE_SIZE of k_nn_f16 (NN_KERNEL overrides the name), two separate --pmc passes
#   (python profiles/timeline.py gpurun_out/prof_TAG_seq/seq_results.db > profiles/TAG_timeline.txt: one registration, dispatch by dispatch)
# Copy what is to be judged into profiles/.
TAG=${1:-r2}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/bench.py --steps 40 --warmup 5 > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_seq -o seq -- python $R/bench.py --steps 40 --warmup 5 --legs "" --cpu-seconds 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_batch -o batch -- python $R/bench.py --steps 2 --warmup 1 --legs batch --cpu-seconds 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof_${TAG}_fetch -o fetch -- python $R/bench.py --steps 8 --warmup 2 --legs "" --cpu-seconds 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof_${TAG}_write -o write -- python $R/bench.py --steps 8 --warmup 2 --legs "" --cpu-seconds 0 > /dev/null 2>&1
cd $R
python profiles/summarize_rocpd.py $(ls gpurun_out/prof_${TAG}_seq/*.db | head -1) 57 > gpurun_out/${TAG}_kernel_stats.txt
python profiles/summarize_rocpd.py $(ls gpurun_out/prof_${TAG}_batch/*.db | head -1) > gpurun_out/${TAG}_batch_kernel_stats.txt
python profiles/summarize_pmc.py $(ls gpurun_out/prof_${TAG}_fetch/*.db | head -1) $(ls gpurun_out/prof_${TAG}_write/*.db | head -1) "${NN_KERNEL:-void k_nn_f16}" > gpurun_out/${TAG}_pmc_nn.json
rm -rf gpurun_out/prof_${TAG}_fetch gpurun_out/prof_${TAG}_write
