#!/bin/bash
# Round-4 evidence, run on the GPU box from the repository root (gpurun -- 'bash profiles/collect_r3.sh TAG'):
#   gpurun_out/TAG_bench.json                 the bench line (all legs)
#   gpurun_out/TAG_kernel_stats.txt           rocprofv3 --kernel-trace of the headline loop (composite step, sequential)
#   gpurun_out/TAG_timeline.txt               one step of that trace, dispatch by dispatch
#   gpurun_out/TAG_batch_kernel_stats.txt     ... of the batched leg (256 pairs)
#   gpurun_out/TAG_solver5k_kernel_stats.txt  ... of the back end alone at L = 5000 (tests/gpu_solver_prof.py)
#   gpurun_out/TAG_dense_solver_kernel_stats.txt   ... at L = 20000
#   gpurun_out/TAG_dense_frontend_kernel_stats.txt ... of the dense legs (50 k-point front end + L = 20000 solver)
#   gpurun_out/TAG_pmc_nn.json                FETCH_SIZE / WRITE_SIZE of k_nn_f16, two separate --pmc passes
#   gpurun_out/TAG_pmc_graph.json             ... of k_graph_build at L = 5000 and 20000 (+ TCC hit / miss)
# Copy what is to be judged into profiles/.  (Every rocprofv3 run sits under `timeout`: one of them once stayed alive for
# a quarter of an hour after its "tool finalization" line; the database is complete by then.)
TAG=${1:-r4}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
python $R/bench.py --steps 40 --warmup 5 > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_seq -o seq -- python $R/bench.py --steps 40 --warmup 5 --legs "" --cpu-seconds 0 > /dev/null 2>&1
timeout 240 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_batch -o batch -- python $R/bench.py --steps 2 --warmup 1 --legs batch --cpu-seconds 0 > /dev/null 2>&1
timeout 240 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_s5k -o s5k -- python $R/tests/gpu_solver_prof.py 5000 20 > $R/gpurun_out/${TAG}_solver5k_run.txt 2>&1
timeout 240 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_s20k -o s20k -- python $R/tests/gpu_solver_prof.py 20000 6 > $R/gpurun_out/${TAG}_solver20k_run.txt 2>&1
timeout 240 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_dense -o dense -- python $R/bench.py --steps 2 --warmup 1 --legs dense --cpu-seconds 0 > /dev/null 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof_${TAG}_fetch -o fetch -- python $R/bench.py --steps 8 --warmup 2 --legs "" --cpu-seconds 0 > /dev/null 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof_${TAG}_write -o write -- python $R/bench.py --steps 8 --warmup 2 --legs "" --cpu-seconds 0 > /dev/null 2>&1
for L in 5000 20000; do
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof_${TAG}_gf$L -o f -- python $R/tests/gpu_solver_prof.py $L 4 > /dev/null 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof_${TAG}_gw$L -o w -- python $R/tests/gpu_solver_prof.py $L 4 > /dev/null 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/prof_${TAG}_gh$L -o h -- python $R/tests/gpu_solver_prof.py $L 4 > /dev/null 2>&1
done
cd $R
db() { ls gpurun_out/prof_${TAG}_$1/*.db | head -1; }
python profiles/summarize_rocpd.py $(db seq) 57 > gpurun_out/${TAG}_kernel_stats.txt
python profiles/timeline.py $(db seq) 30 > gpurun_out/${TAG}_timeline.txt
python profiles/summarize_rocpd.py $(db batch) > gpurun_out/${TAG}_batch_kernel_stats.txt
python profiles/summarize_rocpd.py $(db s5k) 24 > gpurun_out/${TAG}_solver5k_kernel_stats.txt
python profiles/summarize_rocpd.py $(db s20k) 10 > gpurun_out/${TAG}_dense_solver_kernel_stats.txt
python profiles/summarize_rocpd.py $(db dense) > gpurun_out/${TAG}_dense_frontend_kernel_stats.txt
python profiles/summarize_pmc.py $(db fetch) $(db write) "${NN_KERNEL:-void k_nn_f16}" > gpurun_out/${TAG}_pmc_nn.json
python profiles/summarize_pmc_graph.py ${TAG} > gpurun_out/${TAG}_pmc_graph.json
# SQ-level counters of the graph kernel in both forms (where the wave-cycles go)
bash profiles/pmc_sq.sh ${TAG}_graph20k 20000 "void k_graph_build<" > /dev/null 2>&1
bash profiles/pmc_sq.sh ${TAG}_graph5k 5000 "void k_graph_build_tiles" > /dev/null 2>&1
rm -rf gpurun_out/prof_${TAG}_fetch gpurun_out/prof_${TAG}_write gpurun_out/prof_${TAG}_gf* gpurun_out/prof_${TAG}_gw* gpurun_out/prof_${TAG}_gh*
