#!/bin/bash
# Round-6 evidence, run on the GPU box from the repository root (gpurun -- 'bash profiles/collect_r6.sh TAG'):
#   gpurun_out/TAG/bench.json                      the bench line (all legs; connected_l5k beside the headline)
#   gpurun_out/TAG/kernel_stats.txt + timeline.txt rocprofv3 --kernel-trace of the headline loop; one step dispatch by dispatch
#   gpurun_out/TAG/batch_kernel_stats.txt          ... of the batched leg (256 composite pairs)
#   gpurun_out/TAG/dense_step_kernel_stats.txt     BASELINE configs[4] as one registration (tests/gpu_dense_step_prof.py)
#   gpurun_out/TAG/l5k_kernel_stats.txt            the data-connected registrations at L ~ 5 k (tests/gpu_l5k_prof.py)
#   gpurun_out/TAG/solver5k_kernel_stats.txt, dense_solver_kernel_stats.txt, conn20k_kernel_stats.txt   as in round 5
#   gpurun_out/TAG/pmc_seq.json, pmc_batch.json, pmc_dense.json   counters of EVERY kernel of those workloads (profiles/pmc_all.sh)
#   gpurun_out/TAG/kernel_stamps.txt, recheck_stamps.txt   in-kernel stamps (need libquatro_hip_timing.so)
#   gpurun_out/TAG/mfma.txt                        matrix-pipe counters of k_nn_f16 (profiles/pmc_mfma.sh)
#   gpurun_out/TAG/gpu_tests.txt                   pytest -m gpu
# Copy what is to be judged into profiles/ as r6_*.  (Every rocprofv3 run sits under `timeout`.)
TAG=${1:-r6}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest $R/tests -q -m gpu > $O/gpu_tests_full.txt 2>&1; grep -E "passed|failed" $O/gpu_tests_full.txt | tail -1 > $O/gpu_tests.txt
python $R/bench.py --steps 40 --warmup 5 > $O/bench.json 2> $O/bench.err
cd /tmp
prof() { tag=$1; shift; timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o $tag -- "$@" > $O/run_$tag.txt 2>&1; }
prof seq python $R/bench.py --steps 40 --warmup 5 --legs "" --cpu-seconds 0
prof batch python $R/bench.py --steps 2 --warmup 1 --legs batch --cpu-seconds 0
prof s5k python $R/tests/gpu_solver_prof.py 5000 20
prof s20k python $R/tests/gpu_solver_prof.py 20000 6
prof dstep python $R/tests/gpu_dense_step_prof.py 6
prof conn python $R/tests/gpu_conn_diag.py
prof l5k python $R/tests/gpu_l5k_prof.py 4
cd $R
db() { ls $O/prof_$1/*.db | head -1; }
python profiles/summarize_rocpd.py $(db seq) auto > $O/kernel_stats.txt
python profiles/timeline.py $(db seq) 30 > $O/timeline.txt
python profiles/summarize_rocpd.py $(db batch) > $O/batch_kernel_stats.txt
python profiles/summarize_rocpd.py $(db s5k) 24 > $O/solver5k_kernel_stats.txt
python profiles/summarize_rocpd.py $(db s20k) 10 > $O/dense_solver_kernel_stats.txt
python profiles/summarize_rocpd.py $(db dstep) > $O/dense_step_kernel_stats.txt
python profiles/summarize_rocpd.py $(db conn) > $O/conn20k_kernel_stats.txt
python profiles/summarize_rocpd.py $(db l5k) > $O/l5k_kernel_stats.txt
grep "ms per" $O/run_l5k.txt >> $O/l5k_kernel_stats.txt
rm -rf $O/prof_*
bash profiles/pmc_all.sh $TAG seq > $O/pmc_seq.log 2>&1
bash profiles/pmc_all.sh $TAG batch > $O/pmc_batch.log 2>&1
bash profiles/pmc_all.sh $TAG dense > $O/pmc_dense.log 2>&1
bash profiles/pmc_mfma.sh $TAG > $O/mfma.log 2>&1; cp $R/gpurun_out/${TAG}_mfma.txt $O/mfma.txt 2>/dev/null
[ -f $R/quatro_amd/libquatro_hip_timing.so ] && timeout 120 python tests/probe/nn_stamps.py > $O/kernel_stamps.txt 2>&1
[ -f $R/quatro_amd/libquatro_hip_timing.so ] && timeout 120 python tests/probe/recheck_stamps.py > $O/recheck_stamps.txt 2>&1
rm -f $O/run_*.txt
ls -la $O
