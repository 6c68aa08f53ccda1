#!/bin/bash
# Counters of EVERY kernel of one workload (round 6): a clean --kernel-trace run for the durations, then separate --pmc
# passes (SQ wave-cycle split, SQ instruction counts, FETCH_SIZE, WRITE_SIZE, L2 hit / miss) with --kernel-trace only
# beside them.   gpurun -- 'bash profiles/pmc_all.sh TAG WORKLOAD'   ->   gpurun_out/TAG/pmc_WORKLOAD.json
#   WORKLOAD  seq    the headline loop (bench.py --legs "")
#             dense  BASELINE configs[4] as one registration + the dense scene's own correspondences (tests/gpu_dense_step_prof.py)
#             batch  256 composite pairs through the batched entry points (bench.py --legs batch)
#             s5k / s20k   the back end alone at L = 5000 / 20000 (tests/gpu_solver_prof.py)
TAG=${1:-r6}
WL=${2:-seq}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG
mkdir -p $O
case $WL in
  seq)   CMD="python $R/bench.py --steps 8 --warmup 2 --legs  --cpu-seconds 0";;
  dense) CMD="python $R/tests/gpu_dense_step_prof.py 4";;
  batch) CMD="python $R/bench.py --steps 2 --warmup 1 --legs batch --cpu-seconds 0";;
  s5k)   CMD="python $R/tests/gpu_solver_prof.py 5000 12";;
  s20k)  CMD="python $R/tests/gpu_solver_prof.py 20000 4";;
  *) echo "unknown workload $WL"; exit 2;;
esac
cd /tmp
run() { tag=$1; shift; timeout 300 rocprofv3 --kernel-trace "$@" -d $O/prof_${WL}_$tag -o $tag -- $CMD > $O/run_${WL}_$tag.txt 2>&1; }
if [ "$WL" = seq ]; then
  # (an empty --legs argument has to survive word splitting)
  run() { tag=$1; shift; timeout 300 rocprofv3 --kernel-trace "$@" -d $O/prof_${WL}_$tag -o $tag -- python $R/bench.py --steps 8 --warmup 2 --legs "" --cpu-seconds 0 > $O/run_${WL}_$tag.txt 2>&1; }
fi
run clean
run a --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run b --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
run c --pmc FETCH_SIZE
run d --pmc WRITE_SIZE
run e --pmc TCC_HIT_sum TCC_MISS_sum
cd $R
db() { ls $O/prof_${WL}_$1/*.db 2>/dev/null | head -1; }
python profiles/summarize_counters.py $(db clean) $(db a) $(db b) $(db c) $(db d) $(db e) > $O/pmc_$WL.json 2> $O/pmc_$WL.err
python profiles/summarize_rocpd.py $(db clean) $([ "$WL" = seq ] && echo auto) > $O/${WL}_clean_kernel_stats.txt
rm -rf $O/prof_${WL}_*
ls -la $O | head -30
