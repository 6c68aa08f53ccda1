#!/bin/bash
# Round-5 evidence, run on the GPU box from the repository root (gpurun -- 'bash profiles/collect_r5.sh TAG'):
#   gpurun_out/TAG/bench.json                     the bench line (all legs)
#   gpurun_out/TAG/kernel_stats.txt + timeline.txt    rocprofv3 --kernel-trace of the headline loop; one step dispatch by dispatch
#   gpurun_out/TAG/batch_kernel_stats.txt         ... of the batched leg (256 composite pairs)
#   gpurun_out/TAG/solver5k_kernel_stats.txt, dense_solver_kernel_stats.txt    the back end alone at L = 5000 / 20000
#   gpurun_out/TAG/dense_step_kernel_stats.txt    BASELINE configs[4] as one registration (tests/gpu_dense_step_prof.py)
#   gpurun_out/TAG/conn20k_kernel_stats.txt + conn20k_timeline.txt   use_crosscheck = 0: L ~ 20 k of the matcher's own
#   gpurun_out/TAG/rawbatch_kernel_stats.txt      the demo's whole sequence on raw sweeps, batched
#   gpurun_out/TAG/pmc_nn.json                    FETCH_SIZE / WRITE_SIZE of k_nn_f16 (two separate --pmc passes)
#   gpurun_out/TAG/kernel_stamps.txt              in-kernel stamps of k_nn_f16 (tests/probe/nn_stamps.py; needs libquatro_hip_timing.so)
# Copy what is to be judged into profiles/ as r5_*.  (Every rocprofv3 run sits under `timeout`.)
TAG=${1:-r5}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG
mkdir -p $O
python $R/bench.py --steps 40 --warmup 5 > $O/bench.json 2> $O/bench.err
cd /tmp
prof() { tag=$1; shift; timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o $tag -- "$@" > $O/run_$tag.txt 2>&1; }
prof seq python $R/bench.py --steps 40 --warmup 5 --legs "" --cpu-seconds 0
prof batch python $R/bench.py --steps 2 --warmup 1 --legs batch --cpu-seconds 0
prof s5k python $R/tests/gpu_solver_prof.py 5000 20
prof s20k python $R/tests/gpu_solver_prof.py 20000 6
prof dstep python $R/tests/gpu_dense_step_prof.py 6
prof conn python $R/tests/gpu_conn_diag.py
prof rawb python $R/bench.py --steps 2 --warmup 1 --legs rawbatch --cpu-seconds 0
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof_fetch -o fetch -- python $R/bench.py --steps 8 --warmup 2 --legs "" --cpu-seconds 0 > /dev/null 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof_write -o write -- python $R/bench.py --steps 8 --warmup 2 --legs "" --cpu-seconds 0 > /dev/null 2>&1
cd $R
db() { ls $O/prof_$1/*.db | head -1; }
python profiles/summarize_rocpd.py $(db seq) auto > $O/kernel_stats.txt
python profiles/timeline.py $(db seq) 30 > $O/timeline.txt
python profiles/summarize_rocpd.py $(db batch) > $O/batch_kernel_stats.txt
python profiles/summarize_rocpd.py $(db s5k) 24 > $O/solver5k_kernel_stats.txt
python profiles/summarize_rocpd.py $(db s20k) 10 > $O/dense_solver_kernel_stats.txt
python profiles/summarize_rocpd.py $(db dstep) > $O/dense_step_kernel_stats.txt
python profiles/summarize_rocpd.py $(db conn) > $O/conn20k_kernel_stats.txt
python profiles/summarize_rocpd.py $(db rawb) > $O/rawbatch_kernel_stats.txt
python - $(db conn) > $O/conn20k_timeline.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, stream_id from kernels order by start").fetchall()
first = [i for i, r in enumerate(rows) if r[0].startswith("void k2_minmax")]
a = first[-1]  # the last registration of tests/gpu_conn_diag.py: use_crosscheck = 0, third repetition
t0 = rows[a][1]
print("# one registration with use_crosscheck = 0, use_tuple_test = 0 (L ~ 20 k): start_us dur_us stream kernel")
for name, start, end, stream in rows[a:]:
    print(f"{(start - t0) / 1e3:9.1f} {(end - start) / 1e3:8.1f} s{stream:<4} {name[:80]}")
PY
python profiles/summarize_pmc.py $(db fetch) $(db write) "${NN_KERNEL:-void k_nn_f16}" > $O/pmc_nn.json
[ -f $R/quatro_amd/libquatro_hip_timing.so ] && timeout 120 python tests/probe/nn_stamps.py > $O/kernel_stamps.txt 2>&1
rm -rf $O/prof_*
ls -la $O
