#!/bin/bash
# usage (GPU box): tests/gpu_r5_full.sh OUTDIR — the whole GPU suite, the connected configurations' diagnostics, A/B, trace
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r5}
mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q ${2:+-k "$2"} > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log; tail -8 $O/pytest.log
timeout 200 python tests/gpu_conn_diag.py > $O/conn_diag.txt 2>&1; tail -3 $O/conn_diag.txt
bash tests/gpu_ab_lib.sh $R/quatro_amd/libquatro_hip_base.so $R/quatro_amd/libquatro_hip.so 2 > $O/ab.txt 2>&1; cat $O/ab.txt
export TMPDIR=/tmp; cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats -d $O/prof_seq -o seq -- python $R/bench.py --steps 40 --warmup 5 --legs "" --cpu-seconds 0 > /dev/null 2>&1
timeout 240 rocprofv3 --kernel-trace --stats -d $O/prof_conn -o conn -- python $R/tests/gpu_conn_diag.py > /dev/null 2>&1
cd $R
python profiles/summarize_rocpd.py $(ls $O/prof_seq/*.db | head -1) > $O/kernel_stats.txt
python profiles/timeline.py $(ls $O/prof_seq/*.db | head -1) 30 > $O/timeline.txt
python profiles/summarize_rocpd.py $(ls $O/prof_conn/*.db | head -1) > $O/conn_kernel_stats.txt
python - $(ls $O/prof_conn/*.db | head -1) > $O/conn_timeline.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, stream_id from kernels order by start").fetchall()
first = [i for i, r in enumerate(rows) if r[0].startswith("void k2_minmax")]
a = first[-1]  # the last registration of the run: no_cross, third repetition
t0 = rows[a][1]
for name, start, end, stream in rows[a:]:
    print(f"{(start - t0) / 1e3:9.1f} {(end - start) / 1e3:8.1f} s{stream:<4} {name[:80]}")
PY
rm -rf $O/prof_seq $O/prof_conn
grep "k_nn\|k_recheck\|total kernel" $O/kernel_stats.txt
head -12 $O/conn_kernel_stats.txt
