#!/bin/bash
# SQ-level counters of ONE kernel (default k_graph_build) at L = $2 (default 20000): where its wave-cycles go.
#   gpurun -- 'bash profiles/pmc_sq.sh TAG [L] [kernel-name-prefix]'   ->  gpurun_out/TAG_sq.txt
TAG=${1:-sq}
L=${2:-20000}
KERN=${3:-void k_graph_build}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $R/gpurun_out/prof_${TAG}_a -o a -- python $R/tests/gpu_solver_prof.py $L 4 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -d $R/gpurun_out/prof_${TAG}_b -o b -- python $R/tests/gpu_solver_prof.py $L 4 > /dev/null 2>&1
cd $R
python - "$KERN" gpurun_out/prof_${TAG}_a gpurun_out/prof_${TAG}_b > gpurun_out/${TAG}_sq.txt <<'PY'
import glob, sqlite3, sys
kern = sys.argv[1] + "%"
for d in sys.argv[2:]:
    for f in sorted(glob.glob(d + "/*.db")):
        c = sqlite3.connect(f)
        try:
            rows = c.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like ? "
                             "group by counter_name", (kern,)).fetchall()
        except Exception as e:
            print(f, "query failed:", e)
            continue
        for name, n, v in rows:
            print(f"{name:28s} launches {n:3d}  mean {v:16.1f}")
PY
cat gpurun_out/${TAG}_sq.txt
rm -rf gpurun_out/prof_${TAG}_a gpurun_out/prof_${TAG}_b
