#!/bin/bash
# Matrix-pipe and issue counters of k_nn_f16 on the headline loop (two rocprofv3 --pmc passes, --kernel-trace only beside them):
#   gpurun -- 'bash profiles/pmc_mfma.sh TAG'   ->  gpurun_out/TAG_mfma.txt
# SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per v_mfma_f32_32x32x16_f16, per SIMD, summed over the device's 1024 SIMDs);
# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md, PMC section).  The kernel runs one
# wave per SIMD, so a wave's lifetime is its SIMD's time: matrix pipes busy = MFMA_BUSY / (4 x SQ_WAVE_CYCLES), the two
# counters from the two passes (profiles/r5_pmc_mfma.json holds the figures of the round and the derivation; the line the
# script itself prints from GRBM_GUI_ACTIVE prices the launch in that counter's clock and is NOT the figure quoted).
TAG=${1:-r5}
KERN=${2:-void k_nn_f16}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
CMD="python $R/bench.py --steps 8 --warmup 2 --legs  --cpu-seconds 0"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $R/gpurun_out/prof_${TAG}_m1 -o a -- python $R/bench.py --steps 8 --warmup 2 --legs "" --cpu-seconds 0 > $R/gpurun_out/${TAG}_mfma_run1.txt 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAVES GRBM_GUI_ACTIVE -d $R/gpurun_out/prof_${TAG}_m2 -o b -- python $R/bench.py --steps 8 --warmup 2 --legs "" --cpu-seconds 0 > $R/gpurun_out/${TAG}_mfma_run2.txt 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/prof_${TAG}_m3 -o c -- python $R/bench.py --steps 8 --warmup 2 --legs "" --cpu-seconds 0 > $R/gpurun_out/${TAG}_mfma_run3.txt 2>&1
cd $R
python - "$KERN" gpurun_out/prof_${TAG}_m1 gpurun_out/prof_${TAG}_m2 gpurun_out/prof_${TAG}_m3 > gpurun_out/${TAG}_mfma.txt <<'PY'
import glob, sqlite3, sys
kern = sys.argv[1] + "%"
print("# counters of", sys.argv[1], "(means per launch; both directions of the headline loop's pool)")
for d in sys.argv[2:]:
    for f in sorted(glob.glob(d + "/*.db")):
        c = sqlite3.connect(f)
        tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
        try:
            rows = c.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like ? "
                             "group by counter_name", (kern,)).fetchall()
        except Exception as e:
            print(f, "query failed:", e, tabs[:12])
            continue
        vals = {}
        for name, n, v in rows:
            vals[name] = v
            print(f"{d.split('_')[-1]} {name:28s} launches {n:3d}  mean {v:16.1f}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "GRBM_GUI_ACTIVE" in vals:
            print(f"{d.split('_')[-1]} matrix pipes busy: {vals['SQ_VALU_MFMA_BUSY_CYCLES'] / (128.0 * vals['GRBM_GUI_ACTIVE']):.3f} of the launch's SIMD-cycles")
PY
tail -3 gpurun_out/${TAG}_mfma_run1.txt | cut -c1-200
cat gpurun_out/${TAG}_mfma.txt
rm -rf gpurun_out/prof_${TAG}_m1 gpurun_out/prof_${TAG}_m2 gpurun_out/prof_${TAG}_m3
