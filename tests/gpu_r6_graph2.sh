R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6j
T=$R/quatro_amd/libquatro_hip_testengines.so
echo "== mfma: check"; QTR_LIB=$T QTR_GRAPH=mfma timeout 600 python tests/gpu_graph_bench.py check 2>&1 | grep -E "MISMATCH|!=|graph check|Error|error" | head -20
for g in mfma strips mfma strips; do echo "== $g: time"; QTR_LIB=$T QTR_GRAPH=$g timeout 300 python tests/gpu_graph_bench.py time 2>&1 | grep "graph stage"; done
export TMPDIR=/tmp; cd /tmp
for g in mfma strips; do
  QTR_LIB=$T QTR_GRAPH=$g timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY -d $R/gpurun_out/r6j/prof_$g -o p -- python $R/tests/gpu_solver_prof.py 20000 4 > /dev/null 2>&1
  python - $R/gpurun_out/r6j/prof_$g $g <<'PY'
import glob, sqlite3, sys
for f in glob.glob(sys.argv[1] + "/*.db"):
    c = sqlite3.connect(f)
    rows = c.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like 'void k_graph_build%' group by counter_name").fetchall()
    v = {n: a for n, k, a in rows}
    L = 20000
    preds = L * (L + 64) / 2.0
    print(sys.argv[2], {k: round(x) for k, x in v.items()})
    if "SQ_INSTS_VALU" in v:
        print(sys.argv[2], "vector instructions per 64 predicates: %.2f; per wave %.0f; VALU busy %.2f; parked %.2f" % (
            v["SQ_INSTS_VALU"] / (preds / 64.0), v["SQ_INSTS_VALU"] / v["SQ_WAVES"], 4 * v["SQ_ACTIVE_INST_VALU"] / (1024 * v["SQ_BUSY_CYCLES"] / 32),
            v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]))
PY
  rm -rf $R/gpurun_out/r6j/prof_$g
done
