R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6i
bash tests/gpu_r6_graph.sh > gpurun_out/r6i/graph_mfma.txt 2>&1; cat gpurun_out/r6i/graph_mfma.txt
bash tests/gpu_r6_ab.sh libquatro_hip_prev.so libquatro_hip.so > gpurun_out/r6i/ab_recheck5.txt 2>&1; cat gpurun_out/r6i/ab_recheck5.txt
for lib in libquatro_hip_prev.so libquatro_hip.so; do QTR_LIB=$PWD/quatro_amd/$lib timeout 200 python tests/gpu_dense_step_prof.py 6 2>&1 | grep "ms per" | sed "s/^/$lib /"; done | tee gpurun_out/r6i/dense_ab.txt
export TMPDIR=/tmp; cd /tmp
for lib in libquatro_hip.so; do
  QTR_LIB=$R/quatro_amd/$lib timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r6i/prof_$lib -o t -- python $R/bench.py --steps 40 --warmup 5 --legs "" --cpu-seconds 0 > /dev/null 2>&1
  python $R/profiles/summarize_rocpd.py $(ls $R/gpurun_out/r6i/prof_$lib/*.db | head -1) auto > $R/gpurun_out/r6i/seq_stats_$lib.txt
  rm -rf $R/gpurun_out/r6i/prof_$lib
  grep -E "k_recheck_filter|total kernel" $R/gpurun_out/r6i/seq_stats_$lib.txt | cut -c1-140
done
cd $R
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r6i/gpu_tests_full.txt 2>&1; grep -E "passed|failed|rror" gpurun_out/r6i/gpu_tests_full.txt | tail -3
timeout 300 python tests/gpu_fuzz.py 64 120 2>&1 | tail -2 | tee gpurun_out/r6i/fuzz.txt
