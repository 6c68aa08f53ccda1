R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6h
bash tests/gpu_r6_ab.sh libquatro_hip_prev.so libquatro_hip.so > gpurun_out/r6h/ab_recheck4.txt 2>&1; cat gpurun_out/r6h/ab_recheck4.txt
bash tests/gpu_r6_nnlds.sh > gpurun_out/r6h/ab_nnlds.txt 2>&1; cat gpurun_out/r6h/ab_nnlds.txt
export TMPDIR=/tmp; cd /tmp
for lib in libquatro_hip_prev.so libquatro_hip.so; do
  QTR_LIB=$R/quatro_amd/$lib timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r6h/prof_$lib -o t -- python $R/bench.py --steps 40 --warmup 5 --legs "" --cpu-seconds 0 > /dev/null 2>&1
  python $R/profiles/summarize_rocpd.py $(ls $R/gpurun_out/r6h/prof_$lib/*.db | head -1) auto > $R/gpurun_out/r6h/seq_stats_$lib.txt
  rm -rf $R/gpurun_out/r6h/prof_$lib
  grep -E "k_recheck_filter|k2_ranges|k2_spfh|k2_neighbors|k2_fpfh|total kernel" $R/gpurun_out/r6h/seq_stats_$lib.txt | cut -c1-140
done
cd $R
QTR_LIB=$R/quatro_amd/libquatro_hip.so timeout 600 python -m pytest tests -x -q -m gpu -k "match or dense or nn or f16 or batch or adversarial or fpfh or pair" 2>&1 | tail -2
