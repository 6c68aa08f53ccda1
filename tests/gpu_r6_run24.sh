R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6q; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r6q/parity.txt 2>&1; grep -E "passed|failed|rror" gpurun_out/r6q/parity.txt | tail -3
bash tests/gpu_r6_ab2.sh libquatro_hip_prev.so libquatro_hip.so > gpurun_out/r6q/ab_half.txt 2>&1; cut -c1-330 gpurun_out/r6q/ab_half.txt
cd /tmp
for lib in libquatro_hip_prev.so libquatro_hip.so; do
  QTR_LIB=$R/quatro_amd/$lib timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r6q/prof_$lib -o seq -- python $R/bench.py --steps 40 --warmup 5 --legs "" --cpu-seconds 0 > /dev/null 2>&1
  python $R/profiles/timeline.py $(ls $R/gpurun_out/r6q/prof_$lib/*.db | head -1) 30 > $R/gpurun_out/r6q/timeline_$lib.txt
  rm -rf $R/gpurun_out/r6q/prof_$lib
done
cd $R; paste -d'|' <(cut -c1-75 gpurun_out/r6q/timeline_libquatro_hip_prev.so.txt) <(cut -c10-24 gpurun_out/r6q/timeline_libquatro_hip.so.txt)
