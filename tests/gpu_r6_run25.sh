R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6r; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r6r/gpu_tests_full.txt 2>&1; grep -E "passed|failed|rror" gpurun_out/r6r/gpu_tests_full.txt | tail -3
bash tests/gpu_r6_ab2.sh libquatro_hip_prev.so libquatro_hip.so > gpurun_out/r6r/ab_ahead.txt 2>&1; cut -c1-330 gpurun_out/r6r/ab_ahead.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r6r/prof -o seq -- python $R/bench.py --steps 40 --warmup 5 --legs "" --cpu-seconds 0 > /dev/null 2>&1
python $R/profiles/timeline.py $(ls $R/gpurun_out/r6r/prof/*.db | head -1) 30 > $R/gpurun_out/r6r/timeline.txt; rm -rf $R/gpurun_out/r6r/prof
cd $R; cut -c1-100 gpurun_out/r6r/timeline.txt | head -20
timeout 500 python tests/gpu_fuzz.py 69 300 2>&1 | tail -2 | tee gpurun_out/r6r/fuzz.txt
