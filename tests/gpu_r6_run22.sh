R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6q
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r6q/gpu_tests_full.txt 2>&1; grep -E "passed|failed|rror" gpurun_out/r6q/gpu_tests_full.txt | tail -3
bash tests/gpu_r6_ab2.sh libquatro_hip_prev.so libquatro_hip.so > gpurun_out/r6q/ab_tile3.txt 2>&1; cut -c1-330 gpurun_out/r6q/ab_tile3.txt
