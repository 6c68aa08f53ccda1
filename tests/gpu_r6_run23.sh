R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6q
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r6q/parity.txt 2>&1; grep -E "passed|failed|rror" gpurun_out/r6q/parity.txt | tail -3
bash tests/gpu_r6_ab2.sh libquatro_hip_prev.so libquatro_hip.so > gpurun_out/r6q/ab_normals.txt 2>&1; cut -c1-330 gpurun_out/r6q/ab_normals.txt
