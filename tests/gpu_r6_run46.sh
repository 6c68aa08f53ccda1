R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6x
timeout 400 python -m pytest tests -q -m gpu > gpurun_out/r6x/gpu_tests_full.txt 2>&1; grep -E "passed|failed" gpurun_out/r6x/gpu_tests_full.txt | tail -1 | tee gpurun_out/r6x/gpu_tests.txt
