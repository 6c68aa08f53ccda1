R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6p
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "match" > gpurun_out/r6p/match_tests.txt 2>&1; grep -E "passed|failed|rror|assert" gpurun_out/r6p/match_tests.txt | tail -8

timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r6p/gpu_tests_full.txt 2>&1; grep -E "passed|failed|rror" gpurun_out/r6p/gpu_tests_full.txt | tail -3
