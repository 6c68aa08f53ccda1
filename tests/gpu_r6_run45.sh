R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6x
timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "degenerate or core_numbers_when or tiny_pair or max_clique_entry" 2>&1 | grep -E "passed|failed|rror|assert|^E " | tail -12 | tee gpurun_out/r6x/degenerate.txt
