R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6v
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "core_numbers_when or tiny_pair" 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
QTR_LIB=$R/quatro_amd/libquatro_hip_prev.so timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "core_numbers_when or tiny_pair" 2>&1 | grep -E "passed|failed" | sed 's/^/previous library: /'
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r6v/gpu_tests_full.txt 2>&1; grep -E "passed|failed|rror" gpurun_out/r6v/gpu_tests_full.txt | tail -3
for lib in libquatro_hip_prev.so libquatro_hip.so; do for r in 1 2; do QTR_LIB=$PWD/quatro_amd/$lib timeout 200 python tests/gpu_solver_prof.py 5000 40 2>&1 | tail -1 | sed "s/^/$lib /"; done; done | tee gpurun_out/r6v/solver_ab.txt | cut -c1-200
timeout 800 python tests/gpu_fuzz.py 72 600 2>/dev/null | grep "MISMATCH\|fuzz seed" | tee gpurun_out/r6v/fuzz72.txt
