R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6m
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r6m/gpu_tests_full.txt 2>&1; grep -E "passed|failed|rror" gpurun_out/r6m/gpu_tests_full.txt | tail -3
bash tests/gpu_r6_ab.sh libquatro_hip_prev.so libquatro_hip.so > gpurun_out/r6m/ab_cells.txt 2>&1; cat gpurun_out/r6m/ab_cells.txt
for lib in libquatro_hip_prev.so libquatro_hip.so; do QTR_LIB=$PWD/quatro_amd/$lib timeout 200 python tests/gpu_dense_step_prof.py 6 2>&1 | grep "ms per" | sed "s/^/$lib /"; done | tee gpurun_out/r6m/dense_ab.txt
for lib in libquatro_hip_prev.so libquatro_hip.so; do QTR_LIB=$PWD/quatro_amd/$lib timeout 200 python tests/gpu_l5k_prof.py 4 2>&1 | grep "ms per" | sed "s/^/$lib /"; done | tee gpurun_out/r6m/l5k_ab.txt
timeout 500 python tests/gpu_fuzz.py 67 300 2>&1 | tail -2 | tee gpurun_out/r6m/fuzz.txt
