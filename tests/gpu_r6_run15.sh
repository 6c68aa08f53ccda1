R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6n
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r6n/gpu_tests_full.txt 2>&1; grep -E "passed|failed|rror" gpurun_out/r6n/gpu_tests_full.txt | tail -3
bash tests/gpu_r6_ab.sh libquatro_hip_prev.so libquatro_hip.so > gpurun_out/r6n/ab_cells2.txt 2>&1; cat gpurun_out/r6n/ab_cells2.txt
for lib in libquatro_hip_prev.so libquatro_hip.so; do QTR_LIB=$PWD/quatro_amd/$lib timeout 200 python tests/gpu_dense_step_prof.py 6 2>&1 | grep "ms per" | sed "s/^/$lib /"; done | tee gpurun_out/r6n/dense_ab.txt
for lib in libquatro_hip_prev.so libquatro_hip.so; do QTR_LIB=$PWD/quatro_amd/$lib timeout 200 python tests/gpu_solver_prof.py 5000 40 2>&1 | tail -1 | sed "s/^/$lib /"; done | tee gpurun_out/r6n/solver_ab.txt
timeout 700 python tests/gpu_fuzz.py 68 400 2>&1 | tail -2 | tee gpurun_out/r6n/fuzz.txt
