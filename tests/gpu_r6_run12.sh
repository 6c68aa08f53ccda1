R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6k
timeout 600 python tests/gpu_graph_bench.py check 2>&1 | grep -E "MISMATCH|!=|graph check|rror" | head; 
bash tests/gpu_r6_ab.sh libquatro_hip_prev.so libquatro_hip.so > gpurun_out/r6k/ab_graph.txt 2>&1; cat gpurun_out/r6k/ab_graph.txt
for lib in libquatro_hip_prev.so libquatro_hip.so; do QTR_LIB=$PWD/quatro_amd/$lib timeout 200 python tests/gpu_dense_step_prof.py 6 2>&1 | grep "ms per" | sed "s/^/$lib /"; done | tee gpurun_out/r6k/dense_ab.txt
for lib in libquatro_hip_prev.so libquatro_hip.so; do QTR_LIB=$PWD/quatro_amd/$lib timeout 200 python tests/gpu_solver_prof.py 5000 40 2>&1 | tail -2 | sed "s/^/$lib /"; done | tee gpurun_out/r6k/solver_ab.txt
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r6k/gpu_tests_full.txt 2>&1; grep -E "passed|failed|rror" gpurun_out/r6k/gpu_tests_full.txt | tail -3
timeout 500 python tests/gpu_fuzz.py 65 300 2>&1 | tail -2 | tee gpurun_out/r6k/fuzz.txt
