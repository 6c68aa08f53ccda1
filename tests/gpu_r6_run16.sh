R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6o
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r6o/gpu_tests_full.txt 2>&1; grep -E "passed|failed|rror" gpurun_out/r6o/gpu_tests_full.txt | tail -3
bash tests/gpu_r6_ab2.sh libquatro_hip_prev.so libquatro_hip_em.so libquatro_hip_emb.so > gpurun_out/r6o/ab_mean.txt 2>&1; cat gpurun_out/r6o/ab_mean.txt
for lib in libquatro_hip_prev.so libquatro_hip_em.so; do QTR_LIB=$PWD/quatro_amd/$lib timeout 200 python tests/gpu_dense_step_prof.py 6 2>&1 | grep "ms per" | sed "s/^/$lib /"; done | tee gpurun_out/r6o/dense_ab.txt
