R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6q
bash tests/gpu_r6_ab2.sh libquatro_hip_prev.so libquatro_hip_t512.so libquatro_hip_t256.so libquatro_hip_t256h256.so > gpurun_out/r6q/ab_tile2.txt 2>&1; cut -c1-330 gpurun_out/r6q/ab_tile2.txt
for lib in libquatro_hip_t256.so libquatro_hip_t256h256.so; do QTR_LIB=$R/quatro_amd/$lib timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "voxel or register or fixture or extremes" 2>&1 | grep -E "passed|failed" | sed "s/^/$lib /"; done | tee gpurun_out/r6q/tile_parity.txt
