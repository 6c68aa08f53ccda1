R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6u
for lib in libquatro_hip.so libquatro_hip_a29.so libquatro_hip_r5.so; do QTR_LIB=$R/quatro_amd/$lib timeout 120 python tests/gpu_repro_batch_tie.py 2>&1 | grep "npz" | sed "s/^/$lib /"; done | tee gpurun_out/r6u/repro.txt
