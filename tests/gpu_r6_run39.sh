R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6u
timeout 120 python tests/gpu_repro_batch_tie.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6u/repro2.txt | cut -c1-700
