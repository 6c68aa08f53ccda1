R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6x
FUZZ_DUMP_DIR=$R/gpurun_out/r6x timeout 560 python tests/gpu_fuzz.py 74 460 2>/dev/null | grep "MISMATCH\|fuzz seed" | cut -c1-300 | tee gpurun_out/r6x/fuzz74.txt
