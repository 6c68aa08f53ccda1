R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6x
FUZZ_DUMP_DIR=$R/gpurun_out/r6x FUZZ_KINDS=batch,clique timeout 260 python tests/gpu_fuzz.py 84 200 2>/dev/null | grep "MISMATCH\|fuzz seed" | cut -c1-300 | tee gpurun_out/r6x/fuzz84.txt
