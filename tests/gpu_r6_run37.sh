R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6u
FUZZ_DUMP_DIR=$R/gpurun_out/r6u timeout 900 python tests/gpu_fuzz.py 72 640 2>/dev/null | grep "MISMATCH\|fuzz seed" | cut -c1-300
ls -la gpurun_out/r6u
