R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6v
FUZZ_DUMP_DIR=$R/gpurun_out/r6v FUZZ_KINDS=batch timeout 700 python tests/gpu_fuzz.py 80 500 2>/dev/null | grep "MISMATCH\|fuzz seed" | cut -c1-300 | tee gpurun_out/r6v/fuzz_batch80.txt
FUZZ_KINDS=clique,scout,solve timeout 500 python tests/gpu_fuzz.py 81 300 2>/dev/null | grep "MISMATCH\|fuzz seed" | cut -c1-300 | tee gpurun_out/r6v/fuzz_solver81.txt
