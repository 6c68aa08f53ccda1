R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6x
FUZZ_DUMP_DIR=$R/gpurun_out/r6x timeout 600 python tests/gpu_fuzz.py 73 480 2>/dev/null | grep "MISMATCH\|fuzz seed" | cut -c1-300 | tee gpurun_out/r6x/fuzz73.txt
FUZZ_DUMP_DIR=$R/gpurun_out/r6x FUZZ_KINDS=batch timeout 400 python tests/gpu_fuzz.py 82 280 2>/dev/null | grep "MISMATCH\|fuzz seed" | cut -c1-300 | tee gpurun_out/r6x/fuzz82_batch.txt
FUZZ_KINDS=pair,match timeout 400 python tests/gpu_fuzz.py 83 280 2>/dev/null | grep "MISMATCH\|fuzz seed" | cut -c1-300 | tee gpurun_out/r6x/fuzz83_front.txt
