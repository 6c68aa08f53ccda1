R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6t
timeout 900 python tests/gpu_fuzz.py 72 640 > gpurun_out/r6t/fuzz72_full.txt 2> gpurun_out/r6t/fuzz72_err.txt
grep -n "MISMATCH" -A12 gpurun_out/r6t/fuzz72_full.txt | cut -c1-400 | head -80; tail -1 gpurun_out/r6t/fuzz72_full.txt
grep -c "" gpurun_out/r6t/fuzz72_err.txt
