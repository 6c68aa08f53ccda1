R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6t
for s in 71 72; do timeout 800 python tests/gpu_fuzz.py $s 600 2>&1 | tail -1; done | tee gpurun_out/r6t/fuzz_final.txt
