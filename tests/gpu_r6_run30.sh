R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6s
for m in "" 1 run; do
QTR_DENSE_PREALLOC=$m QTR_DENSE_STAGES=1 timeout 200 python tests/gpu_dense_step_prof.py 10 2>&1 | grep "ms per\|stages" | head -2 | sed "s/^/prealloc=$m /"
done | cut -c1-250 | tee gpurun_out/r6s/dense_prealloc.txt
