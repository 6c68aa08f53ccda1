R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6t
timeout 600 python tests/gpu_lane_queues.py 2>&1 | grep "streams alive" | tee gpurun_out/r6t/lane_queues.txt
