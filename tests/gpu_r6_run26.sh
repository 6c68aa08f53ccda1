R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6r
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tests/probe/launch_cost.hip -o /tmp/launch_cost 2>&1 | grep -i error
timeout 60 /tmp/launch_cost | tee gpurun_out/r6r/launch_cost.txt
timeout 60 taskset -c 64 /tmp/launch_cost | sed 's/^/taskset64 /' | tee -a gpurun_out/r6r/launch_cost.txt
