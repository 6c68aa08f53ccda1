# usage (GPU box): where does the process-to-process spread of the headline step come from?  The same library under different
# CPU affinities (the GPU's NUMA-local cores against the others), the voxel stage's time beside the step.
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6o
{
nproc; lscpu | grep -E "NUMA|Model name|Socket|Thread"
for d in /sys/class/drm/card*/device; do echo "$d numa=$(cat $d/numa_node 2>/dev/null) cpus=$(cat $d/local_cpulist 2>/dev/null)"; done
echo "allowed: $(python -c 'import os; print(sorted(os.sched_getaffinity(0)))' | cut -c1-300)"
run() { # label, command prefix
  lab=$1; shift
  for r in 1 2 3; do
    "$@" python bench.py --steps 400 --cpu-seconds 0 --legs "" > /tmp/b.json 2>/tmp/b.err
    python - "$lab" <<'PY'
import json, sys
try:
    d = json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
    rr = sorted([d["ms_per_step"]] + d["repeat_regions"]["ms_per_step"])
    st = d.get("stage_ms", {})
    print(f"{sys.argv[1]:28s} step median {rr[2]:.4f} min {rr[0]:.4f} max {rr[-1]:.4f} | vox {st.get('voxelize'):.4f} fpfh {st.get('fpfh'):.4f} match {st.get('match'):.4f}")
except Exception as e:
    print(sys.argv[1], "FAILED", e, open('/tmp/b.err').read()[-300:])
PY
  done
}
run "free" env
CPUS=$(python - <<'PY'
import os
a = sorted(os.sched_getaffinity(0))
print(a[0], a[len(a)//4], a[len(a)//2], a[-1])
PY
)
for c in $CPUS; do run "taskset -c $c" taskset -c $c; done
} 2>&1 | tee gpurun_out/r6o/affinity.txt
