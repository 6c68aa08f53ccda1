R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6s; export TMPDIR=/tmp
cd /tmp
for m in none 1; do
  mm=$m; [ $m = none ] && mm=""
  QTR_DENSE_PREALLOC=$mm timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r6s/prof_$m -o d -- python $R/tests/gpu_dense_step_prof.py 6 > /dev/null 2>&1
  python $R/profiles/summarize_rocpd.py $(ls $R/gpurun_out/r6s/prof_$m/*.db | head -1) > $R/gpurun_out/r6s/dense_kernels_prealloc_$m.txt
  rm -rf $R/gpurun_out/r6s/prof_$m
done
cd $R
python - <<'PY'
import re
def load(f):
    d = {}
    for line in open(f):
        m = re.match(r'(?:void )?(\w+).*?\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)%', line)
        if m: d[m.group(1)] = (int(m.group(2)), float(m.group(4)))
    return d
a, b = load('gpurun_out/r6s/dense_kernels_prealloc_none.txt'), load('gpurun_out/r6s/dense_kernels_prealloc_1.txt')
for k in sorted(a, key=lambda k: -abs(b.get(k, (0, 0))[1] * b.get(k, (0, 0))[0] - a[k][1] * a[k][0]))[:14]:
    print(f"{k:28s} calls {a[k][0]:4d}  avg_us alone {a[k][1]:9.2f}  after another handle {b.get(k, (0, 0))[1]:9.2f}")
PY
