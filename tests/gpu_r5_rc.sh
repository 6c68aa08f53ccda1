#!/bin/bash
# usage (GPU box): tests/gpu_r5_rc.sh OUTDIR — matcher + dense tests, dense diagnostics (base vs current), A/B, kernel trace
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r5}
mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "nn or match or f16 or adversarial or golden or register or dense or feature_pair or without_cross" > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
for lib in libquatro_hip_base.so libquatro_hip.so; do
  QTR_LIB=$R/quatro_amd/$lib timeout 120 python tests/gpu_dense_diag.py scene 2>&1 | tail -1
  QTR_LIB=$R/quatro_amd/$lib timeout 120 python tests/gpu_dense_diag.py planes 2>&1 | tail -1
done > $O/dense_diag.txt 2>&1
cat $O/dense_diag.txt
bash tests/gpu_ab_lib.sh $R/quatro_amd/libquatro_hip_base.so $R/quatro_amd/libquatro_hip.so 2 > $O/ab.txt 2>&1; cat $O/ab.txt
export TMPDIR=/tmp; cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats -d $O/prof_seq -o seq -- python $R/bench.py --steps 40 --warmup 5 --legs "" --cpu-seconds 0 > /dev/null 2>&1
cd $R
python profiles/summarize_rocpd.py $(ls $O/prof_seq/*.db | head -1) > $O/kernel_stats.txt
rm -rf $O/prof_seq
grep "k_nn\|k_recheck\|total kernel" $O/kernel_stats.txt
