cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_exp7 -o kc -- python $GRAFT_REPO_ROOT/tests/probe/kc_probe.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python profiles/summarize_rocpd.py $(ls gpurun_out/prof_exp7/*.db | head -1) > gpurun_out/exp7_kernel_stats.txt
head -30 gpurun_out/exp7_kernel_stats.txt | cut -c1-130
