cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/exp4_pytest.log 2>&1
grep -E "passed|failed|error" gpurun_out/exp4_pytest.log | tail -5
grep -E "^(FAILED|E  )" gpurun_out/exp4_pytest.log | head -20
timeout 300 python bench.py --steps 40 --warmup 5 --legs batch --cpu-seconds 0 > gpurun_out/exp4_bench.json 2> gpurun_out/exp4_bench.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_exp4 -o seq -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --legs "" --cpu-seconds 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python profiles/summarize_rocpd.py $(ls gpurun_out/prof_exp4/*.db | head -1) 53 > gpurun_out/exp4_kernel_stats.txt
