R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6p; export TMPDIR=/tmp
timeout 300 python tests/gpu_nocross_diag.py 0 1 2 3 2>&1 | grep "^pair" | tee gpurun_out/r6p/nocross_pairs.txt
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r6p/prof_nc2 -o nc2 -- python $R/tests/gpu_nocross_diag.py 2 > /dev/null 2>&1; cd $R
python profiles/summarize_rocpd.py $(ls gpurun_out/r6p/prof_nc2/*.db | head -1) > gpurun_out/r6p/nocross_pair2_kernel_stats.txt; rm -rf gpurun_out/r6p/prof_nc2
head -24 gpurun_out/r6p/nocross_pair2_kernel_stats.txt | cut -c1-160
