R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6l
for lib in libquatro_hip_prev.so libquatro_hip.so; do QTR_LIB=$PWD/quatro_amd/$lib timeout 200 python tests/gpu_l5k_prof.py 4 2>&1 | grep "ms per" | sed "s/^/$lib /"; done | tee gpurun_out/r6l/l5k_ab.txt
for lib in libquatro_hip_prev.so libquatro_hip.so; do QTR_LIB=$PWD/quatro_amd/$lib timeout 200 python tests/gpu_conn_diag.py 2>&1 | tail -4 | sed "s/^/$lib /"; done | tee gpurun_out/r6l/conn_ab.txt
timeout 900 python -m pytest tests -x -q -m gpu -k "clique or connected or solve or second_run or scout or floor or l5k or 5k" 2>&1 | tail -2
timeout 300 python tests/gpu_fuzz.py 66 150 2>&1 | tail -2 | tee gpurun_out/r6l/fuzz.txt
