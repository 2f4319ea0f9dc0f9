cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c38
timeout 1200 python -m pytest tests/test_gpu_k2.py tests/test_gpu_mgpu.py -q -m gpu -x > gpurun_out/c38/k2.log 2>&1; grep "passed\|failed" gpurun_out/c38/k2.log
D2G_LIB=$GRAFT_REPO_ROOT/dashing2_amd/libd2g_trace.so python tools/sp_trace.py 2>&1 | grep -v amdgpu.ids | head -9 | tee gpurun_out/c38/trace_10k.txt
(for m in "" "MATRIX=noise C=1" "N=50000"; do echo -n "$m: "; env $m timeout 300 python tools/k2_time.py 2>&1 | grep step | cut -c1-60; done) | tee gpurun_out/c38/times.txt
tools/kstats.sh c38 python $GRAFT_REPO_ROOT/tools/k2_time.py > /dev/null 2>&1; grep "sparse_k" gpurun_out/c38_kernel_stats.txt
N=50000 tools/kstats.sh c38_50k python $GRAFT_REPO_ROOT/tools/k2_time.py > /dev/null 2>&1; grep "sparse_k" gpurun_out/c38_50k_kernel_stats.txt
