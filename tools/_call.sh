cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c24
timeout 1200 python -m pytest tests/test_gpu_k2.py tests/test_gpu_mgpu.py -q -m gpu -x > gpurun_out/c24/k2.log 2>&1; tail -4 gpurun_out/c24/k2.log
(for m in "N=50000" "N=30000"; do echo -n "$m: "; env $m timeout 300 python tools/k2_time.py 2>&1 | grep step | cut -c1-60; done) | tee gpurun_out/c24/times.txt
N=50000 tools/kstats.sh c24_50k python $GRAFT_REPO_ROOT/tools/k2_time.py > /dev/null 2>&1; grep "rank" gpurun_out/c24_50k_kernel_stats.txt
N=30000 tools/kstats.sh c24_30k python $GRAFT_REPO_ROOT/tools/k2_time.py > /dev/null 2>&1; grep "rank" gpurun_out/c24_30k_kernel_stats.txt
