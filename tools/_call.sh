cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c28
timeout 1200 python -m pytest tests/test_gpu_k2.py -q -m gpu -x -k "split_rank or config4 or full_size or crowded or beyond_65535" > gpurun_out/c28/k2.log 2>&1; tail -5 gpurun_out/c28/k2.log
(for m in "N=50000" "N=50000 D2G_BS_RANK_BINS=0" "N=30000" "N=30000 D2G_BS_RANK_BINS=0"; do echo -n "$m: "; env $m timeout 300 python tools/k2_time.py 2>&1 | grep step | cut -c1-60; done) | tee gpurun_out/c28/times.txt
N=50000 tools/kstats.sh c28_50k python $GRAFT_REPO_ROOT/tools/k2_time.py > /dev/null 2>&1; grep "rank\|bs_bin" gpurun_out/c28_50k_kernel_stats.txt
N=30000 tools/kstats.sh c28_30k python $GRAFT_REPO_ROOT/tools/k2_time.py > /dev/null 2>&1; grep "rank\|bs_bin" gpurun_out/c28_30k_kernel_stats.txt
