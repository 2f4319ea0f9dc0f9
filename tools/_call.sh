cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c22
timeout 1200 python -m pytest tests/test_gpu_k2.py -q -m gpu -x -k "split_rank or config4 or full_size or 50" > gpurun_out/c22/k2.log 2>&1; tail -4 gpurun_out/c22/k2.log
(for m in "N=50000" "N=50000 D2G_BS_RANK_REGS=0" "N=30000" "N=30000 D2G_BS_RANK_REGS=0"; do echo -n "$m: "; env $m timeout 300 python tools/k2_time.py 2>&1 | grep step | cut -c1-60; done) | tee gpurun_out/c22/times.txt
N=50000 tools/kstats.sh c22_50k python $GRAFT_REPO_ROOT/tools/k2_time.py > /dev/null 2>&1; grep "rank" gpurun_out/c22_50k_kernel_stats.txt
N=30000 tools/kstats.sh c22_30k python $GRAFT_REPO_ROOT/tools/k2_time.py > /dev/null 2>&1; grep "rank" gpurun_out/c22_30k_kernel_stats.txt
N=30000 D2G_BS_RANK_REGS=0 tools/kstats.sh c22_30k_old python $GRAFT_REPO_ROOT/tools/k2_time.py > /dev/null 2>&1; grep "rank" gpurun_out/c22_30k_old_kernel_stats.txt
