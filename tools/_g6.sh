cd $GRAFT_REPO_ROOT
D2G_BS_SPARSE_MIN_N=1 timeout 1200 python -m pytest tests/test_gpu_k2.py -x -q 2>&1 | tail -3
MATRIX=stated tools/kstats.sh r04_x python $GRAFT_REPO_ROOT/tools/k2_time.py > /dev/null 2>&1
echo "c3"; grep -h "step" /tmp/ks_r04_x.out | cut -c1-60; head -22 gpurun_out/r04_x_kernel_stats.txt | cut -c1-60,92-150
N=50000 MATRIX=stated tools/kstats.sh r04_x python $GRAFT_REPO_ROOT/tools/k2_time.py > /dev/null 2>&1
echo "c4"; grep -h "step" /tmp/ks_r04_x.out | cut -c1-60; head -12 gpurun_out/r04_x_kernel_stats.txt | cut -c1-60,92-150
