cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_k2.py -x -q -m gpu 2>&1 | grep -a -v "^Extension modules\|^  File" | tail -2
for n in 24000 30000 50000 70000 100000; do echo -n "N=$n "; N=$n MATRIX=stated STEPS=5 timeout 300 python3 tools/k2_time.py 2>/dev/null | grep "^step" | cut -c1-50; done
N=100000 MATRIX=stated STEPS=3 timeout 300 tools/kstats.sh p21_100k python3 $R/tools/k2_time.py > /dev/null 2>&1; grep -E "bs_rank|sp_side|k2_transpose|k2_bitslice" gpurun_out/p21_100k_kernel_stats.txt | cut -c1-40,93-150
