cd $GRAFT_REPO_ROOT
D2G_BS_SPARSE_MIN_N=1 timeout 1500 python -m pytest tests/test_gpu_k2.py -x -q 2>&1 | tail -2
for m in stated unrelated paired skewed; do echo -n "N=10000 $m: "; MATRIX=$m python tools/k2_time.py 2>/dev/null | grep step | cut -c1-44; done
echo -n "N=50000 stated: "; N=50000 MATRIX=stated python tools/k2_time.py 2>/dev/null | grep step | cut -c1-44
MATRIX=stated tools/kstats.sh r04_x python $GRAFT_REPO_ROOT/tools/k2_time.py > /dev/null 2>&1; head -12 gpurun_out/r04_x_kernel_stats.txt | cut -c1-60,92-150
D2G_VERBOSE_EXIT=1 python tools/e2e_cli.py --genomes 10 --threads 16 --sketches 10000 --big-sketches 50000 2>&1 | grep "cmp binary:\|cmp 50000\|cmp: " | cut -c1-230
