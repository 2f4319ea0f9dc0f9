cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
echo "== forced sparse (MIN_N=1)"; D2G_BS_SPARSE_MIN_N=1 timeout 1200 python -m pytest tests/test_gpu_k2.py -x -q 2>&1 | tail -15
for m in stated unrelated paired skewed; do
  MATRIX=$m tools/kstats.sh r04_k2_$m python $GRAFT_REPO_ROOT/tools/k2_time.py > /dev/null 2>&1
  grep -h "step\|pair kernel" /tmp/ks_r04_k2_$m.out; head -12 gpurun_out/r04_k2_${m}_kernel_stats.txt | cut -c1-60,92-150
done
N=50000 MATRIX=stated tools/kstats.sh r04_k2_c4 python $GRAFT_REPO_ROOT/tools/k2_time.py > /dev/null 2>&1
grep -h "step\|pair kernel" /tmp/ks_r04_k2_c4.out; head -14 gpurun_out/r04_k2_c4_kernel_stats.txt | cut -c1-60,92-150
for m in stated unrelated paired skewed; do D2G_BS_SPARSE=0 MATRIX=$m python tools/k2_time.py 2>/dev/null | grep step; done
