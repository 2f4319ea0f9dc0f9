cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c37
for gm in 4 2 1; do
  echo "== grid_mult $gm"
  D2G_SP_GRID_MULT=$gm tools/kstats.sh c37_g$gm python $GRAFT_REPO_ROOT/tools/k2_time.py > /dev/null 2>&1; grep "sparse_k" gpurun_out/c37_g${gm}_kernel_stats.txt
  N=50000 D2G_SP_GRID_MULT=$gm tools/kstats.sh c37_50k_g$gm python $GRAFT_REPO_ROOT/tools/k2_time.py > /dev/null 2>&1; grep "sparse_k" gpurun_out/c37_50k_g${gm}_kernel_stats.txt
  MATRIX=noise C=1 D2G_SP_GRID_MULT=$gm tools/kstats.sh c37_c1_g$gm python $GRAFT_REPO_ROOT/tools/k2_time.py > /dev/null 2>&1; grep "sparse_k" gpurun_out/c37_c1_g${gm}_kernel_stats.txt
done
