cd $GRAFT_REPO_ROOT
for v in u8a u16r u8r; do
N=50000 D2G_LIB=$GRAFT_REPO_ROOT/dashing2_amd/libd2g_$v.so MATRIX=stated tools/kstats.sh r04_x python $GRAFT_REPO_ROOT/tools/k2_time.py > /dev/null 2>&1
echo "$v"; grep -h "step" /tmp/ks_r04_x.out | cut -c1-60; grep sp_mark gpurun_out/r04_x_kernel_stats.txt | cut -c1-60,92-150
done
