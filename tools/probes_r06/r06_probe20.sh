cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
for v in sfq1 sfq2 sfq3 sfq4 sfq6; do
  lib=""; [ -n "$v" ] && lib=$R/dashing2_amd/libd2g_$v.so
  echo "== side fill workgroups per 4 CUs [$v]"
  D2G_LIB=$lib N=50000 MATRIX=stated STEPS=8 timeout 300 python3 tools/k2_time.py 2>/dev/null | grep "^step" | cut -c1-50
  D2G_LIB=$lib N=50000 MATRIX=stated STEPS=5 timeout 300 tools/kstats.sh p20_$v python3 $R/tools/k2_time.py > /dev/null 2>&1; grep -E "bs_rank|sp_side|k2_transpose" gpurun_out/p20_${v}_kernel_stats.txt | cut -c1-40,93-150
  D2G_LIB=$lib N=30000 MATRIX=stated STEPS=8 timeout 300 python3 tools/k2_time.py 2>/dev/null | grep "^step" | cut -c1-50
done
