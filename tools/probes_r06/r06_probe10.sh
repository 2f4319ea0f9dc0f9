#!/bin/bash
cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
for v in "" cmpnt; do
  L=""; if [ -n "$v" ]; then L="D2G_LIB=$R/dashing2_amd/libd2g_$v.so"; fi
  env $L MATRIX=noise C=10 STEPS=20 timeout 200 bash tools/kstats.sh probe10_$v python3 $R/tools/k2_time.py > gpurun_out/probe10_$v.log 2>&1
  echo "== variant '$v'"; grep "^step" /tmp/ks_probe10_$v.out | cut -c1-60; grep "sp_compose\|sparse_kernel\|transpose\|rank" gpurun_out/probe10_${v}_kernel_stats.txt | cut -c1-140
  echo "no profiler: $(env $L MATRIX=noise C=10 STEPS=50 timeout 120 python3 tools/k2_time.py 2>&1 | grep '^step' | cut -c1-50)"
done
MATRIX=stated STEPS=20 timeout 200 bash tools/kstats.sh probe10_clean python3 $R/tools/k2_time.py > gpurun_out/probe10_clean.log 2>&1
echo "== clean"; grep "^step" /tmp/ks_probe10_clean.out | cut -c1-60; head -20 gpurun_out/probe10_clean_kernel_stats.txt | cut -c1-140
