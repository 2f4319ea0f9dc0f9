#!/bin/bash
cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
for v in "" nooo ch8 ch64; do
for c in 1 10; do
  L=""; if [ -n "$v" ]; then L="D2G_LIB=$R/dashing2_amd/libd2g_$v.so"; fi
  env $L MATRIX=noise C=$c STEPS=20 bash tools/kstats.sh probe8_${v}_c$c python3 $R/tools/k2_time.py > gpurun_out/probe8_${v}_c$c.log 2>&1
  echo "== variant '$v' c=$c: $(grep sp_pairs gpurun_out/probe8_${v}_c${c}_kernel_stats.txt | cut -c1-140)"
done; done
