#!/bin/bash
cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_k2.py -x -q -m gpu -k "sparse or fill or bench_matrix" > gpurun_out/probe9_tests.log 2>&1
tail -3 gpurun_out/probe9_tests.log
for w in 0; do
  D2G_SP_BIN_WGS=$w MATRIX=noise C=10 STEPS=20 timeout 200 bash tools/kstats.sh probe9_w$w python3 $R/tools/k2_time.py > gpurun_out/probe9_w$w.log 2>&1
  echo "== bin wgs $w"; grep "^step" /tmp/ks_probe9_w$w.out | cut -c1-60; grep "sp_bin\|sp_permute\|sp_compose\|sp_pairs\|sp_emit" gpurun_out/probe9_w${w}_kernel_stats.txt | cut -c1-140
done
for c in 0 1 3 10; do
  if [ $c = 0 ]; then M=stated; else M=noise; fi
  echo "== no profiler c=$c"; MATRIX=$M C=$c STEPS=50 timeout 120 python3 tools/k2_time.py 2>&1 | grep "^step" | cut -c1-60
done
