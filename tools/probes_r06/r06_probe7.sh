#!/bin/bash
cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_k2.py -x -q -m gpu -k "sparse or fill or bench_matrix" > gpurun_out/probe7_tests.log 2>&1
tail -3 gpurun_out/probe7_tests.log
for c in 1 10; do
  MATRIX=noise C=$c STEPS=20 timeout 200 bash tools/kstats.sh probe7_c$c python3 $R/tools/k2_time.py > gpurun_out/probe7_c$c.log 2>&1
  echo "== c=$c"; grep "^step" /tmp/ks_probe7_c$c.out | cut -c1-60; head -12 gpurun_out/probe7_c${c}_kernel_stats.txt | cut -c1-140
done
for c in 0 1 3 10; do
  if [ $c = 0 ]; then M=stated; else M=noise; fi
  echo "== no profiler c=$c"; MATRIX=$M C=$c STEPS=50 timeout 120 python3 tools/k2_time.py 2>&1 | grep "^step" | cut -c1-60
  echo "== no profiler c=$c form 1"; D2G_SP_LIST_FORM=1 MATRIX=$M C=$c STEPS=50 timeout 120 python3 tools/k2_time.py 2>&1 | grep "^step" | cut -c1-60
done
echo "== paired"; MATRIX=paired STEPS=50 timeout 120 python3 tools/k2_time.py 2>&1 | grep "^step" | cut -c1-60
