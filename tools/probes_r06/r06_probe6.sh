#!/bin/bash
cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_k2.py -x -q -m gpu -k "sparse or fill or bench_matrix or config4 or full_size" > gpurun_out/probe6_tests.log 2>&1
tail -5 gpurun_out/probe6_tests.log
D2G_SP_LIST_FORM=1 timeout 1500 python -m pytest tests/test_gpu_k2.py -x -q -m gpu -k "sparse or fill or bench_matrix" > gpurun_out/probe6_tests_form1.log 2>&1
tail -3 gpurun_out/probe6_tests_form1.log
D2G_SP_LIST_FORM=2 timeout 1500 python -m pytest tests/test_gpu_k2.py -x -q -m gpu -k "sparse or fill or bench_matrix" > gpurun_out/probe6_tests_form2.log 2>&1
tail -3 gpurun_out/probe6_tests_form2.log
for c in 0 1 3 10; do
  if [ $c = 0 ]; then M=stated; else M=noise; fi
  MATRIX=$M C=$c STEPS=20 bash tools/kstats.sh probe6_c$c python3 $R/tools/k2_time.py > gpurun_out/probe6_c$c.log 2>&1
  echo "== c=$c"; grep "^step" /tmp/ks_probe6_c$c.out; head -20 gpurun_out/probe6_c${c}_kernel_stats.txt | cut -c1-140
done
for c in 0 1 3 10; do
  if [ $c = 0 ]; then M=stated; else M=noise; fi
  echo "== no profiler c=$c"; MATRIX=$M C=$c STEPS=50 python3 tools/k2_time.py 2>&1 | grep "^step" | cut -c1-200
  echo "== no profiler c=$c form 1"; D2G_SP_LIST_FORM=1 MATRIX=$M C=$c STEPS=50 python3 tools/k2_time.py 2>&1 | grep "^step" | cut -c1-80
done
