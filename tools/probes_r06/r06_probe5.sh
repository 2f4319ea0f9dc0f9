#!/bin/bash
cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_k2.py -x -q -m gpu -k "sparse or fill or bench_matrix or config4 or full_size" > gpurun_out/probe5_tests.log 2>&1
tail -5 gpurun_out/probe5_tests.log
for c in 0 1 3 10; do
  if [ $c = 0 ]; then M=stated; else M=noise; fi
  MATRIX=$M C=$c STEPS=20 bash tools/kstats.sh probe5_c$c python3 $R/tools/k2_time.py > gpurun_out/probe5_c$c.log 2>&1
  echo "== c=$c"; head -20 gpurun_out/probe5_c${c}_kernel_stats.txt | cut -c1-140
done
MATRIX=paired STEPS=20 bash tools/kstats.sh probe5_paired python3 $R/tools/k2_time.py > gpurun_out/probe5_paired.log 2>&1
echo "== paired"; head -9 gpurun_out/probe5_paired_kernel_stats.txt | cut -c1-140
N=50000 MATRIX=noise C=10 STEPS=5 bash tools/kstats.sh probe5_n50k_c10 python3 $R/tools/k2_time.py > gpurun_out/probe5_n50k_c10.log 2>&1
echo "== n50k c10"; head -10 gpurun_out/probe5_n50k_c10_kernel_stats.txt | cut -c1-140
