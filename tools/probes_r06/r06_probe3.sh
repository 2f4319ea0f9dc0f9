#!/bin/bash
cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_k2.py -x -q -m gpu -k "sparse or fill or bench_matrix or config4 or full_size" > gpurun_out/probe3_tests.log 2>&1
tail -5 gpurun_out/probe3_tests.log
for c in 0 1 10; do
  if [ $c = 0 ]; then M=stated; else M=noise; fi
  MATRIX=$M C=$c STEPS=20 bash tools/kstats.sh probe3_c$c python3 $R/tools/k2_time.py > gpurun_out/probe3_c$c.log 2>&1
  echo "== c=$c"; head -14 gpurun_out/probe3_c${c}_kernel_stats.txt | cut -c1-140
  MATRIX=$M C=$c STEPS=20 D2G_LIB=$R/dashing2_amd/libd2g_t512.so bash tools/kstats.sh probe3_t512_c$c python3 $R/tools/k2_time.py > gpurun_out/probe3_t512_c$c.log 2>&1
  echo "== c=$c emit T=512"; grep "sp_emit" gpurun_out/probe3_t512_c${c}_kernel_stats.txt | cut -c1-140
done
MATRIX=paired STEPS=20 bash tools/kstats.sh probe3_paired python3 $R/tools/k2_time.py > gpurun_out/probe3_paired.log 2>&1
echo "== paired"; head -8 gpurun_out/probe3_paired_kernel_stats.txt | cut -c1-140
N=50000 MATRIX=noise C=10 STEPS=5 bash tools/kstats.sh probe3_n50k_c10 python3 $R/tools/k2_time.py > gpurun_out/probe3_n50k_c10.log 2>&1
echo "== n50k c10"; head -10 gpurun_out/probe3_n50k_c10_kernel_stats.txt | cut -c1-140
