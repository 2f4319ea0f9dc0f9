#!/bin/bash
cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_k2.py -x -q -m gpu -k "sparse or fill or bench_matrix or config4 or full_size" > gpurun_out/probe4_tests.log 2>&1
tail -5 gpurun_out/probe4_tests.log
for c in 0 1 10; do
  if [ $c = 0 ]; then M=stated; else M=noise; fi
  MATRIX=$M C=$c STEPS=20 bash tools/kstats.sh probe4_c$c python3 $R/tools/k2_time.py > gpurun_out/probe4_c$c.log 2>&1
  echo "== c=$c"; head -16 gpurun_out/probe4_c${c}_kernel_stats.txt | cut -c1-140
  for v in t384 t256; do
    MATRIX=$M C=$c STEPS=20 D2G_LIB=$R/dashing2_amd/libd2g_$v.so bash tools/kstats.sh probe4_${v}_c$c python3 $R/tools/k2_time.py > gpurun_out/probe4_${v}_c$c.log 2>&1
    echo "== c=$c emit $v"; grep "sp_emit" gpurun_out/probe4_${v}_c${c}_kernel_stats.txt | cut -c1-140
  done
done
MATRIX=paired STEPS=20 bash tools/kstats.sh probe4_paired python3 $R/tools/k2_time.py > gpurun_out/probe4_paired.log 2>&1
echo "== paired"; head -8 gpurun_out/probe4_paired_kernel_stats.txt | cut -c1-140
