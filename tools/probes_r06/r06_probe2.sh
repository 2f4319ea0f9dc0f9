#!/bin/bash
# round 6 probe: the binned / composed pair list -- parity tests first, then per-kernel durations at c = 0 / 1 / 3 / 10 / 30 and the adversarial matrix
cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_k2.py -x -q -m gpu -k "sparse or fill or bench_matrix or config4 or full_size" > gpurun_out/probe2_tests.log 2>&1
tail -15 gpurun_out/probe2_tests.log
for c in 0 1 3 10 30; do
  if [ $c = 0 ]; then M=stated; else M=noise; fi
  MATRIX=$M C=$c STEPS=20 ${PROBE_ENV} bash tools/kstats.sh probe2_c$c python3 $R/tools/k2_time.py > gpurun_out/probe2_c$c.log 2>&1
  grep "^step" gpurun_out/probe2_c$c.log; head -12 gpurun_out/probe2_c${c}_kernel_stats.txt | cut -c1-140
done
MATRIX=paired STEPS=20 bash tools/kstats.sh probe2_paired python3 $R/tools/k2_time.py > gpurun_out/probe2_paired.log 2>&1
grep "^step" gpurun_out/probe2_paired.log; head -8 gpurun_out/probe2_paired_kernel_stats.txt | cut -c1-140
