#!/bin/bash
# round 6, first probe: per-kernel durations of the K2 step at c = 0 / 1 / 3 / 10 chance collisions per sketch with the pair list's cap loosened (LIST_DIV=4)
cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
for c in 0 1 3 10; do
  if [ $c = 0 ]; then M=stated; else M=noise; fi
  MATRIX=$M C=$c STEPS=20 D2G_SP_LIST_DIV=4 bash tools/kstats.sh probe1_c$c python3 $R/tools/k2_time.py > gpurun_out/probe1_c$c.log 2>&1
done
MATRIX=paired STEPS=20 D2G_SP_LIST_DIV=4 bash tools/kstats.sh probe1_paired python3 $R/tools/k2_time.py > gpurun_out/probe1_paired.log 2>&1
N=50000 MATRIX=noise C=10 STEPS=5 bash tools/kstats.sh probe1_n50k_c10 python3 $R/tools/k2_time.py > gpurun_out/probe1_n50k_c10.log 2>&1
tail -3 gpurun_out/probe1_*.log
