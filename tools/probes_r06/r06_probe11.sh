#!/bin/bash
cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
for m in "noise 100" "skewed -" "stated -"; do
  set -- $m
  for sp in 1 0; do D2G_BS_SPARSE=$sp MATRIX=$1 C=$2 timeout 120 python3 tools/k2_first.py 2>&1 | grep "first step" | cut -c1-330; done
done
MATRIX=noise C=100 REPS=10 timeout 200 bash tools/kstats.sh probe11_first python3 $R/tools/k2_first.py > gpurun_out/probe11_first.log 2>&1
grep "sample\|giveup\|colplan\|planes" gpurun_out/probe11_first_kernel_stats.txt | cut -c1-140
