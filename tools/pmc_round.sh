#!/bin/bash
# usage: tools/pmc_round.sh [OUT.json]   (GPU box)  rocprofv3 --pmc passes over the default bench.py workload -- ONE
# counter set per run, counters only (no trace domains beside --kernel-trace), as MI355X_MICROARCH.md prescribes --
# summarised per kernel and dispatch by tools/pmc_summary.py.  Result: gpurun_out/r05_pmc.json (copy to profiles/).
repo=$(cd "$(dirname "$0")/.." && pwd)
out=$(realpath -m ${1:-$repo/gpurun_out/r05_pmc.json})
mkdir -p $repo/gpurun_out
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmc_round
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR"; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c -f csv -d /tmp/pmc_round -- \
      python $repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config4 --no-matrices > /tmp/pmc_round_last.log 2>&1 || tail -3 /tmp/pmc_round_last.log
done
python3 $repo/tools/pmc_summary.py /tmp/pmc_round $out
