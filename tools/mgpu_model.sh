#!/bin/bash
# usage: tools/mgpu_model.sh [N S]   (GPU box)  -> gpurun_out/r06_mgpu_model[_N<N>].txt
# one rocprofv3 --kernel-trace run per world size of tools/mgpu_model.py --worker (loopback transport on one GPU), then the report
repo=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $repo/gpurun_out
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/mgpu_model
for W in 1 2 4 8; do
  mkdir -p /tmp/mgpu_model/W$W
  timeout 900 rocprofv3 --kernel-trace -f csv -d /tmp/mgpu_model/W$W -- python $repo/tools/mgpu_model.py --worker $W "$@" > /tmp/mgpu_model/W$W/worker.log 2>&1
  grep WORKER /tmp/mgpu_model/W$W/worker.log || tail -5 /tmp/mgpu_model/W$W/worker.log
done
sfx=""; [ -n "$1" ] && sfx="_N$1"
python $repo/tools/mgpu_model.py --report /tmp/mgpu_model "$@" | tee $repo/gpurun_out/r06_mgpu_model$sfx.txt
