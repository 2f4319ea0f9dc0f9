#!/bin/bash
# usage: tools/pmc_k3.sh <ngenomes> [counter sets...] : rocprofv3 PMC passes (one per counter set) over
# tools/k3_time.py; prints per-kernel per-dispatch averages.  FETCH_SIZE / WRITE_SIZE are in KiB.
N=$1; shift
SETS=("$@")
[ ${#SETS[@]} -eq 0 ] && SETS=("FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD")
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmc
for c in "${SETS[@]}"; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -f csv -d /tmp/pmc -- python /root/repo/tools/k3_time.py $N 5e6 21 2048 1 > /dev/null 2>&1
done
python - <<PY
import glob,csv,collections,json
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmc/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k3_" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("::")[1].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out={k: {c: sum(x)/len(x) for c,x in sorted(v.items())} for k,v in agg.items()}
for k,v in out.items():
    if "FETCH_SIZE" in v: v["fetch_bytes_raw"]=v["FETCH_SIZE"]*1024
    if "WRITE_SIZE" in v: v["write_bytes"]=v["WRITE_SIZE"]*1024
print(json.dumps(out, indent=1))
PY
