#!/bin/bash
# usage: tools/pmc_k3.sh <ngenomes> : rocprofv3 PMC passes over tools/k3_time.py, prints per-kernel averages
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmc
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -f csv -d /tmp/pmc -- python /root/repo/tools/k3_time.py $1 5e6 21 2048 1 > /dev/null 2>&1
done
python - <<PY
import glob,csv,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmc/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k3_" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("::")[1].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print(k, {c: "%.3g" % (sum(x)/len(x)) for c,x in sorted(v.items())})
PY
