cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c23; R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_DATA_STALL_CYCLES_sum" "FETCH_SIZE WRITE_SIZE"; do
  i=$((i+1))
  N=50000 ITERS=3 timeout 300 rocprofv3 --kernel-trace --pmc $c -f csv -d /tmp/pmc23/p$i -- python $R/tools/k2_time.py > /tmp/pmc23_$i.log 2>&1 || (echo "pass $i failed: $c"; tail -3 /tmp/pmc23_$i.log)
done
python3 - <<'PY'
import csv, glob, collections, json, os
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmc23/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}
json.dump(res, open(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/c23/pmc_50k.json", "w"), indent=1, sort_keys=True)
for k in ("bs_rank_kernel", "k2_bitslice_sparse_kernel", "sp_fill_kernel"):
    print(k, {c: round(v) for c, v in sorted(res.get(k, {}).items())})
PY
