#!/bin/bash
# usage: tools/kstats.sh LABEL cmd...   -> per-kernel durations (rocprofv3 --kernel-trace --stats) of cmd, printed and
# saved to gpurun_out/LABEL_kernel_stats.txt (GPU box).  Counters are collected separately (tools/pmc_round.sh).
label=$1; shift
repo=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $repo/gpurun_out
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ks_$label
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d /tmp/ks_$label -- "$@" > /tmp/ks_$label.out 2>&1
python3 - "$label" "$repo" <<'PY'
import csv, glob, re, sys, collections
def short(n):
    n = re.sub(r"^void\s+", "", n).replace("(anonymous namespace)::", "")
    return re.split(r"[(<]", n, 1)[0][:90] or n[:90]
label, repo = sys.argv[1], sys.argv[2]
rows = collections.defaultdict(list)
for f in glob.glob(f"/tmp/ks_{label}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in rows.values()) or 1
lines = ["%-92s %6s %12s %12s %12s %6s" % ("kernel", "calls", "avg_us", "min_us", "total_ms", "pct")]
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    lines.append("%-92s %6d %12.1f %12.1f %12.3f %6.2f" % (k, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, sum(v) / 1e6, 100.0 * sum(v) / tot))
txt = "\n".join(lines) + "\n"
open(f"{repo}/gpurun_out/{label}_kernel_stats.txt", "w").write(txt)
print(txt)
PY
tail -2 /tmp/ks_$label.out
