#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs (one pass per counter set) into
profiles/<name>.json: per-kernel averages per dispatch.  FETCH_SIZE / WRITE_SIZE are reported by
rocprofv3 in KiB; MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE counts wide (16 B/lane)
coalesced streaming reads at 1/2 of their bytes -- other access widths and WRITE_SIZE are
uncalibrated -- so both the raw and the doubled read figure are recorded.
usage: pmc_summary.py <dir with *_counter_collection.csv> <out.json>"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    d, out = sys.argv[1], sys.argv[2]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, cs in agg.items():
        import re
        short = re.split(r"[(<]", re.sub(r"^void\s+", "", k).replace("(anonymous namespace)::", ""), 1)[0].strip()
        if not any(x in k for x in ("k0_", "k1_oph", "k2_", "bs_rank", "bs_planes", "bs_colplan", "bs_derive", "k3_", "k3c_", "mg_pack", "sp_")):
            continue
        e = {c: sum(v) / len(v) for c, v in cs.items()}
        e["dispatches"] = max(len(v) for v in cs.values())
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_read_bytes_raw"] = e["FETCH_SIZE"] * 1024
            e["hbm_read_bytes_x2_wide_load_correction"] = e["FETCH_SIZE"] * 2048
            e["hbm_write_bytes"] = e["WRITE_SIZE"] * 1024
            # the LARGEST dispatch (a kernel launched at several sizes in one run, e.g. K1: the 1000-genome launch next to the
            # small launches of the ingest measurement)
            e["largest_dispatch_hbm_read_bytes_raw"] = max(cs["FETCH_SIZE"]) * 1024
            e["largest_dispatch_hbm_write_bytes"] = max(cs["WRITE_SIZE"]) * 1024
        if "TCC_HIT_sum" in e:
            e["l2_hit_rate"] = e["TCC_HIT_sum"] / max(e["TCC_HIT_sum"] + e["TCC_MISS_sum"], 1)
        res[short + " :: " + k[:120]] = e
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, e in res.items():
        print(k[:60], {x: (round(y, 3) if isinstance(y, float) else y) for x, y in e.items() if "bytes" in x or "rate" in x or "VALU" in x})


if __name__ == "__main__":
    main()
