#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + FETCH_SIZE/WRITE_SIZE PMC passes) per kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def find(d, pat):
    r = glob.glob(os.path.join(root, d, "**", pat), recursive=True)
    return r[0] if r else None


def short(n):
    n = n.split("(")[0]
    return n.replace("void cad::", "").replace("cad::", "")


print("== kernel stats (%s) ==" % tag)
f = find(tag + "_stats", "*kernel_stats.csv")
if f:
    rows = list(csv.DictReader(open(f)))
    print("%-60s %8s %12s %12s %7s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
    for r in rows[:25]:
        print("%-60s %8s %12.3f %12.1f %7s" % (short(r["Name"])[:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                               float(r["AverageNs"]) / 1e3, r["Percentage"]))
else:
    print("no kernel_stats.csv found")

for which, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = find(tag + "_" + which, "*counter_collection.csv")
    print("\n== %s per launch (%s; rocprofv3 unit: KiB as reported, see MI355X_MICROARCH.md HBM section) ==" % (ctr, tag))
    if not f:
        print("no counter_collection.csv found")
        continue
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != ctr:
            continue
        k = short(r["Kernel_Name"])
        acc[k][0] += float(r["Counter_Value"])
        acc[k][1] += 1
    print("%-60s %8s %16s" % ("kernel", "launches", "avg_value"))
    for k, (v, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
        print("%-60s %8d %16.1f" % (k[:60], n, v / n))


# per-kernel HBM-side traffic per launch, gfx950 correction applied (MI355X_MICROARCH.md, HBM section):
# FETCH_SIZE is reported in KiB and counts 128-B requests as 64 B for coalesced streams (x2);
# WRITE_SIZE (KiB) is exact.  Kernel template instances are merged by base name.
import json
import re
traffic = {}
vals = {}
for which, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = find(tag + "_" + which, "*counter_collection.csv")
    if not f:
        continue
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != ctr:
            continue
        k = re.sub(r"<.*", "", short(r["Kernel_Name"])).strip()
        acc[k][0] += float(r["Counter_Value"])
        acc[k][1] += 1
    for k, (v, n) in acc.items():
        vals.setdefault(k, {})[ctr] = v / n
for k, d in vals.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d and k.startswith("k_"):
        traffic[k] = {"read_bytes_per_launch": d["FETCH_SIZE"] * 1024 * 2, "write_bytes_per_launch": d["WRITE_SIZE"] * 1024,
                      "bytes_per_launch": d["FETCH_SIZE"] * 1024 * 2 + d["WRITE_SIZE"] * 1024}
json.dump({"tag": tag, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 (gfx950), "
           "same command as bench.py --steps 5 --warmup 2", "kernels": traffic},
          open(os.path.join(root, tag + "_traffic.json"), "w"), indent=1)
