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


# per-kernel HBM-side traffic per launch and per step, gfx950 correction applied (MI355X_MICROARCH.md, HBM section):
# FETCH_SIZE is reported in KiB and counts 128-B requests as 64 B for coalesced streams (x2);
# WRITE_SIZE (KiB) is exact.  Kernel template instances are kept apart (k_final<0|1|2>); steps = launches of k_finalx_consup.
import json
import re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_stamp
traffic = {}
vals = {}
nsteps = {}
for which, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = find(tag + "_" + which, "*counter_collection.csv")
    if not f:
        continue
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != ctr:
            continue
        k = short(r["Kernel_Name"]).strip()
        k = re.sub(r"<(\d).*", r"<\1>", k) if k.startswith("k_final<") or k.startswith("k_riemann1<") else re.sub(r"<.*", "", k)
        acc[k][0] += float(r["Counter_Value"])
        acc[k][1] += 1
    nsteps[ctr] = max(1, acc.get("k_finalx_consup", acc.get("k_trans1", [0, 1]))[1])      # one launch per step
    for k, (v, n) in acc.items():
        vals.setdefault(k, {})[ctr] = (v / n, n)
step_bytes = 0.0
for k, d in vals.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d and k.startswith("k_"):
        rd, wr = d["FETCH_SIZE"][0] * 1024 * 2, d["WRITE_SIZE"][0] * 1024
        per_step = d["FETCH_SIZE"][1] / nsteps["FETCH_SIZE"]
        traffic[k] = {"read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "bytes_per_launch": rd + wr,
                      "launches_per_step": per_step}
        step_bytes += (rd + wr) * per_step
json.dump({"tag": tag, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 (gfx950), "
           "same command as bench.py --steps 5 --warmup 2 --no-contract-leg", "source_stamp": source_stamp(),
           "bytes_per_step": step_bytes, "steps_profiled": nsteps.get("FETCH_SIZE"), "kernels": traffic},
          open(os.path.join(root, tag + "_traffic.json"), "w"), indent=1)
print("\nL2->fabric bytes per step: %.2f GB (%d steps profiled)" % (step_bytes / 1e9, nsteps.get("FETCH_SIZE", 0)))
