"""Reads a rocprofv3 --kernel-trace CSV and prints, for the last replayed steps, the start / end of the exchange kernels (pack,
RCCL, unpack) and of the hydro kernels around them: do the exchange and k_ctoprim on the valid zones run concurrently?"""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
ev = []
for r in rows:
    name = r["Kernel_Name"]
    short = name.split("(")[0].replace("void ", "").replace("cad::", "")[:40]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Stream_Id", "?"), r.get("Queue_Id", "?")))
ev.sort()
# one whole step of the timed batch: between the third- and the second-last k_step_control
ctl = [i for i, e in enumerate(ev) if e[2].startswith("k_step_control")]
last = ev[ctl[-3] + 1:ctl[-2] + 1] if len(ctl) >= 3 else ev[-48:]
t0 = last[0][0]
for s, e, n, st, q in last:
    print("%9.1f %9.1f us  dur %8.1f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n))
