#!/bin/bash
# usage: pmc_generic.sh <tag> "<counter set 1>" "<counter set 2>" ...   (each set = one rocprofv3 pass)
NUMERICS=${NUMERICS:-exact}     # the build that is profiled (environment; pinned on the bench.py line, part of the tag)
TAG=$1_$NUMERICS; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
export TMPDIR=/tmp
cd /tmp
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/${TAG}_p$i -- python3 $REPO/bench.py --numerics $NUMERICS --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-contract-leg > $OUT/${TAG}_p$i.log 2>&1 || tail -3 $OUT/${TAG}_p$i.log
done
cd $REPO
python3 - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0,0]))
for f in glob.glob("gpurun_out/${TAG}_p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void cad::","").replace("cad::","")
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
names = sorted({c for k in acc for c in acc[k]})
for k in sorted(acc):
    if "at::" in k or "rocclr" in k or "sedov" in k or "bc_fill" in k: continue
    print(k)
    for c in names:
        if c in acc[k]: print("    %-40s %16.1f" % (c, acc[k][c][0]/acc[k][c][1]))
PY
find $OUT/${TAG}_p* -type f -size +4M -delete 2>/dev/null
