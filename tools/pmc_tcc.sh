#!/bin/bash
# one PMC pass (L2 counters only) for a given CASTRO_AMD_TILE_ROWS; usage: pmc_tcc.sh <tag> <rows>
TAG=$1; export CASTRO_AMD_TILE_ROWS=$2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $OUT/${TAG} -- python3 $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/${TAG}.log 2>&1
cd $REPO
python3 - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0,0]))
for f in glob.glob("gpurun_out/${TAG}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void cad::","").replace("cad::","")
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print("tile_rows=$2")
for k in sorted(acc):
    if "at::" in k or "rocclr" in k: continue
    d = {c: v[0]/v[1] for c, v in acc[k].items()}
    print("%-28s RD %6.2f GB  WR %6.2f GB  L2hit %5.1f%%  req %7.1fM" % (k[:28], d["TCC_EA0_RDREQ_sum"]*128/1e9, d["TCC_EA0_WRREQ_sum"]*64/1e9,
          100*d["TCC_HIT_sum"]/(d["TCC_HIT_sum"]+d["TCC_MISS_sum"]), (d["TCC_HIT_sum"]+d["TCC_MISS_sum"])/1e6))
PY
find $OUT/${TAG} -type f -size +4M -delete 2>/dev/null
