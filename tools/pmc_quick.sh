#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of every kernel for the current environment, one line per kernel (GB per launch, x2 applied to
# FETCH_SIZE): usage tools/pmc_quick.sh <tag>
# NUMERICS (environment, default contract): the build of the kernel library that is profiled; pinned on every bench.py line and part of the tag
NUMERICS=${NUMERICS:-contract}
TAG=${1:-q}_$NUMERICS
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
export TMPDIR=/tmp
cd /tmp
ARGS="--numerics $NUMERICS --steps 3 --warmup 1 --no-cpu-baseline --no-contract-leg --no-extras"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_fetch -- python3 $REPO/bench.py $ARGS > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_write -- python3 $REPO/bench.py $ARGS > /dev/null 2>&1
cd $REPO
python3 - <<PY
import csv, glob, re
from collections import defaultdict
v = defaultdict(dict)
for which, ctr, mul in (("fetch", "FETCH_SIZE", 2048.0), ("write", "WRITE_SIZE", 1024.0)):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob("gpurun_out/${TAG}_%s/**/*counter_collection.csv" % which, recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == ctr:
                k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void cad::", "").replace("cad::", "")
                acc[k][0] += float(r["Counter_Value"]) * mul; acc[k][1] += 1
    for k, (a, n) in acc.items():
        v[k][ctr] = a / n / 1e9
tot = 0
for k in sorted(v, key=lambda k: -sum(v[k].values())):
    if k.startswith("k_") and "init" not in k and "estdt" not in k:
        print("%-44s R %6.2f W %6.2f GB" % (k[:44], v[k].get("FETCH_SIZE", 0), v[k].get("WRITE_SIZE", 0)))
        tot += sum(v[k].values())
print("numerics = $NUMERICS; sum over hot kernels (one launch each): %.1f GB" % tot)
PY
rm -rf $OUT/${TAG}_fetch $OUT/${TAG}_write
