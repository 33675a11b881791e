#!/bin/bash
# extra PMC passes (each in its own rocprofv3 run with --kernel-trace only)
set -u
# NUMERICS (environment, default contract): the build of the kernel library that is profiled; pinned on every bench.py line and part of the tag
NUMERICS=${NUMERICS:-contract}
TAG=${1:-pmc}_$NUMERICS
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
export TMPDIR=/tmp
cd /tmp
ARGS="--numerics $NUMERICS --steps 3 --warmup 2 --no-cpu-baseline --no-contract-leg --no-extras"
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA TA_BUSY_avr"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/${TAG}_p$i -- python3 $REPO/bench.py $ARGS > $OUT/${TAG}_p$i.log 2>&1
done
cd $REPO
python3 - <<PY > $OUT/${TAG}_pmc_summary.txt
import csv, glob, os
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0,0]))
for f in glob.glob("gpurun_out/${TAG}_p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void cad::","").replace("cad::","")
        if "k_march" in k or "k_linear" in k:
            k = k.split("<")[0] + "<" + r["Kernel_Name"].split("cad::")[2].split(">")[0].split("(")[0][:24] if r["Kernel_Name"].count("cad::")>=2 else k
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
names = sorted({c for k in acc for c in acc[k]})
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES",[0,1])[0]):
    if "at::" in k or "rocclr" in k: continue
    print(k)
    for c in names:
        if c in acc[k]:
            print("    %-28s %18.1f  (per launch, %d launches)" % (c, acc[k][c][0]/acc[k][c][1], acc[k][c][1]))
PY
find $OUT/${TAG}_p* -type f -size +4M -delete 2>/dev/null
