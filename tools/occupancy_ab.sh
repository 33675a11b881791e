#!/bin/bash
# Round 6, item 3: the occupancy variants of the two chain-bound kernels on one box -- per-kernel hipEvent times (bench.py's untimed
# pass) and the SQ wait / issue counters of one PMC pass per variant.  Variants: castro_amd/libvariant_*.so (tools/build_variant.sh
# with NUMERICS=contract) and CASTRO_AMD_TRACE_ONE_ZONE=1 on the shipped library.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
export TMPDIR=/tmp
ARGS="--numerics contract --steps 10 --warmup 3 --no-cpu-baseline --no-contract-leg --no-extras"
run() {   # name, env assignments...
  name=$1; shift
  for rep in 1 2; do
  env "$@" python3 $REPO/bench.py $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k={a: b['ms_per_step'] for a, b in d['roofline']['kernel_utilisation'].items()}
print('%-14s step %.2f ms  ' % ('$name', d['ms_per_step']), {a.replace('k_',''): round(b,2) for a,b in k.items() if b > 0.1})"
  done
}
cd $REPO
run shipped CASTRO_AMD_X=0
run one_zone_trace CASTRO_AMD_TRACE_ONE_ZONE=1
for v in w3trace w3final tilenopf; do run $v CASTRO_AMD_LIB=$REPO/castro_amd/libvariant_$v.so; done
run shipped CASTRO_AMD_X=0
# SQ counters, one pass per variant (the program itself behind `--`: the environment is exported here, not through env)
cd /tmp
pmc() {
  name=$1
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT/occ_$name -- python3 $REPO/bench.py --numerics contract --steps 3 --warmup 2 --no-cpu-baseline --no-contract-leg --no-extras > /dev/null 2>&1
  python3 - <<PY
import csv, glob, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/occ_$name/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void cad::", "").replace("cad::", "")
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc, key=lambda k: -acc[k]["SQ_WAVE_CYCLES"][0]):
    d = {c: v[0] / v[1] for c, v in acc[k].items()}
    if not k.startswith("k_") or "init" in k or d.get("SQ_WAVES", 0) < 1000: continue
    wc = d["SQ_WAVE_CYCLES"]
    print("%-14s %-36s wait_any %5.1f %%  wait_inst %5.1f %%  active_inst %5.1f %%  VALU-active %5.1f %% of wave cycles; %6.0f VALU inst/wave; %5.2f waves/SIMD resident" % (
        "$name", k[:36], d["SQ_WAIT_ANY"] / wc * 100, d["SQ_WAIT_INST_ANY"] / wc * 100, d["SQ_ACTIVE_INST_ANY"] / wc * 100,
        d["SQ_ACTIVE_INST_VALU"] / wc * 100, d["SQ_INSTS_VALU"] / d["SQ_WAVES"], wc / max(d["SQ_BUSY_CYCLES"], 1) / 4.0))
PY
  rm -rf $OUT/occ_$name
}
pmc shipped
export CASTRO_AMD_TRACE_ONE_ZONE=1; pmc one_zone_trace; unset CASTRO_AMD_TRACE_ONE_ZONE
for v in w3trace w3final; do export CASTRO_AMD_LIB=$REPO/castro_amd/libvariant_$v.so; pmc $v; unset CASTRO_AMD_LIB; done
