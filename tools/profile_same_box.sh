#!/bin/bash
# bench.py and the rocprofv3 --kernel-trace --stats summary of the same command on the same box (the kernel times
# of two boxes differ by up to ~8 %): writes gpurun_out/<tag>_bench.json and gpurun_out/<tag>_samebox_stats/
set -u
# NUMERICS (environment, default contract): the build of the kernel library that is profiled; pinned on every bench.py line and part of the tag
NUMERICS=${NUMERICS:-contract}
TAG=${1:-r01}_$NUMERICS
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
export TMPDIR=/tmp
cd $REPO
python3 bench.py --numerics $NUMERICS > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python3 bench.py --numerics $NUMERICS --reference-contract --no-cpu-baseline > $OUT/${TAG}_bench_reference_contract.json 2>/dev/null
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_samebox_stats -- python3 $REPO/bench.py --numerics $NUMERICS --no-cpu-baseline --no-extras --no-contract-leg > $OUT/${TAG}_samebox_stats.log 2>&1
cd $REPO
find $OUT/${TAG}_samebox_stats -type f -size +8M -delete 2>/dev/null
python3 - <<PY
import csv, glob, json
d = json.load(open("$OUT/${TAG}_bench.json"))
ku = d["roofline"]["kernel_utilisation"]
print("bench: ms/step %.3f, contract roofline frac %.4f" % (d["ms_per_step"], d["roofline"]["frac"]))
f = glob.glob("$OUT/${TAG}_samebox_stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
def prof(sub):
    tot = n = 0
    for r in rows:
        if sub in r["Name"]:
            tot += float(r["TotalDurationNs"]); n += int(r["Calls"])
    return (tot / n / 1e6, n) if n else (float("nan"), 0)
for label, sub in (("k_trans1", "k_trans1<"), ("k_trans1_fold", "k_trans1_tile" if any("k_trans1_tile" in r["Name"] for r in rows) else "k_trans1_fold"), ("k_trace", "k_trace_pair"), ("k_finalx_consup", "k_finalx_consup"),
                   ("k_final_y", "k_final<1"), ("k_final_z", "k_final<2")):
    if label in ku:
        a, n = prof(sub)
        print("%-16s hipEvent avg %.4f ms | rocprofv3 avg %.4f ms over %d launches" % (label, ku[label]["avg_launch_ms"], a, n))
PY
