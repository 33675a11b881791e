#!/bin/bash
# bench.py and the rocprofv3 --kernel-trace --stats summary of the same command on the same box (the kernel times
# of two boxes differ by up to ~8 %): writes gpurun_out/<tag>_bench.json and gpurun_out/<tag>_samebox_stats/
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
export TMPDIR=/tmp
cd $REPO
python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python3 bench.py --reference-contract --no-cpu-baseline > $OUT/${TAG}_bench_reference_contract.json 2>/dev/null
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_samebox_stats -- python3 $REPO/bench.py --no-cpu-baseline > $OUT/${TAG}_samebox_stats.log 2>&1
cd $REPO
find $OUT/${TAG}_samebox_stats -type f -size +8M -delete 2>/dev/null
python3 - <<PY
import csv, glob, json
d = json.load(open("$OUT/${TAG}_bench.json"))
print("bench: ms/step %.3f, %s avg launch %.4f ms (hipEvent), frac %.3f" % (d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]))
f = glob.glob("$OUT/${TAG}_samebox_stats/**/*kernel_stats.csv", recursive=True)[0]
tot = n = 0
for r in csv.DictReader(open(f)):
    if "k_final" in r["Name"]:
        tot += float(r["TotalDurationNs"]); n += int(r["Calls"])
print("rocprofv3: k_final avg %.4f ms over %d launches" % (tot / n / 1e6, n))
PY
