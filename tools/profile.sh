#!/bin/bash
# Profiling recipe used for profiles/ (run on the GPU box through gpurun).
# usage: tools/profile.sh <tag>   -> writes gpurun_out/<tag>_{stats,fetch,write}/ and summaries
set -u
# NUMERICS (environment, default contract): the build of the kernel library that is profiled; pinned on every bench.py line and part of the tag
NUMERICS=${NUMERICS:-contract}
TAG=${1:-r01}_$NUMERICS
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
export TMPDIR=/tmp
cd /tmp
ARGS="--numerics $NUMERICS --steps 5 --warmup 2 --no-cpu-baseline --no-contract-leg --no-extras"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $REPO/bench.py $ARGS > $OUT/${TAG}_stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_fetch -- python3 $REPO/bench.py $ARGS > $OUT/${TAG}_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_write -- python3 $REPO/bench.py $ARGS > $OUT/${TAG}_write.log 2>&1
cd $REPO
python3 tools/profile_summary.py $TAG > $OUT/${TAG}_summary.txt 2>&1
# keep only the small files
find $OUT/${TAG}_stats $OUT/${TAG}_fetch $OUT/${TAG}_write -type f -size +8M -delete 2>/dev/null
