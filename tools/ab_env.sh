#!/bin/bash
# A/B of run-time variants inside one box: each argument is a quoted environment assignment list ("" = defaults),
# interleaved twice.  usage: tools/ab_env.sh "" "CASTRO_AMD_FINAL_LDS=0" "CASTRO_AMD_BRICK=32,4,2" ...
# NUMERICS (environment, default contract) is pinned on the bench.py line
NUMERICS=${NUMERICS:-contract}
for rep in 1 2; do
for v in "$@"; do
  env $v python bench.py --numerics $NUMERICS --steps 10 --warmup 3 --no-cpu-baseline --no-contract-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k={a: b['ms_per_step'] for a, b in d['roofline']['kernel_utilisation'].items()}; print('[$v] [$NUMERICS]', round(d['ms_per_step'],2), {a: round(b,2) for a,b in k.items()})"
done; done
