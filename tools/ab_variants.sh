#!/bin/bash
# NUMERICS (environment, default exact): the flag set the variants were built with (tools/build_variant.sh), pinned on the bench.py line
NUMERICS=${NUMERICS:-exact}
# A/B of alternative builds of the kernel library (castro_amd/libvariant_*.so) inside one box, interleaved twice
for rep in 1 2; do
for v in default $(ls castro_amd/libvariant_*.so 2>/dev/null); do
  if [ "$v" = default ]; then unset CASTRO_AMD_LIB; else export CASTRO_AMD_LIB=$PWD/$v; fi
  python bench.py --numerics $NUMERICS --steps 10 --warmup 3 --no-cpu-baseline --no-contract-leg --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k={a: b['ms_per_step'] for a, b in d['roofline']['kernel_utilisation'].items()}; print('$v [$NUMERICS]', round(d['ms_per_step'],2), {a: round(b,2) for a,b in k.items()})"
done; done
