#!/bin/bash
# A/B of run-time variants inside one box: each argument is a quoted environment assignment list ("" = defaults),
# interleaved twice.  usage: tools/ab_env.sh "" "CASTRO_AMD_FINAL_LDS=0" "CASTRO_AMD_BRICK=32,4,2" ...
for rep in 1 2; do
for v in "$@"; do
  env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-contract-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k={a: b['ms_per_step'] for a, b in d['roofline']['kernel_utilisation'].items()}; print('[$v]', round(d['ms_per_step'],2), {a: round(b,2) for a,b in k.items() if a in ('k_trace','k_trace_x','k_trace_y','k_trace_z','k_trace_yz','k_trans1','k_trans1_x','k_trans1_y','k_trans1_z','k_final_x','k_final_y','k_final_z','k_finalx_consup','k_finalxz_consup','k_riemann1','k_consup_clean','k_ctoprim','k_clean_state')})"
done; done
