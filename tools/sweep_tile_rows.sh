#!/bin/bash
# NUMERICS (environment, default contract): the build of the kernel library, pinned on every bench.py line
# sweep the XCD-tiled workgroup order knob: correctness first, then the 256^3 bench per setting
mkdir -p gpurun_out
CASTRO_AMD_TILE_ROWS=8 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
for ty in 0 4 8 16 32 64; do
  echo "== TILE_ROWS=$ty"
  CASTRO_AMD_TILE_ROWS=$ty python bench.py --numerics ${NUMERICS:-contract} --steps 10 --warmup 3 --no-cpu-baseline --no-contract-leg > gpurun_out/sweep_ty$ty.json 2> gpurun_out/sweep_ty$ty.err
  python - <<PY
import json
d=json.load(open("gpurun_out/sweep_ty$ty.json"))
print("ms/step %.2f" % d["ms_per_step"], {k: round(v,2) for k,v in {a: b["ms_per_step"] for a, b in d["roofline"]["kernel_utilisation"].items()}.items()})
PY
done
