#!/bin/bash
for pad in 0 16 32 48 272 1040; do
  echo "== PLANE_PAD=$pad"
  CASTRO_AMD_PLANE_PAD=$pad python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/sweep_pad$pad.json 2> gpurun_out/sweep_pad$pad.err
  python - <<PY
import json
d=json.load(open("gpurun_out/sweep_pad$pad.json"))
print("ms/step %.2f" % d["ms_per_step"], {k: round(v,2) for k,v in d["path_roofline"]["kernel_ms_per_step"].items()})
PY
done
