#!/bin/bash
for pad in 0 50000 70000 100000; do
  echo "== LDS_PAD=$pad"
  CASTRO_AMD_LDS_PAD=$pad python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/sweep_lds$pad.json 2> gpurun_out/sweep_lds$pad.err
  python - <<PY
import json
d=json.load(open("gpurun_out/sweep_lds$pad.json"))
print("ms/step %.2f" % d["ms_per_step"], {k: round(v,2) for k,v in d["path_roofline"]["kernel_ms_per_step"].items()})
PY
done
