#!/bin/bash
# NUMERICS (environment, default contract): the build of the kernel library, pinned on every bench.py line
for n in 64 128 256; do
  python bench.py --numerics ${NUMERICS:-contract} --ncell $n --steps 20 --warmup 5 --no-cpu-baseline --no-contract-leg --no-extras > gpurun_out/sm.json 2> gpurun_out/sm.err || tail -3 gpurun_out/sm.err
  python - <<PY
import json
d=json.load(open("gpurun_out/sm.json"))
k={a: b["ms_per_step"] for a, b in d["roofline"]["kernel_utilisation"].items()}
print("n=$n: ms/step %.3f  kernels %.3f  -> %.3f G cell-updates/s" % (d["ms_per_step"], sum(k.values()), d["value"]/1e9), {a: round(b,3) for a,b in k.items()})
PY
done
