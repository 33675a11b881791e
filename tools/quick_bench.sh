#!/bin/bash
# NUMERICS (environment, default contract): the build of the kernel library, pinned on every bench.py line
# parity tests, then the 256^3 bench with per-kernel times
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
python bench.py --numerics ${NUMERICS:-contract} --steps 10 --warmup 3 --no-cpu-baseline --no-contract-leg > gpurun_out/qb.json 2> gpurun_out/qb.err || tail -5 gpurun_out/qb.err
python - <<PY
import json
d=json.load(open("gpurun_out/qb.json"))
print("ms/step %.2f" % d["ms_per_step"], {k: round(v,2) for k,v in {a: b["ms_per_step"] for a, b in d["roofline"]["kernel_utilisation"].items()}.items()})
PY
