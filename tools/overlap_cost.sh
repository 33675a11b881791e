#!/bin/bash
# NUMERICS (environment, default contract): the build of the kernel library, pinned on every bench.py line
# per-rank cost of the tiled/overlapped path, emulated on one GPU with periodic self-neighbours
for n in 128 256; do
for mode in "--periodic --no-overlap" "--periodic --force-overlap" "--periodic --overlap-tiles"; do
  python bench.py --numerics ${NUMERICS:-contract} --ncell $n --steps 10 --warmup 3 --no-cpu-baseline --no-contract-leg $mode > gpurun_out/ov.json 2> gpurun_out/ov.err || tail -3 gpurun_out/ov.err
  python - <<PY
import json
d=json.load(open("gpurun_out/ov.json"))
k={a: b["ms_per_step"] for a, b in d["roofline"]["kernel_utilisation"].items()}
print("n=$n $mode: ms/step %.3f  kernels %.3f  pack %.3f unpack %.3f" % (d["ms_per_step"], sum(k.values()), k.get("k_pack",0), k.get("k_unpack",0)), d["config"]["overlap_halo"])
PY
done; done
