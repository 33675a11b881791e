#!/usr/bin/env python3
"""256^3 Sedov step time and per-kernel times for non-default options: tools/variant_bench.py ppm_type=0 riemann_solver=2 ..."""
import sys
import time

import torch

sys.path.insert(0, ".")
import castro_amd

kw = {}
for a in sys.argv[1:]:
    k, v = a.split("=")
    kw[k] = float(v) if "." in v or "e" in v else int(v)
n = int(kw.pop("n", 256))
c = castro_amd.Castro((n, n, n), params=castro_amd.default_params(**kw))
c.initData("sedov")
for _ in range(3):
    c.step()
c.hydro.profile(True); c.hydro.profile_reset()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    c.step()
torch.cuda.synchronize(); wall = time.perf_counter() - t0
rep = c.hydro.profile_report()
print(kw, "ms/step %.2f" % (wall * 100), {k: round(v[0] / 10, 2) for k, v in sorted(rep.items(), key=lambda kv: -kv[1][0])})
