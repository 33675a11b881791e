#!/usr/bin/env python3
"""256^3 Sedov with constant gravity (castro.do_grav, ConstantGrav): ms per step and per-kernel hipEvent times -- the path with traced
source terms.  usage: [CASTRO_AMD_NUMERICS=contract] python tools/gravity_bench.py [n] [steps]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import castro_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
c = castro_amd.Castro((n, n, n), do_grav=True, const_grav=-1.0)
c.initData("sedov")
for _ in range(3):
    c.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    c.step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / steps * 1e3
c.hydro.profile(True); c.hydro.profile_reset()
for _ in range(5):
    c.step()
torch.cuda.synchronize()
rep = c.hydro.profile_report()
k = {a: round(ms / 5, 3) for a, (ms, cnt) in sorted(rep.items(), key=lambda kv: -kv[1][0])}
print("[%s] Sedov %d^3 + constant gravity: %.2f ms/step; kernels %.2f ms: %s" % (c.hydro.numerics, n, wall, sum(k.values()), k))
