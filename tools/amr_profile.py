#!/usr/bin/env python3
"""Where a tag-driven AMR coarse step spends its wall time: regridding (tags, host clustering, data movement) vs the rest."""
import sys
import time

import torch

sys.path.insert(0, ".")
import castro_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
mode = sys.argv[3] if len(sys.argv) > 3 else "tags"
a = castro_amd.CastroAmr((n, n, n), refine=[("density", "gradient", 0.05), ("rho_E", "relative_gradient", 0.5)],
                         regrid_int=2, n_error_buf=2, blocking_factor=16 if mode == "cluster" else 8, max_level=2,
                         cluster=mode == "cluster", grid_eff=0.7, max_grid_size=128)
a.initData("sedov")
a.evolve(0.005)
acc = {"regrid": 0.0, "tags": 0.0, "places": 0.0}


def timed(name, fn):
    def w(*args, **kw):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*args, **kw)
        torch.cuda.synchronize()
        acc[name] += time.perf_counter() - t0
        return r
    return w


a.regrid = timed("regrid", a.regrid)
a._tags = timed("tags", a._tags)
a._grid_places = timed("places", a._grid_places)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    a.step()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print("ms per coarse step: total %.2f, regrid %.2f (grid_places %.2f of which tags %.2f)" %
      tuple(1e3 * x / steps for x in (wall, acc["regrid"], acc["places"], acc["tags"])))
print("boxes per level", [len(lev.boxes) for lev in a.levels], "zones per level", [sum(b.n[0] * b.n[1] * b.n[2] for b in lev.boxes) for lev in a.levels])

# per-kernel device time of one coarse step (hipEvent-timed inside the library), all levels together
for h in a.all_hydros():
    h.profile(True)
    h.profile_reset()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    a.step()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
tot = {}
for h in a.all_hydros():
    for k, (ms, n_) in h.profile_report().items():
        e = tot.setdefault(k, [0.0, 0])
        e[0] += ms; e[1] += n_
    h.profile(False)
ksum = sum(v[0] for v in tot.values())
print("with kernel timing on: %.2f ms per coarse step, kernels %.2f ms" % (1e3 * wall / steps, ksum / steps))
for k, (ms, n_) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print("  %-22s %8.3f ms  %6.1f launches per coarse step" % (k, ms / steps, n_ / steps))
