#!/usr/bin/env python3
"""Timing of the two-level AMR driver (first slice): 128^3 base + one 128^3 refined patch
(the central 64^3 coarse zones), Sedov.  Prints zone updates per second counted like the reference's FOM
(coarse zones + 2 x fine zones per coarse step)."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import castro_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nfine = int(sys.argv[3]) if len(sys.argv) > 3 else 1          # number of refined levels (BASELINE config 4: 2)
q = n // 4
patches = [((q, q, q), (3 * q - 1, 3 * q - 1, 3 * q - 1))]
if nfine == 2:                                                # level-2 patch: the central n/2 zones of the level-1 box
    lo1 = 2 * q + q
    patches.append(((lo1, lo1, lo1), (lo1 + 2 * q - 1, lo1 + 2 * q - 1, lo1 + 2 * q - 1)))
mode = sys.argv[4] if len(sys.argv) > 4 else ""                # "tags": one bounding box per level; "cluster": Berger-Rigoutsos boxes
dynamic = mode in ("tags", "cluster")                          # the refined levels follow the tags (regrid every 2 steps)
t_start = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0     # evolve to this time first (a developed blast wave)
kw = dict(a.split("=") for a in sys.argv[6:])                  # bf=<blocking_factor> eff=<grid_eff> mgs=<max_grid_size> grav=<const_grav>; the build: CASTRO_AMD_NUMERICS
grav = dict(do_grav=True, const_grav=float(kw["grav"])) if "grav" in kw else {}
if dynamic:
    a = castro_amd.CastroAmr((n, n, n), refine=[("density", "gradient", 0.05), ("rho_E", "relative_gradient", 0.5)],
                             regrid_int=2, n_error_buf=2, blocking_factor=int(kw.get("bf", 16 if mode == "cluster" else 8)), max_level=nfine,
                             cluster=mode == "cluster", grid_eff=float(kw.get("eff", 0.7)), max_grid_size=int(kw.get("mgs", 128)), **grav)
else:
    a = castro_amd.CastroAmr((n, n, n), patches=patches, **grav)
a.initData("sedov")
if t_start > 0.0:
    a.evolve(t_start)
for _ in range(3):
    a.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
zones = 0
for _ in range(steps):
    a.step()
    zones += a.zones_advanced_per_coarse_step()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
if dynamic:
    print(json.dumps({"workload": "Sedov %d^3 base + up to %d tag-driven refined levels (%s), subcycled, regrid_int 2, from t = %g"
                      % (n, nfine, "Berger-Rigoutsos boxes" if mode == "cluster" else "one box each", t_start),
                      "steps": steps, "ms_per_coarse_step": wall / steps * 1e3, "zone_updates_per_s": zones / wall,
                      "boxes_per_level": [len(lev.boxes) for lev in a.levels],
                      "zones_per_level": [sum(b.n[0] * b.n[1] * b.n[2] for b in lev.boxes) for lev in a.levels],
                      "regrids": a.nregrid, "mass_drift": a.composite_sum(0) - 1.0}))
    sys.exit(0)
zones = (n ** 3 + 2 * (4 * q) ** 3 + (4 * (4 * q) ** 3 if nfine == 2 else 0)) * steps
print(json.dumps({"workload": "Sedov %d^3 base + %d refined level(s), one %d^3 patch each, subcycled" % (n, nfine, 4 * q), "steps": steps,
                  "ms_per_coarse_step": wall / steps * 1e3, "zone_updates_per_s": zones / wall,
                  "mass_drift": a.composite_sum(0) - 1.0}))
