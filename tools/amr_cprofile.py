#!/usr/bin/env python3
"""Host-side profile (cProfile) of coarse AMR steps of config 4 [+ constant gravity]: where the Python driver spends its time.
usage: [CASTRO_AMD_NUMERICS=contract] python tools/amr_cprofile.py [grav=<g>] [eff=<grid_eff>] [steps=<n>]"""
import cProfile
import pstats
import sys
import time

import torch

sys.path.insert(0, ".")
import castro_amd

kw = dict(a.split("=") for a in sys.argv[1:])
grav = dict(do_grav=True, const_grav=float(kw["grav"])) if "grav" in kw else {}
n, steps = 128, int(kw.get("steps", 10))
a = castro_amd.CastroAmr((n, n, n), refine=[("density", "gradient", 0.05), ("rho_E", "relative_gradient", 0.5)],
                         regrid_int=2, n_error_buf=2, blocking_factor=16, max_level=2, cluster=True, grid_eff=float(kw.get("eff", 0.7)),
                         max_grid_size=128, **grav)
a.initData("sedov")
a.evolve(0.005)
for _ in range(3):
    a.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(steps):
    a.step()
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("per coarse step (profiled): host %.2f ms, device done %.2f ms; boxes %s" % ((t1 - t0) / steps * 1e3, (t2 - t0) / steps * 1e3,
                                                                                    [len(l.boxes) for l in a.levels]))
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(32)
