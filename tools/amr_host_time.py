#!/usr/bin/env python3
"""When does the host return from a coarse AMR step, and when is the device done?  (cluster mode, 128^3 base + 2 levels)"""
import sys, time, torch
sys.path.insert(0, ".")
import castro_amd
n = 128
a = castro_amd.CastroAmr((n, n, n), refine=[("density", "gradient", 0.05), ("rho_E", "relative_gradient", 0.5)],
                         regrid_int=2, n_error_buf=2, blocking_factor=16, max_level=2, cluster=True, grid_eff=0.7, max_grid_size=128)
a.initData("sedov")
a.evolve(0.005)
for _ in range(3):
    a.step()
torch.cuda.synchronize()
issue = wall = 0.0
for _ in range(10):
    t0 = time.perf_counter()
    a.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    issue += t1 - t0; wall += t2 - t0
print("per coarse step: host returns after %.2f ms, device done after %.2f ms" % (issue * 100, wall * 100), [len(l.boxes) for l in a.levels])
