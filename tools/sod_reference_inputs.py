#!/usr/bin/env python3
"""The reference's own shock-tube verification runs, as its inputs files specify them (Exec/hydro_tests/Sod/
inputs-sod-x, inputs-test2-x, inputs-test3-x): 32 x 8 x 8 base zones on [0,1] x [0,.25]^2, amr.max_level = 2 with
regrid_int 2, blocking_factor 8, max_grid_size 64, n_error_buf 2 and the density / pressure (/ velocity) refinement
indicators, outflow in x and slip walls in y, z -- compared at stop_time with the 128-point exact solutions of
Exec/hydro_tests/Sod/Verification (the effective resolution of the finest level).  run(case, idir=2 | 3) is the -y / -z
inputs file of the same problem."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import castro_amd

CASES = {   # left (rho, u, p), right, stop_time, cfl, extra indicators
    "sod": ((1.0, 0.0, 1.0), (0.125, 0.0, 0.1), 0.2, 0.9, []),
    "test2": ((1.0, -2.0, 0.4), (1.0, 2.0, 0.4), 0.15, 0.8,
              [("x_velocity", "gradient", 0.01), ("y_velocity", "gradient", 0.01), ("z_velocity", "gradient", 0.01)]),
    "test3": ((1.0, 0.0, 1000.0), (1.0, 0.0, 0.01), 0.012, 0.9, []),
}


def run(case, make_hydro=None, default_params=None, idir=1):
    """idir = 1, 2, 3: the -x, -y, -z inputs files (the same problem with the tube along that direction)"""
    Lst, Rst, stop, cfl, extra = CASES[case]
    dp = default_params or castro_amd.default_params
    refine = [("density", "value_greater", 3.0), ("density", "gradient", 0.01),
              ("pressure", "value_greater", 3.0), ("pressure", "gradient", 0.01)] + extra
    d = idir - 1
    n_cell, prob_hi, bc = [8, 8, 8], [0.25, 0.25, 0.25], [4, 4, 4]
    n_cell[d], prob_hi[d], bc[d] = 32, 1.0, 2
    a = castro_amd.CastroAmr(tuple(n_cell), prob_hi=tuple(prob_hi), lo_bc=tuple(bc), hi_bc=tuple(bc),
                             params=dp(cfl=cfl, init_shrink=0.1, change_max=1.05), make_hydro=make_hydro,
                             refine=refine, regrid_int=2, n_error_buf=2, blocking_factor=8, max_level=2,
                             cluster=True, grid_eff=0.7, max_grid_size=64)
    a.initData("sod", rho_l=Lst[0], u_l=Lst[1], p_l=Lst[2], rho_r=Rst[0], u_r=Rst[1], p_r=Rst[2], idir=idir, frac=0.5)
    m0 = a.composite_sum(0)
    a.evolve(stop)
    # the line through the origin along the tube at the finest spacing: finest data where refined, coarser data repeated
    nf = 32 * 2 ** 2
    line = np.full((8, nf), np.nan)
    t1, t2 = [e for e in range(3) if e != d]
    for l, lev in enumerate(a.levels):
        r = 2 ** (2 - l)
        for b in lev.boxes:
            if b.lo[t1] == 0 and b.lo[t2] == 0:
                sl = [0, 0, 0]
                sl[2 - d] = slice(None)
                S = b.S_new()[(slice(None),) + tuple(sl)].cpu().numpy()
                line[:, b.lo[d] * r:(b.hi[d] + 1) * r] = np.repeat(S, r, axis=1)
    gamma = a.params.eos_gamma
    rho, u, p = line[0], line[1 + d] / line[0], (gamma - 1.0) * line[5]
    ex = np.loadtxt(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "reference_verification",
                                 "%s-exact.out" % case))
    res = dict(rho=float(np.abs(rho - ex[:, 1]).mean() / np.abs(ex[:, 1]).mean()), u=float(np.abs(u - ex[:, 2]).mean()),
               p=float(np.abs(p - ex[:, 3]).mean() / np.abs(ex[:, 3]).mean()), nstep=a.nstep, nregrid=a.nregrid,
               boxes=[len(lev.boxes) for lev in a.levels], zones=[sum(b.n[0] * b.n[1] * b.n[2] for b in lev.boxes) for lev in a.levels],
               mass_drift=float(a.composite_sum(0) - m0), time=a.time, line=line)
    return res, a


if __name__ == "__main__":
    for case in (sys.argv[1:] or list(CASES)):
        res, _ = run(case)
        res.pop("line")
        print(case, res)
