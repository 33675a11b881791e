#!/usr/bin/env python3
"""Sedov with tag-driven AMR (128^3 base + 2 refined levels, BASELINE config 4) run to the reference's stop_time and
compared with the analytic table on the composite grid: radially binned density from the finest data available."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import castro_amd
from tests.util import analytic_bins

def run(n=128, nlev=2, verbose=True, amr_kw=None, params_kw=None):
    kw = dict(refine=[("density", "gradient", 0.05), ("rho_E", "relative_gradient", 0.5)], regrid_int=2, n_error_buf=2,
              blocking_factor=8, max_level=nlev)
    kw.update(amr_kw or {})
    a = castro_amd.CastroAmr((n, n, n), params=castro_amd.default_params(**(params_kw or {})), **kw)
    a.initData("sedov")
    m0, e0 = a.composite_sum(0), a.composite_sum(4)
    t0 = time.time()
    a.evolve(0.01)
    torch.cuda.synchronize()
    wall = time.time() - t0
    if verbose:
        print("%d coarse steps, %d regrids, %.1f s; boxes per level:" % (a.nstep, a.nregrid, wall), [len(lev.boxes) for lev in a.levels],
              "zones per level:", [sum(b.n[0] * b.n[1] * b.n[2] for b in lev.boxes) for lev in a.levels])
    drift = ((a.composite_sum(0) - m0) / m0, (a.composite_sum(4) - e0) / e0)
    if verbose:
        print("composite mass drift %.2e, energy drift %.2e (relative)" % drift)

    # radial bins at the finest resolution; every level contributes its zones that no finer level covers
    nf = n * 2 ** (len(a.levels) - 1)
    dxf = 1.0 / nf
    nb = int(0.36 / dxf)
    tot = torch.zeros(nb + 1, dtype=torch.float64, device="cuda")
    cnt = torch.zeros(nb + 1, dtype=torch.float64, device="cuda")
    for l, lev in enumerate(a.levels):
        dx = lev.geom.dx[0]
        for b in lev.boxes:
            rho = b.S_new()[0]
            ax = [(torch.arange(b.lo[d], b.hi[d] + 1, device="cuda", dtype=torch.float64) + 0.5) * dx - 0.5 for d in range(3)]
            r = torch.sqrt(ax[0][None, None, :] ** 2 + ax[1][None, :, None] ** 2 + ax[2][:, None, None] ** 2)
            w = torch.full_like(rho, dx ** 3)
            if l + 1 < len(a.levels):                      # zones under a finer box do not count
                o = b.lo
                for f in a.levels[l + 1].boxes:
                    p = tuple(max(f.pbox[0][d], b.lo[d]) for d in range(3))
                    q = tuple(min(f.pbox[1][d], b.hi[d]) for d in range(3))
                    if all(p[d] <= q[d] for d in range(3)):
                        w[p[2] - o[2]:q[2] - o[2] + 1, p[1] - o[1]:q[1] - o[1] + 1, p[0] - o[0]:q[0] - o[0] + 1] = 0.0
            idx = torch.clamp((r / dxf).long(), max=nb).ravel()
            tot.index_add_(0, idx, (rho * w).ravel())
            cnt.index_add_(0, idx, w.ravel())
    prof = (tot / cnt)[:nb].cpu().numpy()
    edges = np.arange(nb + 1) * dxf
    rc = 0.5 * (edges[1:] + edges[:-1])
    table = np.loadtxt(os.path.join("tests", "golden", "reference_verification", "spherical_sedov.dat"))
    ref = analytic_bins(edges, table, 2, 1.0)
    ok = np.isfinite(prof)
    wgt = rc ** 2
    res = dict(l1=float((np.abs(prof - ref)[ok] * wgt[ok]).sum() / (ref[ok] * wgt[ok]).sum()), peak=float(np.nanmax(prof)),
               r_peak=float(rc[np.nanargmax(prof)]), r_shock=float(table[np.argmax(table[:, 2]), 1]), drift=drift,
               nstep=a.nstep, nregrid=a.nregrid, levels=[[b.n for b in lev.boxes] for lev in a.levels] if any(len(lev.boxes) > 1 for lev in a.levels)
               else [lev.n for lev in a.levels], boxes=[len(lev.boxes) for lev in a.levels],
               zones=[sum(b.n[0] * b.n[1] * b.n[2] for b in lev.boxes) for lev in a.levels], dx_fine=dxf, seconds=wall)
    if verbose:
        print("composite density: L1 error vs analytic %.4f, peak %.2f at r = %.4f (analytic shock at %.4f)" %
              (res["l1"], res["peak"], res["r_peak"], res["r_shock"]))
    return res


# Exec/hydro_tests/Sedov/inputs.3d.sph.testsuite: 32^3 base, amr.max_level = 3, PLM, blocking_factor 8, max_grid_size 32,
# density / pressure indicators, AMReX's default n_error_buf = 1
TESTSUITE = dict(n=32, nlev=3, params_kw=dict(ppm_type=0),
                 amr_kw=dict(refine=[("density", "value_greater", 3.0), ("density", "gradient", 0.01),
                                     ("pressure", "value_greater", 3.0), ("pressure", "gradient", 0.01)],
                             regrid_int=2, n_error_buf=1, blocking_factor=8, cluster=True, grid_eff=0.7, max_grid_size=32))

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "testsuite":
        run(**TESTSUITE)
    else:
        run(int(sys.argv[1]) if len(sys.argv) > 1 else 128, int(sys.argv[2]) if len(sys.argv) > 2 else 2)
