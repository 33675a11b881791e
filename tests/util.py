"""Shared helpers for the parity tests (tests only)."""
import numpy as np
import torch


def physical_state(rng, lo, hi, smooth=True, vel=1.0, jump=True):
    """A random-but-physical conserved state (NUM_STATE, nz, ny, nx) on box [lo,hi]:
    smooth background + an oblique pressure/density jump so limiters, flattening and every
    Riemann wave pattern are exercised.  Thermodynamically consistent (gamma = 1.4 in E)."""
    nx, ny, nz = (hi[d] - lo[d] + 1 for d in range(3))
    z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    ph = rng.uniform(0, 2 * np.pi, size=6)
    rho = 1.0 + 0.3 * np.sin(0.37 * x + ph[0]) * np.cos(0.29 * y + ph[1]) + 0.2 * np.sin(0.41 * z + ph[2])
    p = 1.0 + 0.4 * np.cos(0.31 * x + 0.23 * y + ph[3]) + 0.2 * np.sin(0.33 * z + ph[4])
    if jump:
        s = (x - nx / 2) + 0.6 * (y - ny / 2) - 0.4 * (z - nz / 2)
        rho = np.where(s > 0, rho * 0.2, rho)
        p = np.where(s > 0, p * 0.05, p)
    u = vel * (0.5 * np.sin(0.21 * x + ph[5]) + rng.uniform(-0.05, 0.05, size=rho.shape))
    v = vel * (0.4 * np.cos(0.27 * y + ph[0]) + rng.uniform(-0.05, 0.05, size=rho.shape))
    w = vel * (0.3 * np.sin(0.19 * z + ph[1]) + rng.uniform(-0.05, 0.05, size=rho.shape))
    if not smooth:
        rho = rho * rng.uniform(0.8, 1.25, size=rho.shape)
        p = p * rng.uniform(0.8, 1.25, size=rho.shape)
    g = 1.4
    eint = p / (g - 1.0)
    U = np.zeros((8, nz, ny, nx))
    U[0] = rho
    U[1] = rho * u
    U[2] = rho * v
    U[3] = rho * w
    U[4] = eint + 0.5 * rho * (u * u + v * v + w * w)
    U[5] = eint
    U[6] = 1.0
    U[7] = rho * rng.uniform(0.999, 1.0, size=rho.shape)
    return np.ascontiguousarray(U)


def max_rel(a, b):
    d = np.abs(a - b).max()
    s = max(np.abs(b).max(), 1e-300)
    return d / s


def ulp_report(a, b):
    """(#different entries, max abs diff, max rel diff)"""
    ne = int((a != b).sum())
    return ne, float(np.abs(a - b).max()), float(max_rel(a, b))


# ---- radial profiles of a Sedov run against the reference's analytic table (the four panels of
# Exec/hydro_tests/Sedov/testsuite_analysis/sedov_3d_sph.py) ----
def sedov_profiles(c, gamma=1.4, rmax=0.36):
    n = c.n_cell[0]
    S = c.S_new()
    rho = S[0]
    x = (torch.arange(n, device=rho.device, dtype=torch.float64) + 0.5) / n - 0.5
    X, Y, Z = x[None, None, :], x[None, :, None], x[:, None, None]
    r = torch.sqrt(X ** 2 + Y ** 2 + Z ** 2)
    vr = (S[1] * X + S[2] * Y + S[3] * Z) / (rho * r)
    p = (gamma - 1.0) * S[5]
    e = S[5] / rho
    dx = 1.0 / n
    nb = int(rmax / dx)
    idx = torch.clamp((r / dx).long(), max=nb).ravel()
    cnt = torch.zeros(nb + 1, device=rho.device, dtype=torch.float64).index_add_(0, idx, torch.ones_like(rho).ravel())
    out = {}
    for name, f in (("density", rho), ("velocity", vr), ("pressure", p), ("eint", e)):
        tot = torch.zeros(nb + 1, device=rho.device, dtype=torch.float64).index_add_(0, idx, f.ravel())
        out[name] = (tot / cnt)[:nb].cpu().numpy()
    edges = np.arange(nb + 1) * dx
    return edges, out


def analytic_bins(edges, table, col, fill):
    """Volume average of column `col` of the analytic table over the radial bins."""
    r_ex, f_ex = table[:, 1], table[:, col]
    rf = np.linspace(0.0, edges[-1], 200001)
    ff = np.interp(rf, r_ex, f_ex, right=fill)
    cum = np.concatenate([[0.0], np.cumsum(0.5 * (ff[1:] * rf[1:] ** 2 + ff[:-1] * rf[:-1] ** 2) * np.diff(rf))])
    vol = rf ** 3 / 3.0
    return np.diff(np.interp(edges, rf, cum)) / np.diff(np.interp(edges, rf, vol))


def sedov_l1_errors(c, table):
    edges, prof = sedov_profiles(c)
    rc = 0.5 * (edges[1:] + edges[:-1])
    w = rc ** 2
    res = {}
    for name, col, fill in (("density", 2, 1.0), ("velocity", 5, 0.0), ("pressure", 4, 1.e-5)):
        ref = analytic_bins(edges, table, col, fill)
        res[name] = float((np.abs(prof[name] - ref) * w).sum() / (np.abs(ref) * w).sum())
    return res, rc, prof


