#!/usr/bin/env python3
"""Randomised parity campaign: one construct_ctu_hydro_source call on random boxes, states, boundary conditions and option
combinations, HIP against the oracle, bit for bit.  usage: tools/fuzz_parity.py [ncases] [seed] [only]
`only` = comma-separated case numbers: the random stream is replayed, the other cases are not computed."""
import itertools
import sys

import numpy as np

sys.path.insert(0, ".")
from castro_amd.hydro import HipHydro
from oracle import oracle_lib as oracle
from tests.test_gpu_parity import _run_both
from tests.util import physical_state, ulp_report

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
only = set(int(x) for x in sys.argv[3].split(",")) if len(sys.argv) > 3 else None
rng = np.random.default_rng(seed)
hip = HipHydro(0)
bad = 0
stats = {}
for case in range(ncases):
    n = [int(rng.integers(1, 15)) for _ in range(3)]
    lo = [int(rng.integers(-3, 4)) for _ in range(3)]
    bxlo, bxhi = tuple(lo), tuple(lo[d] + n[d] - 1 for d in range(3))
    sb_lo, sb_hi = tuple(x - 4 for x in bxlo), tuple(x + 4 for x in bxhi)
    pkw = dict(ppm_type=int(rng.integers(0, 2)), riemann_solver=int(rng.choice([0, 0, 1, 2])), hybrid_riemann=int(rng.integers(0, 2)),
               use_flattening=int(rng.choice([1, 1, 0])), first_order_hydro=int(rng.choice([0, 0, 0, 1])),
               transverse_use_eos=int(rng.integers(0, 2)), transverse_reset_density=int(rng.integers(0, 2)),
               transverse_reset_rhoe=int(rng.integers(0, 2)), ppm_temp_fix=int(rng.choice([0, 0, 2])),
               limit_fluxes_on_small_dens=int(rng.choice([0, 0, 1])), limit_fluxes_on_large_vel=int(rng.choice([0, 0, 1])),
               speed_limit=float(rng.choice([0.0, 3.0])), small_dens=float(rng.choice([1e-200, 0.05])),
               difmag=float(rng.choice([0.1, 0.0])), cg_blend=int(rng.integers(0, 3)))
    if pkw["ppm_type"] == 0:
        pkw.update(plm_iorder=int(rng.choice([1, 2])), plm_limiter=int(rng.choice([1, 2])), use_pslope=int(rng.integers(0, 2)))
    bcs = [int(rng.choice([2, 2, 3, 4, 5, 1])) for _ in range(6)]
    U = physical_state(rng, sb_lo, sb_hi, smooth=bool(rng.integers(0, 2)), vel=float(rng.choice([0.3, 1.5, 3.0])),
                       jump=bool(rng.integers(0, 2)))
    if rng.integers(0, 3) == 0:                              # cold, kinetic-energy dominated
        ke = 0.5 * (U[1] ** 2 + U[2] ** 2 + U[3] ** 2) / U[0]
        U[5] *= 2e-3
        U[4] = U[5] + ke
    src = None
    src_box = None
    if rng.integers(0, 3) == 0:
        src_box = (tuple(x - 3 for x in bxlo), tuple(x + 3 for x in bxhi))
        shp = (7,) + tuple(src_box[1][d] - src_box[0][d] + 1 for d in (2, 1, 0))
        src = rng.normal(scale=0.3, size=shp) * (rng.uniform(size=shp) < rng.choice([1.0, 0.05]))
    tiles = None
    if rng.integers(0, 3) == 0:                              # the box as several tiles of one FAB (mfi.nodaltilebox semantics)
        cuts = [sorted(set([bxlo[d]] + ([int(x) for x in rng.integers(bxlo[d] + 1, bxhi[d] + 1, size=int(rng.integers(0, 3)))]
                                        if bxhi[d] > bxlo[d] else []))) for d in range(3)]
        tiles = []
        for kz, z0 in enumerate(cuts[2]):
            for ky, y0 in enumerate(cuts[1]):
                for kx, x0 in enumerate(cuts[0]):
                    hi = (cuts[0][kx + 1] - 1 if kx + 1 < len(cuts[0]) else bxhi[0],
                          cuts[1][ky + 1] - 1 if ky + 1 < len(cuts[1]) else bxhi[1],
                          cuts[2][kz + 1] - 1 if kz + 1 < len(cuts[2]) else bxhi[2])
                    tiles.append(((x0, y0, z0), hi))
    dx = tuple(float(x) for x in rng.choice([0.01, 0.02, 0.05], size=3))
    dt = float(rng.choice([2e-4, 8e-4, 2e-3]))
    flux_assign = bool(rng.integers(0, 2))
    if only is not None and case not in only:
        continue
    oracle.lib().ora_cg_aborts_reset()
    try:
        out = _run_both(hip, oracle, bxlo, bxhi, U, sb_lo, sb_hi, dt, dx=dx, pkw=pkw, src=src, src_box=src_box,
                        geom_kw=dict(lo_bc=tuple(bcs[:3]), hi_bc=tuple(bcs[3:])), flux_assign=flux_assign, hip_tiles=tiles)
    except AssertionError as e:                              # a state the reference would abort on (rho <= 0 in ctoprim)
        stats["skipped"] = stats.get("skipped", 0) + 1
        continue
    oracle.lib().ora_cg_aborts_count.restype = __import__("ctypes").c_long
    if oracle.lib().ora_cg_aborts_count() > 0 and only is None:  # cg_blend = 0 and no convergence: the reference calls amrex::Error
        stats["reference aborts"] = stats.get("reference aborts", 0) + 1
        continue
    if any(np.isnan(b).any() for _, b in out.values()) and only is None:   # the algorithm itself breaks down on this input
        stats["oracle NaN"] = stats.get("oracle NaN", 0) + 1   # (sqrt of a negative pressure in the HLL wave speeds, ...): not a
        continue                                               # parity case (still compared when asked for by number)
    worst = 0
    for k, (a, b) in out.items():
        if not np.array_equal(a, b, equal_nan=True):
            ne, ad, rd = ulp_report(a, b)
            worst = max(worst, ne)
            print("case %d %s: %d entries differ (max rel %.3e, NaN hip %d oracle %d)  n=%s bc=%s %s src=%s"
                  % (case, k, ne, rd, int(np.isnan(a).sum()), int(np.isnan(b).sum()), n, bcs, pkw, src is not None))
    bad += worst > 0
    if tiles is not None:
        stats["tiled"] = stats.get("tiled", 0) + 1
    key = (pkw["ppm_type"], pkw["riemann_solver"])
    stats[key] = stats.get(key, 0) + 1
print("cases %d, mismatching %d, by (ppm_type, riemann_solver): %s" % (ncases, bad, stats))
sys.exit(1 if bad else 0)
