#!/usr/bin/env python3
"""Randomised driver-level campaign: castro_amd.Castro on the device against the oracle's own level driver (C, an
independent restatement of Castro::advance / do_advance_ctu / retry / dt control): random grids, boundaries (inflow /
outflow, walls), options, problems, constant gravity and rotation on or off, several steps -- same dt sequence, same
retry counts, same state, bit for bit.  usage: tools/fuzz_driver.py [ncases] [seed] [only]
`only` = comma-separated case numbers: the random stream is replayed, the other cases are not computed."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import castro_amd
from oracle import oracle_lib as oracle

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
only = set(int(x) for x in sys.argv[3].split(",")) if len(sys.argv) > 3 else None
bad = 0
for case in range(ncases):
    n = tuple(int(rng.integers(8, 17)) for _ in range(3))
    bcs = [int(rng.choice([2, 2, 3, 4, 5, 1])) for _ in range(6)]     # the oracle's level driver has no periodic wrap
    pkw = dict(ppm_type=int(rng.integers(0, 2)), riemann_solver=int(rng.choice([0, 0, 1, 2])), hybrid_riemann=int(rng.integers(0, 2)),
               cfl=float(rng.choice([0.5, 0.8])), init_shrink=float(rng.choice([0.01, 0.1, 1.0])), change_max=float(rng.choice([1.05, 1.1, 1.3])),
               transverse_reset_rhoe=int(rng.integers(0, 2)), ppm_temp_fix=int(rng.choice([0, 2])),
               limit_fluxes_on_small_dens=int(rng.integers(0, 2)), difmag=float(rng.choice([0.1, 0.0])))
    grav = bool(rng.integers(0, 3) == 0)
    rot = bool(rng.integers(0, 3) == 0)
    gst, cg = int(rng.integers(1, 5)), float(rng.choice([-1.0, -20.0]))
    rkw = dict(center=(0.5, 0.5, 0.5), rot_source_type=int(rng.integers(1, 5)), implicit_rotation_update=int(rng.integers(0, 2)))
    per, ax = float(rng.choice([0.05, 1.0])), int(rng.integers(1, 4))
    ckw = dict(lo_bc=tuple(bcs[:3]), hi_bc=tuple(bcs[3:]))
    if only is not None and case not in only:      # replay the draws below without computing
        if str(rng.choice(["sedov", "sod"])) == "sod":
            rng.integers(1, 4)
        rng.integers(3, 9)
        continue
    c = castro_amd.Castro(n, params=castro_amd.default_params(**pkw), do_grav=grav, const_grav=cg, grav_source_type=gst,
                          rotation=castro_amd.make_rotation(per, ax, **rkw) if rot else None, **ckw)
    lev = oracle.Level(n, oracle.make_geom(n, **ckw), oracle.default_params(**pkw), nthreads=4)
    if grav:
        lev.set_gravity(cg, gst)
    if rot:
        lev.set_rotation(oracle.make_rotation(per, ax, **rkw))
    prob = str(rng.choice(["sedov", "sod"]))
    if prob == "sedov":
        c.initData("sedov", r_init=0.12, nsub=4); lev.init_sedov(r_init=0.12, nsub=4)
    else:
        idir = int(rng.integers(1, 4))
        c.initData("sod", rho_l=1.0, u_l=0.0, p_l=1.0, rho_r=0.125, u_r=0.0, p_r=0.1, idir=idir, frac=0.5)
        lev.init_sod(1.0, 0.0, 1.0, 0.125, 0.0, 0.1, idir=idir)
    info = "n=%s bc=%s grav=%s rot=%s %s %s" % (n, bcs, (grav, gst, cg), (rot, per, ax, rkw), prob, pkw)
    ok = True
    try:
        for step in range(int(rng.integers(3, 9))):
            da, db = c.step(0.5), lev.step(0.5)
            if da != db:
                ok = False
                print("MISMATCH case %d: dt %r vs %r at step %d  %s" % (case, da, db, step, info))
                break
    except Exception as e:                                     # both refuse alike (subcycles too short, ...)? report
        print("case %d raised %s: %s  %s" % (case, type(e).__name__, e, info))
        lev.close()
        continue
    torch.cuda.synchronize()
    if ok and not np.array_equal(c.S_new().cpu().numpy(), lev.state(), equal_nan=True):
        ok = False
        d = np.abs(c.S_new().cpu().numpy() - lev.state())
        print("MISMATCH case %d: state, %d entries, max abs %.3e  %s" % (case, int((d > 0).sum()), float(d.max()), info))
    bad += not ok
    lev.close()
print("cases %d, mismatching %d" % (ncases, bad))
sys.exit(1 if bad else 0)
