#!/usr/bin/env python3
"""Randomised campaign for the `contract` build: castro_amd.Castro(numerics="contract") against the oracle's level driver on random
grids, boundaries (outflow, Symmetry, walls), options -- above all the ones that decide whether a run keeps the five-variable path
(ppm_type, plm_iorder, plm_limiter, use_pslope, use_flattening, first_order_hydro, sources, solvers) --, Sedov or Sod, a few steps:
every conserved component within rtol 1e-10 of the oracle (max |a - b| <= 1e-10 max |b|, the momenta against the largest of the
three), the same dt to 1e-10 -- OR within the run's own conditioning: the scheme has discrete switches (flattening's shifted
stencil and shock test, limiter sign tests), and a coarse grid with a blast a few zones wide sits on their ties: there ONE ULP in the
initial (rho e) moves the EXACT build by 1e-7 ... 1e-4 after a single step (measured: profiles/r06r_*).  So the oracle runs again from FUZZ_NPERT (4) states
perturbed by one ulp each, and a case counts as a mismatch only if the contract build is farther from the oracle
than 1e-10 AND than 100 x that perturbed oracle run (and that run itself has stayed within 1e-9 of the unperturbed one).  Both drivers start from the SAME initial state (the oracle's).
The identities of the five-variable path hold under conditions; this is the net under them: it found the
plm_limiter = 1 / use_pslope = 1 case.   usage: tools/fuzz_contract.py [ncases] [seed]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import castro_amd
from oracle import oracle_lib as oracle

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
only = set(int(x) for x in sys.argv[3].split(",")) if len(sys.argv) > 3 else None      # replay these cases only (verbose)
import os
NUM = os.environ.get("FUZZ_NUMERICS", "contract")
EXTRA = os.environ.get("FUZZ_EXTRA", "0") == "1"
BIG = os.environ.get("FUZZ_BIG", "0") == "1"
NPERT = int(os.environ.get("FUZZ_NPERT", "1" if BIG else "4"))      # oracle runs from states one ulp away        # "exact": the other build through the same campaign (must match bit for bit)
RTOL = 1.e-10
bad, worst, lean, skipped, illcond, onesided = 0, 0.0, 0, 0, 0, 0
for case in range(ncases):
    n = tuple(int(rng.integers(10, 21)) for _ in range(3))
    if BIG:        # FUZZ_BIG=1: boxes of 96 .. 112 zones a side -- the tile form of the transverse kernel (>= 96 rows) with walls and options
        n = tuple(int(rng.integers(96, 113)) for _ in range(3))
    bcs = [int(rng.choice([2, 2, 3, 4])) for _ in range(6)]
    pkw = dict(ppm_type=int(rng.integers(0, 2)), plm_iorder=int(rng.choice([1, 2, 2])), plm_limiter=int(rng.choice([1, 2, 2])),
               use_pslope=int(rng.integers(0, 2)), use_flattening=int(rng.choice([0, 1, 1])), first_order_hydro=int(rng.choice([0, 0, 0, 1])),
               riemann_solver=int(rng.choice([0, 0, 0, 2])), hybrid_riemann=int(rng.choice([0, 0, 0, 1])),
               cfl=float(rng.choice([0.5, 0.8])), init_shrink=float(rng.choice([0.1, 1.0])), change_max=float(rng.choice([1.1, 1.3])),
               difmag=float(rng.choice([0.1, 0.0])), source_term_predictor=int(rng.choice([0, 0, 1])))
    if EXTRA:      # FUZZ_EXTRA=1: the rarer options too (their own draws: the default campaign keeps its random stream)
        pkw.update(ppm_temp_fix=int(rng.choice([0, 0, 2])), transverse_reset_rhoe=int(rng.choice([0, 0, 1])),
                   transverse_reset_density=int(rng.choice([1, 1, 0])), transverse_use_eos=int(rng.choice([0, 0, 1])),
                   limit_fluxes_on_small_dens=int(rng.choice([0, 0, 1])), limit_fluxes_on_large_vel=int(rng.choice([0, 0, 1])),
                   speed_limit=float(rng.choice([0.0, 3.0])))
        if rng.integers(0, 6) == 0:
            pkw.update(riemann_solver=1, cg_blend=int(rng.integers(0, 3)))
    grav = bool(rng.integers(0, 3) == 0)
    rot = bool(rng.integers(0, 4) == 0)
    if not (grav or rot):
        pkw["source_term_predictor"] = 0
    gst, cg = int(rng.integers(1, 5)), float(rng.choice([-1.0, -5.0]))
    rkw = dict(center=(0.5, 0.5, 0.5), rot_source_type=int(rng.integers(1, 5)), implicit_rotation_update=int(rng.integers(0, 2)))
    per, ax = float(rng.choice([0.5, 5.0])), int(rng.integers(1, 4))
    ckw = dict(lo_bc=tuple(bcs[:3]), hi_bc=tuple(bcs[3:]))
    if only is not None and case not in only:      # replay the draws below without computing
        if str(rng.choice(["sedov", "sod"])) == "sod":
            rng.integers(1, 4)
        rng.choice([-1.0, 1.0], size=(n[2], n[1], n[0]))
        rng.integers(4, 13)
        continue
    c = castro_amd.Castro(n, params=castro_amd.default_params(**pkw), do_grav=grav, const_grav=cg, grav_source_type=gst,
                          rotation=castro_amd.make_rotation(per, ax, **rkw) if rot else None, numerics=NUM, **ckw)
    lev = oracle.Level(n, oracle.make_geom(n, **ckw), oracle.default_params(**pkw), nthreads=0 if BIG else 4)
    if grav:
        lev.set_gravity(cg, gst)
    if rot:
        lev.set_rotation(oracle.make_rotation(per, ax, **rkw))
    prob = str(rng.choice(["sedov", "sod"]))
    if prob == "sedov":
        c.initData("sedov", r_init=0.15, nsub=4); lev.init_sedov(r_init=0.15, nsub=4)
    else:
        idir = int(rng.integers(1, 4))
        c.initData("sod", rho_l=1.0, u_l=0.0, p_l=1.0, rho_r=0.125, u_r=0.0, p_r=0.1, idir=idir, frac=0.5)
        lev.init_sod(1.0, 0.0, 1.0, 0.125, 0.0, 0.1, idir=idir)
    # the same initial state on both sides, and a second oracle run one ulp away from it
    S0 = lev.state().copy()
    c.set_state(S0.copy())
    bump = 1.0 + 2.2e-16 * rng.choice([-1.0, 1.0], size=S0[5].shape)
    levs2 = []
    prng = np.random.default_rng(1000 + case)           # the further perturbations draw from a stream of their own
    for kpert in range(NPERT):
        lev2 = oracle.Level(n, oracle.make_geom(n, **ckw), oracle.default_params(**pkw), nthreads=0 if BIG else 4)
        if grav:
            lev2.set_gravity(cg, gst)
        if rot:
            lev2.set_rotation(oracle.make_rotation(per, ax, **rkw))
        S1 = S0.copy()
        S1[5] *= bump if kpert == 0 else 1.0 + 2.2e-16 * prng.choice([-1.0, 1.0], size=S0[5].shape)
        S1[4] = S1[4] - S0[5] + S1[5]
        lev2.state()[...] = S1
        oracle.lib().ora_level_post_init(lev2.h)
        levs2.append(lev2)
    oracle.lib().ora_level_post_init(lev.h)
    info = "n=%s bc=%s grav=%s rot=%s %s %s" % (n, bcs, (grav, gst, cg) if grav else None, (per, ax, rkw) if rot else None, prob, pkw)
    c.hydro.profile(True); c.hydro.profile_reset()
    nsteps = int(rng.integers(4, 13))
    dtdev = 0.0
    gave_up = None
    for step in range(nsteps):
        ea = eb = None
        try:
            da = c.step(0.5)
        except Exception as e:          # AdvanceFailure (too many subcycles, ...): the oracle's driver must give up alike
            ea = type(e).__name__
        try:
            db = lev.step(0.5)
        except Exception as e:
            eb = type(e).__name__
        for lev2 in levs2:
            try:
                lev2.step(0.5)
            except Exception:
                pass
        if ea or eb:
            gave_up = (ea, eb)
            break
        dtdev = max(dtdev, abs(da - db) / db)
    torch.cuda.synchronize()
    if only is not None:
        print("case %d: %s  gave up: %s  failures: %s" % (case, info, gave_up, getattr(c, "last_failure", "")))
    if gave_up is not None:
        if not (gave_up[0] and gave_up[1]):
            # one driver ran out of subcycles (NaNs, negative densities) and the other did not: seen only in configurations where BOTH
            # builds retry in every step and the `exact` build itself ends the same way for another split direction of the same
            # case (strong constant gravity or a rotation period of 0.5 on a cold gas): reported, not counted as a mismatch
            onesided += 1
            print("one driver gave up, case %d: %s  %s" % (case, gave_up, info))
        else:
            skipped += 1
        lev.close(); [l2.close() for l2 in levs2]; c.close()
        continue
    rep = c.hydro.profile_report()
    lean += int("k_trans1_fold" in rep and "k_trans1" not in rep)
    if BIG:
        print("case %d: n=%s %s  kernels %s" % (case, n, {k: pkw[k] for k in ("ppm_type", "riemann_solver", "hybrid_riemann", "use_pslope")}, sorted(k for k in rep if "trans1" in k or "trace" in k)))
    got, want = c.S_new().cpu().numpy(), lev.state()
    dev = {k: np.abs(got[k] - want[k]).max() / max(np.abs(want[k]).max(), 1e-300) for k in range(8)}
    mom = max(max(np.abs(want[k]).max() for k in (1, 2, 3)), 1e-300)
    for k in (1, 2, 3):
        dev[k] = np.abs(got[k] - want[k]).max() / mom
    sens = 0.0
    for lev2 in levs2:          # the largest move of the oracle under NPERT independent one-ulp perturbations (a flip is an event, not a slope)
        pert = lev2.state()
        s_ = max(np.abs(pert[k] - want[k]).max() / (mom if k in (1, 2, 3) else max(np.abs(want[k]).max(), 1e-300)) for k in range(8))
        sens = s_ if s_ != s_ else max(sens, s_)
        if sens != sens:
            break
    m = max(max(dev.values()), dtdev)
    if m <= RTOL:
        worst = max(worst, m)
    elif m <= 100.0 * sens or sens >= 1.e-9 or sens != sens:   # one ulp moves the oracle itself by 1e-9 and more (or into NaNs): the run sits on the switches
        illcond += 1
    else:
        bad += 1
        print("MISMATCH case %d: deviation %.2e (dt %.1e; the oracle one ulp away: %.2e) after %d steps  %s  %s"
              % (case, max(dev.values()), dtdev, sens, nsteps, info, sorted(rep)))
    lev.close(); [l2.close() for l2 in levs2]; c.close()
print("cases %d, mismatches %d, within rtol %g: worst %.2e; beyond it but within 100 x the oracle's own one-ulp sensitivity: %d; runs on the five-variable path %d, given up by both drivers alike %d, by one of them %d" % (ncases, bad, RTOL, worst, illcond, lean, skipped, onesided))
