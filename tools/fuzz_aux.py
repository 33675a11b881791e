#!/usr/bin/env python3
"""Randomised parity campaign for the entry points around the hydro call (clean_state, estdt, physical-boundary fill,
derived fields, AMR building blocks, gravity and rotation sources): each operation on random boxes / states / options
on the device and through the oracle backend, bit for bit.  usage: tools/fuzz_aux.py [ncases] [seed]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import castro_amd
from castro_amd._lib import DERIVE_IDS
from castro_amd.hydro import HipHydro
from oracle import oracle_lib as oracle
from tests.oracle_backend import OracleBackend
from tests.util import physical_state

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
hip, ora = HipHydro(0), OracleBackend()
bad = 0
count = {}


def rbox(lo_min=-6, lo_max=6, n_min=1, n_max=12):
    lo = [int(rng.integers(lo_min, lo_max + 1)) for _ in range(3)]
    n = [int(rng.integers(n_min, n_max + 1)) for _ in range(3)]
    return tuple(lo), tuple(lo[d] + n[d] - 1 for d in range(3))


def grow(b, g):
    return tuple(x - g for x in b[0]), tuple(x + g for x in b[1])


def shape(b, nc):
    return (nc,) + tuple(b[1][d] - b[0][d] + 1 for d in (2, 1, 0))


def both(arr):
    return torch.from_numpy(arr.copy()).cuda(), torch.from_numpy(arr.copy())


def check(name, pairs, info):
    global bad
    count[name] = count.get(name, 0) + 1
    torch.cuda.synchronize()
    for what, (a, b) in pairs.items():
        a = a.cpu().numpy() if hasattr(a, "cpu") else np.asarray(a)
        b = b.numpy() if hasattr(b, "numpy") else np.asarray(b)
        if not np.array_equal(a, b, equal_nan=True):
            bad += 1
            print("MISMATCH %s %s: %d entries, max abs %.3e  %s" % (name, what, int((a != b).sum()), float(np.nanmax(np.abs(a - b))), info))
            return


def state_for(box, **kw):
    U = physical_state(rng, box[0], box[1], smooth=bool(rng.integers(0, 2)), vel=float(rng.choice([0.3, 1.5, 3.0])),
                       jump=bool(rng.integers(0, 2)), **kw)
    if rng.integers(0, 3) == 0:
        U[0] *= rng.uniform(0.01, 1.0, size=U[0].shape)       # very low densities next to normal ones
    return U


for case in range(ncases):
    pkw = dict(small_dens=float(rng.choice([1e-200, 0.05, 0.3])), speed_limit=float(rng.choice([0.0, 0.0, 0.8])),
               small_temp=float(rng.choice([1e-200, 1e-3])), dual_energy_eta2=float(rng.choice([1e-4, 0.1])))
    Ph, Po = castro_amd.default_params(**pkw), oracle.default_params(**pkw)
    bx = rbox()
    gb = grow(bx, 4)
    n = [bx[1][d] - bx[0][d] + 1 for d in range(3)]
    dlo = tuple(bx[0][d] - int(rng.integers(0, 3)) * int(rng.integers(0, 2)) for d in range(3))      # domain may extend past the box
    dn = [bx[1][d] - dlo[d] + 1 + int(rng.integers(0, 3)) * int(rng.integers(0, 2)) for d in range(3)]
    bcs = [int(rng.choice([0, 1, 2, 3, 4, 5])) for _ in range(6)]
    for d in range(3):                                          # periodic comes in pairs
        if bcs[d] == 0 or bcs[d + 3] == 0:
            bcs[d] = bcs[d + 3] = 0
    gk = dict(lo_bc=tuple(bcs[:3]), hi_bc=tuple(bcs[3:]), domlo=dlo)
    ph = [float(x) * float(rng.choice([0.01, 0.05])) for x in dn]
    Gh, Go = castro_amd.make_geom(dn, prob_hi=ph, **gk), oracle.make_geom(dn, probhi=ph, **gk)
    info = "box %s domlo %s n %s bc %s %s" % (bx, dlo, dn, bcs, pkw)

    # clean_state x ntimes, estdt, fused reduce
    U = state_for(gb)
    nt = int(rng.integers(1, 4))
    a, b = both(U)
    hip.clean_state(a, gb, bx[0], bx[1], Ph, ntimes=nt); ora.clean_state(b, gb, bx[0], bx[1], Po, ntimes=nt)
    check("clean_state", {"U": (a, b)}, info)
    ra, rb = torch.full((3,), 1e200, dtype=torch.float64, device="cuda"), torch.full((3,), 1e200, dtype=torch.float64)
    hip.estdt_cfl(a, gb, bx[0], bx[1], Gh, Ph, ra); ora.estdt_cfl(b, gb, bx[0], bx[1], Go, Po, rb)
    check("estdt", {"red": (ra, rb)}, info)
    a, b = both(U)
    ra.fill_(1e200); rb.fill_(1e200)
    hip.clean_state_reduce(a, gb, bx[0], bx[1], Gh, Ph, ra, ntimes=nt); ora.clean_state_reduce(b, gb, bx[0], bx[1], Go, Po, rb, ntimes=nt)
    check("clean_state_reduce", {"U": (a, b), "red": (ra, rb)}, info)

    # physical-boundary fill of a grown FAB (ghost zones of periodic directions hold data already)
    a, b = both(U)
    hip.bc_fill(a, gb, Gh); ora.bc_fill(b, gb, Go)
    check("bc_fill", {"U": (a, b)}, info)

    # derived fields
    g1 = grow(bx, 1)
    name = str(rng.choice(list(DERIVE_IDS)))
    da, db = both(np.zeros(shape(bx, 2)))
    ctr = tuple(float(x) for x in rng.uniform(0, 0.3, size=3))
    a, b = both(np.abs(U) + 0.1 if name == "logden" else U)
    hip.derive(name, a, gb, da, bx, 1, bx[0], bx[1], Gh, Ph, ctr); ora.derive(name, b, gb, db, bx, 1, bx[0], bx[1], Go, Po, ctr)
    if name == "logden":                                       # device log10 vs libm: last-bit differences allowed
        torch.cuda.synchronize()
        assert np.allclose(da.cpu().numpy(), db.numpy(), rtol=1e-14, atol=0), info
    else:
        check("derive " + name, {"der": (da, db)}, info)

    # AMR building blocks
    fb = rbox()
    cb = (tuple(x // 2 - 1 for x in fb[0]), tuple(x // 2 + 1 for x in fb[1]))
    C = state_for(cb)
    ca, cbk = both(C)
    fa, fbk = both(np.zeros(shape(fb, 8)))
    hip.cc_interp(ca, cb, fa, fb, fb[0], fb[1], 8); ora.cc_interp(cbk, cb, fbk, fb, fb[0], fb[1], 8)
    check("cc_interp", {"fine": (fa, fbk)}, "fine %s" % (fb,))
    alo = tuple(-((-x) // 2) for x in fb[0]); ahi = tuple((x + 1) // 2 - 1 for x in fb[1])
    if all(alo[d] <= ahi[d] for d in range(3)):
        aa, ab = both(np.zeros(shape((alo, ahi), 8)))
        hip.avgdown(fa, fb, aa, (alo, ahi), alo, ahi, 8); ora.avgdown(fbk, fb, ab, (alo, ahi), alo, ahi, 8)
        check("avgdown", {"crse": (aa, ab)}, "fine %s" % (fb,))
    kind, val = int(rng.integers(0, 4)), float(rng.choice([0.01, 0.5, 1.0]))
    comp = int(rng.integers(0, 8))
    ta, tb = both(np.zeros(shape(bx, 1)))
    a, b = both(U)
    hip.error_tag(a, gb, comp, ta, bx, bx[0], bx[1], kind, val); ora.error_tag(b, gb, comp, tb, bx, bx[0], bx[1], kind, val)
    check("error_tag", {"tags": (ta, tb)}, info + " kind %d comp %d value %g" % (kind, comp, val))

    # gravity and rotation sources (old-time form) on the same state
    sb = grow(bx, 3)
    gst = int(rng.integers(1, 5))
    grav = tuple(float(x) for x in rng.normal(size=3))
    dt = float(rng.choice([1e-3, 1e-2]))
    sa, sbk = both(np.zeros(shape(sb, 7)))
    hip.old_gravity_source(a, gb, sa, sb, bx[0], bx[1], grav, gst, dt); ora.old_gravity_source(b, gb, sbk, sb, bx[0], bx[1], grav, gst, dt)
    check("old_gravity_source", {"src": (sa, sbk)}, info)
    rkw = dict(center=tuple(float(x) for x in rng.uniform(0, 0.2, size=3)), rot_source_type=int(rng.integers(1, 5)),
               implicit_rotation_update=int(rng.integers(0, 2)), include_centrifugal=int(rng.integers(0, 2)),
               include_coriolis=int(rng.integers(0, 2)))
    per, ax = float(rng.choice([0.5, 5.0])), int(rng.integers(1, 4))
    Rh, Ro = castro_amd.make_rotation(per, ax, **rkw), oracle.make_rotation(per, ax, **rkw)
    sa, sbk = both(np.zeros(shape(sb, 7)))
    hip.old_rotation_source(a, gb, sa, sb, bx[0], bx[1], Rh, Gh, dt); ora.old_rotation_source(b, gb, sbk, sb, bx[0], bx[1], Ro, Go, dt)
    check("old_rotation_source", {"src": (sa, sbk)}, info + " %s" % (rkw,))

print("cases %d, mismatches %d, operations: %s" % (ncases, bad, count))
sys.exit(1 if bad else 0)
