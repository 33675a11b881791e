#!/usr/bin/env python3
"""An isothermal atmosphere at rest under constant gravity (rho = rho0 exp(-z/H), p = rho cT^2, H = cT^2/|g|) on a
single level and with a refined box in the middle: the sources on the refined level, their ghost zones and the reflux must
not disturb the balance more than the discretisation does.  Prints the largest Mach number after a fraction of a sound
crossing time.  usage: tools/amr_hse_validation.py [oracle]   (oracle: run on the CPU with the oracle backend)"""
import math
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import castro_amd

use_oracle = len(sys.argv) > 1 and sys.argv[1] == "oracle"
if use_oracle:
    from oracle import oracle_lib as O
    from tests.oracle_backend import OracleBackend
    mk, params = OracleBackend, O.default_params
else:
    mk, params = None, castro_amd.default_params

g, cT2, gamma = -1.0, 1.0, 1.4
n = (16, 16, 32) if use_oracle else (64, 64, 128)
H = cT2 / abs(g)


def fill(amr):
    for lev in amr.levels:
        for b in lev.boxes:
            S = b.S_new()
            dz = (b.geom.probhi[2] - b.geom.problo[2]) / ((2 ** lev.l) * n[2])
            z = (torch.arange(b.lo[2], b.hi[2] + 1, dtype=torch.float64, device=S.device) + 0.5) * dz
            # zone averages of the exponential, so that the discrete balance is second order
            rho = (torch.exp(-(z - 0.5 * dz) / H) - torch.exp(-(z + 0.5 * dz) / H)) * H / dz
            S.zero_()
            S[0] = rho[:, None, None]
            S[7] = S[0]
            S[5] = S[0] * cT2 / (gamma - 1.0)
            S[4] = S[5]
            b.clean_state(b.S_new_b, 1)


def run(tag, patches):
    kw = dict(prob_hi=(0.5, 0.5, 1.0), lo_bc=(0, 0, 3), hi_bc=(0, 0, 3), do_grav=True, const_grav=g, params=params(cfl=0.5))
    a = castro_amd.CastroAmr(n, patches=patches, make_hydro=mk, **kw)
    fill(a)
    m0 = a.composite_sum(0)
    t_end = 0.25 * 1.0 / math.sqrt(gamma * cT2)
    while a.time < t_end - 1e-14:
        a.step(t_end)
    out = []
    for lev in a.levels:
        mach = 0.0
        for b in lev.boxes:
            S = b.S_new()
            # away from the reflecting lids (their ghost zones are not in hydrostatic balance: the reference has HSE BCs for that)
            k0, k1 = max(0, (2 ** lev.l) * 4 - b.lo[2]), min(b.n[2], (2 ** lev.l) * (n[2] - 4) - b.lo[2])
            if k1 > k0:
                v = (S[1:4, k0:k1] / S[0:1, k0:k1]).abs().max().item()
                mach = max(mach, v / math.sqrt(gamma * cT2))
        out.append(mach)
    print("%-28s steps %4d  max Mach per level %s  mass drift %.1e" % (tag, a.nstep, ["%.2e" % x for x in out], a.composite_sum(0) / m0 - 1))


q = [n[0] // 4, n[1] // 4, 3 * n[2] // 8]
run("single level", [])
run("refined box in the middle", [((q[0], q[1], q[2]), (n[0] - q[0] - 1, n[1] - q[1] - 1, n[2] - q[2] - 1))])
run("two refined levels", [((q[0], q[1], q[2]), (n[0] - q[0] - 1, n[1] - q[1] - 1, n[2] - q[2] - 1)),
                           ((2 * q[0] + 4, 2 * q[1] + 4, 2 * q[2] + 4), (2 * (n[0] - q[0]) - 5, 2 * (n[1] - q[1]) - 5, 2 * (n[2] - q[2]) - 5))])
