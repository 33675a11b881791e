#!/usr/bin/env python3
"""Richardson convergence of the acoustic pulse (Exec/hydro_tests/acoustic_pulse, inputs.64/.128/.256) on the device:
prints the L1 differences between successive resolutions and the rate, like convergence_ppm.sh + RichardsonConvergenceTest."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import castro_amd
from tests.test_driver_cpu import _acoustic_pulse

ppm = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sol = {}
for n, fixed_dt in ((64, 3.0e-3), (128, 1.5e-3), (256, 7.5e-4)):
    c = castro_amd.Castro((n, n, n), lo_bc=(0, 0, 0), hi_bc=(0, 0, 0),
                          params=castro_amd.default_params(init_shrink=0.01, ppm_type=ppm), fixed_dt=fixed_dt)
    c.set_state(_acoustic_pulse(n))
    c.evolve(0.24)
    torch.cuda.synchronize()
    print("n = %d: %d steps to t = %g" % (n, c.nstep, c.time))
    sol[n] = c.S_new().cpu().numpy()


def coarsen(a):
    m = a.shape[-1] // 2
    return a.reshape(a.shape[0], m, 2, m, 2, m, 2).mean(axis=(2, 4, 6))


print("%-8s %14s %14s %8s" % ("field", "L1(64-128)", "L1(128-256)", "rate"))
for comp, name in ((0, "density"), (1, "xmom"), (2, "ymom"), (3, "zmom"), (4, "rho_E"), (5, "rho_e")):
    e_lo = np.abs(coarsen(sol[128])[comp] - sol[64][comp]).mean()
    e_hi = np.abs(coarsen(sol[256])[comp] - sol[128][comp]).mean()
    print("%-8s %14.6e %14.6e %8.3f" % (name, e_lo, e_hi, np.log2(e_lo / e_hi)))
