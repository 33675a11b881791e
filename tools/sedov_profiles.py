#!/usr/bin/env python3
"""Radially binned density, radial velocity, pressure and specific internal energy of a Sedov run against the
reference's analytic table (the four panels of Exec/hydro_tests/Sedov/testsuite_analysis/sedov_3d_sph.py)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import castro_amd


from tests.util import sedov_l1_errors as l1_errors


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    c = castro_amd.Castro((n, n, n))
    c.initData("sedov")
    c.evolve(0.01)
    torch.cuda.synchronize()
    table = np.loadtxt(os.path.join("tests", "golden", "reference_verification", "spherical_sedov.dat"))
    res, rc, prof = l1_errors(c, table)
    print("n = %d, %d steps; volume-weighted L1 errors vs the analytic table:" % (n, c.nstep), {k: round(v, 4) for k, v in res.items()})
    S = c.S_new()
    print("mass - 1 = %.3e, energy = %.15g" % (S[0].sum().item() / n ** 3 - 1.0, S[4].sum().item() / n ** 3))
