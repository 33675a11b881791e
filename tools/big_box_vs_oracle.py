#!/usr/bin/env python3
"""One Sedov step of a large box on the device against the oracle's level driver (the oracle needs ~0.4 KB per zone of
host memory and ~0.2 us per zone and thread).  usage: tools/big_box_vs_oracle.py n [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import castro_amd
from oracle import oracle_lib as O

n = int(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
c = castro_amd.Castro((n, n, n))
c.initData("sedov")
lev = O.Level((n, n, n), O.make_geom((n, n, n)), O.default_params(), nthreads=min(32, os.cpu_count() or 8))
lev.init_sedov()
A, B = c.S_new().cpu().numpy(), lev.state()
print(n, "initial state equal:", np.array_equal(A, B))
for s in range(steps):
    t0 = time.time()
    da, db = c.step(0.01), lev.step(0.01)
    torch.cuda.synchronize()
    A, B = c.S_new().cpu().numpy(), lev.state()
    bad = A != B
    print(n, "step", s, "dt equal", da == db, "values differing", int(bad.sum()), "oracle+device s %.1f" % (time.time() - t0))
    if bad.any():
        idx = np.argwhere(bad)
        print("  first:", idx[:6].tolist(), "k range", idx[:, 1].min(), idx[:, 1].max(), "j range", idx[:, 2].min(), idx[:, 2].max(),
              "i range", idx[:, 3].min(), idx[:, 3].max(), "max abs", float(np.abs(A - B).max()))
        rho = A[0]
        print("  device asym", [float(np.abs(rho - np.flip(rho, axis=d)).max()) for d in range(3)],
              "oracle asym", [float(np.abs(B[0] - np.flip(B[0], axis=d)).max()) for d in range(3)])
        break
lev.close()
