#!/usr/bin/env python3
"""Per-kernel hipEvent times of ONE hydro call repeated on a FIXED state (no stepping): for the timing diagnostics whose
results are wrong on purpose (-DTRACE_DIAG_NOSTORE, -DDIAG_T1_NOSTORE ...), which cannot survive a real run.
usage: [CASTRO_AMD_LIB=castro_amd/libvariant_x.so] python tools/kernel_times.py [ncell] [numerics] [repeats]"""
import sys

import torch

sys.path.insert(0, ".")
import castro_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
numerics = sys.argv[2] if len(sys.argv) > 2 else "contract"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
c = castro_amd.Castro((n, n, n), numerics=numerics)
c.initData("sedov")
dt = c.computeInitialDt(0.01)
c._swap_state_time_levels()
c.expand_state(c.S_old_b, bc=not c.bc_in_hydro)
c.red.fill_(1.e200)
c._whole_step = True
for _ in range(2):
    c._flux_clear = True
    c.construct_ctu_hydro_source(0.0, dt, fuse_clean=True, sborder_clean=2, bc_fill=c.bc_in_hydro)
torch.cuda.synchronize()
c.hydro.profile(True)
c.hydro.profile_reset()
for _ in range(reps):
    c._flux_clear = True
    c.construct_ctu_hydro_source(0.0, dt, fuse_clean=True, sborder_clean=2, bc_fill=c.bc_in_hydro)
torch.cuda.synchronize()
prof = c.hydro.profile_report()
k = {a: round(ms / cnt, 3) for a, (ms, cnt) in sorted(prof.items())}
print("fixed-state call, %d^3 %s: sum %.2f ms" % (n, numerics, sum(k.values())), k)
