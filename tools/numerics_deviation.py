#!/usr/bin/env python3
"""Deviation of a kernel-library build (CASTRO_AMD_LIB or the default) from the CPU oracle on Sedov / Sod runs, per state
component, after 1, 10, 100 steps and at stop_time: the measurement behind the `contract` numerics mode.

  tools/numerics_deviation.py [n=64] [problem=sedov|sod|test2|test3] [stop_time=0.01] [checks=1,10,100] [<castro parameter>=<value> ...]
  (e.g. ppm_type=0 riemann_solver=2: any other key goes to default_params of both sides)

Two figures per component and check point:
  norm = max|a - b| / max|b|        (what AMReX's fcompare prints as the relative error of a plotfile field)
  elem = max over zones of |a - b| / max(|b|, 1e-30 + floor * max|b|), floor = 1e-12 (zones where the field itself is a
         rounding residue of a cancellation -- e.g. the transverse momentum on a symmetry plane -- are measured against
         the field's scale)
"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import castro_amd
from oracle import oracle_lib as O

kw = dict(a.split("=") for a in sys.argv[1:])
n = int(kw.get("n", 64))
problem = kw.get("problem", "sedov")
stop_time = float(kw.get("stop_time", 0.01 if problem == "sedov" else 0.2))
checks = [int(x) for x in kw.get("checks", "1,10,100").split(",")]
names = ["rho", "xmom", "ymom", "zmom", "rho_E", "rho_e", "Temp", "rho_X"]
pkw = {k: (float(v) if ("." in v or "e" in v) else int(v)) for k, v in kw.items() if k not in ("n", "problem", "stop_time", "checks")}

SOD = {"sod": (1.0, 0.0, 1.0, 0.125, 0.0, 0.1, 0.2), "test2": (1.0, -2.0, 0.4, 1.0, 2.0, 0.4, 0.15),
       "test3": (1.0, 0.0, 1000.0, 1.0, 0.0, 0.01, 0.012)}

if problem == "sedov":
    shape = (n, n, n)
    c = castro_amd.Castro(shape, params=castro_amd.default_params(**pkw))
    c.initData("sedov")
    lev = O.Level(shape, O.make_geom(shape), O.default_params(**pkw), nthreads=0)
    lev.init_sedov()
else:
    rl, ul, pl, rr, ur, pr, st = SOD[problem]
    stop_time = float(kw.get("stop_time", st))
    shape = (n, 8, 8)
    c = castro_amd.Castro(shape, prob_hi=(1.0, 8.0 / n, 8.0 / n), params=castro_amd.default_params(**pkw))
    c.initData("sod", rho_l=rl, u_l=ul, p_l=pl, rho_r=rr, u_r=ur, p_r=pr)
    lev = O.Level(shape, O.make_geom(shape, probhi=(1.0, 8.0 / n, 8.0 / n)), O.default_params(**pkw), nthreads=0)
    lev.init_sod(rl, ul, pl, rr, ur, pr)


def report(tag):
    torch.cuda.synchronize()
    a = c.S_new().cpu().numpy()
    b = lev.state()
    out = []
    worst_n, worst_e = 0.0, 0.0
    for k, nm in enumerate(names):
        d = np.abs(a[k] - b[k])
        ref = np.abs(b[k])
        sc = ref.max()
        en = d.max() / sc if sc > 0 else d.max()
        ee = (d / np.maximum(ref, 1e-30 + 1e-12 * sc)).max()
        worst_n, worst_e = max(worst_n, en), max(worst_e, ee)
        out.append("%s %.1e/%.1e" % (nm, en, ee))
    print("%-22s step %5d t=%.6e dt dev %.1e | worst norm %.2e elem %.2e | %s" % (
        tag, c.nstep, c.time, abs(c.dt - lev.dt) / lev.dt, worst_n, worst_e, " ".join(out)), flush=True)


t0 = time.time()
nstep = 0
eps = 2.3e-16
while c.time < stop_time - eps:
    c.step(stop_time)
    lev.step(stop_time)
    nstep += 1
    if nstep in checks:
        report("%s %d^3" % (problem, n))
report("%s %d^3 stop" % (problem, n))
print("wall %.1f s" % (time.time() - t0))
