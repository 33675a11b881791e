#!/usr/bin/env python3
"""Take one case of tools/fuzz_parity.py apart: the oracle hands out the input states of each of its twelve box-level Riemann solves (debug
hook ora_set_debug_riemann_hook), every solve is repeated face by face on both sides through the pointwise entry
points, and the faces whose outputs differ are printed with their inputs at full precision.
usage: tools/fuzz_case_faces.py <ncases> <seed> <case>   (the arguments of the fuzz_parity.py run that reported the case)"""
import ctypes as C
import os
import runpy
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import oracle_lib as O

cap = []
HOOK = C.CFUNCTYPE(None, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), O.A4, O.A4, O.A4, O.A4)


def a4_to_np(a):
    n = [a.hi[d] - a.lo[d] + 1 for d in range(3)]
    buf = np.ctypeslib.as_array(a.p, shape=(a.nc * n[2] * n[1] * n[0],))
    return buf.reshape(a.nc, n[2], n[1], n[0]).copy(), tuple(a.lo[d] for d in range(3))


@HOOK
def hook(idir, lo, hi, ql, qr, qaux, shk):
    cap.append(dict(idir=idir, lo=tuple(lo[d] for d in range(3)), hi=tuple(hi[d] for d in range(3)), ql=a4_to_np(ql), qr=a4_to_np(qr),
                    qaux=a4_to_np(qaux), shk=a4_to_np(shk)))


O.lib().ora_set_debug_riemann_hook.argtypes = [C.c_void_p]
O.lib().ora_set_debug_riemann_hook(C.cast(hook, C.c_void_p))
case = int(sys.argv[3])
sys.argv = ["fuzz_parity.py", sys.argv[1], sys.argv[2], str(case)]
# run_path returns no globals after SystemExit: replay the parameters from a second, quiet pass
import io, contextlib
sys.argv = ["fuzz_parity.py", sys.argv[1], sys.argv[2], str(case)]
KEEP = {}
ns = {"KEEP": KEEP}
src = open(os.path.join("tools", "fuzz_parity.py")).read().replace("sys.exit(1 if bad else 0)", "")
src = src.replace("    try:\n        out = _run_both(", "    KEEP.update(pkw=dict(pkw), bcs=list(bcs), bxlo=bxlo, bxhi=bxhi)\n    try:\n        out = _run_both(", 1)
assert "KEEP.update" in src
with contextlib.redirect_stdout(io.StringIO()):
    exec(compile(src, "fuzz_parity.py", "exec"), ns)
hip, pkw, bcs, bxlo, bxhi = ns["hip"], KEEP["pkw"], KEEP["bcs"], KEEP["bxlo"], KEEP["bxhi"]
import castro_amd
Ph, Po = castro_amd.default_params(**pkw), O.default_params(**pkw)
import torch
comp = [O.QRHO, O.QU, O.QV, O.QW, O.QPRES, O.QREINT, O.QFS]
O.lib().ora_set_debug_riemann_hook(None)
NAMES = ["F^x", "F^y", "F^z", "F^{y|z}", "F^{z|y}", "final x", "F^{z|x}", "F^{x|z}", "final y", "F^{x|y}", "F^{y|x}", "final z"]
for nsolve, c in enumerate(cap):
    idir = c["idir"]
    lo, hi = c["lo"], c["hi"]
    (ql, o), (qr, _), (qaux, oa), (shk, os_) = c["ql"], c["qr"], c["qaux"], c["shk"]
    idx = [(i, j, k) for k in range(lo[2], hi[2] + 1) for j in range(lo[1], hi[1] + 1) for i in range(lo[0], hi[0] + 1)]
    at = lambda arr, org, m, i, j, k: arr[m, k - org[2], j - org[1], i - org[0]]
    sh = [0, 0, 0]
    sh[idir] = 1
    qm = np.array([[at(ql, o, m, *p) for p in idx] for m in comp])
    qp = np.array([[at(qr, o, m, *p) for p in idx] for m in comp])
    cl = np.array([at(qaux, oa, 1, p[0] - sh[0], p[1] - sh[1], p[2] - sh[2]) for p in idx])
    cr = np.array([at(qaux, oa, 1, *p) for p in idx])
    isk = np.array([int(at(shk, os_, 0, p[0] - sh[0], p[1] - sh[1], p[2] - sh[2]) + at(shk, os_, 0, *p) >= 1) for p in idx], dtype=np.int32)
    # walls: bnd_fac = 0 on a face lying on a Symmetry / SlipWall / NoSlipWall boundary of the domain (= the box here)
    bf = np.ones(len(idx))
    for n_, p in enumerate(idx):
        if (p[idir] == bxlo[idir] and bcs[idir] in (3, 4, 5)) or (p[idir] == bxhi[idir] + 1 and bcs[idir + 3] in (3, 4, 5)):
            bf[n_] = 0.0
    want = O.cmpflx_points(idir, qm, qp, cl, cr, bf, Po, is_shock=isk)
    dev = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(hip.device, dt)
    got = hip.cmpflx_points(idir, dev(qm), dev(qp), dev(cl), dev(cr), Ph, bnd_fac=dev(bf), is_shock=dev(isk, torch.int32)).cpu().numpy()
    badf = [n_ for n_ in range(len(idx)) if not np.array_equal(got[:, n_], want[:, n_], equal_nan=True)]
    print("solve %d (%s, direction %d): %d faces, %d differ" % (nsolve, NAMES[nsolve] if len(cap) == 12 else "?", idir, len(idx), len(badf)))
    for n_ in badf[:6]:
        print("  face", idx[n_], "shock" if isk[n_] else "", "bnd_fac", bf[n_])
        print("   qm", [float.hex(float(x)) for x in qm[:, n_]], [float(x) for x in qm[:, n_]])
        print("   qp", [float.hex(float(x)) for x in qp[:, n_]], [float(x) for x in qp[:, n_]])
        print("   cl, cr", float.hex(float(cl[n_])), float.hex(float(cr[n_])), float(cl[n_]), float(cr[n_]))
        print("   oracle", list(want[:, n_]))
        print("   device", list(got[:, n_]))
print("params", pkw)
