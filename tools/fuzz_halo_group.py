#!/usr/bin/env python3
"""Randomised campaign for castro_amd_halo_group / castro_amd_fill_boundary_group: a random domain cut into random unequal boxes
(recursive bisection), random periodic directions, ghost depth and component count, all boxes on one rank -- with and without RCCL
self-send.  Every ghost zone that lies in a box of the level (or in a periodic image of one) must take that box's valid data bit for
bit, every other zone must keep its value.  The message lists come from castro_amd/halo.py (the derivation of the AMReX adapter).
usage: python tools/fuzz_halo_group.py [ncases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import castro_amd
from castro_amd import halo
from castro_amd.hydro import HipHydro

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
h = HipHydro(0)
comm = h.comm_create(1, 0, h.comm_unique_id())


def split(box, depth):
    lo, hi = box
    ext = [hi[d] - lo[d] + 1 for d in range(3)]
    d = int(np.argmax(ext))
    if depth == 0 or ext[d] < 10 or rng.random() < 0.2:
        return [box]
    cut = int(rng.integers(lo[d] + 4, hi[d] - 3))          # both halves at least 4 wide
    a_hi, b_lo = list(hi), list(lo)
    a_hi[d], b_lo[d] = cut, cut + 1
    return split((lo, tuple(a_hi)), depth - 1) + split((tuple(b_lo), hi), depth - 1)


bad = 0
for case in range(ncases):
    n = tuple(int(rng.integers(12, 36)) for _ in range(3))
    dom = ((0, 0, 0), tuple(x - 1 for x in n))
    boxes = split(dom, int(rng.integers(1, 4)))
    if rng.random() < 0.3 and len(boxes) > 1:               # a level that does not cover the domain
        boxes.pop(int(rng.integers(len(boxes))))
    periodic = tuple(bool(rng.random() < 0.5) for _ in range(3))
    ng, ncomp = int(rng.integers(1, 5)), int(rng.choice([1, 7, 8]))
    ss = int(rng.integers(0, 2))
    os.environ["CASTRO_AMD_HALO_SELF_SEND"] = str(ss)
    local, sends, recvs = halo.level_messages(boxes, [0] * len(boxes), 0, ng, dom, periodic)
    group = h.halo_group(comm, len(boxes), sends, recvs, ncomp)
    gb = [(tuple(x - ng for x in lo), tuple(x + ng for x in hi)) for lo, hi in boxes]
    host = [rng.normal(size=(ncomp,) + tuple(g[1][d] - g[0][d] + 1 for d in (2, 1, 0))) for g in gb]
    dev = [torch.from_numpy(a.copy()).to(h.device) for a in host]
    h.fill_boundary_group(group, dev, gb)
    torch.cuda.synchronize()
    G = np.full((ncomp, n[2], n[1], n[0]), np.nan)
    for (lo, hi), a in zip(boxes, host):
        G[:, lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = a[:, ng:-ng, ng:-ng, ng:-ng]
    ok = True
    for (glo, ghi), a, d in zip(gb, host, dev):
        want = a.copy()
        idx = [np.arange(glo[x], ghi[x] + 1) for x in range(3)]
        inside = [np.ones_like(idx[x], dtype=bool) if periodic[x] else ((idx[x] >= 0) & (idx[x] < n[x])) for x in range(3)]
        w = [idx[x] % n[x] for x in range(3)]
        src = G[:, w[2][:, None, None], w[1][None, :, None], w[0][None, None, :]]
        take = inside[2][:, None, None] & inside[1][None, :, None] & inside[0][None, None, :] & ~np.isnan(src[0])
        take[ng:-ng, ng:-ng, ng:-ng] = False
        want[:, take] = src[:, take]
        ok &= np.array_equal(d.cpu().numpy(), want)
    if not ok:
        bad += 1
        print("MISMATCH case %d: n %s boxes %s periodic %s ng %d ncomp %d self_send %d" % (case, n, boxes, periodic, ng, ncomp, ss))
    h.halo_group_destroy(group)
h.comm_destroy(comm)
print("fuzz_halo_group: %d cases, seed %d: %d mismatches" % (ncases, seed, bad))
