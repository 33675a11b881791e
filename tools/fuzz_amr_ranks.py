#!/usr/bin/env python3
"""Randomised campaign for the AMR hierarchy spread over ranks (CPU, gloo, oracle backend): random base grids cut into one or
eight level-0 boxes, random properly nested boxes on one or two refined levels, physical or periodic boundaries, Sedov or Sod,
sometimes gravity / rotation, sometimes tag-driven regridding -- every rank walks the same random stream, the ranks run the
case together, rank 0 also runs it alone (SingleComm) and compares dt sequence, box lists and every box bit for bit.
usage: tools/fuzz_amr_ranks.py [ncases] [seed] [world] [oracle|hip]   (hip: the device path, all processes on one GPU)"""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

sys.path.insert(0, ".")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def random_boxes(rng, region_lo, region_hi, nmax, margin):
    boxes = []
    for _ in range(20):
        if len(boxes) >= nmax:
            break
        lo, hi, ok = [], [], True
        for d in range(3):
            a, b = region_lo[d] + margin[d], region_hi[d] - margin[d]
            if b - a + 1 < 2:
                ok = False
                break
            l = int(rng.integers(a // 2, (b - 1) // 2 + 1)) * 2
            l = max(l, a + (a % 2))
            n = int(rng.integers(1, 4)) * 2
            h = min(l + n - 1, b if (b + 1) % 2 == 0 else b - 1)
            if h < l + 1:
                ok = False
                break
            lo.append(l); hi.append(h)
        if not ok or any(all(lo[d] <= q[d] and p[d] <= hi[d] for d in range(3)) for p, q in boxes):
            continue
        boxes.append((tuple(lo), tuple(hi)))
    return sorted(boxes, key=lambda b: (b[0][2], b[0][1], b[0][0]))


def worker(rank, world, port, ncases, seed, out, backend):
    import torch.distributed as dist
    import castro_amd
    from oracle import oracle_lib as O
    from tests.oracle_backend import OracleBackend
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(1)
    if backend == "hip":                              # the device path, every process on the one GPU of the box (gloo transport)
        torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(seed)
    bad = done = skipped = 0
    for case in range(ncases):
        n = tuple(int(rng.choice([16, 16, 24])) for _ in range(3))
        bcs = [int(rng.choice([2, 2, 3, 4, 0])) for _ in range(6)]
        for d in range(3):
            if bcs[d] == 0 or bcs[d + 3] == 0:
                bcs[d] = bcs[d + 3] = 0
        l1 = random_boxes(rng, (0, 0, 0), tuple(x - 1 for x in n), int(rng.integers(1, 5)), (0, 0, 0))
        patches = [l1] if l1 else None
        if l1 and rng.integers(0, 2):
            b = l1[int(rng.integers(0, len(l1)))]
            l2 = random_boxes(rng, tuple(2 * x for x in b[0]), tuple(2 * x + 1 for x in b[1]), 2, (2, 2, 2))
            if l2:
                patches.append(l2)
        kw = dict(lo_bc=tuple(bcs[:3]), hi_bc=tuple(bcs[3:]), base_grid=(2, 2, 2) if rng.integers(0, 2) else None)
        tagged = bool(rng.integers(0, 3) == 0) or patches is None
        if tagged:
            kw.update(refine=[("density", "gradient", 0.05), ("rho_E", "relative_gradient", 0.5)], regrid_int=int(rng.integers(1, 3)),
                      n_error_buf=int(rng.integers(0, 2)), blocking_factor=4, max_level=int(rng.integers(1, 3)), cluster=True,
                      grid_eff=float(rng.choice([0.5, 0.7, 0.9])), max_grid_size=int(rng.choice([8, 16])))
        else:
            kw.update(patches=patches)
        if rng.integers(0, 4) == 0:
            kw.update(do_grav=True, const_grav=float(rng.uniform(-3, 3)), grav_source_type=int(rng.integers(1, 5)))
            if rng.integers(0, 2):
                kw.update(rotation=O.make_rotation(float(rng.uniform(5, 50)), rot_axis=int(rng.integers(1, 4)),
                                                   rot_source_type=int(rng.integers(1, 5)), implicit_rotation_update=int(rng.integers(0, 2))))
        prob = str(rng.choice(["sedov", "sod"]))
        pkw = dict(init_shrink=float(rng.choice([0.1, 0.3])), ppm_type=int(rng.integers(0, 2)))
        nsteps = int(rng.integers(2, 5))
        idir = int(rng.integers(1, 4))

        def run(comm):
            if backend == "hip":
                kh = dict(kw)
                if "rotation" in kh:
                    r = kh["rotation"]
                    kh["rotation"] = castro_amd.make_rotation(2.0 * np.pi / max(abs(x) for x in r.omega), rot_axis=1 + int(np.argmax([abs(x) for x in r.omega])),
                                                               rot_source_type=r.rot_source_type, implicit_rotation_update=r.implicit_rotation_update)
                a = castro_amd.CastroAmr(n, params=castro_amd.default_params(**pkw), comm=comm, **kh)
            else:
                a = castro_amd.CastroAmr(n, params=O.default_params(**pkw), make_hydro=OracleBackend, comm=comm, **kw)
            if prob == "sedov":
                a.initData("sedov", r_init=0.15, nsub=4)
            else:
                a.initData("sod", rho_l=1.0, u_l=0.2, p_l=1.0, rho_r=0.125, u_r=-0.1, p_r=0.1, idir=idir, frac=0.6)
            res = []
            for _ in range(nsteps):
                try:
                    res.append(a.step())
                except castro_amd.AdvanceFailure as e:
                    res.append("AdvanceFailure: %s" % e)
                    break
            return a, res
        try:
            a, res = run(castro_amd.DistComm())
            ok = True
        except AssertionError:                         # not properly nested: every rank refuses alike (same metadata)
            ok = None
        if ok is None:
            skipped += 1
            continue
        levels = [a.gather_level(l) for l in range(len(a.lev))]
        if rank == 0:
            b, res1 = run(None)
            good = res == res1 and a.boxes == b.boxes
            if good:
                full0 = None
                for l, lev in enumerate(b.lev):
                    if l == 0 and len(lev.boxes) != len(levels[0]):
                        good = False
                        break
                    for (bx, arr), bb in zip(levels[l], lev.boxes):
                        if bx != bb.bx or not np.array_equal(arr, bb.S_new().cpu().numpy(), equal_nan=True):
                            good = False
            if not good:
                bad += 1
                print("MISMATCH case %d: n=%s bc=%s kw=%s %s %s steps=%d" % (case, n, bcs, {k: v for k, v in kw.items() if k != "rotation"}, prob, pkw, nsteps), flush=True)
            done += 1
    if rank == 0:
        print("cases run %d of %d over %d ranks, mismatching %d, refused by every rank alike %d" % (done, ncases, world, bad, skipped), flush=True)
        open(out, "w").write(str(bad))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    world = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    out = "/tmp/fuzz_amr_ranks_%d.txt" % os.getpid()
    backend = sys.argv[4] if len(sys.argv) > 4 else "oracle"
    mp.spawn(worker, args=(world, free_port(), ncases, seed, out, backend), nprocs=world, join=True)
    sys.exit(1 if int(open(out).read()) else 0)
