"""Castro plotfiles: writer and reader.

Reference behaviour restated:
  Castro::writePlotFile / plotFileOutput   Source/driver/Castro_io.cpp:853-1153  (the text `Header`)
  Castro::writeJobInfo                     Source/driver/Castro_io.cpp:513-785   (free-form `job_info`)
  state names                              Source/driver/Castro_setup.cpp:493-556
  derived fields                           Source/driver/Castro_setup.cpp:756-960, Source/driver/Derive.cpp
  VisMF::Write [3P, AMReX 21.07]           Level_0/Cell_H (text) + Level_0/Cell_D_nnnnn (FAB header line +
                                           raw little-endian doubles, i fastest, component slowest); format per
                                           SURVEY.md D.4 -- it cannot be checked against an AMReX reader here.

The field data are produced on the device (state copy + castro_amd_derive_fab); this module only does the
host-side file layout.  One rank = one box = one Cell_D file; rank 0 writes Header and Cell_H.
"""
import os
import re

import numpy as np

# Source/driver/Castro_setup.cpp:493-556 (one species named X: networks/general_null/gammalaw.net [3P])
STATE_NAMES = ["density", "xmom", "ymom", "zmom", "rho_E", "rho_e", "Temp", "rho_X"]
# registration order of Castro_setup.cpp:756-960 for a 3-D pure-hydro build (a derive with several components is
# written as name_0, name_1, ...: Castro_io.cpp:954-965); not provided: entropy (the EOS's entropy is Microphysics's)
DERIVE_NAMES = ["pressure", "kineng", "soundspeed", "Gamma_1", "MachNumber", "magvort", "divu", "eint_E", "eint_e",
                "logden", "StateErr_0", "StateErr_1", "StateErr_2", "X(X)", "abar", "x_velocity", "y_velocity", "z_velocity",
                "magvel", "radvel", "circvel", "magmom", "angular_momentum_x", "angular_momentum_y", "angular_momentum_z"]
_STENCIL = ("magvort", "divu")

FAB_REAL_DESCRIPTOR = "((8, (64 11 52 0 1 12 0 1023)),(8, (8 7 6 5 4 3 2 1)))"   # IEEE-754 binary64, little endian


def _g(x):
    return "%.17g" % x            # Amr::writePlotFile sets HeaderFile.precision(17) [3P]


def _box(lo, hi):
    return "((%d,%d,%d) (%d,%d,%d) (0,0,0))" % (lo[0], lo[1], lo[2], hi[0], hi[1], hi[2])


def plot_data(castro, derive=None):
    """(names, tensor (ncomp, nz, ny, nx)) of this rank's box: the plotMF of Castro_io.cpp:1099-1126
    (state components first, then derived fields, nGrow = 0)."""
    import torch
    derive = list(DERIVE_NAMES if derive is None else derive)
    names = STATE_NAMES + derive
    lo, hi = castro.lo, castro.hi
    S = castro.S_new_b
    if any(d in _STENCIL for d in derive):
        castro.expand_state(S)                                   # derive on grow_box_by_one: FillPatch'ed ghosts
    n = [hi[d] - lo[d] + 1 for d in range(3)]
    out = torch.empty((len(names), n[2], n[1], n[0]), dtype=S.dtype, device=S.device)
    g = [lo[d] - castro.gbox[0][d] for d in range(3)]
    out[:8] = S[:, g[2]:g[2] + n[2], g[1]:g[1] + n[1], g[0]:g[0] + n[0]]
    center = [0.5 * (castro.geom.problo[d] + castro.geom.probhi[d]) for d in range(3)] \
        if getattr(castro, "center", None) is None else castro.center
    for m, name in enumerate(derive):
        castro.hydro.derive(name, S, castro.gbox, out, (lo, hi), 8 + m, lo, hi, castro.geom, castro.params, center)
    return names, out


def write_plotfile(dirname, castro, derive=None, job_info=None):
    """Write `dirname`/{Header, job_info, Level_0/Cell_H, Level_0/Cell_D_nnnnn}."""
    names, data = plot_data(castro, derive)
    comm = castro.comm
    rank, size = comm.rank, comm.size
    arr = data.cpu().numpy() if hasattr(data, "cpu") else np.asarray(data)
    ncomp = arr.shape[0]
    lev = os.path.join(dirname, "Level_0")
    if rank == 0:
        os.makedirs(lev, exist_ok=True)
    comm.barrier()
    fname = "Cell_D_%05d" % rank
    head = "FAB %s%s %d\n" % (FAB_REAL_DESCRIPTOR, _box(castro.lo, castro.hi), ncomp)
    with open(os.path.join(lev, fname), "wb") as f:
        f.write(head.encode("ascii"))
        np.ascontiguousarray(arr, dtype="<f8").tofile(f)
    mine = dict(lo=tuple(castro.lo), hi=tuple(castro.hi), file=fname, offset=0,
                min=[float(arr[n].min()) for n in range(ncomp)], max=[float(arr[n].max()) for n in range(ncomp)])
    boxes = comm.gather_objects(mine)
    if rank == 0:
        geom = castro.geom
        n_cell = castro.n_cell
        dom_lo = [geom.domlo[d] for d in range(3)]
        dom_hi = [geom.domhi[d] for d in range(3)]
        with open(os.path.join(dirname, "Header"), "w") as os_:
            w = os_.write
            w("HyperCLaw-V1.1\n")                                   # Castro::thePlotFileType, Castro_io.cpp:481
            w("%d\n" % ncomp)
            for nm in names:
                w(nm + "\n")
            w("3\n")
            w(_g(castro.time) + "\n")                               # parent->cumTime()
            w("0\n")                                                # finest level
            w(" ".join(_g(geom.problo[d]) for d in range(3)) + " \n")
            w(" ".join(_g(geom.probhi[d]) for d in range(3)) + " \n")
            w("\n")                                                 # refinement ratios: none
            w(_box(dom_lo, dom_hi) + " \n")
            w("%d \n" % castro.nstep)                               # levelSteps
            w(" ".join(_g(geom.dx[d]) for d in range(3)) + " \n")
            w("%d\n" % geom.coord)
            w("0\n")                                                # "Write bndry data."
            w("0 %d %s\n" % (len(boxes), _g(castro.time)))          # level, #grids, cur_time  (:1052)
            w("%d\n" % castro.nstep)
            for b in boxes:
                for d in range(3):
                    w("%s %s\n" % (_g(geom.problo[d] + (b["lo"][d] - dom_lo[d]) * geom.dx[d]),
                                   _g(geom.problo[d] + (b["hi"][d] + 1 - dom_lo[d]) * geom.dx[d])))
            w("Level_0/Cell\n")
        with open(os.path.join(lev, "Cell_H"), "w") as f:
            w = f.write
            w("1\n")                                                # VisMF::Header::Version_v1
            w("0\n")                                                # VisMF::OneFilePerCPU
            w("%d\n" % ncomp)
            w("0\n")                                                # nGrow
            w("(%d 0\n" % len(boxes))
            for b in boxes:
                w(_box(b["lo"], b["hi"]) + "\n")
            w(")\n")
            w("%d\n" % len(boxes))
            for b in boxes:
                w("FabOnDisk: %s %d\n" % (b["file"], b["offset"]))
            w("\n")
            for key in ("min", "max"):
                w("%d,%d\n" % (len(boxes), ncomp))
                for b in boxes:
                    w("".join("%.16e," % v for v in b[key]) + "\n")
                w("\n")
        with open(os.path.join(dirname, "job_info"), "w") as f:     # free-form text in the reference
            f.write("==============================================================================\n")
            f.write(" Castro Job Information (castro_amd, MI355X hydro path)\n")
            f.write("==============================================================================\n")
            f.write("number of MPI processes: %d\n" % size)
            f.write("n_cell: %d %d %d\nnstep: %d\ntime: %s\n" % (n_cell[0], n_cell[1], n_cell[2], castro.nstep, _g(castro.time)))
            f.write("lo_bc: %s\nhi_bc: %s\n" % (list(geom.lo_bc), list(geom.hi_bc)))
            for fld, _ in castro.params._fields_:
                f.write("castro.%s = %s\n" % (fld, getattr(castro.params, fld)))
            if job_info:
                f.write(str(job_info) + "\n")
    comm.barrier()
    return names


def write_plotfile_amr(dirname, amr, derive=None):
    """Multi-level plotfile of a CastroAmr hierarchy (single rank): the same Header layout with finest_level > 0
    (Castro_io.cpp:953-1076 per level) and one Level_l/{Cell_H, Cell_D_00000} per level holding the FABs of all boxes
    of the level one after the other (VisMF [3P]: Cell_H lists the boxes, each FAB's byte offset and its min/max)."""
    levels = amr.levels
    nlev = len(levels)
    per = []
    for l, lev in enumerate(levels):
        lev.alpha = 1.0              # ghost zones for the stencil derives: parent data at the current (new) time
        fabs = []
        for b in lev.boxes:
            names, data = plot_data(b, derive)
            fabs.append(data.cpu().numpy() if hasattr(data, "cpu") else np.asarray(data))
        per.append(fabs)
    ncomp = per[0][0].shape[0]
    os.makedirs(dirname, exist_ok=True)
    g0 = levels[0].geom
    with open(os.path.join(dirname, "Header"), "w") as f:
        w = f.write
        w("HyperCLaw-V1.1\n%d\n" % ncomp)
        for nm in names:
            w(nm + "\n")
        w("3\n" + _g(amr.time) + "\n%d\n" % (nlev - 1))
        w(" ".join(_g(g0.problo[d]) for d in range(3)) + " \n")
        w(" ".join(_g(g0.probhi[d]) for d in range(3)) + " \n")
        w("".join("2 " for _ in range(nlev - 1)) + "\n")
        w("".join(_box([lev.geom.domlo[d] for d in range(3)], [lev.geom.domhi[d] for d in range(3)]) + " " for lev in levels) + "\n")
        w("".join("%d " % (amr.nstep * 2 ** l) for l in range(nlev)) + "\n")
        for lev in levels:
            w(" ".join(_g(lev.geom.dx[d]) for d in range(3)) + " \n")
        w("%d\n0\n" % g0.coord)
        for l, lev in enumerate(levels):
            w("%d %d %s\n%d\n" % (l, len(lev.boxes), _g(amr.time), amr.nstep * 2 ** l))
            for b in lev.boxes:
                for d in range(3):
                    w("%s %s\n" % (_g(lev.geom.problo[d] + b.lo[d] * lev.geom.dx[d]), _g(lev.geom.problo[d] + (b.hi[d] + 1) * lev.geom.dx[d])))
            w("Level_%d/Cell\n" % l)
    for l, (lev, fabs) in enumerate(zip(levels, per)):
        ld = os.path.join(dirname, "Level_%d" % l)
        os.makedirs(ld, exist_ok=True)
        offsets = []
        with open(os.path.join(ld, "Cell_D_00000"), "wb") as f:
            for b, arr in zip(lev.boxes, fabs):
                offsets.append(f.tell())
                f.write(("FAB %s%s %d\n" % (FAB_REAL_DESCRIPTOR, _box(b.lo, b.hi), ncomp)).encode("ascii"))
                np.ascontiguousarray(arr, dtype="<f8").tofile(f)
        ng = len(fabs)
        with open(os.path.join(ld, "Cell_H"), "w") as f:
            f.write("1\n0\n%d\n0\n(%d 0\n" % (ncomp, ng) + "".join(_box(b.lo, b.hi) + "\n" for b in lev.boxes) + ")\n%d\n" % ng)
            for off in offsets:
                f.write("FabOnDisk: Cell_D_00000 %d\n" % off)
            f.write("\n")
            for fn in (np.min, np.max):
                f.write("%d,%d\n" % (ng, ncomp))
                for arr in fabs:
                    f.write("".join("%.16e," % fn(arr[n]) for n in range(ncomp)) + "\n")
                f.write("\n")
    return names


def read_plotfile_amr(dirname):
    """Levels of a multi-level plotfile: names, time and per level dict(dx, boxes=[(lo, hi)], fabs=[(ncomp, nz, ny, nx)]);
    a level of one box also carries box= and data= for that box."""
    with open(os.path.join(dirname, "Header")) as f:
        L = [ln.rstrip("\n") for ln in f]
    ncomp = int(L[1])
    names = L[2:2 + ncomp]
    p = 2 + ncomp
    time, finest = float(L[p + 1]), int(L[p + 2])
    q = p + 8                                   # first dx line
    dxs = [[float(x) for x in L[q + l].split()] for l in range(finest + 1)]
    q += finest + 1 + 2
    out = []
    for l in range(finest + 1):
        ngrids = int(L[q].split()[1])
        path = L[q + 2 + 3 * ngrids]
        q += 3 + 3 * ngrids
        levdir = os.path.join(dirname, os.path.dirname(path))
        with open(os.path.join(levdir, "Cell_H")) as f:
            H = [ln.rstrip("\n") for ln in f]
        assert int(H[4].strip("(").split()[0]) == ngrids
        boxes, fabs = [], []
        for g in range(ngrids):
            v = [int(x) for x in re.findall(r"-?\d+", H[5 + g])]
            boxes.append((v[0:3], v[3:6]))
        r = 5 + ngrids + 2                      # first FabOnDisk line
        for g in range(ngrids):
            _, fname, off = H[r + g].split()
            lo, hi = boxes[g]
            m = [hi[d] - lo[d] + 1 for d in range(3)]
            with open(os.path.join(levdir, fname), "rb") as f:
                f.seek(int(off))
                f.readline()
                fabs.append(np.fromfile(f, dtype="<f8", count=ncomp * m[0] * m[1] * m[2]).reshape(ncomp, m[2], m[1], m[0]))
        lev = dict(dx=dxs[l], boxes=boxes, fabs=fabs)
        if ngrids == 1:
            lev.update(box=boxes[0], data=fabs[0])
        out.append(lev)
    return dict(names=names, time=time, levels=out)


def read_plotfile(dirname):
    """Parse a single-level plotfile written by write_plotfile (or by AMReX with the same layout).
    Returns dict(names, time, nstep, prob_lo, prob_hi, domain=(lo,hi), dx, boxes, data) with
    data[(ncomp, nz, ny, nx)] assembled over the whole domain."""
    with open(os.path.join(dirname, "Header")) as f:
        L = [ln.rstrip("\n") for ln in f]
    assert L[0].startswith("HyperCLaw"), "not a Castro plotfile"
    ncomp = int(L[1])
    names = L[2:2 + ncomp]
    p = 2 + ncomp
    dim = int(L[p]); time = float(L[p + 1]); finest = int(L[p + 2])
    assert dim == 3 and finest == 0, "single-level 3-D plotfiles only"
    prob_lo = [float(x) for x in L[p + 3].split()]
    prob_hi = [float(x) for x in L[p + 4].split()]
    nums = [int(x) for x in re.findall(r"-?\d+", L[p + 6])]
    dom_lo, dom_hi = nums[0:3], nums[3:6]
    nstep = int(L[p + 7].split()[0])
    dx = [float(x) for x in L[p + 8].split()]
    ngrids = int(L[p + 11].split()[1])
    path = L[p + 13 + 3 * ngrids]
    levdir = os.path.join(dirname, os.path.dirname(path))
    with open(os.path.join(levdir, os.path.basename(path) + "_H")) as f:
        H = [ln.rstrip("\n") for ln in f]
    assert int(H[2]) == ncomp
    nb = int(H[4].strip("(").split()[0])
    boxes = []
    for b in range(nb):
        v = [int(x) for x in re.findall(r"-?\d+", H[5 + b])]
        boxes.append((v[0:3], v[3:6]))
    q = 5 + nb + 2
    fods = []
    for b in range(nb):
        _, fn, off = H[q + b].split()
        fods.append((fn, int(off)))
    n = [dom_hi[d] - dom_lo[d] + 1 for d in range(3)]
    data = np.empty((ncomp, n[2], n[1], n[0]))
    for (lo, hi), (fn, off) in zip(boxes, fods):
        with open(os.path.join(levdir, fn), "rb") as f:
            f.seek(off)
            head = f.readline().decode("ascii")
            assert head.startswith("FAB "), head
            nc = int(head.split()[-1])
            m = [hi[d] - lo[d] + 1 for d in range(3)]
            a = np.fromfile(f, dtype="<f8", count=nc * m[0] * m[1] * m[2]).reshape(nc, m[2], m[1], m[0])
        s = [lo[d] - dom_lo[d] for d in range(3)]
        data[:, s[2]:s[2] + m[2], s[1]:s[1] + m[1], s[0]:s[0] + m[0]] = a
    return dict(names=names, time=time, nstep=nstep, prob_lo=prob_lo, prob_hi=prob_hi, domain=(dom_lo, dom_hi),
                dx=dx, boxes=boxes, data=data)
