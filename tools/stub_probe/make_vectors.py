#!/usr/bin/env python3
"""Generate tests/golden/stub_probe/vectors.npz: inputs and outputs of the reference's own per-zone functions.

STUB-COMPILED, NOT oracle/_ref.  The reference cannot be built in this image (AMReX and Microphysics are empty
submodules, Exec/Make.Castro:24-26,44-46), so this script compiles the reference's hydro sources UNMODIFIED and IN PLACE
from /root/reference against the stand-in headers in tools/stub_probe/stub/ (about 250 lines: Array4, Box, ParallelFor,
Geometry, a gamma-law eos(), the castro:: parameters) and runs probe.cpp on seeded inputs.  state_indices.H is produced
by the reference's own Source/driver/set_variables.py into a temporary directory.  What this shows: whether the
oracle's C restatement and the device functions of ppm_reconstruct / ppm_int_profile, uflatten, cmpflx_plus_godunov
(CGF, CG with every cg_blend, HLLC, HLL), actual_trans_single / actual_trans_final, ctoprim and trace_ppm have slipped
from the source text they follow.  What it does NOT show: anything about a real reference binary (AMReX's Array4 /
ParallelFor, Microphysics's EOS arithmetic order) -- parity stays "unpinned" (DESIGN.md section 6).

Only runs where /root/reference exists; nothing of the reference's sources is copied into the repository: the committed
artefacts are this recipe and the vectors (data)."""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("CASTRO_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
from tests.util import physical_state   # noqa: E402


def write_blob(path, arrays):
    with open(path, "wb") as f:
        for name, a in arrays.items():
            a = np.ascontiguousarray(np.atleast_1d(np.asarray(a, dtype=np.float64))).ravel()
            f.write(name.encode().ljust(48, b"\0"))
            f.write(struct.pack("<q", a.size))
            f.write(a.tobytes())


def read_blob(path):
    out = {}
    with open(path, "rb") as f:
        while True:
            h = f.read(48)
            if len(h) < 48:
                break
            n = struct.unpack("<q", f.read(8))[0]
            out[h.split(b"\0")[0].decode()] = np.frombuffer(f.read(8 * n), dtype=np.float64).copy()
    return out


def build(tmp):
    src = os.path.join(REF, "Source")
    subprocess.check_call([sys.executable, "set_variables.py", "--odir", tmp, "--nadv", "0", "--ngroups", "1", "--defines= ",
                           "_variables"], cwd=os.path.join(src, "driver"), stdout=subprocess.DEVNULL)
    inc = ["-I" + os.path.join(HERE, "stub"), "-I" + tmp, "-I" + os.path.join(src, "hydro"), "-I" + os.path.join(src, "driver"),
           "-I" + os.path.join(src, "problems"), "-I" + os.path.join(src, "rotation")]
    flags = ["-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math"]
    objs = []
    for f in ("trans", "flatten", "riemann", "riemann_util", "advection_util", "trace_ppm", "trace_plm", "Castro_ctu", "edge_util"):
        o = os.path.join(tmp, f + ".o")
        subprocess.check_call(["g++"] + flags + inc + ["-c", os.path.join(src, "hydro", f + ".cpp"), "-o", o])
        objs.append(o)
    for f in ("probe", "probe_params"):
        o = os.path.join(tmp, f + ".o")
        subprocess.check_call(["g++"] + flags + inc + ["-c", os.path.join(HERE, f + ".cpp"), "-o", o])
        objs.append(o)
    exe = os.path.join(tmp, "probe")
    subprocess.check_call(["g++", "-o", exe] + objs)
    # the problem initialisers (Exec/hydro_tests/{Sedov,Sod}) in their own executable
    o = os.path.join(tmp, "probe_init.o")
    subprocess.check_call(["g++"] + flags + inc + ["-DREFERENCE_EXEC=" + os.path.join(REF, "Exec", "hydro_tests"), "-c",
                                                   os.path.join(HERE, "probe_init.cpp"), "-o", o])
    subprocess.check_call(["g++", "-o", exe + "_init", o, os.path.join(tmp, "probe_params.o")])
    # the derived fields (Source/driver/Derive.cpp, unmodified)
    od = [os.path.join(tmp, "Derive.o"), os.path.join(tmp, "probe_derive.o"), os.path.join(tmp, "timestep.o")]
    subprocess.check_call(["g++"] + flags + inc + ["-c", os.path.join(src, "driver", "Derive.cpp"), "-o", od[0]])
    subprocess.check_call(["g++"] + flags + inc + ["-c", os.path.join(src, "driver", "timestep.cpp"), "-o", od[2]])
    subprocess.check_call(["g++"] + flags + inc + ["-c", os.path.join(HERE, "probe_derive.cpp"), "-o", od[1]])
    subprocess.check_call(["g++", "-o", exe + "_derive"] + od + [os.path.join(tmp, "probe_params.o")])
    # the rotation sources (Source/rotation/rotation_sources.cpp, Rotation.cpp, unmodified)
    orot = []
    for path in (os.path.join(src, "rotation", "rotation_sources.cpp"), os.path.join(src, "rotation", "Rotation.cpp"),
                 os.path.join(HERE, "probe_rotation.cpp")):
        o = os.path.join(tmp, "rot_" + os.path.basename(path)[:-4] + ".o")
        subprocess.check_call(["g++"] + flags + inc + ["-c", path, "-o", o])
        orot.append(o)
    subprocess.check_call(["g++", "-o", exe + "_rotation"] + orot + [os.path.join(tmp, "probe_params.o")])
    return exe


def edge_states(rng, n, gam=1.4, cold=0.15):
    """(7, n) edge states (rho,u,v,w,p,rhoe,X): jumps of many decades, supersonic flows, some with rho e <= 0 or a tiny
    pressure (the EOS clean-up of load_input_states)"""
    rho = 10.0 ** rng.uniform(-3, 2, n)
    vel = rng.normal(size=(3, n)) * 10.0 ** rng.uniform(-2, 1, n)
    p = 10.0 ** rng.uniform(-5, 3, n)
    rhoe = p / (gam - 1.0) * rng.choice([1.0, 1.0, 1.0, 0.6, 2.5], n)
    bad = rng.uniform(size=n) < cold * 0.2
    rhoe = np.where(bad, -rhoe, rhoe)
    X = rng.uniform(size=n)
    return np.stack([rho, vel[0], vel[1], vel[2], p, rhoe, X])


def flux_records(rng, n):
    """(9, n): rho, mx, my, mz, E, X fluxes, Godunov un, Godunov p, (rho e) flux"""
    f = rng.normal(size=(9, n)) * 10.0 ** rng.uniform(-2, 1, n)
    f[7] = 10.0 ** rng.uniform(-4, 2, n)
    return f


def main():
    rng = np.random.default_rng(20211007)
    A = {}
    # ---- ppm ----
    n = 6000
    kind = rng.integers(0, 5, n)
    x = rng.uniform(0, 6.3, n)
    s = np.empty((5, n))
    for m in range(5):
        smooth = np.sin(x + 0.4 * m) + 2.0
        noisy = rng.normal(size=n)
        mono = np.cumsum(rng.uniform(0, 1, (5, n)), axis=0)[m]
        step = np.where(m < 2 + rng.integers(0, 2, n), 1.0, rng.uniform(0.01, 100.0, n))
        flat = np.full(n, 3.0) + (m == 2) * rng.choice([0.0, 1e-13], n)
        s[m] = np.choose(kind, [smooth, noisy, mono, step, flat])
    A["ppm.s"] = s
    A["ppm.flat"] = rng.choice([0.0, 1.0, 1.0, 0.37], n) * np.where(rng.uniform(size=n) < 0.2, rng.uniform(size=n), 1.0)
    A["ppm.u"] = rng.normal(scale=2.0, size=n)
    A["ppm.c"] = 10.0 ** rng.uniform(-2, 1, n)
    A["ppm.dtdx"] = 0.27
    # ---- flattening along one direction ----
    n = 4000
    base = 10.0 ** rng.uniform(-3, 2, n)
    p7 = base * (1.0 + 0.05 * rng.normal(size=(7, n)))
    jump = rng.integers(0, 8, n)
    for m in range(7):
        p7[m] = np.where(m >= jump, p7[m] * rng.choice([1.0, 1.5, 4.0, 50.0, 0.02], n), p7[m])
    A["flat.p"] = np.abs(p7)
    A["flat.u"] = rng.normal(size=(5, n)) + np.linspace(1.5, -1.5, 5)[:, None] * rng.choice([0.0, 1.0, -1.0], n)
    # ---- cmpflx_plus_godunov ----
    cfgs = [dict(idir=0), dict(idir=1), dict(idir=2, wall=1), dict(idir=0, riemann_solver=1, cg_blend=2),
            dict(idir=1, riemann_solver=1, cg_blend=1), dict(idir=2, riemann_solver=1, cg_blend=1, wall=1),
            dict(idir=0, riemann_solver=2), dict(idir=2, riemann_solver=2, wall=1), dict(idir=0, hybrid_riemann=1),
            dict(idir=1, riemann_solver=2, hybrid_riemann=1), dict(idir=1, small_pres=1e-4, small_dens=1e-2),
            dict(idir=2, riemann_solver=1, cg_blend=2, small_pres=1e-4)]
    for c, cfg in enumerate(cfgs):
        n = 450
        P = "cmpflx%d." % c
        qm, qp = edge_states(rng, n), edge_states(rng, n)
        same = rng.uniform(size=n) < 0.15
        qp[:, same] = qm[:, same] * (1.0 + 1e-3 * rng.normal(size=(7, int(same.sum()))))
        qp[0] = np.abs(qp[0]); qp[4] = np.abs(qp[4])
        cz = 10.0 ** rng.uniform(-2, 1.5, n + 1)
        A[P + "qm"], A[P + "qp"], A[P + "c"] = qm, qp, cz
        A[P + "shk"] = (rng.uniform(size=n + 1) < 0.3).astype(float) if cfg.get("hybrid_riemann") else np.zeros(n + 1)
        A[P + "wall"] = float(cfg.get("wall", 0))
        for k, v in cfg.items():
            if k != "wall":
                A[P + k] = float(v)
    # ---- trans_single ----
    tcfgs = [dict(idir_t=0, idir_n=1), dict(idir_t=1, idir_n=0), dict(idir_t=2, idir_n=0), dict(idir_t=0, idir_n=2),
             dict(idir_t=1, idir_n=2), dict(idir_t=2, idir_n=1), dict(idir_t=0, idir_n=1, transverse_reset_density=0),
             dict(idir_t=1, idir_n=2, transverse_reset_rhoe=1), dict(idir_t=2, idir_n=0, transverse_use_eos=1)]
    for c, cfg in enumerate(tcfgs):
        n = 500
        P = "trans1_%d." % c
        q = edge_states(rng, n, cold=0.0)
        q[5] = np.abs(q[5])
        A[P + "q"] = q
        A[P + "flux"] = flux_records(rng, n + 1) * rng.choice([1.0, 1.0, 30.0], n + 1)     # some large enough to flip the density
        A[P + "qt"] = np.zeros(1)
        A[P + "cdtdx"] = 0.011
        for k, v in cfg.items():
            A[P + k] = float(v)
    # ---- trans_final ----
    fcfgs = [dict(idir_n=0, idir_t1=1, idir_t2=2), dict(idir_n=1, idir_t1=0, idir_t2=2), dict(idir_n=2, idir_t1=0, idir_t2=1),
             dict(idir_n=0, idir_t1=1, idir_t2=2, transverse_reset_density=0), dict(idir_n=2, idir_t1=0, idir_t2=1, transverse_reset_rhoe=1)]
    for c, cfg in enumerate(fcfgs):
        n = 500
        P = "trans2_%d." % c
        q = edge_states(rng, n, cold=0.0)
        q[5] = np.abs(q[5])
        A[P + "q"] = q
        A[P + "flux1"] = flux_records(rng, n + 1) * rng.choice([1.0, 1.0, 20.0], n + 1)
        A[P + "flux2l"] = flux_records(rng, n)
        A[P + "flux2r"] = flux_records(rng, n) * rng.choice([1.0, 1.0, 20.0], n)
        A[P + "cdtdx1"], A[P + "cdtdx2"] = 0.017, 0.013
        for k, v in cfg.items():
            A[P + k] = float(v)
    # ---- a block through ctoprim, uflatten, trace_ppm ----
    nb = 6
    U = physical_state(rng, (-4, -4, -4), (nb + 3, nb + 3, nb + 3), smooth=False, vel=1.5, jump=True)
    A["block.U"], A["block.n"], A["block.dt"], A["block.dx"] = U, float(nb), 7.0e-4, np.array([0.02, 0.025, 0.03])

    # ---- whole tiles of construct_ctu_hydro_source (probe.cpp drives the reference's own functions) ----
    nb = 6
    glo, ghi = (-4, -4, -4), (nb + 3, nb + 3, nb + 3)
    floors = dict(small_dens=1e-8, small_pres=1e-10, small_temp=1e-10, small_ener=1e-12)
    hcfgs = [dict(), dict(riemann_solver=1, cg_blend=2), dict(riemann_solver=2, ppm_type=0), dict(hybrid_riemann=1),
             dict(ppm_temp_fix=2), dict(transverse_reset_rhoe=1, transverse_use_eos=1, transverse_reset_density=0),
             dict(limit_fluxes_on_small_dens=1, limit_fluxes_on_large_vel=1, speed_limit=2.0, small_dens=0.1, cfl=0.5),
             dict(wall_lo=1), dict(src=1), dict(src=1, ppm_type=0, source_term_predictor=1, corr=1, use_pslope=1),
             dict(first_order_hydro=1), dict(use_flattening=0, difmag=0.0, riemann_solver=1, cg_blend=1),
             dict(ppm_type=0, plm_iorder=1), dict(ppm_type=0, plm_limiter=1, hybrid_riemann=1, riemann_solver=2)]
    for c, cfg in enumerate(hcfgs):
        P = "hydro%d." % c
        U = physical_state(rng, glo, ghi, smooth=(c % 3 == 0), vel=1.5 if c % 2 else 0.7, jump=True)
        if cfg.get("wall_lo"):                      # reflect the ghost zones below index 0 like a SlipWall fill would
            for d, ax in enumerate((3, 2, 1)):
                idx = [slice(None)] * 4
                for g in range(4):
                    src_i, dst_i = list(idx), list(idx)
                    src_i[ax], dst_i[ax] = 4 + g, 3 - g
                    U[tuple(dst_i)] = U[tuple(src_i)]
                    U[(1 + d,) + tuple(dst_i[1:])] *= -1.0
        A[P + "U"], A[P + "n"], A[P + "dt"], A[P + "dx"] = U, float(nb), 6.0e-3, np.array([0.05, 0.055, 0.045])
        if cfg.get("src"):
            S = np.zeros((7,) + U.shape[1:])
            g = np.array([0.3, -9.0, 1.5])
            for d in range(3):
                S[1 + d] = U[0] * g[d]
                S[4] += U[1 + d] * g[d]
            A[P + "src"] = S
        if cfg.get("corr"):
            A[P + "corr"] = 0.4 * A[P + "src"] * rng.uniform(0.5, 1.5, size=A[P + "src"].shape)
        for k, v in dict(floors, **cfg).items():
            if k not in ("src", "corr"):
                A[P + k] = float(v)

    # ---- problem initialisers: Exec/hydro_tests/Sedov and Sod (problem_initialize + problem_initialize_state_data) ----
    B = {}
    sed = [dict(n=(16, 16, 16), problo=(0., 0., 0.), probhi=(1., 1., 1.), r_init=0.01, p_ambient=1.e-5, exp_energy=1.0, dens_ambient=1.0, nsub=10),
           dict(n=(12, 10, 14), problo=(-0.5, 0., 0.25), probhi=(0.7, 1.1, 1.3), r_init=0.2, p_ambient=1.e-3, exp_energy=2.5, dens_ambient=0.7, nsub=4),
           dict(n=(8, 8, 8), problo=(0., 0., 0.), probhi=(1., 1., 1.), r_init=0.3, p_ambient=1.e-5, exp_energy=1.0, dens_ambient=1.0, nsub=5)]
    for c, cfg in enumerate(sed):
        for k, v in cfg.items():
            B["sedov%d.%s" % (c, k)] = np.atleast_1d(np.asarray(v, dtype=np.float64))
    sod = [dict(n=(32, 4, 4), problo=(0., 0., 0.), probhi=(1., 0.125, 0.125), idir=1, frac=0.5, left=(1.0, 0.0, 1.0), right=(0.125, 0.0, 0.1)),
           dict(n=(4, 24, 4), problo=(0.1, -0.2, 0.), probhi=(0.35, 1.3, 0.25), idir=2, frac=0.3, left=(1.0, 0.75, 1.0), right=(0.125, -2.0, 0.4)),
           dict(n=(4, 6, 20), problo=(0., 0., 0.), probhi=(0.2, 0.3, 1.0), idir=3, frac=0.5, left=(5.99924, 19.5975, 460.894), right=(5.99242, -6.19633, 46.0950))]
    for c, cfg in enumerate(sod):
        for k, v in cfg.items():
            B["sod%d.%s" % (c, k)] = np.atleast_1d(np.asarray(v, dtype=np.float64))

    with tempfile.TemporaryDirectory() as tmp:
        exe = build(tmp)
        write_blob(os.path.join(tmp, "in.bin"), A)
        subprocess.check_call([exe, os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")])
        O = read_blob(os.path.join(tmp, "out.bin"))
        write_blob(os.path.join(tmp, "in2.bin"), B)
        subprocess.check_call([exe + "_init", os.path.join(tmp, "in2.bin"), os.path.join(tmp, "out2.bin")])
        O.update(read_blob(os.path.join(tmp, "out2.bin")))
        # ---- derived fields on one box (state with one ghost zone for the vorticity and the divergence) ----
        dn = (10, 9, 8)
        D = {"derive.n": np.array(dn, dtype=np.float64), "derive.dx": np.array([0.05, 0.04, 0.0625]), "derive.problo": np.array([-0.2, 0.1, 0.0]),
             "derive.center": np.array([0.05, 0.28, 0.25]),
             "derive.U": physical_state(rng, (-1, -1, -1), dn, smooth=False, vel=1.5, jump=True)}
        write_blob(os.path.join(tmp, "in3.bin"), D)
        subprocess.check_call([exe + "_derive", os.path.join(tmp, "in3.bin"), os.path.join(tmp, "out3.bin")])
        for k, v in read_blob(os.path.join(tmp, "out3.bin")).items():
            if k == "derive.estdt":
                O[k] = v
                continue
            nc = v.size // ((dn[0] + 2) * (dn[1] + 2) * (dn[2] + 2))
            O[k] = v.reshape(nc, dn[2] + 2, dn[1] + 2, dn[0] + 2)[:, 1:-1, 1:-1, 1:-1].copy()
        B.update(D)
        # ---- rotation sources: rsrc on the old state, corrrsrc with old and new state and the mass fluxes ----
        R = {}
        rn = (7, 6, 5)
        rcfg = [dict(axis=3, rot_source_type=4, implicit=1, centrifugal=1, coriolis=1), dict(axis=1, rot_source_type=1, implicit=0, centrifugal=1, coriolis=1),
                dict(axis=2, rot_source_type=2, implicit=1, centrifugal=0, coriolis=1), dict(axis=3, rot_source_type=3, implicit=1, centrifugal=1, coriolis=0),
                dict(axis=3, rot_source_type=4, implicit=0, centrifugal=1, coriolis=1), dict(axis=1, rot_source_type=3, implicit=0, centrifugal=1, coriolis=1)]
        for c, cfg in enumerate(rcfg):
            P = "rot%d." % c
            R[P + "n"], R[P + "dx"], R[P + "problo"] = np.array(rn, dtype=np.float64), np.array([0.1, 0.12, 0.15]), np.array([-0.3, 0.0, 0.2])
            R[P + "center"], R[P + "period"], R[P + "dt"] = np.array([0.05, 0.36, 0.55]), np.array([2.5]), np.array([0.02])
            hi = tuple(x - 1 for x in rn)
            R[P + "uold"] = physical_state(rng, (0, 0, 0), hi, smooth=False, vel=1.5, jump=True)
            R[P + "unew"] = R[P + "uold"] * rng.uniform(0.9, 1.1, size=R[P + "uold"].shape)
            for d in range(3):
                shp = [rn[2], rn[1], rn[0]]
                shp[2 - d] += 1
                R[P + "mflux%d" % d] = rng.normal(scale=1e-4, size=shp)
            for k, v in cfg.items():
                R[P + k] = np.array([float(v)])
        write_blob(os.path.join(tmp, "in4.bin"), R)
        subprocess.check_call([exe + "_rotation", os.path.join(tmp, "in4.bin"), os.path.join(tmp, "out4.bin")])
        O.update(read_blob(os.path.join(tmp, "out4.bin")))
        B.update(R)
    A.update(B)
    # whole-tile outputs: keep the zones and faces the call defines (everything lives on the box grown by 4 in the probe)
    m = nb + 8
    for c in range(len(hcfgs)):
        P = "hydro%d." % c
        O[P + "unew"] = O[P + "unew"].reshape(8, m, m, m)[:, 4:4 + nb, 4:4 + nb, 4:4 + nb].copy()
        for d in range(3):
            sl = [slice(4, 4 + nb + (1 if ax == d else 0)) for ax in (2, 1, 0)]
            O[P + "flux%d" % d] = O[P + "flux%d" % d].reshape(8, m, m, m)[(slice(None),) + tuple(sl)].copy()
            O[P + "qe%d" % d] = O[P + "qe%d" % d].reshape(4, m, m, m)[(slice(None),) + tuple(sl)].copy()
        for k in ("div", "shk", "srcq"):
            O.pop(P + k)
    dst = os.path.join(ROOT, "tests", "golden", "stub_probe")
    os.makedirs(dst, exist_ok=True)
    allv = {"in:" + k: np.atleast_1d(np.asarray(v, dtype=np.float64)) for k, v in A.items()}
    allv.update({"out:" + k: v for k, v in O.items()})
    np.savez_compressed(os.path.join(dst, "vectors.npz"), **allv)
    print("wrote %s: %d input arrays, %d output arrays, %.1f KB" % (
        os.path.join(dst, "vectors.npz"), len(A), len(O), os.path.getsize(os.path.join(dst, "vectors.npz")) / 1024.0))
    for k in sorted(O):
        print("  %-16s %8d values, %d NaN" % (k, O[k].size, int(np.isnan(O[k]).sum())))


if __name__ == "__main__":
    main()
