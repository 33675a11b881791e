"""GPU parity tests of the `contract` numerics mode (castro_amd/libcastro_hydro_amd_contract.so: FMA contraction,
reciprocal-based division, rsq-based sqrt) against the CPU oracle, at the north star's tolerance.

Tolerance, written here once: for EVERY field f of the plotfile (the 8 state components and the 25 derived fields the
reference registers, Source/driver/Castro_setup.cpp:756-960)

        max |f_hip - f_oracle|  <=  RTOL * scale(f),        RTOL = 1e-10,

with scale(f) = max |f_oracle| -- the relative error AMReX's fcompare prints for a plotfile field -- except for the three
derives that are differences of nearly equal terms: divu and magvort (differences of neighbouring zones' velocities: scale
max |velocity| / dx) and circvel = sqrt(|v|^2 - v_r^2), which is compared through its square (scale max |velocity|^2):
on a spherical blast |v|^2 - v_r^2 is a rounding residue of |v|^2, so the field itself is sqrt(1e-16) |v| of noise in ANY
build -- two compilations of the reference differ there by as much.  The `exact`
mode is held to bit equality by tests/test_gpu_parity.py; this file holds `contract` to the tolerance and prints the
measured deviation per check point (pytest -s), after 1, 10, 100 steps and at the stop time, as SURVEY.md section 7 asks.
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-10


def _oracle_fields(oracle, lev, G, P):
    """name -> array for the 33 plotfile fields of the oracle's state"""
    from castro_amd import plotfile as pf
    from castro_amd._lib import DERIVE_IDS
    S = lev.state()
    nx, ny, nz = lev.n
    lo1, hi1 = (-1, -1, -1), (nx, ny, nz)
    Sg = np.zeros((8, nz + 2, ny + 2, nx + 2))
    Sg[:, 1:-1, 1:-1, 1:-1] = S
    oracle.lib().ora_bc_fill(oracle.a4(Sg, lo1, hi1), C.byref(G))
    ctr = (C.c_double * 3)(*[0.5 * (G.problo[d] + G.probhi[d]) for d in range(3)])      # problem.center of the inputs files
    out = {nm: S[m].copy() for m, nm in enumerate(pf.STATE_NAMES)}
    for nm in pf.DERIVE_NAMES:
        want = np.zeros((1, nz, ny, nx))
        rc = oracle.lib().ora_derive(DERIVE_IDS[nm], oracle.i3((0, 0, 0)), oracle.i3((nx - 1, ny - 1, nz - 1)),
                                     oracle.a4(Sg, lo1, hi1), oracle.a4(want, (0, 0, 0), (nx - 1, ny - 1, nz - 1)),
                                     C.byref(G), C.byref(P), C.byref(ctr))
        assert rc == 0, nm
        out[nm] = want[0]
    return out


def _hip_fields(c, tmp_path=None, tag=None):
    """name -> array of the plotMF the HIP run would write (castro_amd.plotfile.plot_data: the state components and the
    derived fields of Castro_io.cpp:1099-1126; the on-disk format is covered by test_gpu_parity / test_plotfile_cpu)"""
    import torch
    from castro_amd import plotfile as pf
    names, data = pf.plot_data(c)
    torch.cuda.synchronize()
    arr = data.cpu().numpy()
    return dict(zip(names, arr))


def field_deviation(got, want, dx):
    """{field: deviation / scale}; see the module docstring for scale"""
    vmax = max(np.abs(want[k]).max() for k in ("x_velocity", "y_velocity", "z_velocity"))
    dev = {}
    for nm, b in want.items():
        a = got[nm]
        scale = np.abs(b).max()
        if nm in ("divu", "magvort"):
            scale = max(scale, vmax / dx)
        if nm == "circvel":
            a, b, scale = a * a, b * b, max(scale * scale, vmax * vmax)
        d = np.abs(a - b).max()
        dev[nm] = d / scale if scale > 0.0 else d
    return dev


# Zone by zone.  The deviation of the `contract` build is an ABSOLUTE error of <= 1e-13 x scale(f) (the norm-relative figure above:
# rounding differences of order 1e-16 x scale accumulated over the steps), so a relative bound per zone can only hold above a floor:
#   |f_hip - f_oracle| <= 1e-10 |f_oracle|   on zones with |f_oracle| >= 1e-3 x scale(f)      (asserted)
#   |f_hip - f_oracle| <= 1e-8  |f_oracle|   on zones with |f_oracle| >= 1e-6 x scale(f)      (asserted; measured worst 7.7e-10: a
#       velocity of 2.6e-5 where the field reaches 10 -- Sedov 128^3 at t = 0.01, profiles/r06d_*)
ELEM_BOUNDS = ((1e-3, RTOL), (1e-6, 1e-8))       # (floor as a fraction of scale(f), relative bound)
ELEM_FLOOR = ELEM_BOUNDS[0][0]


def elementwise_deviation(got, want, dx, floor=ELEM_FLOOR):
    """{field: (max over the zones with |f_oracle| >= floor * scale(f) of |f_hip - f_oracle| / |f_oracle|, zone index (k, j, i), value there)}
    -- the zone-by-zone relative error above a floor, next to the norm-relative figure of field_deviation (round 6: asserted, not
    only printed by tools/numerics_deviation.py).  scale(f) as in the module docstring; circvel through its square; logden =
    log10(rho) is measured against max(|log10 rho|, 1): next to rho = 1 the logarithm is the rounding residue of the density
    itself (a density right to 1e-16 gives log10 rho = 1.4e-7 to 7e-10 only), so its error is that of rho, not a fraction of itself."""
    vmax = max(np.abs(want[k]).max() for k in ("x_velocity", "y_velocity", "z_velocity"))
    out = {}
    for nm, b in want.items():
        a = got[nm]
        scale = np.abs(b).max()
        if nm in ("divu", "magvort"):
            scale = max(scale, vmax / dx)
        if nm == "circvel":
            a, b, scale = a * a, b * b, max(scale * scale, vmax * vmax)
        mask = np.abs(b) >= floor * scale
        if scale == 0.0 or not mask.any():
            out[nm] = (0.0, None, 0.0)
            continue
        den = np.maximum(np.abs(b), 1.0) if nm == "logden" else np.abs(b)
        rel = np.where(mask, np.abs(a - b) / np.where(mask, den, 1.0), 0.0)
        idx = np.unravel_index(np.argmax(rel), rel.shape)
        out[nm] = (float(rel[idx]), tuple(int(x) for x in idx), float(b[idx]))
    return out


def _check(c, lev, oracle, G, P, tmp_path, tag, dx, elementwise=False):
    got = _hip_fields(c, tmp_path, tag)
    want = _oracle_fields(oracle, lev, G, P)
    dev = field_deviation(got, want, dx)
    worst = max(dev, key=dev.get)
    print("contract vs oracle, %-28s step %4d t = %.6e: max deviation %.2e (%s); dt deviation %.1e"
          % (tag, c.nstep, c.time, dev[worst], worst, abs(c.dt - lev.dt) / lev.dt))
    bad = {k: v for k, v in dev.items() if not v <= RTOL}
    assert not bad, "%s: fields beyond rtol %g: %s" % (tag, RTOL, bad)
    assert abs(c.time - lev.time) <= RTOL * lev.time
    if elementwise:
        for floor, bound in ELEM_BOUNDS:
            el = elementwise_deviation(got, want, dx, floor)
            w = max(el, key=lambda k: el[k][0])
            print("    elementwise above %g x scale:      worst zone %s of %s (value %.3e): relative deviation %.2e (bound %g)"
                  % (floor, el[w][1], w, el[w][2], el[w][0], bound))
            bad = {k: v for k, v in el.items() if not v[0] <= bound}
            assert not bad, "%s: zones above %g x scale beyond the elementwise rtol %g: %s" % (tag, floor, bound, bad)
    return dev[worst]


def test_contract_library_is_the_contract_build():
    import castro_amd
    h = castro_amd.HipHydro(0, numerics="contract")
    assert h.numerics == "contract" and b"numerics=contract" in h.lib.castro_amd_version()
    e = castro_amd.HipHydro(0, numerics="exact")
    assert e.numerics == "exact"
    h.close()
    e.close()


def test_contract_sedov_64_plotfile_fields_within_rtol_after_1_10_100_steps_and_at_stop_time(tmp_path, oracle):
    """Config 1 (Sedov 3-D 64^3): all 33 plotfile fields after 1, 10, 100 steps and at t = 0.01."""
    import castro_amd
    n = (64, 64, 64)
    c = castro_amd.Castro(n, numerics="contract")
    assert c.hydro.numerics == "contract"
    c.initData("sedov")
    G, P = oracle.make_geom(n), oracle.default_params()
    lev = oracle.Level(n, G, P, nthreads=0)
    lev.init_sedov()
    worst = 0.0
    while c.time < 0.01 - 2.3e-16:
        c.step(0.01)
        lev.step(0.01)
        if c.nstep in (1, 10, 100):
            worst = max(worst, _check(c, lev, oracle, G, P, tmp_path, "sedov64_step%d" % c.nstep, 1.0 / 64, elementwise=True))
    assert c.nstep == lev.nstep
    worst = max(worst, _check(c, lev, oracle, G, P, tmp_path, "sedov64_t0.01", 1.0 / 64, elementwise=True))
    assert worst > 0.0, "the contract build gave the exact build's bits: is it the right library?"
    lev.close()


def test_contract_sedov_128_against_the_oracle_all_the_way_to_the_stop_time(tmp_path, oracle):
    """The end state above 64^3 against the ORACLE (until round 6 the only stop-time comparison above 64^3 was GPU against GPU):
    Sedov 128^3 -- a size that takes the large-box kernels of the `contract` build (k_trans1_tile from 96 rows up) -- with the
    reference's inputs (Exec/hydro_tests/Sedov/inputs.3d.sph: stop_time 0.01) on both sides, about 370 steps, all 33 plotfile
    fields after 1, 10, 100 steps and at t = 0.01, norm-relative and zone by zone."""
    import castro_amd
    n = (128, 128, 128)
    c = castro_amd.Castro(n, numerics="contract")
    assert c.hydro.numerics == "contract"
    c.initData("sedov")
    G, P = oracle.make_geom(n), oracle.default_params()
    lev = oracle.Level(n, G, P, nthreads=0)
    lev.init_sedov()
    while c.time < 0.01 - 2.3e-16:
        c.step(0.01)
        lev.step(0.01)
        if c.nstep in (1, 10, 100):
            _check(c, lev, oracle, G, P, tmp_path, "sedov128_step%d" % c.nstep, 1.0 / 128, elementwise=True)
    assert c.nstep == lev.nstep and c.nstep > 300
    _check(c, lev, oracle, G, P, tmp_path, "sedov128_t0.01", 1.0 / 128, elementwise=True)
    lev.close()


@pytest.mark.parametrize("name,rl,ul,pl,rr,ur,pr,stop", [("sod", 1.0, 0.0, 1.0, 0.125, 0.0, 0.1, 0.2),
                                                        ("test2", 1.0, -2.0, 0.4, 1.0, 2.0, 0.4, 0.15),
                                                        ("test3", 1.0, 0.0, 1000.0, 1.0, 0.0, 0.01, 0.012)])
def test_contract_shock_tubes_within_rtol(tmp_path, oracle, name, rl, ul, pl, rr, ur, pr, stop):
    """The reference's shock tubes (Exec/hydro_tests/Sod: inputs-sod-x, inputs-test2-x, inputs-test3-x) on 128 x 8 x 8 zones to
    their stop times, checked after 1, 10, 100 steps and at the end."""
    import castro_amd
    n = (128, 8, 8)
    hi = (1.0, 8.0 / 128, 8.0 / 128)
    c = castro_amd.Castro(n, prob_hi=hi, numerics="contract")
    c.initData("sod", rho_l=rl, u_l=ul, p_l=pl, rho_r=rr, u_r=ur, p_r=pr)
    G, P = oracle.make_geom(n, probhi=hi), oracle.default_params()
    lev = oracle.Level(n, G, P, nthreads=0)
    lev.init_sod(rl, ul, pl, rr, ur, pr)
    while c.time < stop - 2.3e-16:
        c.step(stop)
        lev.step(stop)
        if c.nstep in (1, 10, 100):
            _check(c, lev, oracle, G, P, tmp_path, "%s_step%d" % (name, c.nstep), 1.0 / 128)
    assert c.nstep == lev.nstep
    _check(c, lev, oracle, G, P, tmp_path, "%s_stop" % name, 1.0 / 128)
    lev.close()


def test_contract_sedov_256_four_steps_against_oracle_and_to_stop_time_against_exact(tmp_path, oracle):
    """Config 2 (the bench configuration): 4 steps against the oracle at 256^3, then both GPU builds to t = 0.01 (about 730
    steps; the exact build is bit-identical to the oracle, tests/test_gpu_parity.py::test_256_cubed_against_oracle) and all
    plotfile fields of the two compared at the same tolerance."""
    import torch
    import castro_amd
    from castro_amd import plotfile as pf
    n = (256, 256, 256)
    c = castro_amd.Castro(n, numerics="contract")
    c.initData("sedov")
    G, P = oracle.make_geom(n), oracle.default_params()
    lev = oracle.Level(n, G, P, nthreads=0)
    lev.init_sedov()
    for _ in range(4):                    # the quiet start; the developed state has its own test below (step 700 + 3 against the oracle)
        c.step(0.01)
        lev.step(0.01)
    torch.cuda.synchronize()
    # the state against the oracle (the derived fields are zone-local functions of it, checked at 64^3 above)
    got, want = c.S_new().cpu().numpy(), lev.state()
    dev = {nm: np.abs(got[m] - want[m]).max() / np.abs(want[m]).max() for m, nm in enumerate(pf.STATE_NAMES)}
    worst = max(dev, key=dev.get)
    print("contract vs oracle, Sedov 256^3 after 4 steps: max deviation %.2e (%s); dt deviation %.1e"
          % (dev[worst], worst, abs(c.dt - lev.dt) / lev.dt))
    assert all(v <= RTOL for v in dev.values()), dev
    lev.close()
    del got, want
    e = castro_amd.Castro(n, numerics="exact")
    e.initData("sedov")
    e.evolve(0.01)
    c.evolve(0.01)
    torch.cuda.synchronize()
    assert c.nstep == e.nstep and abs(c.time - e.time) <= RTOL * e.time
    # all 33 plotfile fields of the two builds, compared on the device
    names, a = pf.plot_data(c)
    _, b = pf.plot_data(e)
    vmax = max(b[names.index(k)].abs().max().item() for k in ("x_velocity", "y_velocity", "z_velocity"))
    dev = {}
    for m, nm in enumerate(names):
        scale = b[m].abs().max().item()
        if nm in ("divu", "magvort"):
            scale = max(scale, vmax * 256.0)
        if nm == "circvel":
            scale = max(scale * scale, vmax * vmax)
            d = (a[m] * a[m] - b[m] * b[m]).abs().max().item()
            dev[nm] = d / scale if scale > 0.0 else d
            continue
        d = (a[m] - b[m]).abs().max().item()
        dev[nm] = d / scale if scale > 0.0 else d
    worst = max(dev, key=dev.get)
    print("contract vs exact (GPU), Sedov 256^3 at t = %.4f after %d steps: max deviation %.2e (%s)" % (c.time, c.nstep, dev[worst], worst))
    bad = {k: v for k, v in dev.items() if not v <= RTOL}
    assert not bad, bad


def test_contract_sedov_256_developed_state_three_steps_against_oracle(tmp_path, oracle):
    """The bench configuration on the developed blast wave against the ORACLE (not only against the other GPU build): 700 steps of
    the `contract` build at 256^3, the state handed to the oracle's level driver, three more steps on both, all 33 plotfile
    fields within rtol 1e-10."""
    import torch
    import castro_amd
    n = (256, 256, 256)
    c = castro_amd.Castro(n, numerics="contract")
    c.initData("sedov")
    c.evolve(0.01, max_step=700)
    torch.cuda.synchronize()
    assert c.nstep == 700 and c.hydro.numerics == "contract"
    G, P = oracle.make_geom(n), oracle.default_params()
    lev = oracle.Level(n, G, P, nthreads=0)
    lev.set_state(c.S_new().cpu().numpy(), c.time, c.dt, c.nstep)
    for _ in range(3):
        c.step(0.01)
        lev.step(0.01)
    torch.cuda.synchronize()
    _check(c, lev, oracle, G, P, tmp_path, "sedov256_developed", 1.0 / 256, elementwise=True)
    lev.close()
    del c
    torch.cuda.empty_cache()


def test_contract_one_256_cubed_box_of_the_512_cubed_decomposition_within_rtol(oracle):
    """Config 3's per-rank box with its neighbours' ghost data on a developed state (tests/test_gpu_parity.py has the `exact`
    twin): every output array of one `contract` call on the 256^3 corner box of a 512^3 Sedov run against the oracle."""
    from castro_amd.hydro import HipHydro
    from tests.test_gpu_parity import _run_both, _corner_box_of_512
    U, dt, t = _corner_box_of_512()
    hip = HipHydro(0, numerics="contract")
    out = _run_both(hip, oracle, (0, 0, 0), (255, 255, 255), U, (-4, -4, -4), (259, 259, 259), dt, dx=(1.0 / 512,) * 3)
    dev = _outputs_deviation(out)
    worst = max(dev, key=dev.get)
    print("contract vs oracle, 256^3 corner box of 512^3 at t = %.3e: max deviation %.2e (%s)" % (t, dev[worst], worst))
    assert all(v <= RTOL for v in dev.values()), dev
    hip.close()


@pytest.mark.parametrize("case", ["plm", "plm_iorder1", "plm_limiter1", "plm_nopslope", "hllc", "hybrid", "cg", "gravity"])
def test_contract_non_default_options_within_rtol(oracle, case):
    """The `contract` build on the option sets that leave its default-solver path (whose edge states carry no (rho e) plane,
    gamma_law_edges): HLLC, the hybrid solver, Colella-Glaz; and on those that keep it through identities of their own: a
    constant-gravity source (traced source terms) and -- since round 6 -- PLM with its slope options (trace_plm_dir<D, SRC, GL>):
    Sedov 32^3, 30 steps, the conserved state and dt against the oracle at the same tolerance."""
    import torch
    import castro_amd
    pkw = {"plm": dict(ppm_type=0), "plm_iorder1": dict(ppm_type=0, plm_iorder=1), "plm_limiter1": dict(ppm_type=0, plm_limiter=1),
           "plm_nopslope": dict(ppm_type=0, use_pslope=0), "hllc": dict(riemann_solver=2), "hybrid": dict(hybrid_riemann=1),
           "cg": dict(riemann_solver=1), "gravity": {}}[case]
    n = (32, 32, 32)
    grav = dict(do_grav=True, const_grav=-2.0) if case == "gravity" else {}
    c = castro_amd.Castro(n, params=castro_amd.default_params(**pkw), numerics="contract", **grav)
    c.initData("sedov", r_init=0.1, nsub=4)
    lev = oracle.Level(n, oracle.make_geom(n), oracle.default_params(**pkw), nthreads=0)
    if case == "gravity":
        lev.set_gravity(-2.0)
    lev.init_sedov(r_init=0.1, nsub=4)
    for _ in range(30):
        c.step(0.05)
        lev.step(0.05)
    torch.cuda.synchronize()
    got, want = c.S_new().cpu().numpy(), lev.state()
    dev = {k: np.abs(got[k] - want[k]).max() / max(np.abs(want[k]).max(), 1e-300) for k in range(8)}
    # the momentum components of a blast centred in the box are of one size: each is held to the largest of the three
    mom = max(np.abs(want[k]).max() for k in (1, 2, 3))
    for k in (1, 2, 3):
        dev[k] = np.abs(got[k] - want[k]).max() / mom
    print("contract vs oracle, %-8s 30 steps: max deviation %.2e; dt deviation %.1e" % (case, max(dev.values()), abs(c.dt - lev.dt) / lev.dt))
    assert all(v <= RTOL for v in dev.values()), dev
    assert abs(c.dt - lev.dt) <= RTOL * lev.dt
    lev.close()


def test_contract_amr_two_refined_levels_within_rtol_of_exact():
    """Config 4's structure (base + two refined levels, subcycled, refluxed; a level is one grid per kernel) in the `contract`
    mode against the `exact` mode (which the AMR parity tests tie to the oracle backend bit for bit): every box of every level
    at the tolerance after eight coarse steps, same dt."""
    import torch
    import castro_amd
    from castro_amd.hydro import HipHydro
    kw = dict(patches=[[((4, 4, 4), (11, 11, 11))], [((10, 10, 10), (15, 15, 15)), ((16, 10, 10), (19, 15, 15))]])
    runs = {}
    for mode in ("exact", "contract"):
        a = castro_amd.CastroAmr((16, 16, 16), params=castro_amd.default_params(init_shrink=0.1),
                                 make_hydro=lambda m=mode: HipHydro(0, numerics=m), **kw)
        a.initData("sedov", r_init=0.1, nsub=4)
        dts = [a.step() for _ in range(8)]
        torch.cuda.synchronize()
        runs[mode] = (dts, [[b.S_new().cpu().numpy() for b in lev.boxes] for lev in a.levels], a.levels[0].boxes[0].hydro.numerics)
    assert runs["exact"][2] == "exact" and runs["contract"][2] == "contract"
    worst = 0.0
    for le, lc in zip(runs["exact"][1], runs["contract"][1]):
        for be, bc in zip(le, lc):
            mom = max(np.abs(be[k]).max() for k in (1, 2, 3))
            for k in range(8):
                scale = mom if k in (1, 2, 3) else np.abs(be[k]).max()
                worst = max(worst, np.abs(bc[k] - be[k]).max() / scale)
    ddt = max(abs(x - y) / y for x, y in zip(runs["contract"][0], runs["exact"][0]))
    print("contract vs exact, AMR 16^3 + 2 levels, 8 coarse steps: max deviation %.2e; dt deviation %.1e" % (worst, ddt))
    assert 0.0 < worst <= RTOL and ddt <= RTOL


def _outputs_deviation(out):
    """max deviation of every output array of one construct_ctu_hydro_source call from the oracle's, each component scaled by
    the largest magnitude of its kind in the oracle's array (momenta and momentum fluxes share one scale: a component that
    vanishes by symmetry has no scale of its own)"""
    dev = {}
    for name, (a, b) in out.items():
        if name.startswith("qe"):
            groups = [(0, 1, 2), (3,)]                              # Godunov velocities, Godunov pressure
        elif a.shape[0] == 8:
            groups = [(0,), (1, 2, 3), (4,), (5,), (6,), (7,)]      # rho, momenta, rho E, rho e, Temp, rho X
        else:
            groups = [tuple(range(a.shape[0]))]
        worst = 0.0
        for g in groups:
            scale = max(np.abs(b[k]).max() for k in g)
            d = max(np.abs(a[k] - b[k]).max() for k in g)
            worst = max(worst, d / scale if scale > 0.0 else d)
        dev[name] = worst
    return dev


@pytest.mark.parametrize("tiled", [False, True])
def test_contract_single_call_every_output_array_within_rtol(oracle, tiled):
    """Not only the plotfile: S_new, the three flux arrays, the mass fluxes and the Godunov states (qe) of ONE
    castro_amd_ctu_hydro_fab call of the `contract` build on a noisy state, whole box and the overlapped path's seven tiles,
    against the oracle at the tolerance; and the tiled call gives the bits of the whole-box call (same build)."""
    import torch
    from castro_amd.hydro import HipHydro
    from tests.test_gpu_parity import _run_both
    from tests.util import physical_state
    hip = HipHydro(0, numerics="contract")
    assert hip.numerics == "contract"
    rng = np.random.default_rng(11)
    bxlo, bxhi = (0, 0, 0), (19, 17, 15)
    sb_lo, sb_hi = (-4, -4, -4), (23, 21, 19)
    U = physical_state(rng, sb_lo, sb_hi, smooth=False, vel=2.0)
    tiles = [((4, 4, 4), (15, 13, 11)),
             ((0, 0, 0), (19, 17, 3)), ((0, 0, 12), (19, 17, 15)),
             ((0, 0, 4), (19, 3, 11)), ((0, 14, 4), (19, 17, 11)),
             ((0, 4, 4), (3, 13, 11)), ((16, 4, 4), (19, 13, 11))]
    kw = dict(dx=(0.02, 0.02, 0.02))
    whole = _run_both(hip, oracle, bxlo, bxhi, U, sb_lo, sb_hi, 8.0e-4, **kw)
    out = _run_both(hip, oracle, bxlo, bxhi, U, sb_lo, sb_hi, 8.0e-4, tile=(1024, 8, 8), hip_tiles=tiles, **kw) if tiled else whole
    dev = _outputs_deviation(out)
    worst = max(dev, key=dev.get)
    print("contract vs oracle, one call (%s): max deviation %.2e (%s)" % ("7 tiles" if tiled else "whole box", dev[worst], worst))
    assert all(v <= RTOL for v in dev.values()), dev
    assert any(not np.array_equal(a, b) for a, b in out.values())      # it IS the other build
    if tiled:
        for k in out:
            assert np.array_equal(out[k][0], whole[k][0]), k
    hip.close()


@pytest.mark.parametrize("case", ["rotation_rst4", "rotation_rst1", "predictor", "rotation_rst4_plm", "predictor_plm"])
def test_contract_traced_sources_on_the_five_variable_path_within_rtol(oracle, case):
    """Round 5: with traced source terms the `contract` build keeps its gamma-law identities (k_trace<SRC, PLM, GL>: the p source of
    a gamma-law gas is (gamma - 1) times its (rho e) source, so the traced (rho e) is still the traced p over (gamma - 1)).  The
    gravity case of test_contract_non_default_options_within_rtol has no (rho e) source at all; these do: rotation (two energy forms,
    implicit Coriolis update) with gravity on a perturbed state, and castro.source_term_predictor = 1 on a stratified atmosphere with
    random velocities -- conserved state and dt against the oracle at the tolerance."""
    import torch
    import castro_amd
    # _plm (round 6): the same with the PLM trace, which keeps the identities too (trace_plm_dir<D, SRC, GL>) -- with Symmetry and wall
    # faces, whose reflecting fix-up writes both edge states of a face from the zone inside
    plm = dict(ppm_type=0, use_pslope=0) if case.endswith("_plm") else {}       # use_pslope = 1 with a source keeps the 7-variable kernels
    case = case[:-4] if case.endswith("_plm") else case
    if case == "predictor":
        from tests.test_driver_cpu import _hse_atmosphere
        n = (8, 8, 32)
        bc = dict(lo_bc=(4, 2, 3), hi_bc=(4, 2, 3))
        prob_hi = (0.25, 0.25, 1.0)
        S0 = _hse_atmosphere(n)
        rng = np.random.default_rng(8)
        for d in (1, 2, 3):
            S0[d] = S0[0] * 0.1 * rng.uniform(-1, 1, size=S0[0].shape)
        S0[4] += 0.5 * (S0[1] ** 2 + S0[2] ** 2 + S0[3] ** 2) / S0[0]
        pkw = dict(source_term_predictor=1, init_shrink=1.0, change_max=1.05, **plm)
        c = castro_amd.Castro(n, params=castro_amd.default_params(**pkw), do_grav=True, const_grav=-20.0, prob_hi=prob_hi,
                              numerics="contract", **bc)
        lev = oracle.Level(n, oracle.make_geom(n, probhi=prob_hi, **bc), oracle.default_params(**pkw), nthreads=0)
        lev.set_gravity(-20.0, 4)
        nsteps, stop = 10, 1.0
    else:
        rst = 4 if case == "rotation_rst4" else 1
        n = (12, 10, 8)
        pkw = dict(cfl=0.5, init_shrink=1.0, change_max=1.1, **plm)
        rng = np.random.default_rng(8)
        S0 = np.zeros((8,) + n[::-1])
        S0[0] = 1.0 + 0.1 * rng.uniform(-1, 1, size=S0[0].shape)
        S0[7] = S0[0]
        S0[6] = 1.0
        for d in (1, 2, 3):
            S0[d] = S0[0] * 0.05 * rng.uniform(-1, 1, size=S0[0].shape)
        S0[5] = 2.5
        S0[4] = S0[5] + 0.5 * (S0[1] ** 2 + S0[2] ** 2 + S0[3] ** 2) / S0[0]
        bc = dict(lo_bc=(2, 4, 3), hi_bc=(2, 4, 2))
        rkw = dict(center=(0.4, 0.5, 0.6), rot_source_type=rst, implicit_rotation_update=1)
        c = castro_amd.Castro(n, params=castro_amd.default_params(**pkw), rotation=castro_amd.make_rotation(5.0, 2, **rkw),
                              do_grav=True, const_grav=-0.5, numerics="contract", **bc)
        lev = oracle.Level(n, oracle.make_geom(n, **bc), oracle.default_params(**pkw), nthreads=0)
        lev.set_rotation(oracle.make_rotation(5.0, 2, **rkw))
        lev.set_gravity(-0.5)
        nsteps, stop = 6, 2.0
    assert c.hydro.numerics == "contract"
    c.set_state(S0.copy())
    lev.state()[...] = S0
    oracle.lib().ora_level_post_init(lev.h)
    for _ in range(nsteps):
        c.step(stop)
        lev.step(stop)
    torch.cuda.synchronize()
    got, want = c.S_new().cpu().numpy(), lev.state()
    dev = {k: np.abs(got[k] - want[k]).max() / max(np.abs(want[k]).max(), 1e-300) for k in range(8)}
    mom = max(np.abs(want[k]).max() for k in (1, 2, 3))
    for k in (1, 2, 3):
        dev[k] = np.abs(got[k] - want[k]).max() / mom
    print("contract vs oracle, %-14s %2d steps: max deviation %.2e; dt deviation %.1e" % (case, nsteps, max(dev.values()), abs(c.dt - lev.dt) / lev.dt))
    assert all(v <= RTOL for v in dev.values()), dev
    assert abs(c.dt - lev.dt) <= RTOL * lev.dt
    assert not np.array_equal(got, want)               # it IS the other build
    lev.close()


@pytest.mark.parametrize("shape", [(37, 23, 19), (16, 12, 10), (130, 9, 11), (12, 8, 64)])
@pytest.mark.parametrize("form", ["fold_tile_4x2", "fold_tile_2x4", "fold_and_final_tile", "final_tile_only"])
def test_contract_tile_kernels_on_ragged_boxes(oracle, shape, form):
    """The tile forms of round 5 -- k_trans1_tile (a 4 x 2 / 2 x 4 tile of rows per workgroup, first-stage records through LDS;
    the default of large boxes) and k_final_tile (the final stage in one zone-centred launch; a variant) -- on boxes whose row
    counts are no multiples of the tile (partial tiles, waves that straddle two tiles, rows shorter than a wave): every output
    array of one call against the oracle at the tolerance, and against the row-form kernels of the same build (k_trans1_fold_lds,
    k_final<y,z> + k_finalx_consup) at a rounding level."""
    from castro_amd.hydro import HipHydro
    from tests.test_gpu_parity import _run_both
    from tests.util import physical_state
    env = {"fold_tile_4x2": ("1", "0"), "fold_tile_2x4": ("2", "0"), "fold_and_final_tile": ("1", "1"), "final_tile_only": ("0", "1")}[form]
    rng = np.random.default_rng(5)
    bxlo = (2, -3, 1)
    bxhi = tuple(bxlo[d] + shape[d] - 1 for d in range(3))
    sb_lo, sb_hi = tuple(x - 4 for x in bxlo), tuple(x + 4 for x in bxhi)
    U = physical_state(rng, sb_lo, sb_hi, smooth=False, vel=1.5)
    kw = dict(dx=(0.02, 0.017, 0.023))
    outs = {}
    try:
        for name, (ft, fin) in (("rows", ("0", "0")), ("tile", env)):
            os.environ["CASTRO_AMD_FOLD_TILE"], os.environ["CASTRO_AMD_FINAL_TILE"] = ft, fin
            hip = HipHydro(0, numerics="contract")          # the knobs are read when a context is created
            outs[name] = _run_both(hip, oracle, bxlo, bxhi, U, sb_lo, sb_hi, 6.0e-4, **kw)
            hip.close()
    finally:
        os.environ["CASTRO_AMD_FOLD_TILE"], os.environ["CASTRO_AMD_FINAL_TILE"] = "-1", "0"
        HipHydro(0, numerics="contract").close()           # back to the defaults for the tests that follow
        del os.environ["CASTRO_AMD_FOLD_TILE"], os.environ["CASTRO_AMD_FINAL_TILE"]
    dev = _outputs_deviation(outs["tile"])
    worst = max(dev, key=dev.get)
    print("contract %s vs oracle, box %s: max deviation %.2e (%s)" % (form, shape, dev[worst], worst))
    assert all(v <= RTOL for v in dev.values()), dev
    same = _outputs_deviation({k: (outs["tile"][k][0], outs["rows"][k][0]) for k in outs["tile"]})
    assert all(v <= 1e-13 for v in same.values()), same


@pytest.mark.parametrize("use_retry", [False, True])
def test_contract_staged_overlap_and_step_graph_equal_the_plain_contract_run(use_retry):
    """Inside one build the launch partition must not matter: the staged halo overlap (stage A on the valid zones while the
    exchange is in flight, stage B for the rest) and the host-free, graph-replayed batch give the bits of the stepwise,
    un-staged `contract` run."""
    import torch
    import castro_amd
    n = (48, 40, 32)
    # use_retry: the device-side step control has to form the single subcycle (time + dt) - time like the host does -- the
    # `contract` build's reassociation folded it to dt until round 6 (k_step_control)
    kw = dict(lo_bc=(0, 0, 0), hi_bc=(0, 0, 0), use_retry=use_retry, numerics="contract")
    plain = castro_amd.Castro(n, overlap=False, **kw)
    staged = castro_amd.Castro(n, overlap="staged", **kw)
    batch = castro_amd.Castro(n, overlap="staged", **kw)
    light = castro_amd.Castro(n, overlap=True, **kw)            # round 6: the light split (valid zones / rest)
    light_batch = castro_amd.Castro(n, overlap=True, **kw)
    for c in (plain, staged, batch, light, light_batch):
        c.initData("sedov", r_init=0.1, nsub=4)
        assert c.hydro.numerics == "contract"
    assert staged.overlap and staged._comm_stream is not None and staged.neighbors and not plain.overlap
    assert light._light_overlap() and light_batch.host_free_ok() and (batch.host_free_ok() or use_retry)
    for _ in range(7):
        plain.step()
        staged.step()
        light.step()
    batch.run_steps(7)
    light_batch.run_steps(7)
    torch.cuda.synchronize()
    for c in (staged, batch, light, light_batch):
        assert c.time == plain.time and c.dt == plain.dt and c.nstep == plain.nstep
        assert torch.equal(c.S_new_b, plain.S_new_b)


def test_contract_randomised_campaign_against_the_oracle_driver():
    """tools/fuzz_contract.py (round 6): 60 random single-level runs of the `contract` build -- random grids, outflow / Symmetry / wall
    boundaries, PPM / PLM with their slope options, CGF / HLLC, hybrid, flattening on or off, constant gravity and rotation, Sedov or
    Sod, 4 to 12 steps -- against the oracle's level driver: every conserved component within rtol 1e-10, or within the conditioning of
    the run (the oracle a second time from a state one ulp away: coarse grids sit on the ties of the scheme's discrete switches).  The
    campaign that found -fassociative-math producing densities of -1e31 and HLLC picking its star state on wall faces by a rounding."""
    import os
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(__file__), "..")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_contract.py"), "60", "11"], cwd=root, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "cases 60, mismatches 0," in r.stdout, r.stdout[-3000:]
