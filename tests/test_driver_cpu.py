"""CPU tests of the host logic in castro_amd.Castro (decomposition, FillPatch halo exchange over
torch.distributed, clean_state ordering, dt control, interior/shell tiling) with the oracle as the
per-FAB backend (tests/oracle_backend.py).  The N>1 path runs with gloo, world_size 2 and 4."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests.oracle_backend import OracleBackend


def _make(n, oracle, comm=None, grid=None, lo_bc=(2, 2, 2), hi_bc=(2, 2, 2), **pkw):
    import castro_amd
    return castro_amd.Castro(n, lo_bc=lo_bc, hi_bc=hi_bc, params=oracle.default_params(**pkw),
                             hydro=OracleBackend(), comm=comm, grid=grid)


def test_driver_reproduces_reference_ordering(oracle):
    """castro_amd.Castro cleans the valid zones twice and THEN fills ghosts; the reference cleans S_old,
    fills Sborder, then cleans Sborder including ghosts (Castro_advance.cpp:311,186).  Bitwise equal."""
    n = (16, 16, 16)
    c = _make(n, oracle)
    c.initData("sedov", r_init=0.1, nsub=4)
    lev = oracle.Level(n, oracle.make_geom(n), oracle.default_params(), nthreads=4)
    lev.init_sedov(r_init=0.1, nsub=4)
    assert np.array_equal(c.S_new().numpy(), lev.state())
    for _ in range(6):
        c.step(0.01)
        lev.step(0.01)
        assert c.dt == lev.dt
        assert np.array_equal(c.S_new().numpy(), lev.state())
    for d in range(3):
        assert np.array_equal(c.fluxes[d].numpy(), lev.flux(d))
    lev.close()


def test_reflecting_walls_ordering(oracle):
    n = (8, 24, 8)
    bc = dict(lo_bc=(4, 2, 3), hi_bc=(5, 2, 4))
    c = _make(n, oracle, cfl=0.9, init_shrink=0.1, change_max=1.05, **bc)
    kw = dict(rho_l=1.0, u_l=0.3, p_l=1.0, rho_r=0.125, u_r=-0.1, p_r=0.1, idir=2)
    c.initData("sod", **kw)
    lev = oracle.Level(n, oracle.make_geom(n, **bc), oracle.default_params(cfl=0.9, init_shrink=0.1, change_max=1.05), nthreads=2)
    lev.init_sod(1.0, 0.3, 1.0, 0.125, -0.1, 0.1, idir=2)
    for _ in range(8):
        c.step()
        lev.step()
        assert c.dt == lev.dt
    assert np.array_equal(c.S_new().numpy(), lev.state())
    lev.close()


def test_interior_plus_shell_tiles_equal_whole_box(oracle):
    n = (20, 18, 16)
    a = _make(n, oracle)
    b = _make(n, oracle)
    for c in (a, b):
        c.initData("sedov", r_init=0.15, nsub=3)
        c.step()
    interior, shells = b._shell_tiles()
    assert interior == ((4, 4, 4), (15, 13, 11)) and len(shells) == 6
    ncells = sum(np.prod([t[1][d] - t[0][d] + 1 for d in range(3)]) for t in [interior] + shells)
    assert ncells == 20 * 18 * 16
    # one more advance by hand: whole box vs tiles
    for c, tiles in ((a, None), (b, [interior] + shells)):
        c.S_old_b, c.S_new_b = c.S_new_b, c.S_old_b
        c.clean_state(c.S_old_b, 2)
        for d in range(3):
            c.fluxes[d].zero_()
        c.expand_state(c.S_old_b)
        c.construct_ctu_hydro_source(0.0, 1e-5, tiles=tiles)
    assert np.array_equal(a.S_new().numpy(), b.S_new().numpy())
    for d in range(3):
        assert np.array_equal(a.fluxes[d].numpy(), b.fluxes[d].numpy())
        assert np.array_equal(a.mass_fluxes[d].numpy(), b.mass_fluxes[d].numpy())


def test_default_grid():
    import castro_amd
    assert castro_amd.default_grid(1) == (1, 1, 1)
    assert castro_amd.default_grid(2) == (1, 1, 2)
    assert castro_amd.default_grid(4) == (1, 2, 2)
    assert castro_amd.default_grid(8) == (2, 2, 2)


# ---------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, bcs, nsteps, out_path):
    import torch.distributed as dist
    import castro_amd
    from oracle import oracle_lib as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c = castro_amd.Castro(n, lo_bc=bcs[0], hi_bc=bcs[1], params=O.default_params(), hydro=OracleBackend(),
                              comm=castro_amd.DistComm())
        c.initData("sedov", r_init=0.12, nsub=3)
        dts = []
        for _ in range(nsteps):
            dts.append(c.step())
        # gather the valid regions on rank 0
        mine = c.S_new().contiguous()
        parts = [torch.zeros_like(mine) for _ in range(world)] if rank == 0 else None
        dist.gather(mine, parts, dst=0)
        boxes = [None] * world
        dist.all_gather_object(boxes, (c.lo, c.hi))
        if rank == 0:
            full = np.zeros((8, n[2], n[1], n[0]))
            for p, (lo, hi) in zip(parts, boxes):
                full[:, lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = p.numpy()
            np.savez(out_path, S=full, dts=np.array(dts))
    finally:
        dist.destroy_process_group()


def _single(n, bcs, nsteps, oracle):
    c = _make(n, oracle, lo_bc=bcs[0], hi_bc=bcs[1])
    c.initData("sedov", r_init=0.12, nsub=3)
    dts = [c.step() for _ in range(nsteps)]
    return c.S_new().numpy().copy(), np.array(dts)


@pytest.mark.parametrize("world,n,bcs", [
    (2, (16, 16, 16), ((2, 2, 2), (2, 2, 2))),          # outflow, split in z
    (2, (16, 8, 16), ((0, 4, 0), (0, 4, 0))),           # periodic x,z (self-wrap in x, peer-wrap in z) + walls in y
    (4, (8, 16, 16), ((2, 3, 2), (2, 2, 5))),           # 1x2x2 grid: faces + an edge neighbour
    (8, (16, 16, 16), ((2, 2, 2), (2, 2, 2))),          # 2x2x2 grid (the 8-GPU layout): faces, edges and the corner
])
def test_decomposed_run_is_bitwise_identical_gloo(tmp_path, oracle, world, n, bcs):
    """Decomposition independence: ghost data are exact copies, so N ranks give the
    same bits as one rank."""
    nsteps = 4
    out = str(tmp_path / "dist.npz")
    mp.spawn(_worker, args=(world, _free_port(), n, bcs, nsteps, out), nprocs=world, join=True)
    got = np.load(out)
    want_S, want_dts = _single(n, bcs, nsteps, oracle)
    assert np.array_equal(got["dts"], want_dts)
    assert np.array_equal(got["S"], want_S)


def test_retry_and_subcycling_match_the_oracle_driver(oracle):
    """castro.use_retry (Castro_advance_ctu.cpp:403-768): a step that fails the timestep-validity check is redone
    as two half steps; state, old state, accumulated fluxes and step sizes equal the oracle's C restatement."""
    import castro_amd
    n = (16, 16, 16)
    kw = dict(cfl=0.9, init_shrink=1.0, change_max=1.02)
    c = _make(n, oracle, **kw)
    c.initData("sedov", r_init=0.1, nsub=4)
    lev = oracle.Level(n, oracle.make_geom(n), oracle.default_params(**kw), nthreads=4)
    lev.init_sedov(r_init=0.1, nsub=4)
    seen_retry = False
    for _ in range(4):
        c.step(0.05)
        lev.step(0.05)
        assert (c.dt, c.nsubcycles, c.nretries) == (lev.dt, lev.nsubcycles, lev.nretries)
        seen_retry |= c.nretries > 0
        assert np.array_equal(c.S_new().numpy(), lev.state())
        g = 4
        assert np.array_equal(c.S_old_b[:, g:-g, g:-g, g:-g].numpy(), lev.old_state())
        for d in range(3):
            assert np.array_equal(c.fluxes[d].numpy(), lev.flux(d))
    assert seen_retry and "timestep validity" in c.last_failure
    lev.close()

    # without retries the same step aborts (amrex::Abort("Advance was unsuccessful.") in the reference)
    c = castro_amd.Castro(n, params=oracle.default_params(**kw), hydro=OracleBackend(), use_retry=False)
    c.initData("sedov", r_init=0.1, nsub=4)
    with pytest.raises(castro_amd.AdvanceFailure, match="timestep validity"):
        c.step(0.05)

    # small-density failures halve the step until max_subcycles is exceeded
    c = _make(n, oracle, small_dens=0.9, init_shrink=1.0)
    c.initData("sedov", r_init=0.1, nsub=4)
    with pytest.raises(castro_amd.AdvanceFailure, match="too many subcycles"):
        for _ in range(12):
            c.step(0.05)
    assert "density" in c.last_failure


def _hse_atmosphere(n, H=0.25, g=-1.0, gamma=1.4):
    """isothermal atmosphere in hydrostatic equilibrium along z: rho = exp(-z/H), p = rho * (-g H)"""
    z = (np.arange(n[2]) + 0.5) / n[2]
    rho = np.exp(-z / H)[:, None, None] * np.ones((n[2], n[1], n[0]))
    S = np.zeros((8,) + rho.shape)
    S[0] = rho
    S[4] = S[5] = rho * (-g * H) / (gamma - 1.0)
    S[6] = 1.0
    S[7] = rho
    return S


@pytest.mark.parametrize("pkw,gtype", [(dict(ppm_type=1), 4), (dict(ppm_type=0), 4), (dict(ppm_type=1), 2), (dict(ppm_type=1), 3)])
def test_constant_gravity_sources_match_the_oracle_driver(oracle, pkw, gtype):
    """castro.do_grav with ConstantGrav: old source -> traced in the predictor -> hydro -> new-time corrector
    (Castro_advance_ctu.cpp:113-143,256-274; Castro_gravity.cpp:234-614): driver == oracle level driver, bitwise;
    the atmosphere stays close to rest, and PLM with the well-balanced pressure slope does so 50x better."""
    import castro_amd
    n = (4, 4, 32)
    bc = dict(lo_bc=(4, 4, 3), hi_bc=(4, 4, 3))
    geo = dict(prob_hi=(0.125, 0.125, 1.0))
    c = castro_amd.Castro(n, params=oracle.default_params(**pkw), hydro=OracleBackend(), do_grav=True, const_grav=-1.0,
                          grav_source_type=gtype, **bc, **geo)
    c.set_state(_hse_atmosphere(n))
    lev = oracle.Level(n, oracle.make_geom(n, probhi=geo["prob_hi"], **bc), oracle.default_params(**pkw), nthreads=4)
    lev.set_gravity(-1.0, gtype)
    lev.state()[...] = _hse_atmosphere(n)
    oracle.lib().ora_level_post_init(lev.h)
    for _ in range(12):
        c.step(1.0)
        lev.step(1.0)
        assert c.dt == lev.dt
    assert np.array_equal(c.S_new().numpy(), lev.state())
    S = lev.state()
    mach = np.abs(S[3] / S[0]).max() / np.sqrt(1.4 * 0.25)
    assert mach < (3e-4 if pkw["ppm_type"] == 0 else 3e-2)
    assert np.abs(S[1]).max() == 0.0 and np.abs(S[2]).max() == 0.0
    lev.close()


@pytest.mark.parametrize("init_shrink", [0.1, 1.0])
def test_source_term_predictor_matches_the_oracle_driver_and_changes_the_answer(oracle, init_shrink):
    """castro.source_term_predictor = 1 (Castro_ctu.cpp:493-497, Castro.cpp:3780-3818): the momentum sources traced in the
    predictor get dt/2 x the lagged dS/dt = 2 x (new-time corrector of the last advance) / lastDt.  Driver == oracle level
    driver, bitwise, over steps that include retries (init_shrink = 1); and the result differs from the run without it."""
    import castro_amd
    n = (4, 4, 32)
    bc = dict(lo_bc=(4, 4, 3), hi_bc=(4, 4, 3))
    geo = dict(prob_hi=(0.125, 0.125, 1.0))
    S0 = _hse_atmosphere(n)
    rng = np.random.default_rng(5)
    S0[3] = S0[0] * 0.2 * rng.uniform(-1, 1, size=S0[0].shape)
    S0[4] += 0.5 * S0[3] ** 2 / S0[0]
    out = {}
    for pred in (1, 0):
        pkw = dict(source_term_predictor=pred, init_shrink=init_shrink, change_max=1.05)
        c = castro_amd.Castro(n, params=oracle.default_params(**pkw), hydro=OracleBackend(), do_grav=True, const_grav=-20.0, **bc, **geo)
        c.set_state(S0.copy())
        lev = oracle.Level(n, oracle.make_geom(n, probhi=geo["prob_hi"], **bc), oracle.default_params(**pkw), nthreads=4)
        lev.set_gravity(-20.0, 4)
        lev.state()[...] = S0
        oracle.lib().ora_level_post_init(lev.h)
        retries = 0
        for _ in range(10):
            c.step(1.0)
            lev.step(1.0)
            assert c.dt == lev.dt and c.nretries == lev.nretries
            retries += c.nretries
        assert np.array_equal(c.S_new().numpy(), lev.state())
        out[pred] = (lev.state().copy(), retries)
        lev.close()
    assert not np.array_equal(out[1][0], out[0][0])
    if init_shrink == 1.0:
        assert out[1][1] > 0                      # the retry path (no new corrector on the attempt after a rejection) ran


def _grav_worker(rank, world, port, n, nsteps, out_path):
    import torch.distributed as dist
    import castro_amd
    from oracle import oracle_lib as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c = castro_amd.Castro(n, params=O.default_params(), hydro=OracleBackend(), comm=castro_amd.DistComm(),
                              do_grav=True, const_grav=-1.0, lo_bc=(0, 4, 3), hi_bc=(0, 4, 3), prob_hi=(0.25, 0.25, 1.0))
        c.set_state(_hse_atmosphere(n))
        for _ in range(nsteps):
            c.step(1.0)
        mine = c.S_new().contiguous()
        parts = [torch.zeros_like(mine) for _ in range(world)] if rank == 0 else None
        dist.gather(mine, parts, dst=0)
        boxes = c.comm.gather_objects((c.lo, c.hi))
        if rank == 0:
            full = np.zeros((8, n[2], n[1], n[0]))
            for p, (lo, hi) in zip(parts, boxes):
                full[:, lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = p.numpy()
            np.save(out_path, full)
    finally:
        dist.destroy_process_group()


def test_gravity_run_on_two_ranks_is_bitwise_identical_gloo(tmp_path, oracle):
    """The Source_Type FillPatch (3 ghost layers, 7 components) goes through the same halo exchange as the state:
    two ranks split along z (the direction of gravity), periodic in x, give the single-rank bits."""
    import castro_amd
    n, nsteps = (8, 8, 32), 5
    out = str(tmp_path / "grav.npy")
    mp.spawn(_grav_worker, args=(2, _free_port(), n, nsteps, out), nprocs=2, join=True)
    c = castro_amd.Castro(n, params=oracle.default_params(), hydro=OracleBackend(), do_grav=True, const_grav=-1.0,
                          lo_bc=(0, 4, 3), hi_bc=(0, 4, 3), prob_hi=(0.25, 0.25, 1.0))
    c.set_state(_hse_atmosphere(n))
    for _ in range(nsteps):
        c.step(1.0)
    assert np.array_equal(np.load(out), c.S_new().numpy())


@pytest.mark.parametrize("rst,implicit", [(4, 1), (1, 0), (2, 1), (3, 1)])
def test_rotation_sources_inertial_oscillation_and_oracle_driver(oracle, rst, implicit):
    """castro.do_rotation (Source/rotation): a uniform gas moving in the rotating frame, Coriolis force only: the
    velocity vector turns with angular frequency 2 Omega (known answer), Coriolis does no work, and the driver equals
    the oracle level driver bit for bit; with the centrifugal term and a non-uniform state the two drivers still agree."""
    import math
    import castro_amd
    n = (8, 8, 8)
    kw = dict(cfl=0.5, init_shrink=1.0, change_max=1.1)
    S0 = np.zeros((8,) + n[::-1])
    S0[0] = 1.0; S0[1] = 0.1; S0[5] = 2.5; S0[4] = 2.5 + 0.5 * 0.01; S0[6] = 1.0; S0[7] = 1.0
    T = 20.0
    rot = castro_amd.make_rotation(T, 3, include_centrifugal=0, rot_source_type=rst, implicit_rotation_update=implicit)
    c = castro_amd.Castro(n, params=oracle.default_params(**kw), hydro=OracleBackend(), rotation=rot)
    c.set_state(S0)
    lev = oracle.Level(n, oracle.make_geom(n), oracle.default_params(**kw), nthreads=2)
    lev.set_rotation(oracle.make_rotation(T, 3, include_centrifugal=0, rot_source_type=rst, implicit_rotation_update=implicit))
    lev.state()[...] = S0
    oracle.lib().ora_level_post_init(lev.h)
    for _ in range(15):
        c.step(2.0)
        lev.step(2.0)
        assert c.dt == lev.dt
    S = c.S_new().numpy()
    assert np.array_equal(S, lev.state())
    u, v = S[1][4, 4, 4], S[2][4, 4, 4]
    w = 2.0 * (2.0 * math.pi / T) * c.time
    assert abs(u - 0.1 * math.cos(w)) < 2e-5 and abs(v + 0.1 * math.sin(w)) < 2e-5
    assert abs(S[4][4, 4, 4] - S[5][4, 4, 4] - 0.5 * (u * u + v * v)) < 1e-15
    lev.close()
    # centrifugal + Coriolis on a non-uniform state: driver == oracle driver
    rng = np.random.default_rng(5)
    S1 = S0.copy()
    S1[0] *= 1.0 + 0.1 * rng.uniform(-1, 1, size=S1[0].shape)
    S1[7] = S1[0]
    for d in (1, 2, 3):
        S1[d] = S1[0] * 0.05 * rng.uniform(-1, 1, size=S1[0].shape)
    S1[4] = S1[5] + 0.5 * (S1[1] ** 2 + S1[2] ** 2 + S1[3] ** 2) / S1[0]
    rot = castro_amd.make_rotation(5.0, 2, center=(0.4, 0.5, 0.6), rot_source_type=rst, implicit_rotation_update=implicit)
    c = castro_amd.Castro(n, params=oracle.default_params(**kw), hydro=OracleBackend(), rotation=rot, do_grav=True, const_grav=-0.5)
    c.set_state(S1)
    lev = oracle.Level(n, oracle.make_geom(n), oracle.default_params(**kw), nthreads=2)
    lev.set_rotation(oracle.make_rotation(5.0, 2, center=(0.4, 0.5, 0.6), rot_source_type=rst, implicit_rotation_update=implicit))
    lev.set_gravity(-0.5)
    lev.state()[...] = S1
    oracle.lib().ora_level_post_init(lev.h)
    for _ in range(5):
        c.step(2.0)
        lev.step(2.0)
    assert np.array_equal(c.S_new().numpy(), lev.state())
    lev.close()


def _acoustic_pulse(n, gamma=1.4, rho0=1.4, drho0=0.14):
    """Exec/hydro_tests/acoustic_pulse/problem_initialize_state_data.H:29-75 (McCorquodale & Colella 2011)."""
    x = (np.arange(n) + 0.5) / n
    zz, yy, xx = np.meshgrid(x, x, x, indexing="ij")
    dist = np.sqrt((0.5 - xx) ** 2 + (0.5 - yy) ** 2 + (0.5 - zz) ** 2)
    rho = np.where(dist <= 0.5, rho0 + drho0 * np.exp(-16.0 * dist * dist) * np.cos(np.pi * dist) ** 6, rho0)
    p = (rho / rho0) ** gamma
    S = np.zeros((8, n, n, n))
    S[0] = rho; S[4] = p / (gamma - 1.0); S[5] = S[4]; S[7] = rho
    return S


@pytest.mark.parametrize("fixed_dt,steps", [(3.0e-3, 81), (1.5e-3, 161)])
def test_fixed_dt_step_counts_of_the_acoustic_pulse_scripts(oracle, fixed_dt, steps):
    """Exec/hydro_tests/acoustic_pulse/convergence_ppm.sh compares plt00081 (inputs.64, fixed_dt = 3e-3), plt00161
    (inputs.128, 1.5e-3) and plt00321 (inputs.256, 7.5e-4) at stop_time = 0.24: the first step is init_shrink x fixed_dt
    (Castro::initialTimeStep), every later one fixed_dt without the change_max limit (Castro.cpp:1655), the last one
    is clipped to stop_time.  The counts do not depend on the resolution, so they are checked on 8^3 zones."""
    import castro_amd
    c = castro_amd.Castro((8, 8, 8), lo_bc=(0, 0, 0), hi_bc=(0, 0, 0), params=oracle.default_params(init_shrink=0.01),
                          hydro=OracleBackend(), fixed_dt=fixed_dt)
    c.set_state(_acoustic_pulse(8))
    dts = []
    while c.time < 0.24 - 2.220446049250313e-16:
        dts.append(c.step(0.24))
    assert c.nstep == steps and c.time == 0.24
    assert dts[0] == 0.01 * fixed_dt and all(d == fixed_dt for d in dts[1:-1]) and 0.0 < dts[-1] <= fixed_dt
    assert c.nretries == 0


def test_max_dt_caps_the_step(oracle):
    """castro.max_dt (Castro.cpp:1515-1530): the hydro estimate is capped, also inside the validity check."""
    import castro_amd
    c = castro_amd.Castro((8, 8, 8), lo_bc=(0, 0, 0), hi_bc=(0, 0, 0), params=oracle.default_params(init_shrink=1.0),
                          hydro=OracleBackend(), max_dt=1.0e-3)
    c.set_state(_acoustic_pulse(8))
    dts = [c.step() for _ in range(3)]
    assert dts == [1.0e-3] * 3 and c.nretries == 0
    free = castro_amd.Castro((8, 8, 8), lo_bc=(0, 0, 0), hi_bc=(0, 0, 0), params=oracle.default_params(init_shrink=1.0),
                             hydro=OracleBackend())
    free.set_state(_acoustic_pulse(8))
    assert free.step() > 1.0e-2
