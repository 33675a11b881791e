"""Two-level AMR (first slice) with the oracle as the per-FAB backend: interpolation and
flux-register building blocks, conservation over the composite grid, consistency of coarse data under the patch,
preservation of a uniform state, agreement with a uniformly fine run."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.oracle_backend import OracleBackend


def _amr(oracle, n=(16, 16, 16), patch=((4, 4, 4), (11, 11, 11)), **pkw):
    import castro_amd
    return castro_amd.CastroAmr(n, patch, params=oracle.default_params(**pkw), make_hydro=OracleBackend)


def test_interpolation_is_conservative_linear_and_bounded(oracle):
    rng = np.random.default_rng(0)
    clo, chi = (-1, -1, -1), (6, 6, 6)
    lin = np.fromfunction(lambda n, k, j, i: 1.0 + 0.5 * i - 0.25 * j + 2.0 * k + n, (2, 8, 8, 8))
    noisy = rng.uniform(1.0, 2.0, size=(2, 8, 8, 8))
    for crse, is_linear in ((lin, True), (noisy, False)):
        fine = np.zeros((2, 12, 12, 12))
        flo, fhi = (0, 0, 0), (11, 11, 11)
        oracle.lib().ora_cc_interp(oracle.i3(flo), oracle.i3(fhi), oracle.a4(crse, clo, chi), oracle.a4(fine, flo, fhi), 2)
        back = np.zeros((2, 6, 6, 6))
        oracle.lib().ora_avgdown(oracle.i3((0, 0, 0)), oracle.i3((5, 5, 5)), oracle.a4(fine, flo, fhi),
                                 oracle.a4(back, (0, 0, 0), (5, 5, 5)), 2)
        assert np.allclose(back, crse[:, 1:7, 1:7, 1:7], rtol=1e-15, atol=1e-15)             # conservative
        if is_linear:                                                                      # exact for linear data
            # numpy index = coarse index + 1 (the coarse array starts at -1); fine centre in coarse units = (i + .5)/2 - .5
            want = np.fromfunction(lambda n, k, j, i: 1.0 + 0.5 * ((i + 0.5) / 2 + 0.5) - 0.25 * ((j + 0.5) / 2 + 0.5)
                                   + 2.0 * ((k + 0.5) / 2 + 0.5) + n, (2, 12, 12, 12))
            assert np.allclose(fine, want, rtol=1e-14)
        else:                                                                              # no new extrema
            assert fine.min() >= crse.min() - 1e-15 and fine.max() <= crse.max() + 1e-15


def test_uniform_state_is_preserved(oracle):
    a = _amr(oracle)
    for lev in a.levels:
        S = lev.S_new_b
        S.zero_()
        S[0] = 1.0; S[1] = 0.3; S[2] = -0.2; S[3] = 0.1
        S[5] = 2.5; S[4] = 2.5 + 0.5 * (0.09 + 0.04 + 0.01); S[6] = 1.0; S[7] = 1.0
        lev.clean_state(S, 1)
    ref = a.crse.S_new().clone()
    for _ in range(3):
        a.step()
    for lev in a.levels:
        for comp in (0, 1, 2, 3, 4, 5, 7):
            assert torch.allclose(lev.S_new()[comp], ref[comp][0, 0, 0].expand_as(lev.S_new()[comp]), rtol=1e-13, atol=1e-14)


def test_sedov_amr_conserves_and_tracks_the_fine_solution(oracle):
    import castro_amd
    a = _amr(oracle, init_shrink=0.1)
    a.initData("sedov", r_init=0.1, nsub=4)
    m0, e0 = a.composite_sum(0), a.composite_sum(4)
    assert m0 == pytest.approx(1.0, rel=1e-13)
    a.evolve(0.025)           # the shock crosses the coarse-fine boundary (r = 0.25) at t ~ 0.022; nothing has left the domain yet
    # reflux makes the composite update conservative (outflow boundaries are still quiescent)
    assert abs(a.composite_sum(0) - m0) <= 1e-12 * m0
    assert abs(a.composite_sum(4) - e0) <= 1e-12 * e0
    # coarse data under the patch is the average of the fine data (avgDown + clean_state)
    f, c = a.fine.S_new().numpy(), a.crse.S_new().numpy()
    avg = f.reshape(8, 8, 2, 8, 2, 8, 2).mean(axis=(2, 4, 6))
    assert np.allclose(c[0][4:12, 4:12, 4:12], avg[0], rtol=1e-13)
    # a uniformly fine run to the same time: the patch tracks it
    u = castro_amd.Castro((32, 32, 32), params=oracle.default_params(init_shrink=0.1), hydro=OracleBackend())
    u.initData("sedov", r_init=0.1, nsub=4)
    u.evolve(a.time)
    assert u.time == pytest.approx(a.time, rel=1e-12)
    uf = u.S_new().numpy()[0][8:24, 8:24, 8:24]
    err = np.abs(f[0] - uf).mean() / np.abs(uf).mean()
    assert err < 0.05, err
    # the shock has crossed the coarse-fine boundary: coarse zones outside the patch are disturbed
    outside = c[0].copy()
    outside[4:12, 4:12, 4:12] = 1.0
    assert np.abs(outside - 1.0).max() > 0.05 and f[0].min() < 0.3


def test_tag_driven_patch_follows_the_shock_and_conserves(oracle):
    """amr.refinement_indicators restated (gradient / value_greater on density); the refined region is the aligned
    bounding box of the tags, regridded every 2 coarse steps.  Mass and energy of the composite grid survive every
    regrid (interpolation and avgDown are conservative) and the patch grows with the blast."""
    import castro_amd
    a = castro_amd.CastroAmr((24, 24, 24), params=oracle.default_params(init_shrink=0.1), make_hydro=OracleBackend,
                             refine=[("density", "gradient", 0.05), ("density", "value_greater", 2.5)], regrid_int=2,
                             n_error_buf=1, blocking_factor=4)
    a.initData("sedov", r_init=0.1, nsub=4)
    # at t = 0 density is uniform: nothing to tag on density, so refine on the energy jump instead for the start
    assert a.fine is None
    a.refine = a.refine + [("rho_E", "relative_gradient", 0.5)]
    a.initData("sedov", r_init=0.1, nsub=4)
    assert a.fine is not None
    first = (a.plo, a.phi)
    m0, e0 = a.composite_sum(0), a.composite_sum(4)
    sizes = []
    while a.time < 0.012:
        a.step(0.02)
        sizes.append(tuple(a.phi[d] - a.plo[d] + 1 for d in range(3)))
        assert abs(a.composite_sum(0) - m0) <= 1e-12 * m0, (a.nstep, a.composite_sum(0) - m0)
        assert abs(a.composite_sum(4) - e0) <= 1e-12 * e0
    assert a.nregrid >= 1 and sizes[-1][0] > (first[1][0] - first[0][0] + 1), (first, sizes[-1])
    for d in range(3):                                         # aligned to blocking_factor (fine) = 2 coarse zones
        assert a.plo[d] % 2 == 0 and (a.phi[d] + 1) % 2 == 0
    # symmetric problem, symmetric patch
    assert a.plo == tuple(24 - 1 - x for x in a.phi)


def test_three_levels_with_subcycling_conserve_and_track_the_fine_solution(oracle):
    """amr.max_level = 2: the Amr::timeStep recursion (1 coarse step = 2 level-1 steps = 4 level-2 steps), FillPatch of
    level 2 from the time-interpolated level-1 data (whose own ghost zones come from level 0), two flux registers."""
    import castro_amd
    a = castro_amd.CastroAmr((16, 16, 16), patches=[((4, 4, 4), (11, 11, 11)), ((12, 12, 12), (19, 19, 19))],
                             params=oracle.default_params(init_shrink=0.1), make_hydro=OracleBackend)
    a.initData("sedov", r_init=0.08, nsub=4)
    assert [lev.n for lev in a.levels] == [(16, 16, 16), (16, 16, 16), (16, 16, 16)]
    assert a.levels[2].geom.dx[0] == pytest.approx(1.0 / 64)
    m0, e0 = a.composite_sum(0), a.composite_sum(4)
    a.evolve(0.012)          # the shock (r = 0.2 at this time) has crossed the faces of the level-2 patch (half width 0.125)
    assert abs(a.composite_sum(0) - m0) <= 1e-12 * m0
    assert abs(a.composite_sum(4) - e0) <= 1e-12 * e0
    f2, f1, c = (lev.S_new().numpy() for lev in (a.levels[2], a.levels[1], a.levels[0]))
    assert np.allclose(f1[0][4:12, 4:12, 4:12], f2[0].reshape(8, 2, 8, 2, 8, 2).mean(axis=(1, 3, 5)), rtol=1e-13)
    assert np.allclose(c[0][4:12, 4:12, 4:12], f1[0].reshape(8, 2, 8, 2, 8, 2).mean(axis=(1, 3, 5)), rtol=1e-13)
    outside2 = f1[0].copy()
    outside2[4:12, 4:12, 4:12] = 1.0
    assert np.abs(outside2 - 1.0).max() > 0.5 and f2[0].min() < 0.2      # shell on level 1, evacuated centre on level 2
    u = castro_amd.Castro((64, 64, 64), params=oracle.default_params(init_shrink=0.1), hydro=OracleBackend())
    u.initData("sedov", r_init=0.08, nsub=4)
    u.evolve(a.time)
    uf = u.S_new().numpy()[0][24:40, 24:40, 24:40]
    assert np.abs(f2[0] - uf).mean() / np.abs(uf).mean() < 0.08


def test_tag_driven_three_levels_stay_nested_and_conserve(oracle):
    """Amr::grid_places restated for one box per level: tags from the finest level down, every new box covers the
    (coarsened, buffered) box of the level above it; a regrid moves both refined levels with the blast."""
    import castro_amd
    a = castro_amd.CastroAmr((16, 16, 16), params=oracle.default_params(init_shrink=0.1), make_hydro=OracleBackend,
                             refine=[("density", "gradient", 0.05), ("rho_E", "relative_gradient", 0.5)], regrid_int=2,
                             n_error_buf=1, blocking_factor=4, max_level=2)
    a.initData("sedov", r_init=0.08, nsub=4)
    assert len(a.levels) == 3                                   # bldFineLevels: one level per pass
    first = list(a.pbox)
    m0, e0 = a.composite_sum(0), a.composite_sum(4)
    while a.time < 0.004:
        a.step(0.02)
        assert abs(a.composite_sum(0) - m0) <= 1e-12 * m0 and abs(a.composite_sum(4) - e0) <= 1e-12 * e0
        for l in range(2, len(a.levels)):                       # level l inside level l-1, aligned to blocking_factor
            (plo, phi), (qlo, qhi) = a.pbox[l - 1], a.pbox[l]
            for d in range(3):
                assert 2 * plo[d] <= qlo[d] and qhi[d] <= 2 * phi[d] + 1
                assert qlo[d] % 2 == 0 and (qhi[d] + 1) % 2 == 0
    assert a.nregrid >= 1 and a.pbox != first and len(a.levels) == 3
    # coarse data under each patch is the average of the finer data
    for l in (2, 1):
        f, c = a.levels[l].S_new().numpy()[0], a.levels[l - 1]
        (plo, phi) = a.pbox[l]
        nz, ny, nx = f.shape
        avg = f.reshape(nz // 2, 2, ny // 2, 2, nx // 2, 2).mean(axis=(1, 3, 5))
        o = c.lo
        got = c.S_new().numpy()[0][plo[2] - o[2]:phi[2] - o[2] + 1, plo[1] - o[1]:phi[1] - o[1] + 1, plo[0] - o[0]:phi[0] - o[0] + 1]
        assert np.allclose(got, avg, rtol=1e-13)
    # a level disappears when nothing is tagged any more
    a.refine = [("density", "value_greater", 1e9)]
    assert a.regrid() and len(a.levels) == 1


def _assemble(level, lo, n):
    """Valid data of all boxes of a level pasted into one array covering [lo, lo + n)."""
    out = np.full((8, n[2], n[1], n[0]), np.nan)
    for b in level.boxes:
        o = [b.lo[d] - lo[d] for d in range(3)]
        out[:, o[2]:o[2] + b.n[2], o[1]:o[1] + b.n[1], o[0]:o[0] + b.n[0]] = b.S_new().numpy()
    return out


def test_splitting_a_refined_level_into_boxes_changes_nothing(oracle):
    """The same refined region as one box, as two boxes and as eight boxes: same-level ghost copies, one flux register
    per box face (fine-fine faces are refluxed into covered coarse zones and overwritten by avgDown), level-wide dt and
    retry -- bit for bit the same coarse and fine data after the shock has crossed the coarse-fine boundary."""
    import castro_amd
    kw = dict(params=oracle.default_params(init_shrink=0.1), make_hydro=OracleBackend)
    one = castro_amd.CastroAmr((16, 16, 16), patches=[((4, 4, 4), (11, 11, 11))], **kw)
    two = castro_amd.CastroAmr((16, 16, 16), patches=[[((4, 4, 4), (7, 11, 11)), ((8, 4, 4), (11, 11, 11))]], **kw)
    eight = castro_amd.CastroAmr((16, 16, 16), patches=[[((4 + 4 * i, 4 + 4 * j, 4 + 4 * k), (7 + 4 * i, 7 + 4 * j, 7 + 4 * k))
                                                         for k in range(2) for j in range(2) for i in range(2)]], **kw)
    for a in (one, two, eight):
        a.initData("sedov", r_init=0.1, nsub=4)
    assert len(eight.fine.boxes) == 8 and eight.pbox[1] == eight.boxes[1]
    while one.time < 0.025 - 1e-15:
        d = one.step(0.025)
        assert two.step(0.025) == d and eight.step(0.025) == d
    ref_c, ref_f = one.crse.S_new().numpy(), one.fine.S_new().numpy()
    assert np.abs(ref_c[0] - 1.0).max() > 0.05                      # the shock is outside the refined region
    for a in (two, eight):
        assert np.array_equal(a.crse.S_new().numpy(), ref_c)
        assert np.array_equal(_assemble(a.fine, (8, 8, 8), (16, 16, 16)), ref_f)
        assert abs(a.composite_sum(0) - 1.0) <= 1e-12


def test_a_box_over_two_parent_boxes(oracle):
    """Level 2 sits across the two boxes of level 1: its coarse data, flux-register sources, reflux targets and avgDown
    targets come from both parents.  Same result as with level 1 in one piece."""
    import castro_amd
    kw = dict(params=oracle.default_params(init_shrink=0.1), make_hydro=OracleBackend)
    l2 = ((12, 12, 12), (19, 19, 19))
    one = castro_amd.CastroAmr((16, 16, 16), patches=[((4, 4, 4), (11, 11, 11)), l2], **kw)
    two = castro_amd.CastroAmr((16, 16, 16), patches=[[((4, 4, 4), (11, 7, 11)), ((4, 8, 4), (11, 11, 11))], l2], **kw)
    for a in (one, two):
        a.initData("sedov", r_init=0.08, nsub=4)
    assert len(two.levels[2].boxes[0].csrc) == 2 and len(two.levels[2].boxes[0].avg_to) == 2
    for _ in range(10):
        assert one.step(0.05) == two.step(0.05)
    assert np.abs(one.levels[1].S_new().numpy()[0] - 1.0).max() > 0.5   # the shock has left level 2
    assert np.array_equal(two.crse.S_new().numpy(), one.crse.S_new().numpy())
    assert np.array_equal(_assemble(two.levels[1], (8, 8, 8), (16, 16, 16)), one.levels[1].S_new().numpy())
    assert np.array_equal(two.levels[2].S_new().numpy(), one.levels[2].S_new().numpy())


def test_clustered_boxes_follow_the_shock_and_conserve(oracle):
    """cluster=True: Berger-Rigoutsos boxes around the tagged shell instead of its bounding box, regridded every two
    coarse steps; composite mass and energy survive every regrid, the boxes are disjoint, nested and aligned."""
    import castro_amd
    from castro_amd import cluster as CL
    a = castro_amd.CastroAmr((32, 32, 32), params=oracle.default_params(init_shrink=0.1), make_hydro=OracleBackend,
                             refine=[("density", "gradient", 0.1), ("rho_E", "relative_gradient", 0.5)], regrid_int=2,
                             n_error_buf=1, blocking_factor=4, cluster=True, grid_eff=0.7, max_grid_size=16)
    a.initData("sedov", r_init=0.1, nsub=4)
    assert a.fine is not None
    m0, e0 = a.composite_sum(0), a.composite_sum(4)
    nbox = []
    while a.time < 0.012:
        a.step(0.02)
        assert abs(a.composite_sum(0) - m0) <= 1e-12 * m0 and abs(a.composite_sum(4) - e0) <= 1e-12 * e0
        bl = a.boxes[1]
        nbox.append(len(bl))
        cov = np.zeros((32, 32, 32), dtype=np.int32)
        for lo, hi in bl:
            assert all(l % 2 == 0 and (h + 1) % 2 == 0 and h - l + 1 <= 8 for l, h in zip(lo, hi))
            cov[lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] += 1
        assert cov.max() == 1
    assert a.nregrid >= 2 and max(nbox) > 8
    # the evacuated centre is no longer refined once the shell has moved out: fewer zones than the bounding box
    lo = [min(b[0][d] for b in bl) for d in range(3)]
    hi = [max(b[1][d] for b in bl) for d in range(3)]
    assert cov.sum() < 0.9 * np.prod([hi[d] - lo[d] + 1 for d in range(3)])
    # every currently tagged coarse zone is refined
    tags = a._tags(0)[0].numpy() > 0.5
    assert tags.any() and not (tags & (cov == 0)).any()


def test_multilevel_plotfile_round_trip(tmp_path, oracle):
    from castro_amd import plotfile as pf
    a = _amr(oracle, init_shrink=0.1)
    a.initData("sedov", r_init=0.1, nsub=4)
    for _ in range(3):
        a.step()
    d = str(tmp_path / "plt_amr")
    names = pf.write_plotfile_amr(d, a)
    H = open(d + "/Header").read().split("\n")
    p = 2 + len(names)
    assert H[p + 2] == "1" and H[p + 5].strip() == "2"
    assert H[p + 6].strip() == "((0,0,0) (15,15,15) (0,0,0)) ((0,0,0) (31,31,31) (0,0,0))"
    assert H[p + 7].split() == ["3", "6"]
    r = pf.read_plotfile_amr(d)
    assert r["names"] == names and r["time"] == a.time and len(r["levels"]) == 2
    assert r["levels"][1]["box"] == ([8, 8, 8], [23, 23, 23]) and r["levels"][1]["dx"][0] == 1.0 / 32
    for lev, got in zip(a.levels, r["levels"]):
        assert np.array_equal(got["data"][:8], lev.S_new().numpy())
        assert np.array_equal(got["data"][names.index("x_velocity")], got["data"][1] / got["data"][0])


def test_plotfile_of_a_level_with_several_boxes(tmp_path, oracle):
    import castro_amd
    from castro_amd import plotfile as pf
    a = castro_amd.CastroAmr((16, 16, 16), patches=[[((4, 4, 4), (7, 11, 11)), ((8, 4, 4), (11, 11, 11))]],
                             params=oracle.default_params(init_shrink=0.1), make_hydro=OracleBackend)
    a.initData("sedov", r_init=0.1, nsub=4)
    for _ in range(2):
        a.step()
    d = str(tmp_path / "plt_mb")
    names = pf.write_plotfile_amr(d, a)
    H = open(d + "/Header").read().split("\n")
    p = 2 + len(names)
    q = p + 8 + 2 + 2
    assert H[q].split()[:2] == ["0", "1"] and H[q + 6].split()[:2] == ["1", "2"]      # level 1 lists two grids
    assert H[q + 8].split() == ["0.25", "0.5"] and H[q + 11].split() == ["0.5", "0.75"]
    cell_h = open(d + "/Level_1/Cell_H").read().split("\n")
    assert cell_h[4] == "(2 0" and cell_h[5] == "((8,8,8) (15,23,23) (0,0,0))" and cell_h[6] == "((16,8,8) (23,23,23) (0,0,0))"
    r = pf.read_plotfile_amr(d)
    assert [len(lv["fabs"]) for lv in r["levels"]] == [1, 2]
    for b, got, box in zip(a.fine.boxes, r["levels"][1]["fabs"], r["levels"][1]["boxes"]):
        assert (tuple(box[0]), tuple(box[1])) == b.bx
        assert np.array_equal(got[:8], b.S_new().numpy())
        assert np.array_equal(got[names.index("x_velocity")], got[1] / got[0])


def test_level_wide_retry_on_a_refined_level_stays_conservative(oracle):
    """init_shrink = 1 makes the first step fail its validity check: every level goes through retry_advance_ctu as a
    whole (all boxes of the level back to the old state, fluxes cleared, two subcycles); the fluxes that reach the
    flux registers are the sums over the subcycles, so the composite update stays conservative."""
    import castro_amd
    a = castro_amd.CastroAmr((16, 16, 16), patches=[[((4, 4, 4), (7, 11, 11)), ((8, 4, 4), (11, 11, 11))]],
                             params=oracle.default_params(init_shrink=1.0, cfl=0.9), make_hydro=OracleBackend)
    a.initData("sedov", r_init=0.1, nsub=4)
    m0, e0 = a.composite_sum(0), a.composite_sum(4)
    a.step()
    assert [(lev.nsubcycles, lev.nretries) for lev in a.levels] == [(2, 1), (2, 1)]
    assert "validity" in a.levels[1].last_failure
    for _ in range(3):
        a.step()
    assert abs(a.composite_sum(0) - m0) <= 1e-12 * m0 and abs(a.composite_sum(4) - e0) <= 1e-12 * e0


def _spy_regrids(a):
    calls = []
    orig = a.regrid

    def spy(lbase=0, alpha=1.0):
        r = orig(lbase, alpha)
        calls.append((a.nstep, lbase, alpha, r))
        return r
    a.regrid = spy
    return calls


def test_levels_regrid_on_their_own_step_counts(oracle):
    """Amr::level_count: with regrid_int = 1 level 1 regrids the level above it at the start of each of its own
    steps, i.e. also in the middle of a coarse step (its ghost zones are then filled at alpha = 1/2 of the coarse
    interval); with four levels and regrid_int = 2, level 2 regrids level 3 at a level-1 boundary inside the coarse
    step.  Composite mass and energy survive all of them."""
    import castro_amd
    kw = dict(params=oracle.default_params(init_shrink=0.1), make_hydro=OracleBackend,
              refine=[("density", "gradient", 0.05), ("rho_E", "relative_gradient", 0.5)], n_error_buf=1, blocking_factor=4)
    a = castro_amd.CastroAmr((16, 16, 16), regrid_int=1, max_level=2, **kw)
    a.initData("sedov", r_init=0.08, nsub=4)
    calls = _spy_regrids(a)
    m0, e0 = a.composite_sum(0), a.composite_sum(4)
    for _ in range(8):
        a.step()
        assert abs(a.composite_sum(0) - m0) <= 1e-12 * m0 and abs(a.composite_sum(4) - e0) <= 1e-12 * e0
    assert [(c[1], c[2]) for c in calls[:3]] == [(1, 0.5), (0, 0.0), (1, 0.5)]          # step 0: only the mid-step one
    assert any(c[1] == 1 and c[2] == 0.5 and c[3] for c in calls), "no mid-step regrid changed the grids"

    b = castro_amd.CastroAmr((16, 16, 16), regrid_int=2, max_level=3, **kw)
    b.initData("sedov", r_init=0.08, nsub=4)
    assert len(b.levels) == 4
    calls = _spy_regrids(b)
    m0, e0 = b.composite_sum(0), b.composite_sum(4)
    for _ in range(3):
        b.step()
    assert abs(b.composite_sum(0) - m0) <= 1e-12 * m0 and abs(b.composite_sum(4) - e0) <= 1e-12 * e0
    # coarse step 0: level 2 has taken two steps when level 1 starts its second one; coarse step 1 starts with level 1's
    # turn (two steps taken), coarse step 2 with level 0's
    assert [(c[0], c[1]) for c in calls] == [(0, 2), (1, 1), (1, 2), (2, 0), (2, 2)]
    for l in range(2, 4):
        (plo, phi), (qlo, qhi) = b.pbox[l - 1], b.pbox[l]
        assert all(2 * plo[d] <= qlo[d] and qhi[d] <= 2 * phi[d] + 1 for d in range(3))


def _rolled_copy(one, two, shift_crse):
    """Give `two` (patches = the periodic image of `one`'s centred patch, rolled by shift_crse coarse zones in x) the
    initial data of `one` rolled by that shift."""
    import torch
    two.crse.S_new()[:] = torch.roll(one.crse.S_new(), shift_crse, dims=3)
    f = one.fine.S_new()
    nfx = 2 * one.n_cell[0]
    o = one.fine.lo[0]
    for b in two.fine.boxes:
        xs = [(x - 2 * shift_crse) % nfx - o for x in range(b.lo[0], b.hi[0] + 1)]       # where these zones sit in `one`
        assert xs == list(range(xs[0], xs[0] + len(xs))) and 0 <= xs[0] and xs[-1] < f.shape[3]
        b.S_new()[:] = f[:, :, :, xs[0]:xs[-1] + 1]


def test_periodic_domain_with_a_refined_region_across_the_boundary(oracle):
    """Translation invariance on a periodic domain: the centred Sedov problem with a centred refined box, and the same
    data rolled by half a domain with the refined region now in two boxes on either side of the periodic boundary
    (same-level copies and reflux through the periodic images of the boxes): the rolled solution, bit for bit."""
    import castro_amd
    import torch
    kw = dict(params=oracle.default_params(init_shrink=0.1), make_hydro=OracleBackend, lo_bc=(0, 0, 0), hi_bc=(0, 0, 0))
    one = castro_amd.CastroAmr((16, 16, 16), patches=[((4, 4, 4), (11, 11, 11))], **kw)
    two = castro_amd.CastroAmr((16, 16, 16), patches=[[((0, 4, 4), (3, 11, 11)), ((12, 4, 4), (15, 11, 11))]], **kw)
    one.initData("sedov", r_init=0.1, nsub=4)
    _rolled_copy(one, two, 8)
    m0 = two.composite_sum(0)
    assert m0 == one.composite_sum(0)
    while one.time < 0.025 - 1e-15:                      # the shock crosses the coarse-fine boundary
        assert one.step(0.025) == two.step(0.025)
    assert torch.equal(two.crse.S_new(), torch.roll(one.crse.S_new(), 8, dims=3))
    f = one.fine.S_new()
    for b in two.fine.boxes:
        x0 = (b.lo[0] - 16) % 32 - one.fine.lo[0]
        assert torch.equal(b.S_new(), f[:, :, :, x0:x0 + b.n[0]])
    assert abs(two.composite_sum(0) - m0) <= 1e-12 * m0
    assert np.abs(one.crse.S_new().numpy()[0] - 1.0).max() > 0.05


def test_a_box_that_touches_a_level_two_below_is_refused(oracle):
    """Proper nesting: a level-2 box flush with the edge of its level-1 parent would border level 0 directly and lose
    its flux correction there (composite sums drift); the layout is refused when the levels are bound.  The same box
    two level-1 zones further in is accepted and conserves."""
    import castro_amd
    kw = dict(params=oracle.default_params(init_shrink=0.1), make_hydro=OracleBackend)
    lev1 = [((0, 4, 4), (3, 11, 11)), ((12, 4, 4), (15, 11, 11))]
    with pytest.raises(AssertionError, match="not properly nested"):
        castro_amd.CastroAmr((16, 16, 16), patches=[lev1, ((24, 10, 10), (29, 19, 19))], **kw)
    a = castro_amd.CastroAmr((16, 16, 16), patches=[lev1, ((26, 10, 10), (29, 19, 19))], **kw)
    a.initData("sedov", r_init=0.3, nsub=4)
    m0 = a.composite_sum(0)
    for _ in range(2):
        a.step()
    assert abs(a.composite_sum(0) - m0) <= 1e-12 * m0


def test_source_fillpatch_of_a_refined_level(oracle):
    """AmrLevel::FillPatch of Source_Type on a refined level (Castro_advance_ctu.cpp:138-140): ghost zones under another
    box of the level take its valid data, the others the coarse Source_Type data interpolated in time ((1 - alpha) old
    + alpha new) and space -- exact for fields linear in space."""
    import castro_amd
    a = castro_amd.CastroAmr((16, 16, 16), params=oracle.default_params(), make_hydro=OracleBackend, do_grav=True,
                             const_grav=-1.0, patches=[[((4, 4, 4), (7, 11, 11)), ((8, 4, 4), (11, 11, 11))]])
    c = a.crse.boxes[0]
    (x0, y0, z0), (x1, y1, z1) = c.sbox
    z, y, x = np.meshgrid(np.arange(z0, z1 + 1) + 0.5, np.arange(y0, y1 + 1) + 0.5, np.arange(x0, x1 + 1) + 0.5, indexing="ij")
    for n in range(7):
        c.old_source[n] = torch.from_numpy((n + 1) * (1.0 + 0.5 * x - 0.25 * y + 0.125 * z))
        c.new_source_g[n] = torch.from_numpy((n + 1) * (-2.0 + 0.25 * x + 0.5 * y - 0.75 * z))
    fine = a.fine
    for i, b in enumerate(fine.boxes):
        b.old_source[:] = 100.0 + i
    fine.alpha = 0.5
    fine.fill_source("old_source")
    for i, b in enumerate(fine.boxes):
        (x0, y0, z0), (x1, y1, z1) = b.sbox
        z, y, x = np.meshgrid(*[(np.arange(p, q + 1) + 0.5) / 2 for p, q in ((z0, z1), (y0, y1), (x0, x1))], indexing="ij")
        got = b.old_source.numpy()
        other = fine.boxes[1 - i]
        for n in range(7):
            want = (n + 1) * 0.5 * ((1.0 + 0.5 * x - 0.25 * y + 0.125 * z) + (-2.0 + 0.25 * x + 0.5 * y - 0.75 * z))
            for (lo, hi), val in ((b.bx, 100.0 + i), (other.bx, 100.0 + 1 - i)):
                it = castro_amd.cluster.intersect((lo, hi), b.sbox)
                want[it[0][2] - z0:it[1][2] - z0 + 1, it[0][1] - y0:it[1][1] - y0 + 1, it[0][0] - x0:it[1][0] - x0 + 1] = val
            assert np.abs(got[n] - want).max() <= 1e-13 * np.abs(want).max()


def test_source_term_predictor_on_refined_levels(oracle):
    """castro.source_term_predictor = 1 with AMR (round 6; Castro::create_source_corrector, Castro.cpp:3780-3818, on every level:
    the FillPatch of the last advance's new-time momentum sources x 2 / lastDt).  Level 1 in one box or in two gives the same
    bits (the corrector's ghost zones then come from the neighbour box and from both parents), the predictor changes the
    solution, and a single-level CastroAmr equals castro_amd.Castro with the predictor."""
    import castro_amd
    kw = dict(make_hydro=OracleBackend, do_grav=True, const_grav=-3.0, lo_bc=(0, 0, 0), hi_bc=(0, 0, 0))
    P = lambda pred: oracle.default_params(init_shrink=0.1, source_term_predictor=pred)
    l2 = ((12, 12, 12), (19, 19, 19))
    one = castro_amd.CastroAmr((16, 16, 16), params=P(1), patches=[((4, 4, 4), (11, 11, 11)), l2], **kw)
    two = castro_amd.CastroAmr((16, 16, 16), params=P(1), patches=[[((4, 4, 4), (11, 7, 11)), ((4, 8, 4), (11, 11, 11))], l2], **kw)
    off = castro_amd.CastroAmr((16, 16, 16), params=P(0), patches=[((4, 4, 4), (11, 11, 11)), l2], **kw)
    for a in (one, two, off):
        a.initData("sedov", r_init=0.08, nsub=4)
    for _ in range(5):
        dt = one.step(0.05)
        assert dt == two.step(0.05)
        off.step(0.05)
    assert np.array_equal(two.crse.S_new().numpy(), one.crse.S_new().numpy())
    assert np.array_equal(_assemble(two.levels[1], (8, 8, 8), (16, 16, 16)), one.levels[1].S_new().numpy())
    assert np.array_equal(two.levels[2].S_new().numpy(), one.levels[2].S_new().numpy())
    for l in range(3):
        assert one.levels[l].lastDt < 1.0                                   # Castro::lastDt of every level is the last advance's dt
        c = one.levels[l].boxes[0].source_corrector.numpy()
        assert np.abs(c[3]).max() > 0.0 and not c[0].any() and not c[4:].any()      # momentum components only
    assert not np.array_equal(one.levels[2].S_new().numpy(), off.levels[2].S_new().numpy())
    # one level: the AMR driver with the predictor is the single-level driver with the predictor
    flat = castro_amd.CastroAmr((16, 16, 16), params=P(1), patches=[], **kw)
    ref = castro_amd.Castro((16, 16, 16), params=P(1), hydro=OracleBackend(), do_grav=True, const_grav=-3.0, lo_bc=(0, 0, 0), hi_bc=(0, 0, 0))
    for a in (flat, ref):
        a.initData("sedov", r_init=0.08, nsub=4)
    for _ in range(5):
        assert flat.step(0.05) == ref.step(0.05)
    assert np.array_equal(flat.crse.S_new().numpy(), ref.S_new().numpy())


def test_gravity_and_rotation_on_refined_levels(oracle):
    """Constant gravity and rotation on three levels: level 1 in one box or in two (the Source_Type ghost zones of a box
    then come from its neighbour and from both parents) -- bit for bit the same; mass is conserved, the momentum gained
    is the impulse of gravity to first order, and the sources are felt on every level."""
    import castro_amd
    rot = oracle.make_rotation(rotational_period=20.0, rot_axis=3)
    kw = dict(params=oracle.default_params(init_shrink=0.1), make_hydro=OracleBackend, do_grav=True, const_grav=-3.0, rotation=rot,
              lo_bc=(0, 0, 0), hi_bc=(0, 0, 0))
    l2 = ((12, 12, 12), (19, 19, 19))
    one = castro_amd.CastroAmr((16, 16, 16), patches=[((4, 4, 4), (11, 11, 11)), l2], **kw)
    two = castro_amd.CastroAmr((16, 16, 16), patches=[[((4, 4, 4), (11, 7, 11)), ((4, 8, 4), (11, 11, 11))], l2], **kw)
    for a in (one, two):
        a.initData("sedov", r_init=0.08, nsub=4)
    m0 = one.composite_sum(0)
    for _ in range(6):
        assert one.step(0.05) == two.step(0.05)
    assert np.array_equal(two.crse.S_new().numpy(), one.crse.S_new().numpy())
    assert np.array_equal(_assemble(two.levels[1], (8, 8, 8), (16, 16, 16)), one.levels[1].S_new().numpy())
    assert np.array_equal(two.levels[2].S_new().numpy(), one.levels[2].S_new().numpy())
    assert abs(one.composite_sum(0) - m0) <= 1e-12 * m0            # periodic: the falling gas stays in the box
    pz = one.composite_sum(3)
    assert abs(pz / (m0 * -3.0 * one.time) - 1.0) < 0.02
    far = one.levels[2].S_new().numpy()[:, 0, 0, 0]                 # a quiet corner zone of the finest level falls freely
    assert abs(far[3] / (far[0] * -3.0 * one.time) - 1.0) < 1e-3


def test_proper_nesting_domain(oracle):
    """Amr::grid_places' proper nesting domain restated (CastroAmr._nesting_cells): the region of the level that keeps its
    boxes, shrunk by one blocking cell -- not at a physical boundary, and not across a periodic one when the level
    continues on the other side -- then refined and shrunk again for every level above."""
    import castro_amd
    kw = dict(params=oracle.default_params(), make_hydro=OracleBackend, blocking_factor=4)
    # level 1 (zones of level 1: x 0..15 and 24..31, y 8..23, z 0..31): walls in x, periodic in y and z
    a = castro_amd.CastroAmr((16, 16, 16), patches=[[((0, 4, 0), (7, 11, 15)), ((12, 4, 0), (15, 11, 15))]],
                             lo_bc=(4, 0, 0), hi_bc=(4, 0, 0), **kw)
    cells = a._nesting_cells(1, 1, (0, 4, 0), (16, 8, 16), 2)          # cells of 2 zones of level 1, [z, y, x]
    want = np.zeros((16, 8, 16), dtype=bool)
    want[:, 1:7, 0:7] = True        # x: flush with the wall at 0, one cell in from the edge at cell 7; y: one cell in on both sides
    want[:, 1:7, 13:16] = True      # the second box: one cell in from its inner edge, flush with the wall at the top
    assert np.array_equal(cells, want)                                   # z: the level spans the periodic domain, nothing lost
    # one level up: refined, and another cell (of 2 zones of level 2) taken off
    up = a._nesting_cells(2, 1, (0, 8, 0), (32, 16, 32), 2)
    want2 = np.zeros((32, 16, 32), dtype=bool)
    want2[:, 3:13, 0:13] = True
    want2[:, 3:13, 27:32] = True
    assert np.array_equal(up, want2)
    # periodic in x with the level on both sides of the boundary: the boxes continue into each other
    b = castro_amd.CastroAmr((16, 16, 16), patches=[[((0, 4, 0), (7, 11, 15)), ((12, 4, 0), (15, 11, 15))]],
                             lo_bc=(0, 0, 0), hi_bc=(0, 0, 0), **kw)
    assert np.array_equal(b._nesting_cells(1, 1, (0, 4, 0), (16, 8, 16), 2), want)
    # ... and with the level on one side only, the periodic boundary is an edge like any other
    c = castro_amd.CastroAmr((16, 16, 16), patches=[[((0, 4, 0), (7, 11, 15))]], lo_bc=(0, 0, 0), hi_bc=(0, 0, 0), **kw)
    one = c._nesting_cells(1, 1, (0, 4, 0), (16, 8, 8), 2)
    want3 = np.zeros((16, 8, 8), dtype=bool)
    want3[:, 1:7, 1:7] = True
    assert np.array_equal(one, want3)
    assert a._nesting_cells(1, 0, (0, 4, 0), (16, 8, 16), 2) is None     # level 0 covers the domain: nothing to respect


# ---- the orchestration against an INDEPENDENT restatement of it (oracle/ora_amr_level.c) -----------------------------
# CastroAmr (castro_amd/amr.py: overlap tables, per-box Castro objects, batched operations) and ora_amr_level.c (whole-level
# arrays in C, following Castro::advance / finalize_advance / post_timestep / reflux / avgDown / computeNewDt and AMReX's
# Amr::timeStep recursion) share the per-zone arithmetic here (the oracle's) and nothing of the orchestration.
def _both(oracle, patches, boxes, n=(16, 16, 16), bc=(2, 2, 2), **pkw):
    import castro_amd
    a = castro_amd.CastroAmr(n, patches=patches, params=oracle.default_params(**pkw), make_hydro=OracleBackend, lo_bc=bc, hi_bc=bc)
    b = oracle.Amr(boxes, oracle.make_geom(n, lo_bc=bc, hi_bc=bc), oracle.default_params(**pkw), nthreads=4)
    return a, b


def _assert_same(a, b, nsteps, stop_time=-1.0):
    for step in range(nsteps):
        da, db = a.step(stop_time), b.step(stop_time)
        assert da == db, "coarse dt differs at step %d: %r vs %r" % (step, da, db)
        for l, lev in enumerate(a.levels):
            A, B = lev.S_new().cpu().numpy(), b.state(l)
            assert np.array_equal(A, B), "level %d differs after step %d: %d values, max %.3e" % (
                l, step, int((A != B).sum()), float(np.abs(A - B).max()))


TWO = ([((4, 4, 4), (11, 11, 11))], [((0, 0, 0), (15, 15, 15)), ((8, 8, 8), (23, 23, 23))])
THREE = ([((4, 4, 4), (11, 11, 11)), ((12, 12, 12), (19, 19, 19))],
         [((0, 0, 0), (15, 15, 15)), ((8, 8, 8), (23, 23, 23)), ((24, 24, 24), (39, 39, 39))])


@pytest.mark.parametrize("hier", [TWO, THREE], ids=["two-levels", "three-levels"])
def test_sedov_amr_equals_the_independent_orchestration_oracle(oracle, hier):
    a, b = _both(oracle, *hier, init_shrink=0.1)
    a.initData("sedov", r_init=0.1, nsub=4)
    b.init_sedov(r_init=0.1, nsub=4)
    for l, lev in enumerate(a.levels):
        assert np.array_equal(lev.S_new().cpu().numpy(), b.state(l))
    _assert_same(a, b, 6)
    b.close()


def test_sod_amr_with_walls_and_a_patch_at_the_boundary_equals_the_orchestration_oracle(oracle):
    """the refined box touches the low-y wall: coarse zones outside the domain come from the coarse level's boundary
    fill, the register on that face refluxes nothing"""
    patches = [((6, 0, 4), (11, 5, 11))]
    boxes = [((0, 0, 0), (15, 15, 15)), ((12, 0, 8), (23, 11, 23))]
    a, b = _both(oracle, patches, boxes, bc=(2, 4, 4), init_shrink=0.1, cfl=0.8)
    a.initData("sod", rho_l=1.0, u_l=0.0, p_l=1.0, rho_r=0.125, u_r=0.0, p_r=0.1, idir=1, frac=0.5)
    b.init_sod(1.0, 0.0, 1.0, 0.125, 0.0, 0.1, idir=1, frac=0.5)
    _assert_same(a, b, 8)
    b.close()


def test_amr_orchestration_oracle_on_data_where_clean_state_is_not_idempotent(oracle):
    """random composition and momenta: rho X != rho, so every clean_state application matters (normalize_species moves
    rho X, the temperature follows the composition) -- the number and the place of the clean_state calls of the two
    orchestrations must agree exactly: initialize_advance, Sborder, do_advance_ctu, post_timestep on every level."""
    a, b = _both(oracle, *THREE, init_shrink=0.1)
    a.initData("sedov", r_init=0.1, nsub=4)
    b.init_sedov(r_init=0.1, nsub=4)
    rng = np.random.default_rng(3)
    for l, lev in enumerate(a.levels):
        S = lev.S_new().cpu().numpy().copy()
        S[7] = S[0] * rng.uniform(0.2, 0.999, size=S[0].shape)
        S[1] = 0.3 * S[0] * rng.uniform(-1, 1, size=S[0].shape)
        S[4] = S[5] + 0.5 * S[1] ** 2 / S[0]
        lev.S_new().copy_(torch.from_numpy(S))
        b.set_state(l, S)
    for l in range(len(a.lev) - 1, 0, -1):
        a.avgDown(l)
    b.post_init(False)
    _assert_same(a, b, 5)
    b.close()


# ---- the boxes of the refined levels spread over ranks (gloo, oracle backend) -----------------------------------------
import itertools
import os
import socket

import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


# level 1: four boxes (two ranks -> boxes 0, 2 here, 1, 3 there), level 2: three boxes over two of them, one at the edge
# of its parent so that its ghost shell and its flux registers reach into boxes of the other rank
_MR_PATCHES = [[((2, 2, 2), (7, 7, 13)), ((8, 2, 2), (13, 7, 13)), ((2, 8, 2), (7, 13, 13)), ((8, 8, 2), (13, 13, 13))],
               [((8, 8, 8), (15, 15, 23)), ((16, 8, 8), (23, 15, 23)), ((8, 16, 8), (23, 19, 19))]]


def _mr_run(comm, nsteps, bc):
    import castro_amd
    from oracle import oracle_lib as O
    a = castro_amd.CastroAmr((16, 16, 16), patches=_MR_PATCHES, params=O.default_params(init_shrink=0.1), make_hydro=OracleBackend,
                             lo_bc=bc[0], hi_bc=bc[1], comm=comm)
    a.initData("sedov", r_init=0.1, nsub=4)
    dts = [a.step() for _ in range(nsteps)]
    return a, dts


def _mr_worker(rank, world, port, nsteps, bc, out_path):
    import torch.distributed as dist
    import castro_amd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a, dts = _mr_run(castro_amd.DistComm(), nsteps, bc)
        owned = [[b.owned for b in lev.boxes] for lev in a.lev]
        assert all(any(o) for o in owned[1:]) and not all(all(o) for o in owned[1:])     # every rank holds a part only
        levels = [a.gather_level(l) for l in range(len(a.lev))]
        if rank == 0:
            np.savez(out_path, dts=np.array(dts), **{"L%d_%d" % (l, i): arr for l, lv in enumerate(levels) for i, (bx, arr) in enumerate(lv)})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,bc", [(2, ((2, 2, 2), (2, 2, 2))), (3, ((2, 4, 2), (2, 4, 3)))])
def test_amr_with_boxes_spread_over_ranks_is_bitwise_identical_gloo(tmp_path, oracle, world, bc):
    """CastroAmr(comm=DistComm()): three levels, four and three boxes on the refined ones, dealt round-robin to 2 and to
    3 ranks.  Coarse data under fine ghost shells, sibling ghost zones, coarse fluxes into registers, registers into
    the reflux and averaged-down zones all cross ranks; dt sequence and every box of every level equal the one-rank
    run bit for bit."""
    nsteps = 4
    out = str(tmp_path / "amr_ranks.npz")
    mp.spawn(_mr_worker, args=(world, _free_port(), nsteps, bc, out), nprocs=world, join=True)
    got = np.load(out)
    a, dts = _mr_run(None, nsteps, bc)
    assert np.array_equal(got["dts"], np.array(dts))
    for l, lev in enumerate(a.lev):
        for i, b in enumerate(lev.boxes):
            assert np.array_equal(got["L%d_%d" % (l, i)], b.S_new().cpu().numpy()), "level %d box %d" % (l, i)


def _mr_tag_run(comm, nsteps, base_grid=None):
    import castro_amd
    from oracle import oracle_lib as O
    a = castro_amd.CastroAmr((16, 16, 16), params=O.default_params(init_shrink=0.3), make_hydro=OracleBackend, base_grid=base_grid,
                             refine=[("density", "gradient", 0.05), ("rho_E", "relative_gradient", 0.5)], regrid_int=2,
                             n_error_buf=1, blocking_factor=4, max_level=2, cluster=True, grid_eff=0.7, max_grid_size=16, comm=comm)
    a.initData("sedov", r_init=0.08, nsub=4)
    dts, boxes = [], [a.boxes[1:]]
    for _ in range(nsteps):
        dts.append(a.step())
        boxes.append(a.boxes[1:])
    return a, dts, boxes


def _mr_tag_worker(rank, world, port, nsteps, out_path, base_grid=None):
    import pickle
    import torch.distributed as dist
    import castro_amd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a, dts, boxes = _mr_tag_run(castro_amd.DistComm(), nsteps, base_grid)
        owned = [[b.owned for b in lev.boxes] for lev in a.lev]
        levels = [a.gather_level(l) for l in range(len(a.lev))]
        if rank == 0:
            pickle.dump(dict(dts=dts, boxes=boxes, nregrid=a.nregrid, owned=owned,
                             data=[[arr for bx, arr in lv] for lv in levels]), open(out_path, "wb"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,base_grid", [(2, None), (3, None), (3, (2, 2, 2))])
def test_tag_driven_amr_regrids_with_boxes_spread_over_ranks_gloo(tmp_path, oracle, world, base_grid):
    """Regridding with the boxes of every level dealt over ranks: each rank tags its own boxes, the buffered tags reduced
    to blocking cells are combined over the ranks (a maximum, like the buffering and the pooling), every rank clusters
    the same array into the same box lists, and the data of a new level come from the parents' and the old boxes'
    owners through the staged transfers.  Box lists after every step, dt sequence and every box equal the one-rank
    run bit for bit, with Berger-Rigoutsos boxes on two refined levels and regrids in between."""
    import pickle
    nsteps = 6
    out = str(tmp_path / "amr_tag_ranks.pkl")
    mp.spawn(_mr_tag_worker, args=(world, _free_port(), nsteps, out, base_grid), nprocs=world, join=True)
    got = pickle.load(open(out, "rb"))
    a, dts, boxes = _mr_tag_run(None, nsteps)           # the reference run keeps level 0 in one box
    if base_grid is not None:                           # level 0 came back as 8 boxes: paste them together
        full = np.empty_like(a.lev[0].boxes[0].S_new().cpu().numpy())
        assert len(got["data"][0]) == 8 and not all(got["owned"][0])
        for i, (kz, jy, ix) in enumerate(itertools.product(range(2), range(2), range(2))):
            full[:, 8 * kz:8 * kz + 8, 8 * jy:8 * jy + 8, 8 * ix:8 * ix + 8] = got["data"][0][i]
        got["data"][0] = [full]
    assert a.nregrid >= 2 and got["nregrid"] == a.nregrid and len(a.lev) == 3
    assert got["boxes"] == boxes and boxes[0] != boxes[-1]
    assert got["dts"] == dts
    assert sum(len(o) for o in got["owned"]) > world and not all(all(o) for o in got["owned"][1:])
    for l, lev in enumerate(a.lev):
        assert len(got["data"][l]) == len(lev.boxes)
        for i, b in enumerate(lev.boxes):
            assert np.array_equal(got["data"][l][i], b.S_new().cpu().numpy()), "level %d box %d" % (l, i)


@pytest.mark.parametrize("case", ["patches", "periodic_tags", "gravity"])
def test_base_level_cut_into_boxes_is_bitwise_identical(oracle, case):
    """CastroAmr(base_grid=(2, 2, 2)): level 0 as eight boxes (same-level copies incl. periodic images + physical boundaries,
    no coarse data, no flux registers) instead of one box that fills its own ghost zones; the refined levels take their
    coarse data, coarse fluxes and reflux targets from several level-0 boxes.  Fixed patches, tag-driven regridding in a
    periodic domain, constant gravity with sources on every level: dt sequence and every zone equal the one-box run."""
    import castro_amd
    kw = dict(params=oracle.default_params(init_shrink=0.1), make_hydro=OracleBackend)
    if case == "patches":
        kw.update(patches=_MR_PATCHES)
    elif case == "periodic_tags":
        kw.update(refine=[("density", "gradient", 0.05), ("rho_E", "relative_gradient", 0.5)], regrid_int=2, n_error_buf=1, blocking_factor=4,
                  max_level=2, cluster=True, grid_eff=0.7, max_grid_size=16, lo_bc=(0, 0, 0), hi_bc=(0, 0, 0),
                  params=oracle.default_params(init_shrink=0.3))
    else:
        kw.update(patches=[((4, 4, 4), (11, 11, 11))], do_grav=True, const_grav=-2.0, lo_bc=(2, 2, 4), hi_bc=(2, 2, 4))
    runs = []
    for bg in (None, (2, 2, 2)):
        a = castro_amd.CastroAmr((16, 16, 16), base_grid=bg, **kw)
        a.initData("sedov", r_init=0.1 if case != "periodic_tags" else 0.08, nsub=4)
        dts = [a.step() for _ in range(5)]
        runs.append((a, dts))
    (a1, d1), (a2, d2) = runs
    assert d1 == d2 and a1.boxes == a2.boxes and len(a2.lev[0].boxes) == 8
    if case == "periodic_tags":
        assert a1.nregrid >= 1 and a1.nregrid == a2.nregrid
    full = a1.lev[0].boxes[0].S_new().numpy()
    for b in a2.lev[0].boxes:
        sl = (slice(None),) + tuple(slice(b.lo[d], b.hi[d] + 1) for d in (2, 1, 0))
        assert np.array_equal(full[sl], b.S_new().numpy()), b.bx
    for l in range(1, len(a1.lev)):
        for x, y in zip(a1.lev[l].boxes, a2.lev[l].boxes):
            assert np.array_equal(x.S_new().numpy(), y.S_new().numpy()), (l, x.bx)


def _mr_grav_run(comm, nsteps, base_grid, predictor=0):
    import castro_amd
    from oracle import oracle_lib as O
    a = castro_amd.CastroAmr((16, 16, 16), patches=_MR_PATCHES, params=O.default_params(init_shrink=0.1, source_term_predictor=predictor),
                             make_hydro=OracleBackend,
                             do_grav=True, const_grav=-2.0, lo_bc=(2, 2, 4), hi_bc=(2, 2, 4), comm=comm, base_grid=base_grid,
                             rotation=castro_amd.make_rotation(3.0, rot_axis=3, center=(0.5, 0.5, 0.5)))
    a.initData("sedov", r_init=0.1, nsub=4)
    dts = [a.step() for _ in range(nsteps)]
    return a, dts


def _mr_grav_worker(rank, world, port, nsteps, base_grid, out_path, predictor=0):
    import pickle
    import torch.distributed as dist
    import castro_amd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a, dts = _mr_grav_run(castro_amd.DistComm(), nsteps, base_grid, predictor)
        levels = [a.gather_level(l) for l in range(len(a.lev))]
        if rank == 0:
            pickle.dump(dict(dts=dts, data=[[(bx, arr) for bx, arr in lv] for lv in levels]), open(out_path, "wb"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,base_grid,predictor", [(2, None, 0), (3, (2, 2, 2), 0), (3, (2, 2, 2), 1)])
def test_amr_with_sources_and_boxes_spread_over_ranks_gloo(tmp_path, oracle, world, base_grid, predictor):
    """Constant gravity and rotation on three levels with the boxes (and, with base_grid, level 0 too) dealt over ranks:
    the Source_Type FillPatch of a refined level -- coarse sources interpolated in time and space, the siblings' sources --
    goes through the staged transfers like the state (predictor = 1: so does the FillPatch of Castro::source_corrector).  dt
    sequence and every box equal the one-rank run bit for bit."""
    import pickle
    nsteps = 4
    out = str(tmp_path / "amr_grav_ranks.pkl")
    mp.spawn(_mr_grav_worker, args=(world, _free_port(), nsteps, base_grid, out, predictor), nprocs=world, join=True)
    got = pickle.load(open(out, "rb"))
    a, dts = _mr_grav_run(None, nsteps, base_grid, predictor)
    assert got["dts"] == dts
    for l, lev in enumerate(a.lev):
        assert [bx for bx, _ in got["data"][l]] == [b.bx for b in lev.boxes]
        for (bx, arr), b in zip(got["data"][l], lev.boxes):
            assert np.array_equal(arr, b.S_new().cpu().numpy()), "level %d box %s" % (l, bx)


# two refined boxes that touch through the periodic x boundary and one in the middle; level 0 in eight boxes
_PER_PATCHES = [[((0, 4, 4), (3, 11, 11)), ((6, 4, 4), (9, 11, 11)), ((12, 4, 4), (15, 11, 11))]]


def _mr_periodic_run(comm, nsteps):
    import castro_amd
    from oracle import oracle_lib as O
    a = castro_amd.CastroAmr((16, 16, 16), patches=_PER_PATCHES, params=O.default_params(init_shrink=0.1), make_hydro=OracleBackend,
                             lo_bc=(0, 0, 2), hi_bc=(0, 0, 2), comm=comm, base_grid=(2, 2, 2))
    a.initData("sod", rho_l=1.0, u_l=0.3, p_l=1.0, rho_r=0.125, u_r=-0.2, p_r=0.1, idir=1, frac=0.8)
    dts = [a.step() for _ in range(nsteps)]
    return a, dts


def _mr_periodic_worker(rank, world, port, nsteps, out_path):
    import pickle
    import torch.distributed as dist
    import castro_amd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a, dts = _mr_periodic_run(castro_amd.DistComm(), nsteps)
        levels = [a.gather_level(l) for l in range(len(a.lev))]
        if rank == 0:
            pickle.dump(dict(dts=dts, data=[[(bx, arr) for bx, arr in lv] for lv in levels]), open(out_path, "wb"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_periodic_amr_with_boxes_spread_over_ranks_gloo(tmp_path, oracle, world):
    """A periodic domain (x and y) with the boxes on several ranks: the periodic images of the level-0 boxes and of two
    refined boxes that touch through the x boundary are read from other ranks (a staged transfer of a box shifted by the
    domain extent), the reflux of a fine face on the boundary lands in the coarse zone at the other end of the domain.
    A Sod discontinuity at 0.8 of the x extent with moving gas on both sides crosses the boundary within the run.  Bitwise
    equal to one rank."""
    import pickle
    nsteps = 5
    out = str(tmp_path / "amr_periodic_ranks.pkl")
    mp.spawn(_mr_periodic_worker, args=(world, _free_port(), nsteps, out), nprocs=world, join=True)
    got = pickle.load(open(out, "rb"))
    a, dts = _mr_periodic_run(None, nsteps)
    assert got["dts"] == dts
    for l, lev in enumerate(a.lev):
        for (bx, arr), b in zip(got["data"][l], lev.boxes):
            assert bx == b.bx and np.array_equal(arr, b.S_new().cpu().numpy()), "level %d box %s" % (l, bx)
