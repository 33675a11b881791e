"""Grid generation from tags (castro_amd/cluster.py, Berger-Rigoutsos 1991): every tag covered, boxes disjoint, aligned,
inside the allowed region, filled to grid_eff, no longer than max_grid_size."""
import numpy as np
import pytest

from castro_amd import cluster as CL


def _paint(boxes, shape, origin=(0, 0, 0)):
    cov = np.zeros(shape, dtype=np.int32)
    for lo, hi in boxes:
        cov[lo[2] - origin[2]:hi[2] - origin[2] + 1, lo[1] - origin[1]:hi[1] - origin[1] + 1, lo[0] - origin[0]:hi[0] - origin[0] + 1] += 1
    return cov


def _shell(n, r0, r1):
    z, y, x = np.mgrid[0:n, 0:n, 0:n]
    r = np.sqrt((x - 0.5 * (n - 1)) ** 2 + (y - 0.5 * (n - 1)) ** 2 + (z - 0.5 * (n - 1)) ** 2)
    return (r > r0) & (r < r1)


def test_dilate_and_coarsen():
    t = np.zeros((8, 8, 8), dtype=bool)
    t[3, 4, 5] = True
    d = CL.dilate(t, 1)
    assert d.sum() == 27 and d[2:5, 3:6, 4:7].all()
    assert CL.dilate(t, 2).sum() == 125
    c = CL.coarsen_any(t, 2)
    assert c.shape == (4, 4, 4) and c.sum() == 1 and c[1, 2, 2]
    m = np.ones((8, 8, 8), dtype=bool)
    m[0, 0, 0] = False
    assert CL.coarsen_all(m, 2).sum() == 63


def test_two_separate_blobs_become_two_boxes():
    t = np.zeros((16, 16, 32), dtype=bool)
    t[2:6, 3:7, 1:5] = True
    t[9:13, 8:12, 20:30] = True
    b = CL.berger_rigoutsos(t, grid_eff=0.7)
    assert sorted(b) == [((1, 3, 2), (4, 6, 5)), ((20, 8, 9), (29, 11, 12))]          # a hole in the signature splits them


@pytest.mark.parametrize("eff", [0.5, 0.7, 0.9])
def test_spherical_shell_is_covered_by_disjoint_aligned_boxes(eff):
    n, a = 64, 4
    tags = _shell(n, 20, 23)
    boxes = CL.make_boxes(tags, (0, 0, 0), n_error_buf=1, blocking=a, grid_eff=eff, max_size=32)
    cov = _paint(boxes, tags.shape)
    assert cov.max() == 1                                                           # disjoint
    assert not (CL.dilate(tags, 1) & (cov == 0)).any()                              # every buffered tag covered
    ct = CL.coarsen_any(CL.dilate(tags, 1), a)
    for lo, hi in boxes:
        assert all(l % a == 0 and (h + 1) % a == 0 for l, h in zip(lo, hi))
        assert all(h - l + 1 <= 32 for l, h in zip(lo, hi))
        sub = ct[lo[2] // a:(hi[2] + 1) // a, lo[1] // a:(hi[1] + 1) // a, lo[0] // a:(hi[0] + 1) // a]
        assert sub.any()
    # the shell is hollow: far fewer zones than its bounding box
    if eff >= 0.7:
        assert cov.sum() < 0.8 * 48 ** 3
    assert boxes == sorted(boxes, key=lambda b: (b[0][2], b[0][1], b[0][0]))


def test_boxes_stay_inside_an_l_shaped_region():
    mask = np.zeros((16, 16, 16), dtype=bool)
    mask[:, :8, :] = True
    mask[:, 8:, :8] = True                                                          # L-shaped union of two parent boxes
    tags = np.zeros_like(mask)
    tags[4:12, 2:14, 2:6] = True
    tags[4:12, 2:6, 2:14] = True                                                    # an L inside the L: its bounding box leaves the mask
    boxes = CL.make_boxes(tags, (100, 200, 300), mask, n_error_buf=0, blocking=2, grid_eff=0.3)
    cov = _paint(boxes, mask.shape, (100, 200, 300))
    assert cov.max() == 1 and not (tags & (cov == 0)).any()
    assert not ((cov > 0) & ~mask).any()
    assert len(boxes) >= 2


def test_grid_eff_zero_gives_the_bounding_box_and_max_size_chops_it():
    tags = _shell(32, 8, 10)
    b = CL.make_boxes(tags, (0, 0, 0), n_error_buf=0, blocking=2, grid_eff=0.0)
    assert b == [((6, 6, 6), (25, 25, 25))]
    c = CL.chop_max_size(b, 8)
    assert len(c) == 27 and _paint(c, (32, 32, 32)).max() == 1 and _paint(c, (32, 32, 32)).sum() == 20 ** 3
    assert all(h - l + 1 <= 8 for lo, hi in c for l, h in zip(lo, hi))


def test_native_clustering_gives_the_boxes_of_the_numpy_form():
    """castro_amd_berger_rigoutsos (csrc/cluster_host.hip: the host routine the AMR driver uses since round 6) against the numpy
    statement of the algorithm in castro_amd/cluster.py: random tag clouds, shells, unions of boxes and sparse tags, with and without
    an allowed region, several efficiencies and minimum sizes -- the same boxes in the same order, box for box."""
    from castro_amd import cluster as CL
    rng = np.random.default_rng(12)
    for case in range(120):
        shape = tuple(int(rng.integers(1, 36)) for _ in range(3))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            t = rng.random(shape) < rng.uniform(0.01, 0.6)
        elif kind == 1:
            z, y, x = np.meshgrid(*[np.arange(n) + 0.5 - n / 2 for n in shape], indexing="ij")
            t = np.abs(np.sqrt(x * x + y * y + z * z) - rng.uniform(2, max(shape) / 2)) < rng.uniform(0.6, 2.5)
        elif kind == 2:
            t = np.zeros(shape, dtype=bool)
            for _ in range(int(rng.integers(1, 5))):
                lo = [int(rng.integers(0, n)) for n in shape]
                hi = [int(rng.integers(l, n)) for l, n in zip(lo, shape)]
                t[lo[0]:hi[0] + 1, lo[1]:hi[1] + 1, lo[2]:hi[2] + 1] = True
        else:
            t = rng.random(shape) < 0.02
        m = None if rng.random() < 0.4 else (rng.random(shape) < rng.uniform(0.7, 1.0))
        eff, mc = float(rng.choice([0.5, 0.7, 0.9, 0.95])), int(rng.choice([1, 1, 2, 4]))
        want = CL.berger_rigoutsos_numpy(t, m, eff, mc)
        got = CL.berger_rigoutsos_native(t, m, eff, mc)
        assert got == want, (case, shape, kind, eff, mc)
        # and the properties either form has to have: every allowed tag covered exactly once, boxes inside the allowed region
        cover = np.zeros(shape, dtype=int)
        for (x0, y0, z0), (x1, y1, z1) in got:
            cover[z0:z1 + 1, y0:y1 + 1, x0:x1 + 1] += 1
            if m is not None:
                assert m[z0:z1 + 1, y0:y1 + 1, x0:x1 + 1].all()
        allowed = t if m is None else (t & m)
        assert (cover[allowed] == 1).all() and cover.max(initial=0) <= 1
