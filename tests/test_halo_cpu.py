"""castro_amd/halo.py -- the message derivation of the many-box ghost exchange (the Python twin of
include/castro_hydro_amd_amrex.H::fill_boundary, round 6) -- without a GPU: pairing of sends and receives across ranks, and
agreement with the sibling tables of castro_amd/amr.py's level FillPatch."""
import numpy as np
import pytest

from castro_amd import halo

BOXES = [((0, 0, 0), (7, 15, 15)), ((8, 0, 0), (15, 7, 15)), ((8, 8, 0), (15, 15, 9)), ((8, 8, 10), (15, 15, 15))]
DOMAIN = ((0, 0, 0), (15, 15, 15))


def _size(b):
    return int(np.prod([b[1][d] - b[0][d] + 1 for d in range(3)]))


@pytest.mark.parametrize("owners", [[0, 0, 0, 0], [0, 1, 0, 1], [0, 1, 2, 2], [3, 2, 1, 0]])
@pytest.mark.parametrize("periodic", [(False, False, False), (True, False, True), (True, True, True)])
def test_every_send_has_its_receive_on_the_peer(owners, periodic):
    """Unequal boxes, several per rank: every message one rank sends is received by exactly one receive of its peer with the same
    tag and the same number of zones, tags are unique between a pair of ranks, and a rank's receives cover exactly the ghost zones
    that lie in some box of the level (or a periodic image of one)."""
    nranks = max(owners) + 1
    per_rank = {r: halo.level_messages(BOXES, owners, r, 4, DOMAIN, periodic) for r in range(nranks)}
    for r, (local, sends, recvs) in per_rank.items():
        assert local == [b for b in range(len(BOXES)) if owners[b] == r]
        for f, peer, box, tag in sends:
            twin = [x for x in per_rank[peer][2] if x[1] == r and x[3] == tag]
            assert len(twin) == 1 and _size(twin[0][2]) == _size(box)
            assert halo._intersect(box, BOXES[local[f]]) == box                      # sends come from valid zones
        for f, peer, box, tag in recvs:
            assert len([x for x in per_rank[peer][1] if x[1] == r and x[3] == tag]) == 1
            assert halo._intersect(box, BOXES[local[f]]) is None                     # receives go to ghost zones only
        for peer in range(nranks):
            for msgs in (sends, recvs):
                tags = [t for _, p, _, t in msgs if p == peer]
                assert len(tags) == len(set(tags))
        # coverage: the union of a box's receive regions = its ghost zones that wrap into some box of the level
        ext = [DOMAIN[1][d] + 1 for d in range(3)]
        owner_of = -np.ones((ext[2], ext[1], ext[0]), dtype=int)
        for b, (lo, hi) in enumerate(BOXES):
            owner_of[lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = b
        for f, m in enumerate(local):
            glo, ghi = tuple(x - 4 for x in BOXES[m][0]), tuple(x + 4 for x in BOXES[m][1])
            cov = np.zeros(tuple(ghi[d] - glo[d] + 1 for d in (2, 1, 0)), dtype=int)
            for ff, _, (lo, hi), _ in recvs:
                if ff == f:
                    cov[lo[2] - glo[2]:hi[2] - glo[2] + 1, lo[1] - glo[1]:hi[1] - glo[1] + 1, lo[0] - glo[0]:hi[0] - glo[0] + 1] += 1
            want = np.zeros_like(cov)
            for k in range(glo[2], ghi[2] + 1):
                for j in range(glo[1], ghi[1] + 1):
                    for i in range(glo[0], ghi[0] + 1):
                        z = [i, j, k]
                        if all(BOXES[m][0][d] <= z[d] <= BOXES[m][1][d] for d in range(3)):
                            continue
                        ok = True
                        for d in range(3):
                            if z[d] < 0 or z[d] >= ext[d]:
                                if periodic[d]:
                                    z[d] %= ext[d]
                                else:
                                    ok = False
                        if ok and owner_of[z[2], z[1], z[0]] >= 0:
                            want[k - glo[2], j - glo[1], i - glo[0]] = 1
            assert np.array_equal(cov, want)


def test_receives_are_the_sibling_table_of_the_amr_driver(oracle):
    """The receive regions of halo.level_messages are the sibling overlaps castro_amd/amr.py derives for the FillPatch of a level of
    several boxes (b.sib: ghost zones of a box under a sibling, periodic images included) -- the plan the C++ adapter has to match."""
    import castro_amd
    from tests.oracle_backend import OracleBackend
    a = castro_amd.CastroAmr((16, 16, 16), params=oracle.default_params(), make_hydro=OracleBackend, lo_bc=(0, 2, 2), hi_bc=(0, 2, 2),
                             patches=[[((0, 4, 4), (3, 11, 11)), ((4, 4, 4), (9, 9, 11)), ((12, 4, 4), (15, 11, 9))]])
    lev = a.levels[1]
    boxes = [b.bx for b in lev.boxes]
    dom = ((0, 0, 0), (31, 31, 31))
    local, sends, recvs = halo.level_messages(boxes, [0] * len(boxes), 0, 4, dom, (True, False, False))
    got = sorted((f, box) for f, _, box, _ in recvs)
    want = sorted((i, (tuple(lo), tuple(hi))) for i, b in enumerate(lev.boxes) for _, (lo, hi), _ in b.sib)
    assert got == want and len(got) >= 4
