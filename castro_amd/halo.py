"""Messages of a same-level ghost exchange over a BoxArray of ANY shape -- several boxes per rank, unequal sizes, periodic
images (AmrLevel::FillPatch's same-level part / MultiFab::FillBoundary as Castro::expand_state uses it,
Source/driver/Castro.cpp:4201-4209).  The host-side twin of include/castro_hydro_amd_amrex.H::fill_boundary: the same derivation
in Python, for hosts and tests that have no AMReX; the exchange itself is castro_amd_fill_boundary_group of the C ABI.

For the local box m and any box b of the level seen under the periodic shift s (b == m only with s != 0):
  receive   grow(m, ng) & (b + s)          the zones of b that lie in m's ghost region            tag T(b -> m, s)
  send      m & (grow(b, ng) + s)          the zones of m that lie in b's ghost region as b, shifted the same way, sees them;
                                           b receives them under the shift -s                      tag T(m -> b, -s)
with T(src, dst, s) = (src * nboxes + dst) * 27 + code(s): both ends compute the same number for a message, and the two regions of
a message have the same shape by construction (one is the other translated by s).
"""
import itertools


def _intersect(a, b):
    lo = tuple(max(a[0][d], b[0][d]) for d in range(3))
    hi = tuple(min(a[1][d], b[1][d]) for d in range(3))
    return (lo, hi) if all(lo[d] <= hi[d] for d in range(3)) else None


def _shift(box, s):
    return (tuple(box[0][d] + s[d] for d in range(3)), tuple(box[1][d] + s[d] for d in range(3)))


def _grow(box, ng):
    return (tuple(x - ng for x in box[0]), tuple(x + ng for x in box[1]))


def shift_code(s, period):
    """0 .. 26 from the signs of a periodic shift (one period per direction at most)"""
    c = 0
    for d in range(3):
        k = 0 if s[d] == 0 else (1 if s[d] > 0 else -1)
        assert s[d] == k * period[d]
        c += (k + 1) * 3 ** d
    return c


def message_tag(src, dst, s, period, nboxes):
    t = (src * nboxes + dst) * 27 + shift_code(s, period)
    assert t < 2 ** 31, "too many boxes for 32-bit message tags"
    return t


def level_messages(boxes, owners, rank, ng, domain, periodic):
    """boxes: [(lo, hi)] of the level (valid boxes, disjoint); owners[b]: rank of box b; domain: (lo, hi); periodic: 3 flags.
    Returns (local, sends, recvs): local = indices of this rank's boxes in the order of the FAB list, sends / recvs =
    [(fab, peer, (lo, hi), tag)] with fab an index into `local` -- the arguments of HipHydro.halo_group."""
    nb = len(boxes)
    period = tuple(domain[1][d] - domain[0][d] + 1 for d in range(3))
    shifts = [tuple(k[d] * period[d] for d in range(3)) for k in itertools.product(*[((-1, 0, 1) if periodic[d] else (0,)) for d in range(3)])]
    local = [b for b in range(nb) if owners[b] == rank]
    sends, recvs = [], []
    for f, m in enumerate(local):
        vm, gm = boxes[m], _grow(boxes[m], ng)
        for b in range(nb):
            for s in shifts:
                if b == m and s == (0, 0, 0):
                    continue
                r = _intersect(gm, _shift(boxes[b], s))
                if r is not None:
                    recvs.append((f, owners[b], r, message_tag(b, m, s, period, nb)))
                snd = _intersect(vm, _shift(_grow(boxes[b], ng), s))
                if snd is not None:
                    sends.append((f, owners[b], snd, message_tag(m, b, tuple(-x for x in s), period, nb)))
    return local, sends, recvs
