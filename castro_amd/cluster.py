"""Grid generation from tagged zones: the point-clustering algorithm of Berger & Rigoutsos (IEEE Trans. Systems,
Man and Cybernetics 21(5), 1991), which AMReX's `Amr::grid_places` / `ClusterList::chop` [3P] apply to the tags that
`Castro::errorEst` (Castro.cpp:3131-3164) sets.  Restated from the paper, not from AMReX's sources (which are absent
from the reference tree): box lists can differ from an AMReX run's in where a cut is placed; every list produced here
covers all tags, stays inside the allowed region and respects blocking factor and maximum box size.

On the host: castro_amd_berger_rigoutsos of the kernel library (C++, no device) with this file's numpy form as its reference and fallback.
All boxes are (lo, hi) with inclusive integer bounds, index order (x, y, z); arrays are indexed [z, y, x].
"""
import numpy as np


def intersect(a, b):
    """Intersection of two boxes or None."""
    lo = tuple(max(a[0][d], b[0][d]) for d in range(3))
    hi = tuple(min(a[1][d], b[1][d]) for d in range(3))
    return (lo, hi) if all(lo[d] <= hi[d] for d in range(3)) else None


def dilate(tags, n):
    """Tags grown by n zones in every direction (amr.n_error_buf): a separable maximum over a (2n+1)^3 cube."""
    out = tags
    for ax in range(3):
        acc = out.copy()
        for s in range(1, n + 1):
            lo = [slice(None)] * 3
            hi = [slice(None)] * 3
            lo[ax], hi[ax] = slice(s, None), slice(None, -s)
            acc[tuple(hi)] |= out[tuple(lo)]
            acc[tuple(lo)] |= out[tuple(hi)]
        out = acc
    return out


def coarsen_any(tags, a):
    """A coarse cell (a^3 zones) is tagged when any of its zones is.  Shape must be a multiple of a."""
    if a == 1:
        return tags
    nz, ny, nx = tags.shape
    return tags.reshape(nz // a, a, ny // a, a, nx // a, a).any(axis=(1, 3, 5))


def coarsen_all(mask, a):
    if a == 1:
        return mask
    nz, ny, nx = mask.shape
    return mask.reshape(nz // a, a, ny // a, a, nx // a, a).all(axis=(1, 3, 5))


def _bbox(t):
    """Bounding box ((z0, y0, x0), (z1, y1, x1)) of the True entries of t, or None."""
    if not t.any():
        return None
    out_lo, out_hi = [], []
    for ax in range(3):
        s = t.any(axis=tuple(x for x in range(3) if x != ax))
        idx = np.nonzero(s)[0]
        out_lo.append(int(idx[0])); out_hi.append(int(idx[-1]))
    return tuple(out_lo), tuple(out_hi)


def _find_cut(sig):
    """Cut position along one axis from the signature (number of tags per plane): the widest hole nearest the
    centre, else the strongest inflection of the discrete Laplacian; None when neither exists.  A cut at c splits
    [0, c-1] | [c, n-1]."""
    n = len(sig)
    if n < 2:
        return None, 0
    mid = 0.5 * (n - 1)
    holes = np.nonzero(sig == 0)[0]
    if len(holes):
        c = int(holes[np.argmin(np.abs(holes - mid))])
        return max(c, 1), 2                      # quality 2: a hole always wins
    if n < 4:
        return None, 0
    lap = sig[:-2] - 2 * sig[1:-1] + sig[2:]     # lap[i] belongs to plane i+1
    best, cut = 0, None
    for i in range(len(lap) - 1):
        if lap[i] * lap[i + 1] < 0:
            jump = abs(int(lap[i + 1]) - int(lap[i]))
            if jump > best or (jump == best and cut is not None and abs(i + 2 - mid) < abs(cut - mid)):
                best, cut = jump, i + 2          # between planes i+1 and i+2
    return cut, (1 if cut is not None else 0)


def _chop(t, m, off, eff, min_cells, out):
    bb = _bbox(t)
    if bb is None:
        return
    (z0, y0, x0), (z1, y1, x1) = bb
    sub = t[z0:z1 + 1, y0:y1 + 1, x0:x1 + 1]
    subm = m[z0:z1 + 1, y0:y1 + 1, x0:x1 + 1]
    o = (off[0] + z0, off[1] + y0, off[2] + x0)
    n = sub.shape
    inside = bool(subm.all())
    if inside and (sub.sum() >= eff * sub.size or max(n) <= min_cells):
        out.append(((o[2], o[1], o[0]), (o[2] + n[2] - 1, o[1] + n[1] - 1, o[0] + n[0] - 1)))
        return
    # best cut over the three axes: holes first, then inflections, else bisect the longest side
    cand = []
    for ax in range(3):
        sig = sub.sum(axis=tuple(x for x in range(3) if x != ax)).astype(np.int64)
        c, q = _find_cut(sig)
        if c is not None:
            cand.append((q, n[ax], ax, c))
    if cand:
        cand.sort(key=lambda v: (-v[0], -v[1], v[2]))
        _, _, ax, c = cand[0]
    else:
        ax = int(np.argmax(n))
        c = n[ax] // 2
        if c == 0:                               # a single cell outside the allowed region cannot happen (tags &= mask)
            out.append(((o[2], o[1], o[0]), (o[2], o[1], o[0])))
            return
    lo = [slice(None)] * 3
    hi = [slice(None)] * 3
    lo[ax], hi[ax] = slice(0, c), slice(c, None)
    o2 = list(o)
    o2[ax] += c
    _chop(sub[tuple(lo)], subm[tuple(lo)], o, eff, min_cells, out)
    _chop(sub[tuple(hi)], subm[tuple(hi)], tuple(o2), eff, min_cells, out)


def berger_rigoutsos_numpy(tags, mask=None, grid_eff=0.7, min_cells=1):
    """Boxes (in the index space of `tags`, origin 0) that cover every tagged cell, lie inside `mask`, and are each
    filled to at least grid_eff with tags (or are no larger than min_cells a side).  The numpy statement of the algorithm:
    the reference form of castro_amd_berger_rigoutsos (tests/test_cluster_cpu.py holds the two to the same boxes)."""
    tags = np.asarray(tags, dtype=bool)
    mask = np.ones_like(tags) if mask is None else np.asarray(mask, dtype=bool)
    out = []
    _chop(tags & mask, mask, (0, 0, 0), grid_eff, min_cells, out)
    return out


def berger_rigoutsos_native(tags, mask=None, grid_eff=0.7, min_cells=1):
    """The same through castro_amd_berger_rigoutsos of the kernel library (host code: csrc/cluster_host.hip; no device needed):
    50x faster than the numpy form, which cost a tenth of an AMR coarse step (round 6)."""
    import ctypes as C
    from . import _lib as L
    lib = L.load()
    t = np.ascontiguousarray(np.asarray(tags, dtype=bool), dtype=np.uint8)
    m = None if mask is None else np.ascontiguousarray(np.asarray(mask, dtype=bool), dtype=np.uint8)
    nz, ny, nx = t.shape
    cap = 256
    while True:
        buf = np.empty((cap, 6), dtype=np.int32)
        nb = lib.castro_amd_berger_rigoutsos(t.ctypes.data, None if m is None else m.ctypes.data, nz, ny, nx, float(grid_eff),
                                             int(min_cells), buf.ctypes.data, cap)
        if nb < 0:
            raise RuntimeError("castro_amd_berger_rigoutsos failed (%d)" % nb)
        if nb <= cap:
            break
        cap = nb
    # (z0, y0, x0, z1, y1, x1) -> ((x0, y0, z0), (x1, y1, z1))
    return [((int(b[2]), int(b[1]), int(b[0])), (int(b[5]), int(b[4]), int(b[3]))) for b in buf[:nb]]


def berger_rigoutsos(tags, mask=None, grid_eff=0.7, min_cells=1):
    """The clustering used by the AMR driver: the library's host routine, or the numpy form where the library cannot be loaded
    (or CASTRO_AMD_CLUSTER_NATIVE=0).  Same boxes either way."""
    import os
    if os.environ.get("CASTRO_AMD_CLUSTER_NATIVE", "1") != "0":
        try:
            return berger_rigoutsos_native(tags, mask, grid_eff, min_cells)
        except (ImportError, OSError, AttributeError):
            pass
    return berger_rigoutsos_numpy(tags, mask, grid_eff, min_cells)


def chop_max_size(boxes, max_size):
    """Split boxes longer than max_size into equal parts (amr.max_grid_size)."""
    out = []
    for lo, hi in boxes:
        parts = [[(lo[d], hi[d])] for d in range(3)]
        for d in range(3):
            n = hi[d] - lo[d] + 1
            if n > max_size:
                k = -(-n // max_size)
                base, extra = divmod(n, k)
                edges, s = [], lo[d]
                for i in range(k):
                    e = s + base + (1 if i < extra else 0) - 1
                    edges.append((s, e)); s = e + 1
                parts[d] = edges
        for zr in parts[2]:
            for yr in parts[1]:
                for xr in parts[0]:
                    out.append(((xr[0], yr[0], zr[0]), (xr[1], yr[1], zr[1])))
    return out


def boxes_from_coarse(ct, cm, origin, blocking, grid_eff=0.7, max_size=None):
    """Clustering of already buffered tags given per blocking^3 cell (ct: any zone tagged, cm: all zones allowed):
    sorted boxes in level-l zones, aligned to `blocking`."""
    a = int(blocking)
    boxes = berger_rigoutsos(ct, cm, grid_eff)
    if max_size is not None:
        boxes = chop_max_size(boxes, max(int(max_size) // a, 1))
    out = [(tuple(origin[d] + a * lo[d] for d in range(3)), tuple(origin[d] + a * (hi[d] + 1) - 1 for d in range(3)))
           for lo, hi in boxes]
    return sorted(out, key=lambda b: (b[0][2], b[0][1], b[0][0]))


def make_boxes(tags, origin, mask=None, n_error_buf=1, blocking=1, grid_eff=0.7, max_size=None):
    """Level-l tags (array over the region starting at `origin`, shape a multiple of `blocking`) -> sorted list of
    boxes in level-l zones, aligned to `blocking` zones, covering the buffered tags inside `mask`."""
    tags = np.asarray(tags, dtype=bool)
    mask = np.ones_like(tags) if mask is None else np.asarray(mask, dtype=bool)
    t = dilate(tags, n_error_buf) & mask if n_error_buf > 0 else tags & mask
    a = int(blocking)
    return boxes_from_coarse(coarsen_any(t, a), coarsen_all(mask, a), origin, a, grid_eff, max_size)
