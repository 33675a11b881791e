// cluster_host.hip -- Berger-Rigoutsos point clustering on the HOST (no device code): the grid generation of a regrid,
// Amr::grid_places / ClusterList::chop [3P: AMReX, absent from the reference tree] applied to the tags Castro::errorEst sets
// (Source/driver/Castro.cpp:3131-3164).  Restated from the paper (Berger & Rigoutsos, IEEE Trans. SMC 21(5), 1991) exactly as
// castro_amd/cluster.py states it -- same signatures, same cut rules, same tie breaks, so the two give the same boxes
// (tests/test_cluster_cpu.py) -- because the numpy form cost 1-2 ms per regrid of a level of ~50 boxes, a tenth of an AMR
// coarse step (round 6).  Arrays are [z][y][x], one byte per cell; boxes come back as (z0, y0, x0, z1, y1, x1), inclusive.
#include <cstdint>
#include <cstdlib>
#include <vector>
#include "../../include/castro_hydro_amd.h"

namespace {

struct View { const unsigned char* t; const unsigned char* m; long sy, sz; int o[3]; int n[3]; };   // o: offset of the view in the arrays; axes 0 = z, 1 = y, 2 = x

inline unsigned char at(const unsigned char* a, const View& v, int z, int y, int x)
{
    return a[(long)(v.o[0] + z) * v.sz + (long)(v.o[1] + y) * v.sy + (v.o[2] + x)];
}

// signature of the tags along `ax` (number of tagged cells per plane)
void signature(const View& v, int ax, std::vector<long long>& sig)
{
    sig.assign((size_t)v.n[ax], 0);
    for (int z = 0; z < v.n[0]; ++z)
        for (int y = 0; y < v.n[1]; ++y)
            for (int x = 0; x < v.n[2]; ++x)
                if (at(v.t, v, z, y, x)) sig[(size_t)(ax == 0 ? z : ax == 1 ? y : x)] += 1;
}

// cluster.py::_find_cut: the hole nearest the centre (quality 2), else the strongest sign change of the discrete Laplacian
// (quality 1; on equal strength the one nearer the centre), else none (quality 0)
int find_cut(const std::vector<long long>& sig, int& cut)
{
    const int n = (int)sig.size();
    cut = -1;
    if (n < 2) return 0;
    const double mid = 0.5 * (n - 1);
    int hole = -1;
    double best_d = 0.0;
    for (int i = 0; i < n; ++i)
        if (sig[(size_t)i] == 0) {
            const double d = std::abs(i - mid);
            if (hole < 0 || d < best_d) { hole = i; best_d = d; }       // the first of equally near holes (numpy argmin)
        }
    if (hole >= 0) { cut = hole > 1 ? hole : 1; return 2; }
    if (n < 4) return 0;
    std::vector<long long> lap((size_t)n - 2);
    for (int i = 0; i < n - 2; ++i) lap[(size_t)i] = sig[(size_t)i] - 2 * sig[(size_t)i + 1] + sig[(size_t)i + 2];
    long long best = 0;
    for (int i = 0; i + 1 < n - 2; ++i)
        if (lap[(size_t)i] * lap[(size_t)i + 1] < 0) {
            const long long jump = std::llabs(lap[(size_t)i + 1] - lap[(size_t)i]);
            if (jump > best || (jump == best && cut >= 0 && std::abs(i + 2 - mid) < std::abs(cut - mid))) { best = jump; cut = i + 2; }
        }
    return cut >= 0 ? 1 : 0;
}

void chop(const View& v0, double eff, int min_cells, std::vector<int>& out)
{
    // bounding box of the tags
    int lo[3] = { v0.n[0], v0.n[1], v0.n[2] }, hi[3] = { -1, -1, -1 };
    for (int z = 0; z < v0.n[0]; ++z)
        for (int y = 0; y < v0.n[1]; ++y)
            for (int x = 0; x < v0.n[2]; ++x)
                if (at(v0.t, v0, z, y, x)) {
                    if (z < lo[0]) lo[0] = z;
                    if (z > hi[0]) hi[0] = z;
                    if (y < lo[1]) lo[1] = y;
                    if (y > hi[1]) hi[1] = y;
                    if (x < lo[2]) lo[2] = x;
                    if (x > hi[2]) hi[2] = x;
                }
    if (hi[0] < 0) return;
    View v = v0;
    for (int d = 0; d < 3; ++d) { v.o[d] = v0.o[d] + lo[d]; v.n[d] = hi[d] - lo[d] + 1; }
    long long ntag = 0, size = (long long)v.n[0] * v.n[1] * v.n[2];
    bool inside = true;
    for (int z = 0; z < v.n[0]; ++z)
        for (int y = 0; y < v.n[1]; ++y)
            for (int x = 0; x < v.n[2]; ++x) {
                if (at(v.t, v, z, y, x)) ++ntag;
                if (!at(v.m, v, z, y, x)) inside = false;
            }
    const int nmax = v.n[0] > v.n[1] ? (v.n[0] > v.n[2] ? v.n[0] : v.n[2]) : (v.n[1] > v.n[2] ? v.n[1] : v.n[2]);
    auto emit = [&](const View& w) { for (int d = 0; d < 3; ++d) out.push_back(w.o[d]); for (int d = 0; d < 3; ++d) out.push_back(w.o[d] + w.n[d] - 1); };
    if (inside && ((double)ntag >= eff * (double)size || nmax <= min_cells)) { emit(v); return; }
    // best cut over the three axes: quality first, then the longer side, then the lower axis
    int bq = 0, bn = 0, bax = -1, bc = -1;
    std::vector<long long> sig;
    for (int ax = 0; ax < 3; ++ax) {
        signature(v, ax, sig);
        int c;
        const int q = find_cut(sig, c);
        if (c < 0) continue;
        if (bax < 0 || q > bq || (q == bq && v.n[ax] > bn)) { bq = q; bn = v.n[ax]; bax = ax; bc = c; }
    }
    if (bax < 0) {
        bax = 0;
        for (int ax = 1; ax < 3; ++ax) if (v.n[ax] > v.n[bax]) bax = ax;          // the first of the longest sides (numpy argmax)
        bc = v.n[bax] / 2;
        if (bc == 0) { View w = v; for (int d = 0; d < 3; ++d) w.n[d] = 1; emit(w); return; }
    }
    View a = v, b = v;
    a.n[bax] = bc;
    b.o[bax] = v.o[bax] + bc; b.n[bax] = v.n[bax] - bc;
    chop(a, eff, min_cells, out);
    chop(b, eff, min_cells, out);
}

} // namespace

extern "C" int castro_amd_berger_rigoutsos(const unsigned char* tags, const unsigned char* mask, int nz, int ny, int nx,
                                           double grid_eff, int min_cells, int* boxes, int max_boxes)
{
    if (!tags || nz < 1 || ny < 1 || nx < 1 || max_boxes < 0 || (max_boxes > 0 && !boxes)) return CASTRO_AMD_ERR_ARG;
    const size_t n = (size_t)nz * ny * nx;
    // tags &= mask (cluster.py: berger_rigoutsos); a missing mask allows everything
    std::vector<unsigned char> t(n), m(n, 1);
    for (size_t i = 0; i < n; ++i) { if (mask) m[i] = mask[i] ? 1 : 0; t[i] = (tags[i] && m[i]) ? 1 : 0; }
    View v;
    v.t = t.data(); v.m = m.data(); v.sy = nx; v.sz = (long)nx * ny;
    v.o[0] = v.o[1] = v.o[2] = 0; v.n[0] = nz; v.n[1] = ny; v.n[2] = nx;
    std::vector<int> out;
    chop(v, grid_eff, min_cells, out);
    const int nb = (int)(out.size() / 6);
    for (int i = 0; i < nb && i < max_boxes; ++i)
        for (int d = 0; d < 6; ++d) boxes[6 * i + d] = out[(size_t)6 * i + d];
    return nb;
}
