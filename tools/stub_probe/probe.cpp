// probe.cpp -- calls the reference's per-zone hydro functions, compiled UNMODIFIED and IN PLACE from /root/reference
// against the stand-in headers of tools/stub_probe/stub/, on inputs prepared by make_vectors.py.
// STUB-COMPILED, NOT oracle/_ref: AMReX and Microphysics are replaced by stand-ins, so this pins nothing about the
// reference BINARY; it shows whether the oracle's (and the device code's) restatement of these functions has slipped.
#include <Castro.H>
#include <ppm.H>
#include <riemann_solvers.H>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <vector>

using Arr = std::vector<double>;
static std::map<std::string, Arr> in, out;

static void read_blob(const char* path)
{
    std::ifstream f(path, std::ios::binary);
    while (f) {
        char name[48];
        int64_t n;
        if (!f.read(name, 48)) break;
        f.read(reinterpret_cast<char*>(&n), 8);
        Arr a((size_t)n);
        f.read(reinterpret_cast<char*>(a.data()), 8 * n);
        in[std::string(name)] = a;
    }
}

static void write_blob(const char* path)
{
    std::ofstream f(path, std::ios::binary);
    for (auto& kv : out) {
        char name[48] = {0};
        std::strncpy(name, kv.first.c_str(), 47);
        int64_t n = (int64_t)kv.second.size();
        f.write(name, 48);
        f.write(reinterpret_cast<const char*>(&n), 8);
        f.write(reinterpret_cast<const char*>(kv.second.data()), 8 * n);
    }
}

static double scalar(const std::string& k) { return in.at(k)[0]; }

static void set_params(const std::string& p)
{
    auto g = [&](const char* k, double dflt) { auto it = in.find(p + k); return it == in.end() ? dflt : it->second[0]; };
    castro::riemann_solver = (int)g("riemann_solver", 0);
    castro::cg_blend = (int)g("cg_blend", 2);
    castro::hybrid_riemann = (int)g("hybrid_riemann", 0);
    castro::ppm_temp_fix = (int)g("ppm_temp_fix", 0);
    castro::transverse_reset_density = (int)g("transverse_reset_density", 1);
    castro::transverse_reset_rhoe = (int)g("transverse_reset_rhoe", 0);
    castro::transverse_use_eos = (int)g("transverse_use_eos", 0);
    castro::small_dens = g("small_dens", 1.e-100);
    castro::small_pres = g("small_pres", 1.e-100);
    castro::small_temp = g("small_temp", 1.e-100);
    castro::small_ener = g("small_ener", 1.e-100);
    castro::ppm_type = (int)g("ppm_type", 1);
    castro::use_flattening = (int)g("use_flattening", 1);
    castro::first_order_hydro = (int)g("first_order_hydro", 0);
    castro::difmag = g("difmag", 0.1);
    castro::cfl = g("cfl", 0.5);
    castro::limit_fluxes_on_small_dens = (int)g("limit_fluxes_on_small_dens", 0);
    castro::limit_fluxes_on_large_vel = (int)g("limit_fluxes_on_large_vel", 0);
    castro::speed_limit = g("speed_limit", 0.0);
    castro::plm_iorder = (int)g("plm_iorder", 2);
    castro::plm_limiter = (int)g("plm_limiter", 2);
    castro::use_pslope = (int)g("use_pslope", 1);
    castro::source_term_predictor = (int)g("source_term_predictor", 0);
    castro::dual_energy_eta1 = g("dual_energy_eta1", 1.0);
    castro::dual_energy_eta2 = g("dual_energy_eta2", 1.0e-4);
}

// the box algebra of one tile of the reference's hydro driver, with the stand-in Box
static Box grown(const Box& b, int gx, int gy, int gz)
{
    const int lo[3] = {b.lo_[0] - gx, b.lo_[1] - gy, b.lo_[2] - gz}, hi[3] = {b.hi_[0] + gx, b.hi_[1] + gy, b.hi_[2] + gz};
    return Box(lo, hi);
}
static Box nodal(const Box& b, int d)
{
    int hi[3] = {b.hi_[0], b.hi_[1], b.hi_[2]};
    hi[d] += 1;
    return Box(b.lo_, hi);
}

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    read_blob(argv[1]);
    Castro castro_obj;
    for (int d = 0; d < 3; ++d) {
        Castro::geom.d.prob_lo[d] = 0.0; Castro::geom.d.prob_hi[d] = 1.0; Castro::geom.d.dx[d] = 1.0;
        Castro::geom.d.domain.lo_[d] = -1000; Castro::geom.d.domain.hi_[d] = 1000;
        Castro::phys_bc.l[d] = Outflow; Castro::phys_bc.h[d] = Outflow;
    }

    // ---- ppm_reconstruct + ppm_int_profile (ppm.H:54-211) ----
    if (in.count("ppm.s")) {
        const Arr &s = in["ppm.s"], &fl = in["ppm.flat"], &u = in["ppm.u"], &c = in["ppm.c"];
        const long n = (long)fl.size();
        const double dtdx = scalar("ppm.dtdx");
        Arr o(8 * n);
        for (long p = 0; p < n; ++p) {
            Real st[5], sm, sp, Ip[3], Im[3];
            for (int m = 0; m < 5; ++m) st[m] = s[m * n + p];
            ppm_reconstruct(st, fl[p], sm, sp);
            ppm_int_profile(sm, sp, st[2], u[p], c[p], dtdx, Ip, Im);
            o[0 * n + p] = sm; o[1 * n + p] = sp;
            for (int w = 0; w < 3; ++w) { o[(2 + w) * n + p] = Ip[w]; o[(5 + w) * n + p] = Im[w]; }
        }
        out["ppm.out"] = o;
    }

    // ---- Castro::uflatten (flatten.cpp:12-166) on lines that vary along x only ----
    if (in.count("flat.p")) {
        const Arr &pv = in["flat.p"], &uv = in["flat.u"];
        const long n = (long)pv.size() / 7;
        Arr o(n);
        for (long p = 0; p < n; ++p) {
            const int lo[3] = {-3, -3, -3}, hi[3] = {3, 3, 3};
            Arr q((size_t)NQ * 343, 0.0), fo(343, 0.0);
            Array4<Real> qa(q.data(), lo, hi, NQ), fa(fo.data(), lo, hi, 1);
            for (int k = -3; k <= 3; ++k) for (int j = -3; j <= 3; ++j) for (int i = -3; i <= 3; ++i) {
                qa(i, j, k, QPRES) = pv[(i + 3) * n + p];
                qa(i, j, k, QU) = (i >= -2 && i <= 2) ? uv[(i + 2) * n + p] : 0.0;
                qa(i, j, k, QRHO) = 1.0;
            }
            const int z[3] = {0, 0, 0};
            castro_obj.uflatten(Box(z, z), Array4<Real const>(qa), fa, QPRES);
            o[p] = fa(0, 0, 0);
        }
        out["flat.out"] = o;
    }

    // ---- Castro::cmpflx_plus_godunov (riemann.cpp:15-206), lines of faces along idir ----
    for (int cfg = 0; cfg < 64; ++cfg) {
        const std::string P = "cmpflx" + std::to_string(cfg) + ".";
        if (!in.count(P + "qm")) continue;
        set_params(P);
        const int idir = (int)scalar(P + "idir");
        const Arr &qm = in[P + "qm"], &qp = in[P + "qp"], &cz = in[P + "c"], &shk = in[P + "shk"];
        const long n = (long)qm.size() / 7;                 // faces 1..n, zones 0..n
        int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
        hi[idir] = (int)n;
        const long nz = n + 1;
        Arr QM((size_t)NQ * nz, 0.0), QP((size_t)NQ * nz, 0.0), AUX((size_t)NQAUX * nz), SHK(nz), FLX((size_t)NUM_STATE * nz, 0.0), GD((size_t)NGDNV * nz, 0.0);
        const int comp[7] = {QRHO, QU, QV, QW, QPRES, QREINT, QFS};
        for (long f = 1; f <= n; ++f)
            for (int m = 0; m < 7; ++m) { QM[comp[m] * nz + f] = qm[m * n + (f - 1)]; QP[comp[m] * nz + f] = qp[m * n + (f - 1)]; }
        for (long z = 0; z < nz; ++z) { AUX[QGAMC * nz + z] = stub_eos::gamma; AUX[QC * nz + z] = cz[z]; SHK[z] = shk[z]; }
        // a SlipWall on the first face when asked for (bnd_fac = 0 there)
        Castro::phys_bc.l[idir] = (int)scalar(P + "wall") ? SlipWall : Outflow;
        Castro::geom.d.domain.lo_[idir] = (int)scalar(P + "wall") ? 1 : -1000;
        int flo[3] = {0, 0, 0}, fhi[3] = {0, 0, 0};
        flo[idir] = 1; fhi[idir] = (int)n;
        castro_obj.cmpflx_plus_godunov(Box(flo, fhi), Array4<Real>(QM.data(), lo, hi, NQ), Array4<Real>(QP.data(), lo, hi, NQ),
                                       Array4<Real>(FLX.data(), lo, hi, NUM_STATE), Array4<Real>(GD.data(), lo, hi, NGDNV),
                                       Array4<Real const>(AUX.data(), lo, hi, NQAUX), Array4<Real const>(SHK.data(), lo, hi, 1),
                                       idir, false);
        Castro::phys_bc.l[idir] = Outflow;
        Castro::geom.d.domain.lo_[idir] = -1000;
        const int it = (idir == 0) ? 1 : 0, itt = (idir == 2) ? 1 : 2;
        Arr o(11 * n);
        for (long f = 1; f <= n; ++f) {
            const long p = f - 1;
            o[0 * n + p] = FLX[URHO * nz + f]; o[1 * n + p] = FLX[(UMX + idir) * nz + f]; o[2 * n + p] = FLX[(UMX + it) * nz + f];
            o[3 * n + p] = FLX[(UMX + itt) * nz + f]; o[4 * n + p] = FLX[UEDEN * nz + f]; o[5 * n + p] = FLX[UEINT * nz + f];
            o[6 * n + p] = FLX[UFS * nz + f]; o[7 * n + p] = GD[(GDU + idir) * nz + f]; o[8 * n + p] = GD[(GDU + it) * nz + f];
            o[9 * n + p] = GD[(GDU + itt) * nz + f]; o[10 * n + p] = GD[GDPRES * nz + f];
        }
        out[P + "out"] = o;
    }

    // ---- Castro::actual_trans_single (trans.cpp:66-437), plus states (d = 0) of a line of zones along idir_t ----
    for (int cfg = 0; cfg < 16; ++cfg) {
        const std::string P = "trans1_" + std::to_string(cfg) + ".";
        if (!in.count(P + "q")) continue;
        set_params(P);
        const int T = (int)scalar(P + "idir_t"), N = (int)scalar(P + "idir_n");
        const Arr &q = in[P + "q"], &fx = in[P + "flux"], &qt = in[P + "qt"];
        const long n = (long)q.size() / 7;                  // zones 0..n-1, T-faces 0..n
        int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}, zhi[3] = {0, 0, 0};
        hi[T] = (int)n; zhi[T] = (int)n - 1;
        const long nf = n + 1;
        Arr Q((size_t)NQ * nf, 0.0), QO((size_t)NQ * nf, 0.0), AUX((size_t)NQAUX * nf, stub_eos::gamma), F((size_t)NUM_STATE * nf, 0.0), G((size_t)NGDNV * nf, 0.0);
        const int comp[7] = {QRHO, QU, QV, QW, QPRES, QREINT, QFS};
        for (long z = 0; z < n; ++z) for (int m = 0; m < 7; ++m) Q[comp[m] * nf + z] = q[m * n + z];
        // flux record (rho, mx, my, mz, E, X, Godunov un, Godunov p) + the (rho e) flux as a 9th row
        const int fcomp[6] = {URHO, UMX, UMY, UMZ, UEDEN, UFS};
        for (long f = 0; f < nf; ++f) {
            for (int m = 0; m < 6; ++m) F[fcomp[m] * nf + f] = fx[m * nf + f];
            F[UEINT * nf + f] = fx[8 * nf + f];
            G[(GDU + T) * nf + f] = fx[6 * nf + f];
            G[GDPRES * nf + f] = fx[7 * nf + f];
        }
        (void)qt;
        castro_obj.actual_trans_single(Box(lo, zhi), T, N, 0, Array4<Real const>(Q.data(), lo, hi, NQ), Array4<Real>(QO.data(), lo, hi, NQ),
                                       Array4<Real const>(AUX.data(), lo, hi, NQAUX), Array4<Real const>(F.data(), lo, hi, NUM_STATE),
                                       Array4<Real const>(G.data(), lo, hi, NGDNV), 0.0, scalar(P + "cdtdx"));
        // Castro_ctu_hydro.cpp:742-744 etc.: every transverse correction is followed by reset_edge_state_thermo
        castro_obj.reset_edge_state_thermo(Box(lo, zhi), Array4<Real>(QO.data(), lo, hi, NQ));
        Arr o(7 * n);
        for (long z = 0; z < n; ++z) for (int m = 0; m < 7; ++m) o[m * n + z] = QO[comp[m] * nf + z];
        out[P + "out"] = o;
    }

    // ---- Castro::actual_trans_final (trans.cpp:498-862), plus states of a line of zones along idir_t1 ----
    for (int cfg = 0; cfg < 16; ++cfg) {
        const std::string P = "trans2_" + std::to_string(cfg) + ".";
        if (!in.count(P + "q")) continue;
        set_params(P);
        const int N = (int)scalar(P + "idir_n"), T1 = (int)scalar(P + "idir_t1"), T2 = (int)scalar(P + "idir_t2");
        const Arr &q = in[P + "q"], &f1 = in[P + "flux1"], &f2l = in[P + "flux2l"], &f2r = in[P + "flux2r"];
        const long n = (long)q.size() / 7;
        int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}, zhi[3] = {0, 0, 0};
        hi[T1] = (int)n; hi[T2] = 1; zhi[T1] = (int)n - 1;
        const long nf = n + 1, tot = nf * 2;                // (T1 index 0..n) x (T2 index 0..1), N index 0
        Arr Q((size_t)NQ * tot, 0.0), QO((size_t)NQ * tot, 0.0), AUX((size_t)NQAUX * tot, stub_eos::gamma);
        Arr F1((size_t)NUM_STATE * tot, 0.0), G1((size_t)NGDNV * tot, 0.0), F2((size_t)NUM_STATE * tot, 0.0), G2((size_t)NGDNV * tot, 0.0);
        Array4<Real> Qa(Q.data(), lo, hi, NQ), F1a(F1.data(), lo, hi, NUM_STATE), G1a(G1.data(), lo, hi, NGDNV),
                     F2a(F2.data(), lo, hi, NUM_STATE), G2a(G2.data(), lo, hi, NGDNV);
        const int comp[7] = {QRHO, QU, QV, QW, QPRES, QREINT, QFS};
        const int fcomp[6] = {URHO, UMX, UMY, UMZ, UEDEN, UFS};
        auto at = [&](long a, int b, int* idx) { idx[0] = idx[1] = idx[2] = 0; idx[T1] = (int)a; idx[T2] = b; };
        int ix[3];
        for (long z = 0; z < n; ++z) { at(z, 0, ix); for (int m = 0; m < 7; ++m) Qa(ix[0], ix[1], ix[2], comp[m]) = q[m * n + z]; }
        for (long f = 0; f < nf; ++f) {
            at(f, 0, ix);
            for (int m = 0; m < 6; ++m) F1a(ix[0], ix[1], ix[2], fcomp[m]) = f1[m * nf + f];
            F1a(ix[0], ix[1], ix[2], UEINT) = f1[8 * nf + f];
            G1a(ix[0], ix[1], ix[2], GDU + T1) = f1[6 * nf + f];
            G1a(ix[0], ix[1], ix[2], GDPRES) = f1[7 * nf + f];
        }
        for (long z = 0; z < n; ++z)
            for (int side = 0; side < 2; ++side) {
                const Arr& src = side ? f2r : f2l;
                at(z, side, ix);
                for (int m = 0; m < 6; ++m) F2a(ix[0], ix[1], ix[2], fcomp[m]) = src[m * n + z];
                F2a(ix[0], ix[1], ix[2], UEINT) = src[8 * n + z];
                G2a(ix[0], ix[1], ix[2], GDU + T2) = src[6 * n + z];
                G2a(ix[0], ix[1], ix[2], GDPRES) = src[7 * n + z];
            }
        castro_obj.actual_trans_final(Box(lo, zhi), N, T1, T2, 0, Array4<Real const>(Qa), Array4<Real>(QO.data(), lo, hi, NQ),
                                      Array4<Real const>(AUX.data(), lo, hi, NQAUX), Array4<Real const>(F1a), Array4<Real const>(F2a),
                                      Array4<Real const>(G1a), Array4<Real const>(G2a), scalar(P + "cdtdx1"), scalar(P + "cdtdx2"));
        castro_obj.reset_edge_state_thermo(Box(lo, zhi), Array4<Real>(QO.data(), lo, hi, NQ));
        Array4<Real> QOa(QO.data(), lo, hi, NQ);
        Arr o(7 * n);
        for (long z = 0; z < n; ++z) { at(z, 0, ix); for (int m = 0; m < 7; ++m) o[m * n + z] = QOa(ix[0], ix[1], ix[2], comp[m]); }
        out[P + "out"] = o;
    }

    // ---- a block through Castro::ctoprim -> uflatten -> trace_ppm x, y, z (advection_util.cpp:26-200, flatten.cpp,
    //      trace_ppm.cpp:15-594) ----
    if (in.count("block.U")) {
        set_params("block.");
        const int nb = (int)scalar("block.n");              // valid zones per side; U on the box grown by 4
        const double dt = scalar("block.dt");
        for (int d = 0; d < 3; ++d) Castro::geom.d.dx[d] = in["block.dx"][d];
        const int lo[3] = {-4, -4, -4}, hi[3] = {nb + 3, nb + 3, nb + 3};
        const long nt = (long)(nb + 8) * (nb + 8) * (nb + 8);
        Arr U = in["block.U"], Q((size_t)NQ * nt, 0.0), AUX((size_t)NQAUX * nt, 0.0), FL(nt, 0.0), SRC((size_t)NQSRC * nt, 0.0);
        castro_obj.ctoprim(Box(lo, hi), 0.0, Array4<Real const>(U.data(), lo, hi, NUM_STATE), Array4<Real>(Q.data(), lo, hi, NQ),
                           Array4<Real>(AUX.data(), lo, hi, NQAUX));
        const int l1[3] = {-1, -1, -1}, h1[3] = {nb, nb, nb}, vlo[3] = {0, 0, 0}, vhi[3] = {nb - 1, nb - 1, nb - 1};
        castro_obj.uflatten(Box(l1, h1), Array4<Real const>(Q.data(), lo, hi, NQ), Array4<Real>(FL.data(), lo, hi, 1), QPRES);
        out["block.q"] = Q; out["block.qaux"] = AUX; out["block.flatn"] = FL;
        for (int idir = 0; idir < 3; ++idir) {
            Arr QM((size_t)NQ * nt, 0.0), QP((size_t)NQ * nt, 0.0);
            castro_obj.trace_ppm(Box(l1, h1), idir, Array4<Real const>(Q.data(), lo, hi, NQ), Array4<Real const>(AUX.data(), lo, hi, NQAUX),
                                 Array4<Real const>(SRC.data(), lo, hi, NQSRC), Array4<Real const>(FL.data(), lo, hi, 1),
                                 Array4<Real>(QM.data(), lo, hi, NQ), Array4<Real>(QP.data(), lo, hi, NQ), Box(vlo, vhi), dt);
            out["block.qm" + std::to_string(idir)] = QM;
            out["block.qp" + std::to_string(idir)] = QP;
        }
    }

    // ---- one whole tile of Castro::construct_ctu_hydro_source: the reference's own member functions called in the order
    //      and on the boxes of the MFIter body (Castro_ctu_hydro.cpp:130-1480, 3-D, no radiation); the driver below is a
    //      restatement of that orchestration, every function it calls is the reference's ----
    for (int cfg = 0; cfg < 32; ++cfg) {
        const std::string P = "hydro" + std::to_string(cfg) + ".";
        if (!in.count(P + "U")) continue;
        set_params(P);
        const int nb = (int)scalar(P + "n");
        const double dt = scalar(P + "dt");
        for (int d = 0; d < 3; ++d) {
            Castro::geom.d.dx[d] = in[P + "dx"][d];
            const bool wall = in.count(P + "wall_lo") && scalar(P + "wall_lo") != 0.0;
            Castro::phys_bc.l[d] = wall ? SlipWall : Outflow;
            Castro::geom.d.domain.lo_[d] = wall ? 0 : -1000;
        }
        const double* dx = Castro::geom.d.dx;
        const int vlo[3] = {0, 0, 0}, vhi[3] = {nb - 1, nb - 1, nb - 1};
        const Box bx(vlo, vhi), obx = grown(bx, 1, 1, 1), qbx = grown(bx, 4, 4, 4), qbx3 = grown(bx, 3, 3, 3);
        const int* lo = qbx.lo_; const int* hi = qbx.hi_;                 // every array lives on grow(bx, 4)
        const long nt = (long)(nb + 8) * (nb + 8) * (nb + 8);
        auto A = [&](Arr& a, int nc) { return Array4<Real>(a.data(), lo, hi, nc); };
        auto CA = [&](Arr& a, int nc) { return Array4<Real const>(a.data(), lo, hi, nc); };
        Arr U = in[P + "U"], Unew = U, Q((size_t)NQ * nt, 0.0), AUX((size_t)NQAUX * nt, 0.0), FLT(nt, 0.0), SHK(nt, 0.0), DIV(nt, 0.0);
        Arr SRC((size_t)NSRC * nt, 0.0), CORR((size_t)NSRC * nt, 0.0), SRCQ((size_t)NQSRC * nt, 0.0);
        if (in.count(P + "src")) SRC = in[P + "src"];
        if (in.count(P + "corr")) CORR = in[P + "corr"];
        Arr E[3][2];                                                      // traced states: [dir][minus, plus]
        for (auto& d : E) for (auto& a : d) a.assign((size_t)NQ * nt, 0.0);
        Arr T[3][3][2];                                                   // [normal][transverse] after trans_single
        for (int n = 0; n < 3; ++n) for (int t = 0; t < 3; ++t) if (n != t) for (auto& a : T[n][t]) a.assign((size_t)NQ * nt, 0.0);
        Arr QL((size_t)NQ * nt, 0.0), QR((size_t)NQ * nt, 0.0), F1((size_t)NUM_STATE * nt, 0.0), F2((size_t)NUM_STATE * nt, 0.0),
            G1((size_t)NGDNV * nt, 0.0), G2((size_t)NGDNV * nt, 0.0);
        Arr FX[3], QE[3], AREA[3], VOL(nt, dx[0] * dx[1] * dx[2]);
        for (int d = 0; d < 3; ++d) {
            FX[d].assign((size_t)NUM_STATE * nt, 0.0); QE[d].assign((size_t)NGDNV * nt, 0.0);
            AREA[d].assign(nt, d == 0 ? dx[1] * dx[2] : d == 1 ? dx[0] * dx[2] : dx[0] * dx[1]);
        }

        castro_obj.ctoprim(qbx, 0.0, CA(U, NUM_STATE), A(Q, NQ), A(AUX, NQAUX));
        if (castro::first_order_hydro == 1) { /* FLT stays 0 */ }
        else if (castro::use_flattening == 1) castro_obj.uflatten(obx, CA(Q, NQ), A(FLT, 1), QPRES);
        else FLT.assign(nt, 1.0);
        if (castro::hybrid_riemann == 1) castro_obj.shock(obx, CA(Q, NQ), A(SHK, 1));
        castro_obj.src_to_prim(qbx3, dt, CA(Q, NQ), CA(SRC, NSRC), CA(CORR, NSRC), A(SRCQ, NQSRC));
        if (castro::ppm_type == 0)
            castro_obj.ctu_plm_states(obx, bx, CA(Q, NQ), CA(FLT, 1), CA(AUX, NQAUX), CA(SRCQ, NQSRC), A(E[0][0], NQ), A(E[0][1], NQ),
                                      A(E[1][0], NQ), A(E[1][1], NQ), A(E[2][0], NQ), A(E[2][1], NQ), dt);
        else
            castro_obj.ctu_ppm_states(obx, bx, CA(Q, NQ), CA(FLT, 1), CA(AUX, NQAUX), CA(SRCQ, NQSRC), A(E[0][0], NQ), A(E[0][1], NQ),
                                      A(E[1][0], NQ), A(E[1][1], NQ), A(E[2][0], NQ), A(E[2][1], NQ), dt);
        castro_obj.divu(obx, CA(Q, NQ), A(DIV, 1));

        const Box fb[3] = {nodal(bx, 0), nodal(bx, 1), nodal(bx, 2)};
        const double hdt = 0.5 * dt, cdt[3] = {dt / dx[0] / 3.0, dt / dx[1] / 3.0, dt / dx[2] / 3.0};
        const double hdtd[3] = {0.5 * dt / dx[0], 0.5 * dt / dx[1], 0.5 * dt / dx[2]};
        auto riemann = [&](const Box& b, Arr& qm, Arr& qp, Arr& f, Arr& g, int idir) {
            castro_obj.cmpflx_plus_godunov(b, A(qm, NQ), A(qp, NQ), A(f, NUM_STATE), A(g, NGDNV), CA(AUX, NQAUX), CA(SHK, 1), idir, false);
        };
        // first solves, each followed by the two transverse corrections that use its flux (x, y, z)
        for (int t = 0; t < 3; ++t) {
            int gr[3] = {1, 1, 1};
            gr[t] = 0;
            riemann(grown(fb[t], gr[0], gr[1], gr[2]), E[t][0], E[t][1], F1, G1, t);
            for (int n = 0; n < 3; ++n) {
                if (n == t) continue;
                int g2[3] = {1, 1, 1};                      // faces of n, not grown along n nor along t
                g2[n] = 0; g2[t] = 0;
                const Box tb = grown(fb[n], g2[0], g2[1], g2[2]);
                castro_obj.trans_single(tb, t, n, CA(E[n][0], NQ), A(T[n][t][0], NQ), CA(E[n][1], NQ), A(T[n][t][1], NQ), CA(AUX, NQAUX),
                                        CA(F1, NUM_STATE), CA(G1, NGDNV), hdt, cdt[t]);
                castro_obj.reset_edge_state_thermo(tb, A(T[n][t][0], NQ));
                castro_obj.reset_edge_state_thermo(tb, A(T[n][t][1], NQ));
            }
        }
        // final stage per normal direction: the two solves on the singly corrected transverse states, trans_final, solve
        for (int n = 0; n < 3; ++n) {
            const int t1 = (n == 0) ? 1 : 0, t2 = (n == 2) ? 1 : 2;
            int g1[3] = {0, 0, 0}, g2[3] = {0, 0, 0};
            g1[n] = 1; g2[n] = 1;                           // faces of t1 (t2) grown along n only
            if (n == 0) {
                riemann(grown(fb[1], g1[0], g1[1], g1[2]), T[1][2][0], T[1][2][1], F1, G1, 1);      // F^{y|z}
                riemann(grown(fb[2], g2[0], g2[1], g2[2]), T[2][1][0], T[2][1][1], F2, G2, 2);      // F^{z|y}
                castro_obj.trans_final(fb[0], 0, 1, 2, CA(E[0][0], NQ), A(QL, NQ), CA(E[0][1], NQ), A(QR, NQ), CA(AUX, NQAUX),
                                       CA(F1, NUM_STATE), CA(F2, NUM_STATE), CA(G1, NGDNV), CA(G2, NGDNV), hdtd[1], hdtd[2]);
            } else if (n == 1) {
                riemann(grown(fb[2], g2[0], g2[1], g2[2]), T[2][0][0], T[2][0][1], F1, G1, 2);      // F^{z|x}
                riemann(grown(fb[0], g1[0], g1[1], g1[2]), T[0][2][0], T[0][2][1], F2, G2, 0);      // F^{x|z}
                castro_obj.trans_final(fb[1], 1, 0, 2, CA(E[1][0], NQ), A(QL, NQ), CA(E[1][1], NQ), A(QR, NQ), CA(AUX, NQAUX),
                                       CA(F2, NUM_STATE), CA(F1, NUM_STATE), CA(G2, NGDNV), CA(G1, NGDNV), hdtd[0], hdtd[2]);
            } else {
                riemann(grown(fb[0], g1[0], g1[1], g1[2]), T[0][1][0], T[0][1][1], F1, G1, 0);      // F^{x|y}
                riemann(grown(fb[1], g2[0], g2[1], g2[2]), T[1][0][0], T[1][0][1], F2, G2, 1);      // F^{y|x}
                castro_obj.trans_final(fb[2], 2, 0, 1, CA(E[2][0], NQ), A(QL, NQ), CA(E[2][1], NQ), A(QR, NQ), CA(AUX, NQAUX),
                                       CA(F1, NUM_STATE), CA(F2, NUM_STATE), CA(G1, NGDNV), CA(G2, NGDNV), hdtd[0], hdtd[1]);
            }
            castro_obj.reset_edge_state_thermo(fb[n], A(QL, NQ));
            castro_obj.reset_edge_state_thermo(fb[n], A(QR, NQ));
            riemann(fb[n], QL, QR, FX[n], QE[n], n);
        }
        for (int d = 0; d < 3; ++d) {
            Array4<Real> f = A(FX[d], NUM_STATE);
            for (int k = fb[d].lo_[2]; k <= fb[d].hi_[2]; ++k) for (int j = fb[d].lo_[1]; j <= fb[d].hi_[1]; ++j)
                for (int i = fb[d].lo_[0]; i <= fb[d].hi_[0]; ++i) f(i, j, k, UTEMP) = 0.0;
            castro_obj.apply_av(fb[d], d, CA(DIV, 1), CA(U, NUM_STATE), f);
            if (castro::limit_fluxes_on_small_dens == 1)
                castro_obj.limit_hydro_fluxes_on_small_dens(fb[d], d, CA(U, NUM_STATE), CA(Q, NQ), CA(VOL, 1), f, CA(AREA[d], 1), dt);
            if (castro::limit_fluxes_on_large_vel == 1)
                castro_obj.limit_hydro_fluxes_on_large_vel(fb[d], d, CA(U, NUM_STATE), CA(Q, NQ), CA(VOL, 1), f, CA(AREA[d], 1), dt);
            castro_obj.normalize_species_fluxes(fb[d], f);
        }
        castro_obj.consup_hydro(bx, A(Unew, NUM_STATE), A(FX[0], NUM_STATE), CA(QE[0], NGDNV), A(FX[1], NUM_STATE), CA(QE[1], NGDNV),
                                A(FX[2], NUM_STATE), CA(QE[2], NGDNV), dt);
        for (int d = 0; d < 3; ++d) castro_obj.scale_flux(fb[d], A(FX[d], NUM_STATE), CA(AREA[d], 1), dt);
        out[P + "unew"] = Unew;
        for (int d = 0; d < 3; ++d) { out[P + "flux" + std::to_string(d)] = FX[d]; out[P + "qe" + std::to_string(d)] = QE[d]; }
        out[P + "div"] = DIV; out[P + "shk"] = SHK; out[P + "srcq"] = SRCQ;
        for (int d = 0; d < 3; ++d) { Castro::phys_bc.l[d] = Outflow; Castro::geom.d.domain.lo_[d] = -1000; }
    }

    write_blob(argv[2]);
    return 0;
}
