// probe_derive.cpp -- the reference's derived-field functions (Source/driver/Derive.cpp, compiled UNMODIFIED and IN PLACE
// from /root/reference against the stand-in headers) on one box of seeded state data, and Castro::estdt_cfl
// (Source/driver/timestep.cpp) on the same state.  STUB-COMPILED, NOT oracle/_ref.
// Every function gets the state components its registration in Castro_setup.cpp hands it (the whole State_Type, or
// (rho, momenta), (rho, one momentum), (momenta), (rho, rho X), (rho, Temp, rho X)).
#include <Castro.H>
#include <Derive.H>
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

using DerFn = void (*)(const Box&, FArrayBox&, int, int, const FArrayBox&, const Geometry&, Real, const int*, int);

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    read_blob(argv[1]);
    int n[3];
    for (int d = 0; d < 3; ++d) {
        n[d] = (int)in["derive.n"][d];
        Castro::geom.d.prob_lo[d] = in["derive.problo"][d];
        Castro::geom.d.dx[d] = in["derive.dx"][d];
        Castro::geom.d.prob_hi[d] = in["derive.problo"][d] + n[d] * in["derive.dx"][d];
        Castro::geom.d.domain.lo_[d] = 0; Castro::geom.d.domain.hi_[d] = n[d] - 1;
        problem::center[d] = in["derive.center"][d];
    }
    const int glo[3] = {-1, -1, -1}, ghi[3] = {n[0], n[1], n[2]}, lo[3] = {0, 0, 0}, hi[3] = {n[0] - 1, n[1] - 1, n[2] - 1};
    const Box gbx(glo, ghi), bx(lo, hi);
    const long ng = (long)(n[0] + 2) * (n[1] + 2) * (n[2] + 2);
    Arr U = in["derive.U"];                                  // NUM_STATE components on the box grown by 1
    auto pick = [&](std::initializer_list<int> comps) {
        Arr a;
        for (int c : comps) a.insert(a.end(), U.begin() + (size_t)c * ng, U.begin() + (size_t)(c + 1) * ng);
        return a;
    };
    struct Item { const char* name; DerFn fn; std::initializer_list<int> comps; int nout; };
    const Item items[] = {
        {"pressure", ca_derpres, {0, 1, 2, 3, 4, 5, 6, 7}, 1}, {"eint_E", ca_dereint1, {0, 1, 2, 3, 4, 5, 6, 7}, 1},
        {"eint_e", ca_dereint2, {0, 1, 2, 3, 4, 5, 6, 7}, 1}, {"logden", ca_derlogden, {URHO}, 1},
        {"soundspeed", ca_dersoundspeed, {0, 1, 2, 3, 4, 5, 6, 7}, 1}, {"Gamma_1", ca_dergamma1, {0, 1, 2, 3, 4, 5, 6, 7}, 1},
        {"MachNumber", ca_dermachnumber, {0, 1, 2, 3, 4, 5, 6, 7}, 1},
        {"x_velocity", ca_dervel, {URHO, UMX}, 1}, {"y_velocity", ca_dervel, {URHO, UMY}, 1}, {"z_velocity", ca_dervel, {URHO, UMZ}, 1},
        {"magvel", ca_dermagvel, {URHO, UMX, UMY, UMZ}, 1}, {"radvel", ca_derradialvel, {URHO, UMX, UMY, UMZ}, 1},
        {"circvel", ca_dercircvel, {URHO, UMX, UMY, UMZ}, 1}, {"magmom", ca_dermagmom, {UMX, UMY, UMZ}, 1},
        {"angular_momentum_x", ca_derangmomx, {URHO, UMX, UMY, UMZ}, 1}, {"angular_momentum_y", ca_derangmomy, {URHO, UMX, UMY, UMZ}, 1},
        {"angular_momentum_z", ca_derangmomz, {URHO, UMX, UMY, UMZ}, 1}, {"kineng", ca_derkineng, {URHO, UMX, UMY, UMZ}, 1},
        {"X(X)", ca_derspec, {URHO, UFS}, 1}, {"abar", ca_derabar, {URHO, UFS}, 1},
        {"magvort", ca_dermagvort, {URHO, UMX, UMY, UMZ}, 1}, {"divu", ca_derdivu, {URHO, UMX, UMY, UMZ}, 1},
        {"StateErr", ca_derstate, {URHO, UTEMP, UFS}, 3},
    };
    for (const Item& it : items) {
        Arr dat = pick(it.comps), der((size_t)it.nout * ng, 0.0);
        FArrayBox datfab(dat.data(), gbx, (int)it.comps.size()), derfab(der.data(), gbx, it.nout);
        it.fn(bx, derfab, 0, it.nout, datfab, Castro::geom, 0.0, nullptr, 0);
        out[std::string("derive.") + it.name] = der;
    }
    // ---- Castro::estdt_cfl (Source/driver/timestep.cpp:22-137, compiled unmodified) over the valid zones ----
    {
        const long nv = (long)n[0] * n[1] * n[2];
        Arr V((size_t)NUM_STATE * nv);
        Array4<Real> g(U.data(), glo, ghi, NUM_STATE), v(V.data(), lo, hi, NUM_STATE);
        for (int c = 0; c < NUM_STATE; ++c)
            for (int k = 0; k < n[2]; ++k) for (int j = 0; j < n[1]; ++j) for (int i = 0; i < n[0]; ++i) v(i, j, k, c) = g(i, j, k, c);
        MultiFab mf(V.data(), bx, NUM_STATE);
        Castro castro_obj;
        castro_obj.state_mf = &mf;
        out["derive.estdt"] = {castro_obj.estdt_cfl(0.0)};
    }
    write_blob(argv[2]);
    return 0;
}
