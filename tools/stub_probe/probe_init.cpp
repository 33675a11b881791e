// probe_init.cpp -- the reference's problem initialisers (Exec/hydro_tests/{Sedov,Sod}/problem_initialize.H and
// problem_initialize_state_data.H, included UNMODIFIED and IN PLACE from /root/reference) against the stand-in headers.
// STUB-COMPILED, NOT oracle/_ref (see README.md).  Both problems define the same two function names, so each pair of
// headers is included inside its own namespace.
#include <Castro.H>
#include <prob_parameters.H>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <vector>

static const Geometry& DefaultGeometry() { return Castro::geom; }

#define STR2(x) #x
#define STR(x) STR2(x)
namespace sedov_ref {
#include STR(REFERENCE_EXEC/Sedov/problem_initialize.H)
#include STR(REFERENCE_EXEC/Sedov/problem_initialize_state_data.H)
}
#undef problem_initialize_H
#undef problem_initialize_state_data_H
namespace sod_ref {
#include STR(REFERENCE_EXEC/Sod/problem_initialize.H)
#include STR(REFERENCE_EXEC/Sod/problem_initialize_state_data.H)
}

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

static void set_geom(const std::string& P, int n[3])
{
    for (int d = 0; d < 3; ++d) {
        n[d] = (int)in[P + "n"][d];
        Castro::geom.d.prob_lo[d] = in[P + "problo"][d];
        Castro::geom.d.prob_hi[d] = in[P + "probhi"][d];
        Castro::geom.d.dx[d] = (in[P + "probhi"][d] - in[P + "problo"][d]) / n[d];
        Castro::geom.d.domain.lo_[d] = 0; Castro::geom.d.domain.hi_[d] = n[d] - 1;
    }
}

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    read_blob(argv[1]);
    for (int cfg = 0; cfg < 16; ++cfg) {
        const std::string P = "sedov" + std::to_string(cfg) + ".";
        if (!in.count(P + "n")) continue;
        int n[3];
        set_geom(P, n);
        problem::r_init = in[P + "r_init"][0]; problem::p_ambient = in[P + "p_ambient"][0]; problem::exp_energy = in[P + "exp_energy"][0];
        problem::dens_ambient = in[P + "dens_ambient"][0]; problem::nsub = (int)in[P + "nsub"][0]; problem::temp_ambient = -1.e2;
        sedov_ref::problem_initialize();
        const int lo[3] = {0, 0, 0}, hi[3] = {n[0] - 1, n[1] - 1, n[2] - 1};
        Arr S((size_t)NUM_STATE * n[0] * n[1] * n[2], 0.0);
        Array4<Real> s(S.data(), lo, hi, NUM_STATE);
        const GeometryData gd = Castro::geom.data();
        for (int k = 0; k < n[2]; ++k) for (int j = 0; j < n[1]; ++j) for (int i = 0; i < n[0]; ++i)
            sedov_ref::problem_initialize_state_data(i, j, k, s, gd);
        out[P + "state"] = S;
        out[P + "const"] = {problem::e_exp, problem::e_ambient, problem::temp_ambient, problem::center[0], problem::center[1], problem::center[2]};
    }
    for (int cfg = 0; cfg < 16; ++cfg) {
        const std::string P = "sod" + std::to_string(cfg) + ".";
        if (!in.count(P + "n")) continue;
        int n[3];
        set_geom(P, n);
        problem::rho_l = in[P + "left"][0]; problem::u_l = in[P + "left"][1]; problem::p_l = in[P + "left"][2];
        problem::rho_r = in[P + "right"][0]; problem::u_r = in[P + "right"][1]; problem::p_r = in[P + "right"][2];
        problem::idir = (int)in[P + "idir"][0]; problem::frac = in[P + "frac"][0]; problem::use_Tinit = 0;
        sod_ref::problem_initialize();
        const int lo[3] = {0, 0, 0}, hi[3] = {n[0] - 1, n[1] - 1, n[2] - 1};
        Arr S((size_t)NUM_STATE * n[0] * n[1] * n[2], 0.0);
        Array4<Real> s(S.data(), lo, hi, NUM_STATE);
        const GeometryData gd = Castro::geom.data();
        for (int k = 0; k < n[2]; ++k) for (int j = 0; j < n[1]; ++j) for (int i = 0; i < n[0]; ++i)
            sod_ref::problem_initialize_state_data(i, j, k, s, gd);
        out[P + "state"] = S;
    }
    write_blob(argv[2]);
    return 0;
}
