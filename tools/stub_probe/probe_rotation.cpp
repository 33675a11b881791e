// probe_rotation.cpp -- the reference's rotation sources (Source/rotation/rotation_sources.cpp, Rotation.cpp, Rotation.H,
// compiled UNMODIFIED and IN PLACE from /root/reference against the stand-in headers): Castro::rsrc on the old state and
// Castro::corrrsrc with the potentials of Castro::fill_rotational_potential, as construct_old/new_rotation_source call
// them (Castro_rotation.cpp:29-33, :88-100).  STUB-COMPILED, NOT oracle/_ref.
#include <Castro.H>
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

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    read_blob(argv[1]);
    Castro castro_obj;
    for (int cfg = 0; cfg < 16; ++cfg) {
        const std::string P = "rot" + std::to_string(cfg) + ".";
        if (!in.count(P + "n")) continue;
        int n[3];
        for (int d = 0; d < 3; ++d) {
            n[d] = (int)in[P + "n"][d];
            Castro::geom.d.prob_lo[d] = in[P + "problo"][d];
            Castro::geom.d.dx[d] = in[P + "dx"][d];
            Castro::geom.d.prob_hi[d] = in[P + "problo"][d] + n[d] * in[P + "dx"][d];
            Castro::geom.d.domain.lo_[d] = 0; Castro::geom.d.domain.hi_[d] = n[d] - 1;
            problem::center[d] = in[P + "center"][d];
        }
        castro::do_rotation = 1;
        castro::rotational_period = in[P + "period"][0];
        castro::rot_axis = (int)in[P + "axis"][0];
        castro::rot_source_type = (int)in[P + "rot_source_type"][0];
        castro::implicit_rotation_update = (int)in[P + "implicit"][0];
        castro::rotation_include_centrifugal = (int)in[P + "centrifugal"][0];
        castro::rotation_include_coriolis = (int)in[P + "coriolis"][0];
        const double dt = in[P + "dt"][0];
        const int glo[3] = {-1, -1, -1}, ghi[3] = {n[0], n[1], n[2]}, lo[3] = {0, 0, 0}, hi[3] = {n[0] - 1, n[1] - 1, n[2] - 1};
        const Box bx(lo, hi), gbx(glo, ghi);
        const long nz = (long)n[0] * n[1] * n[2], ng = (long)(n[0] + 2) * (n[1] + 2) * (n[2] + 2);
        Arr Uold = in[P + "uold"], Unew = in[P + "unew"];             // NUM_STATE x valid box
        Arr S1((size_t)NSRC * nz, 0.0), S2((size_t)NSRC * nz, 0.0), PHI(ng, 0.0), VOL(nz, in[P + "dx"][0] * in[P + "dx"][1] * in[P + "dx"][2]);
        castro_obj.rsrc(bx, Array4<Real const>(Uold.data(), lo, hi, NUM_STATE), Array4<Real>(S1.data(), lo, hi, NSRC), dt);
        castro_obj.fill_rotational_potential(gbx, Array4<Real>(PHI.data(), glo, ghi, 1), 0.0);
        Arr F[3];
        int fhi[3][3];
        for (int d = 0; d < 3; ++d) {
            F[d] = in[P + "mflux" + std::to_string(d)];
            for (int e = 0; e < 3; ++e) fhi[d][e] = hi[e] + (e == d ? 1 : 0);
        }
        castro_obj.corrrsrc(bx, Array4<Real const>(PHI.data(), glo, ghi, 1), Array4<Real const>(PHI.data(), glo, ghi, 1),
                            Array4<Real const>(Uold.data(), lo, hi, NUM_STATE), Array4<Real const>(Unew.data(), lo, hi, NUM_STATE),
                            Array4<Real>(S2.data(), lo, hi, NSRC), Array4<Real const>(F[0].data(), lo, fhi[0], 1),
                            Array4<Real const>(F[1].data(), lo, fhi[1], 1), Array4<Real const>(F[2].data(), lo, fhi[2], 1), dt,
                            Array4<Real const>(VOL.data(), lo, hi, 1));
        out[P + "old"] = S1; out[P + "new"] = S2; out[P + "phi"] = PHI;
    }
    write_blob(argv[2]);
    return 0;
}
