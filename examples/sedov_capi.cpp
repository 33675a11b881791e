// sedov_capi.cpp -- the Sedov 3-D problem driven from plain C++ through the C ABI only
// (include/castro_hydro_amd.h + hipMalloc): what a compiled host such as Castro's own driver does, without
// PyTorch and without AMReX.  Single level, single box, outflow boundaries.
//
//   sedov_capi <n> <nsteps> [state.bin]
//
// The time-step loop is Castro::advance for max_level = 0 (Source/driver/Castro_advance.cpp:19-121,
// Castro_advance_ctu.cpp:15-397,507-768 with use_retry = 1 and no rejected step) + computeNewDt
// (Source/driver/Castro.cpp:1629-1866).  Prints cell-updates/s and, if asked, dumps S_new (NUM_STATE, nz, ny, nx)
// so that tests/test_gpu_parity.py can compare it bit for bit with the Python driver.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/castro_hydro_amd.h"

#define CK(call) do { int rc_ = (call); if (rc_ != 0) { std::fprintf(stderr, "%s failed: %d\n", #call, rc_); return 2; } } while (0)
#define HK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 2; } } while (0)

static castro_amd_fab make_fab(double* p, const int lo[3], const int hi[3], int nc)
{
    castro_amd_fab f;
    f.p = p;
    for (int d = 0; d < 3; ++d) { f.lo[d] = lo[d]; f.hi[d] = hi[d]; }
    f.ncomp = nc;
    return f;
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? std::atoi(argv[1]) : 64;
    const int nsteps = argc > 2 ? std::atoi(argv[2]) : 10;
    const char* dump = argc > 3 ? argv[3] : nullptr;
    const double stop_time = 0.01;
    const int NS = CASTRO_AMD_NUM_STATE, NG = CASTRO_AMD_NUM_GROW;

    castro_amd_ctx* ctx = nullptr;
    CK(castro_amd_ctx_create(&ctx, 0));
    CK(castro_amd_ctx_reserve(ctx, n, n, n));

    castro_amd_params P;
    castro_amd_default_params(&P);
    castro_amd_geom G;
    for (int d = 0; d < 3; ++d) {
        G.dx[d] = 1.0 / n; G.problo[d] = 0.0; G.probhi[d] = 1.0; G.domlo[d] = 0; G.domhi[d] = n - 1;
        G.lo_bc[d] = 2; G.hi_bc[d] = 2;                        // Outflow (inputs.3d.sph:17-18)
    }
    G.coord = 0;

    const int lo[3] = { 0, 0, 0 }, hi[3] = { n - 1, n - 1, n - 1 };
    const int glo[3] = { -NG, -NG, -NG }, ghi[3] = { n - 1 + NG, n - 1 + NG, n - 1 + NG };
    const size_t ng = (size_t)(n + 2 * NG) * (n + 2 * NG) * (n + 2 * NG);
    double *A = nullptr, *B = nullptr, *d_red = nullptr;
    HK(hipMalloc(&A, ng * NS * sizeof(double)));
    HK(hipMalloc(&B, ng * NS * sizeof(double)));
    HK(hipMalloc(&d_red, 3 * sizeof(double)));
    HK(hipMemset(A, 0, ng * NS * sizeof(double)));
    HK(hipMemset(B, 0, ng * NS * sizeof(double)));
    double* fl[3]; double* mf[3];
    castro_amd_fab flux[3], mass[3], noqe[3];
    for (int d = 0; d < 3; ++d) {
        int fhi[3] = { hi[0], hi[1], hi[2] };
        fhi[d] += 1;
        const size_t nf = (size_t)(fhi[0] + 1) * (fhi[1] + 1) * (fhi[2] + 1);
        HK(hipMalloc(&fl[d], nf * NS * sizeof(double)));
        HK(hipMalloc(&mf[d], nf * sizeof(double)));
        flux[d] = make_fab(fl[d], lo, fhi, NS);
        mass[d] = make_fab(mf[d], lo, fhi, 1);
        noqe[d] = make_fab(nullptr, lo, fhi, CASTRO_AMD_NGDNV);
    }
    castro_amd_fab nosrc = make_fab(nullptr, lo, hi, 0);
    castro_amd_fab S_new = make_fab(A, glo, ghi, NS), S_old = make_fab(B, glo, ghi, NS);

    // initData + post-init clean_state (Castro.cpp:934-1160)
    CK(castro_amd_sedov_init_fab(ctx, &S_new, lo, hi, &G, &P, 0.01, 1.e-5, 1.0, 1.0, 10, nullptr));
    CK(castro_amd_clean_state_fab(ctx, &S_new, lo, hi, &P, 1, nullptr));

    auto reduce = [&](double out[2]) -> int {           // [min dx/(c+|u|), min rho] of S_new
        const double init[2] = { 1.e200, 1.e200 };
        if (hipMemcpy(d_red, init, sizeof(init), hipMemcpyHostToDevice) != hipSuccess) return 1;
        if (castro_amd_estdt_fab(ctx, &S_new, lo, hi, &G, &P, d_red, nullptr) != 0) return 1;
        return hipMemcpy(out, d_red, 2 * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess;
    };

    double time = 0.0, dt = 0.0, est[2];
    HK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    for (int step = 0; step < nsteps; ++step) {
        if (step == 0) {
            // computeInitialDt (Castro.cpp:1822-1866)
            if (reduce(est)) return 2;
            dt = P.init_shrink * std::min(1.e200, est[0] * P.cfl);
            if (time + dt > stop_time - 0.001 * dt) dt = stop_time - time;
        } else {
            // computeNewDt (Castro.cpp:1629-1819); est[0] still holds the estimate of the last advance
            double dt0 = std::min(std::min(1.e200, est[0] * P.cfl), P.change_max * dt);
            if (time + dt0 >= stop_time - 2.220446049250313e-16) dt0 = stop_time - time;
            dt = dt0;
        }
        // initialize_advance: swap the time levels; subcycle_advance_ctu hands (time + dt) - time to do_advance_ctu
        std::swap(S_new.p, S_old.p);
        const double dts = (time + dt) - time;
        // clean_state(S_old) + clean_state(Sborder) on the valid zones, then FillPatch (DESIGN.md "clean_state order")
        CK(castro_amd_clean_state_fab(ctx, &S_old, lo, hi, &P, 2, nullptr));
        CK(castro_amd_bc_fill_fab(ctx, &S_old, &G, nullptr));
        // hydro update + S_new.min(URHO) + clean_state(S_new) + estTimeStep in one pass
        const double init[3] = { 1.e200, 1.e200, 1.e200 };   // [estimate after the last clean, min density, estimate after the first]
        HK(hipMemcpyAsync(d_red, init, sizeof(init), hipMemcpyHostToDevice, nullptr));
        CK(castro_amd_ctu_hydro_clean_fab(ctx, lo, hi, lo, hi, &S_old, &nosrc, &S_new, flux, mass, noqe, &G, &P, time, dts,
                                          CASTRO_AMD_UPDATE_FROM_SBORDER | CASTRO_AMD_FLUX_ASSIGN, 1, d_red, nullptr));
        HK(hipMemcpy(est, d_red, sizeof(est), hipMemcpyDeviceToHost));
        if (est[1] < P.small_dens) { std::fprintf(stderr, "small/negative density %g: retry not handled in this example\n", est[1]); return 3; }
        if (P.change_max * std::min(1.e200, est[0] * P.cfl) < dts) { std::fprintf(stderr, "timestep validity check failed\n"); return 3; }
        time += dt;
    }
    HK(hipDeviceSynchronize());
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (castro_amd_ctx_status(ctx, nullptr) != 0) std::fprintf(stderr, "warning: device status bits set\n");
    std::printf("n=%d steps=%d time=%.17g dt=%.17g cell-updates/s=%.4g\n", n, nsteps, time, dt, (double)n * n * n * nsteps / wall);

    if (dump) {
        std::vector<double> h(ng * NS);
        HK(hipMemcpy(h.data(), S_new.p, ng * NS * sizeof(double), hipMemcpyDeviceToHost));
        std::FILE* f = std::fopen(dump, "wb");
        if (!f) return 2;
        const size_t gx = n + 2 * NG;
        for (int c = 0; c < NS; ++c)
            for (int k = 0; k < n; ++k)
                for (int j = 0; j < n; ++j)
                    std::fwrite(&h[(size_t)c * ng + ((size_t)(k + NG) * gx + (j + NG)) * gx + NG], sizeof(double), n, f);
        std::fclose(f);
    }
    for (int d = 0; d < 3; ++d) { (void)hipFree(fl[d]); (void)hipFree(mf[d]); }
    (void)hipFree(A); (void)hipFree(B); (void)hipFree(d_red);
    castro_amd_ctx_destroy(ctx);
    return 0;
}
