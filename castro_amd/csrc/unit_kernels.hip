// unit_kernels.hip -- pointwise entry points: the per-interface / per-zone device functions of the path
// (hydro_device.h) applied to flat lists of inputs, one list element per thread.  They mirror the reference's
// per-point functions (cmpflx_plus_godunov's body, ppm_reconstruct + ppm_int_profile, actual_trans_single /
// actual_trans_final) so that known-answer vectors recorded at that level can be replayed against the device code.
// All arrays are device pointers, component-major: a[comp * n + point].
#include <hip/hip_runtime.h>
#include "../../include/castro_hydro_amd.h"
#include "hydro_device.h"
#include "ctu_kernels.h"

namespace cad {

// riemann.cpp:62-203 for one interface of direction D
template <int D>
__device__ __forceinline__ void cmpflx_point(long n, long p, const double* qm, const double* qp, const double* cl, const double* cr,
                                             const double* bnd_fac, const int* is_shock, const DevParams& P, double* out)
{
    RState ql, qr;
    double Xl, Xr, em[NEDGE], ep[NEDGE];
#pragma unroll
    for (int m = 0; m < NEDGE; ++m) { em[m] = qm[m * n + p]; ep[m] = qp[m * n + p]; }
    ql.rho = em[PRHO]; ql.un = em[PU + RDir<D>::n]; ql.ut = em[PU + RDir<D>::t]; ql.utt = em[PU + RDir<D>::tt];
    ql.p = em[PP]; ql.rhoe = em[PRE]; ql.gamc = P.gamma; Xl = em[PX];
    qr.rho = ep[PRHO]; qr.un = ep[PU + RDir<D>::n]; qr.ut = ep[PU + RDir<D>::t]; qr.utt = ep[PU + RDir<D>::tt];
    qr.p = ep[PP]; qr.rhoe = ep[PRE]; qr.gamc = P.gamma; Xr = ep[PX];
    IFlux f;
    interface_flux<D>(ql, qr, Xl, Xr, cl[p], cr[p], bnd_fac ? bnd_fac[p] : 1.0, is_shock ? is_shock[p] != 0 : false, P, f);
    out[0 * n + p] = f.rho; out[1 * n + p] = f.mn; out[2 * n + p] = f.mt; out[3 * n + p] = f.mtt;
    out[4 * n + p] = f.E; out[5 * n + p] = f.eint; out[6 * n + p] = f.X;
    out[7 * n + p] = f.ugd; out[8 * n + p] = f.ut; out[9 * n + p] = f.utt; out[10 * n + p] = f.pgd;
}

__global__ void __launch_bounds__(256) k_cmpflx_points(long n, int idir, const double* qm, const double* qp, const double* cl,
                                                       const double* cr, const double* bnd_fac, const int* is_shock,
                                                       DevParams P, double* out)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    if (idir == 0) cmpflx_point<0>(n, p, qm, qp, cl, cr, bnd_fac, is_shock, P, out);
    else if (idir == 1) cmpflx_point<1>(n, p, qm, qp, cl, cr, bnd_fac, is_shock, P, out);
    else cmpflx_point<2>(n, p, qm, qp, cl, cr, bnd_fac, is_shock, P, out);
}

// ppm.H:54-139 + :157-252: parabola limits of a five-point stencil, then the integrals under the three waves
__global__ void __launch_bounds__(256) k_ppm_points(long n, const double* s, const double* flat, const double* u, const double* c,
                                                    double dtdx, double* out)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    double st[5], sm, sp;
#pragma unroll
    for (int m = 0; m < 5; ++m) st[m] = s[m * n + p];
    ppm_reconstruct(st, flat[p], sm, sp);
    const double s6 = 6.0 * st[2] - 3.0 * (sm + sp);
    double Ip[3], Im[3];
    ppm_int_wave(sm, sp, s6, u[p] - c[p], dtdx, Ip[0], Im[0]);
    ppm_int_wave(sm, sp, s6, u[p], dtdx, Ip[1], Im[1]);
    ppm_int_wave(sm, sp, s6, u[p] + c[p], dtdx, Ip[2], Im[2]);
    out[0 * n + p] = sm; out[1 * n + p] = sp;
#pragma unroll
    for (int w = 0; w < 3; ++w) { out[(2 + w) * n + p] = Ip[w]; out[(5 + w) * n + p] = Im[w]; }
}

// flatten.cpp:12-166 along one direction: p[-3..3], u[-2..2]
__global__ void __launch_bounds__(256) k_flatten_points(long n, const double* pv, const double* uv, double* out)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    double a[7], b[5];
#pragma unroll
    for (int m = 0; m < 7; ++m) a[m] = pv[m * n + p];
#pragma unroll
    for (int m = 0; m < 5; ++m) b[m] = uv[m * n + p];
    out[p] = flatten_1d(a, b);
}

// trans.cpp:66-437 (ntrans = 1) and :498-862 (ntrans = 2): q = (rho,u,v,w,p,rhoe,X); flux records in the scratch order
// (rho, mx, my, mz, E, X, Godunov un, Godunov p) at the high (r) and low (l) transverse face
__global__ void __launch_bounds__(256) k_trans_points(long n, int ntrans, int tdir, const double* q, const double* f1r,
                                                      const double* f1l, const double* f2r, const double* f2l,
                                                      const double* fe, double cdtdx1, double cdtdx2, DevParams P, double* out)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    double qi[NEDGE], qo[NEDGE], a[NF1], b[NF1], c[NF1], d[NF1];
#pragma unroll
    for (int m = 0; m < NEDGE; ++m) qi[m] = q[m * n + p];
#pragma unroll
    for (int m = 0; m < NF1; ++m) { a[m] = f1r[m * n + p]; b[m] = f1l[m * n + p]; }
    if (ntrans == 2) {
#pragma unroll
        for (int m = 0; m < NF1; ++m) { c[m] = f2r[m * n + p]; d[m] = f2l[m * n + p]; }
        // fe: (rho e) fluxes at the faces 1r, 1l, 2r, 2l (rows of n), read with transverse_reset_rhoe = 1
        trans_final(qi, a, b, c, d, P.gamma, cdtdx1, cdtdx2, P, qo, fe ? fe[p] : 0.0, fe ? fe[n + p] : 0.0,
                    fe ? fe[2 * n + p] : 0.0, fe ? fe[3 * n + p] : 0.0);
    } else if (tdir == 0) {
        trans_single<0>(qi, a, b, P.gamma, cdtdx1, P, qo, fe ? fe[p] : 0.0, fe ? fe[n + p] : 0.0);
    } else if (tdir == 1) {
        trans_single<1>(qi, a, b, P.gamma, cdtdx1, P, qo, fe ? fe[p] : 0.0, fe ? fe[n + p] : 0.0);
    } else {
        trans_single<2>(qi, a, b, P.gamma, cdtdx1, P, qo, fe ? fe[p] : 0.0, fe ? fe[n + p] : 0.0);
    }
#pragma unroll
    for (int m = 0; m < NEDGE; ++m) out[m * n + p] = qo[m];
}

DevParams unit_devparams(const castro_amd_params* p);    // capi.hip

} // namespace cad

using namespace cad;

extern "C" {

int castro_amd_cmpflx_points(long long n, int idir, const double* qm, const double* qp, const double* cl, const double* cr,
                             const double* bnd_fac, const int* is_shock, const castro_amd_params* params, double* out,
                             void* stream)
{
    if (n < 0 || idir < 0 || idir > 2 || !qm || !qp || !cl || !cr || !params || !out) return CASTRO_AMD_ERR_ARG;
    if (n == 0) return CASTRO_AMD_OK;
    hipLaunchKernelGGL(k_cmpflx_points, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long)n, idir,
                       qm, qp, cl, cr, bnd_fac, is_shock, unit_devparams(params), out);
    return hipGetLastError() == hipSuccess ? CASTRO_AMD_OK : CASTRO_AMD_ERR_HIP;
}

int castro_amd_ppm_points(long long n, const double* s, const double* flatn, const double* u, const double* c, double dtdx,
                          double* out, void* stream)
{
    if (n < 0 || !s || !flatn || !u || !c || !out) return CASTRO_AMD_ERR_ARG;
    if (n == 0) return CASTRO_AMD_OK;
    hipLaunchKernelGGL(k_ppm_points, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long)n, s, flatn, u, c,
                       dtdx, out);
    return hipGetLastError() == hipSuccess ? CASTRO_AMD_OK : CASTRO_AMD_ERR_HIP;
}

int castro_amd_flatten_points(long long n, const double* p7, const double* u5, double* out, void* stream)
{
    if (n < 0 || !p7 || !u5 || !out) return CASTRO_AMD_ERR_ARG;
    if (n == 0) return CASTRO_AMD_OK;
    hipLaunchKernelGGL(k_flatten_points, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long)n, p7, u5, out);
    return hipGetLastError() == hipSuccess ? CASTRO_AMD_OK : CASTRO_AMD_ERR_HIP;
}

int castro_amd_trans_points(long long n, int ntrans, int tdir, const double* q, const double* f1r, const double* f1l,
                            const double* f2r, const double* f2l, const double* fe, double cdtdx1, double cdtdx2,
                            const castro_amd_params* params, double* out, void* stream)
{
    if (n < 0 || (ntrans != 1 && ntrans != 2) || tdir < 0 || tdir > 2 || !q || !f1r || !f1l || !params || !out) return CASTRO_AMD_ERR_ARG;
    if (ntrans == 2 && (!f2r || !f2l)) return CASTRO_AMD_ERR_ARG;
    if (n == 0) return CASTRO_AMD_OK;
    hipLaunchKernelGGL(k_trans_points, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long)n, ntrans, tdir,
                       q, f1r, f1l, f2r, f2l, fe, cdtdx1, cdtdx2, unit_devparams(params), out);
    return hipGetLastError() == hipSuccess ? CASTRO_AMD_OK : CASTRO_AMD_ERR_HIP;
}

} // extern "C"
