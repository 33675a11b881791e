// capi.hip -- the C ABI (include/castro_hydro_amd.h) over the HIP kernels.
// Plain pointers and sizes only; no torch, no AMReX.  One context per (device, stream).
#include <hip/hip_runtime.h>
#include <vector>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include "../../include/castro_hydro_amd.h"
#include <cstdlib>
#include "ctu_kernels.h"
namespace cad { extern int g_tile_rows; extern int g_xpad; extern int g_fuse_consup; extern int g_fused_tile_rows; extern int g_trace_tile_rows; extern int g_side_stream; extern int g_fold_r1; extern int g_fold_tile; extern int g_final_tile; extern int g_gl_sources; extern int g_gl_plm; extern int g_fold_tile_rows; extern int g_wg; extern int g_final_wg; extern int g_fused_wg; extern int g_trace_one_zone; extern int g_divu_in_trace; }

using namespace cad;

struct castro_amd_ctx {
    int device = 0;
    double* arena = nullptr;
    size_t arena_doubles = 0;
    int* d_status = nullptr;
    int* h_status = nullptr;   // pinned
    castro_amd_fab src_corr = { nullptr, { 0, 0, 0 }, { 0, 0, 0 }, 0 };     // Castro::source_corrector
    Profiler prof;
    // a second stream for launches that depend on nothing the main stream is about to produce (k_divu beside the trace
    // kernel), forked from and joined to the caller's stream with events inside one call: CASTRO_AMD_SIDE_STREAM=0 turns it off
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    FabOpsArena ops_arena;                          // device table of castro_amd_fab_ops_p calls with more than 16 operations
    FabOpsArena level_arena;                        // device table of a level-wide hydro launch (castro_amd_ctu_hydro_mf)
    hipEvent_t mf_fork = nullptr, mf_join = nullptr;   // castro_amd_ctu_hydro_mf: fork from / join to the caller's stream
};

namespace cad {

// ---- profiler -------------------------------------------------------------------------
static hipEvent_t prof_event(Profiler* p)
{
    if (!p->pool.empty()) { hipEvent_t e = p->pool.back(); p->pool.pop_back(); return e; }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}

void prof_begin(Profiler* p, const char* name, hipStream_t s)
{
    if (!p || !p->enabled) return;
    int idx = -1;
    for (size_t n = 0; n < p->recs.size(); ++n) if (p->recs[n].name == name) { idx = (int)n; break; }
    if (idx < 0) { p->recs.push_back(Profiler::Rec()); p->recs.back().name = name; idx = (int)p->recs.size() - 1; }
    p->cur = idx;
    p->cur_e0 = prof_event(p);
    p->cur_e1 = prof_event(p);
    hipEventRecord(p->cur_e0, s);
}

void prof_end(Profiler* p, hipStream_t s)
{
    if (!p || !p->enabled || p->cur < 0) return;
    hipEventRecord(p->cur_e1, s);
    p->pending.push_back({ p->cur, p->cur_e0, p->cur_e1 });
    p->cur = -1;
}

void prof_collect(Profiler* p)
{
    for (auto& pe : p->pending) {
        hipEventSynchronize(pe.e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, pe.e0, pe.e1);
        p->recs[pe.rec].total_ms += ms;
        p->recs[pe.rec].launches += 1;
        p->pool.push_back(pe.e0);
        p->pool.push_back(pe.e1);
    }
    p->pending.clear();
}

} // namespace cad

// ---- helpers ------------------------------------------------------------------------------
static DFab to_dfab(const castro_amd_fab* f)
{
    DFab d;
    if (!f || !f->p) { d.p = nullptr; d.lo[0] = d.lo[1] = d.lo[2] = 0; d.sy = d.sz = d.sn = 0; return d; }
    d.p = f->p;
    long nx = f->hi[0] - f->lo[0] + 1, ny = f->hi[1] - f->lo[1] + 1, nz = f->hi[2] - f->lo[2] + 1;
    for (int n = 0; n < 3; ++n) d.lo[n] = f->lo[n];
    d.sy = nx; d.sz = nx * ny; d.sn = nx * ny * nz;
    return d;
}

static bool fab_contains(const castro_amd_fab* f, const int lo[3], const int hi[3])
{
    for (int d = 0; d < 3; ++d) if (f->lo[d] > lo[d] || f->hi[d] < hi[d]) return false;
    return true;
}

static DevParams to_devparams(const castro_amd_params* p)
{
    DevParams P;
    P.gamma = p->eos_gamma;
    P.small_dens = p->small_dens; P.small_pres = p->small_pres;
    P.small_temp = p->small_temp; P.small_ener = p->small_ener;
    P.small_dens_ener = p->small_dens * p->small_ener;
    P.difmag = p->difmag;
    P.cg_tol = p->cg_tol;
    P.eta1 = p->dual_energy_eta1; P.eta2 = p->dual_energy_eta2;
    P.small_x = p->small_x;
    P.abar = p->abar;
    P.riemann_solver = p->riemann_solver; P.use_flattening = p->use_flattening;
    P.first_order_hydro = p->first_order_hydro; P.hybrid_riemann = p->hybrid_riemann;
    P.cg_maxiter = p->cg_maxiter; P.cg_blend = p->cg_blend;
    P.reset_density = p->transverse_reset_density; P.reset_rhoe = p->transverse_reset_rhoe;
    P.use_eos = p->transverse_use_eos;
    P.ppm_temp_fix = p->ppm_temp_fix;
    P.ppm_type = p->ppm_type; P.plm_iorder = p->plm_iorder; P.plm_limiter = p->plm_limiter; P.use_pslope = p->use_pslope;
    P.pslope_cutoff_density = p->pslope_cutoff_density;
    P.cfl = p->cfl; P.speed_limit = p->speed_limit;
    P.limit_small_dens = p->limit_fluxes_on_small_dens; P.limit_large_vel = p->limit_fluxes_on_large_vel;
    P.source_term_predictor = p->source_term_predictor;
    P.dtp = nullptr;
    return P;
}

namespace cad { DevParams unit_devparams(const castro_amd_params* p) { return to_devparams(p); } }

static DevGeom to_devgeom(const castro_amd_geom* g)
{
    DevGeom G;
    for (int d = 0; d < 3; ++d) {
        G.dx[d] = g->dx[d];
        G.domlo[d] = g->domlo[d]; G.domhi[d] = g->domhi[d];
        G.wall_lo[d] = (g->lo_bc[d] >= 3) ? 1 : 0;
        G.wall_hi[d] = (g->hi_bc[d] >= 3) ? 1 : 0;
        G.sym_lo[d] = (g->lo_bc[d] == 3) ? 1 : 0;
        G.sym_hi[d] = (g->hi_bc[d] == 3) ? 1 : 0;
    }
    return G;
}

// x extent of a scratch plane: the tile grown by 4, preceded by g_xpad unused columns and rounded up to a multiple of
// 16 doubles when g_xpad > 0, so that zone lo[0] - 4 + (4 + g_xpad) starts a 128-byte line in every row
static int scratch_nx(int nx) { return cad::g_xpad > 0 ? ((nx + 8 + cad::g_xpad + 15) & ~15) : nx + 8; }

static size_t plane_doubles(int nx, int ny, int nz)
{
    size_t n = (size_t)scratch_nx(nx) * (ny + 8) * (nz + 8);
    return (n + 31) & ~(size_t)31;     // keep every component plane 256-byte aligned
}

// number of component planes in the scratch arena
static constexpr int kPlanes = NPRIM + 2 + 6 + 6 * NEDGE + 3 * NF1 + 6 * NF1 + 3 * NFIN;
static constexpr int kPlanesResetRhoe = 9;     // F1E[3] + F2E[6], only with transverse_reset_rhoe = 1

extern "C" {

#ifdef CAD_NUMERICS_CONTRACT
#define CAD_NUMERICS_NAME "contract"
#else
#define CAD_NUMERICS_NAME "exact"
#endif
const char* castro_amd_version(void) { return "castro_hydro_amd " CASTRO_AMD_RELEASE " (gfx950, numerics=" CAD_NUMERICS_NAME ")"; }
int castro_amd_abi_version(void) { return CASTRO_AMD_ABI_VERSION; }
const char* castro_amd_numerics(void) { return CAD_NUMERICS_NAME; }

// Source/driver/_cpp_parameters defaults + Exec/hydro_tests/Sedov/inputs.3d.sph(.testsuite)
void castro_amd_default_params(castro_amd_params* p)
{
    std::memset(p, 0, sizeof(*p));
    p->ppm_type = 1; p->riemann_solver = 0; p->use_flattening = 1; p->hybrid_riemann = 0;
    p->first_order_hydro = 0; p->cg_maxiter = 12; p->cg_blend = 2;
    p->transverse_use_eos = 0; p->transverse_reset_density = 1; p->transverse_reset_rhoe = 0;
    p->ppm_temp_fix = 0;
    p->plm_iorder = 2; p->plm_limiter = 2; p->use_pslope = 1; p->pslope_cutoff_density = -1.e20;
    p->limit_fluxes_on_small_dens = 0; p->limit_fluxes_on_large_vel = 0; p->speed_limit = 0.0;
    p->source_term_predictor = 0;
    p->difmag = 0.1;
    p->small_dens = -1.e200; p->small_temp = -1.e200; p->small_pres = -1.e200; p->small_ener = -1.e200;
    p->cg_tol = 1.0e-5;
    p->dual_energy_eta1 = 1.0; p->dual_energy_eta2 = 1.0e-4;
    p->cfl = 0.5; p->init_shrink = 0.01; p->change_max = 1.1;
    p->eos_gamma = 1.4; p->small_x = 1.e-30; p->T_guess = 1.e8; p->abar = 1.0;
    castro_amd_finalize_params(p);
}

// Castro_setup.cpp:222-236 and :259-288
void castro_amd_finalize_params(castro_amd_params* p)
{
    if (p->small_dens < 0.0) p->small_dens = 1.e-100;
    if (p->small_temp < 0.0) p->small_temp = 1.e-100;
    if (p->small_pres < 0.0) p->small_pres = 1.e-100;
    if (p->small_ener < 0.0) p->small_ener = 1.e-100;
    // eos(eos_input_rt) at (small_dens, small_temp)
    // xn = 1 / NumSpec (Castro_setup.cpp:279): mu = abar = 1 / sum_k(xn_k / A_k)
    const double mu = 1.0 / (1.0 * (1.0 / p->abar));
    double e = K_B * p->small_temp / ((p->eos_gamma - 1.0) * (mu * M_U));
    double pr = (p->eos_gamma - 1.0) * p->small_dens * e;
    if (p->small_pres < pr) p->small_pres = pr;
    if (p->small_ener < e) p->small_ener = e;
}

int castro_amd_ctx_create(castro_amd_ctx** out, int device)
{
    if (!out) return CASTRO_AMD_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        std::fprintf(stderr, "castro_hydro_amd: no HIP device available -- this library has no CPU fallback\n");
        return CASTRO_AMD_ERR_HIP;
    }
    if (device < 0 || device >= ndev) return CASTRO_AMD_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return CASTRO_AMD_ERR_HIP;
    castro_amd_ctx* c = new (std::nothrow) castro_amd_ctx();
    if (!c) return CASTRO_AMD_ERR_NOMEM;
    c->device = device;
    if (hipMalloc(&c->d_status, sizeof(int)) != hipSuccess) { delete c; return CASTRO_AMD_ERR_NOMEM; }
    hipMemset(c->d_status, 0, sizeof(int));
    if (hipHostMalloc(&c->h_status, sizeof(int)) != hipSuccess) { hipFree(c->d_status); delete c; return CASTRO_AMD_ERR_NOMEM; }
    // tuning knobs (ctu_kernels.hip), process-wide, read afresh by every context creation: a variable that is not set puts its
    // knob back to the default (until round 6 a knob kept the last value it had been given, so "unset" did not undo "set")
    auto knob = [](const char* name, int dflt) { const char* e = std::getenv(name); return e ? std::atoi(e) : dflt; };
    auto wg_knob = [&](const char* name, int dflt) { const int v = knob(name, dflt); return (v == 64 || v == 128 || v == 256) ? v : dflt; };
    g_tile_rows = knob("CASTRO_AMD_TILE_ROWS", 32);
    g_fused_tile_rows = knob("CASTRO_AMD_FUSED_TILE_ROWS", 16);
    g_fuse_consup = knob("CASTRO_AMD_FUSE_CONSUP", 1);                 // 0: k_final<x> + k_consup
    g_trace_tile_rows = knob("CASTRO_AMD_TRACE_TILE_ROWS", 64);
    g_xpad = knob("CASTRO_AMD_XPAD", 0);                               // unused columns in front of every scratch row
    g_side_stream = knob("CASTRO_AMD_SIDE_STREAM", 0);
    g_trace_one_zone = knob("CASTRO_AMD_TRACE_ONE_ZONE", 0);
#ifdef CAD_NUMERICS_CONTRACT
    g_divu_in_trace = knob("CASTRO_AMD_DIVU_IN_TRACE", 1);
#else
    g_divu_in_trace = knob("CASTRO_AMD_DIVU_IN_TRACE", 0);
#endif
    g_fold_r1 = knob("CASTRO_AMD_FOLD_R1", 2);
    g_fold_tile_rows = knob("CASTRO_AMD_FOLD_TILE_ROWS", -1);
    g_fold_tile = knob("CASTRO_AMD_FOLD_TILE", -1);
    g_final_tile = knob("CASTRO_AMD_FINAL_TILE", 0);
    g_gl_sources = knob("CASTRO_AMD_GL_SOURCES", 1);
    g_gl_plm = knob("CASTRO_AMD_GL_PLM", 1);
    g_wg = wg_knob("CASTRO_AMD_WG", 256);
    g_fused_wg = wg_knob("CASTRO_AMD_FUSED_WG", 128);
    g_final_wg = wg_knob("CASTRO_AMD_FINAL_WG", 0);
    if (g_side_stream) {
        if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
            castro_amd_ctx_destroy(c);
            return CASTRO_AMD_ERR_HIP;
        }
    }
    *out = c;
    return CASTRO_AMD_OK;
}

void castro_amd_ctx_destroy(castro_amd_ctx* c)
{
    if (!c) return;
    hipSetDevice(c->device);
    if (c->side) { hipStreamSynchronize(c->side); hipStreamDestroy(c->side); }
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    if (c->mf_fork) hipEventDestroy(c->mf_fork);
    if (c->mf_join) hipEventDestroy(c->mf_join);
    if (c->ops_arena.p) hipFree(c->ops_arena.p);
    if (c->level_arena.p) hipFree(c->level_arena.p);
    prof_collect(&c->prof);
    for (auto e : c->prof.pool) hipEventDestroy(e);
    if (c->arena) hipFree(c->arena);
    if (c->d_status) hipFree(c->d_status);
    if (c->h_status) hipHostFree(c->h_status);
    delete c;
}

static int reserve_planes(castro_amd_ctx* c, int nx, int ny, int nz, int planes)
{
    if (!c || nx <= 0 || ny <= 0 || nz <= 0) return CASTRO_AMD_ERR_ARG;
    size_t need = plane_doubles(nx, ny, nz) * (size_t)planes;
    if (need <= c->arena_doubles) return CASTRO_AMD_OK;
    hipSetDevice(c->device);
    if (c->arena) { hipDeviceSynchronize(); hipFree(c->arena); c->arena = nullptr; c->arena_doubles = 0; }
    if (hipMalloc(&c->arena, need * sizeof(double)) != hipSuccess) return CASTRO_AMD_ERR_NOMEM;
    c->arena_doubles = need;
    return CASTRO_AMD_OK;
}

int castro_amd_ctx_reserve(castro_amd_ctx* c, int nx, int ny, int nz)
{
    return reserve_planes(c, nx, ny, nz, kPlanes);
}

long long castro_amd_ctx_scratch_bytes(const castro_amd_ctx* c)
{
    return c ? (long long)(c->arena_doubles * sizeof(double)) : 0;
}

int castro_amd_ctx_status(castro_amd_ctx* c, void* stream)
{
    if (!c) return CASTRO_AMD_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    hipMemcpyAsync(c->h_status, c->d_status, sizeof(int), hipMemcpyDeviceToHost, s);
    hipMemsetAsync(c->d_status, 0, sizeof(int), s);
    if (hipStreamSynchronize(s) != hipSuccess) return CASTRO_AMD_ERR_HIP;
    return *c->h_status;
}

int castro_amd_ctu_hydro_fab(castro_amd_ctx* c, const int bxlo[3], const int bxhi[3],
                             const int vbxlo[3], const int vbxhi[3],
                             const castro_amd_fab* Sborder, const castro_amd_fab* src,
                             const castro_amd_fab* S_new, const castro_amd_fab flux_out[3],
                             const castro_amd_fab mass_flux_out[3], const castro_amd_fab qe_out[3],
                             const castro_amd_geom* geom, const castro_amd_params* params,
                             double time, double dt, int flags, void* stream)
{
    return castro_amd_ctu_hydro_clean_fab(c, bxlo, bxhi, vbxlo, vbxhi, Sborder, src, S_new, flux_out, mass_flux_out, qe_out,
                                          geom, params, time, dt, flags, 0, nullptr, stream);
}

int castro_amd_ctu_hydro_clean_fab(castro_amd_ctx* c, const int bxlo[3], const int bxhi[3],
                                   const int vbxlo[3], const int vbxhi[3],
                                   const castro_amd_fab* Sborder, const castro_amd_fab* src,
                                   const castro_amd_fab* S_new, const castro_amd_fab flux_out[3],
                                   const castro_amd_fab mass_flux_out[3], const castro_amd_fab qe_out[3],
                                   const castro_amd_geom* geom, const castro_amd_params* params,
                                   double time, double dt, int flags, int clean_ntimes, double* d_out, void* stream)
{
    castro_amd_hydro_opts o;
    o.flags = flags; o.clean_ntimes = clean_ntimes; o.d_out = d_out; o.sborder_clean_ntimes = 0; o.d_dt = nullptr;
    return castro_amd_ctu_hydro_fab_ex(c, bxlo, bxhi, vbxlo, vbxhi, Sborder, src, S_new, flux_out, mass_flux_out, qe_out,
                                       geom, params, time, dt, &o, stream);
}

// One box of a hydro call, checked and translated: what castro_amd_ctu_hydro_fab_ex and the level-wide launch of
// castro_amd_ctu_hydro_mf have in common.
struct PreparedBox {
    Tile t;
    DFab dS, dN, dSrc, dF[3], dM[3], dQ[3], dCorr;
    int acc_hi[3];
    int nx, ny, nz;
    bool reset_rhoe;
};

static int prepare_box(castro_amd_ctx* c, const int bxlo[3], const int bxhi[3], const int vbxlo[3], const int vbxhi[3],
                       const castro_amd_fab* Sborder, const castro_amd_fab* src, const castro_amd_fab* S_new,
                       const castro_amd_fab flux_out[3], const castro_amd_fab mass_flux_out[3], const castro_amd_fab qe_out[3],
                       const castro_amd_geom* geom, const castro_amd_params* params, const castro_amd_hydro_opts* opts, PreparedBox& B)
{
    if (!opts) return CASTRO_AMD_ERR_ARG;
    const int flags = opts->flags, clean_ntimes = opts->clean_ntimes, sb_clean = opts->sborder_clean_ntimes;
    if (clean_ntimes < 0 || sb_clean < 0) return CASTRO_AMD_ERR_ARG;
    const int light = flags & (CASTRO_AMD_STAGE_VALID | CASTRO_AMD_STAGE_REST);
    if (light == (CASTRO_AMD_STAGE_VALID | CASTRO_AMD_STAGE_REST)) return CASTRO_AMD_ERR_ARG;
    if ((light || (flags & CASTRO_AMD_BC_FILL)) && (flags & (CASTRO_AMD_STAGE_A | CASTRO_AMD_STAGE_B))) return CASTRO_AMD_ERR_ARG;
    if (sb_clean > 0 || light || (flags & CASTRO_AMD_BC_FILL)) {
        // in-place cleaning of Sborder, the valid / rest split and the fused boundary fill: whole-box calls only, never with the
        // round-2 staging (see the header)
        if (sb_clean > 0 && (flags & (CASTRO_AMD_STAGE_A | CASTRO_AMD_STAGE_B))) return CASTRO_AMD_ERR_ARG;
        if (vbxlo && vbxhi)
            for (int d = 0; d < 3; ++d) if (vbxlo[d] != bxlo[d] || vbxhi[d] != bxhi[d]) return CASTRO_AMD_ERR_ARG;
        // cleaned valid zones + a separate k_bc_fill + cleaning the shell would clean the boundary zones twice over
        if (sb_clean > 0 && (flags & CASTRO_AMD_STAGE_REST) && !(flags & CASTRO_AMD_BC_FILL)) return CASTRO_AMD_ERR_ARG;
    }
    if (!c || !bxlo || !bxhi || !Sborder || !Sborder->p || !S_new || !S_new->p || !geom || !params)
        return CASTRO_AMD_ERR_ARG;
    if (Sborder->ncomp != NUM_STATE || S_new->ncomp != NUM_STATE) return CASTRO_AMD_ERR_ARG;
    if (geom->coord != 0) return CASTRO_AMD_ERR_UNSUPPORTED;
    if (params->ppm_type != 0 && params->ppm_type != 1) return CASTRO_AMD_ERR_ARG;
    if (params->riemann_solver < 0 || params->riemann_solver > 2) return CASTRO_AMD_ERR_ARG;
    if (params->hybrid_riemann != 0 && params->hybrid_riemann != 1) return CASTRO_AMD_ERR_ARG;
    if (params->ppm_temp_fix < 0 || params->ppm_temp_fix > 2) return CASTRO_AMD_ERR_ARG;    // 1 is a no-op in the CTU path

    Tile& t = B.t;
    int glo[3], ghi[3];
    for (int d = 0; d < 3; ++d) {
        if (bxhi[d] < bxlo[d]) return CASTRO_AMD_ERR_ARG;
        t.lo[d] = bxlo[d]; t.hi[d] = bxhi[d];
        t.glo[d] = glo[d] = bxlo[d] - CASTRO_AMD_NUM_GROW;
        ghi[d] = bxhi[d] + CASTRO_AMD_NUM_GROW;
    }
    const int nx = bxhi[0] - bxlo[0] + 1, ny = bxhi[1] - bxlo[1] + 1, nz = bxhi[2] - bxlo[2] + 1;
    B.nx = nx; B.ny = ny; B.nz = nz;
    t.glo[0] -= g_xpad;
    t.NX = scratch_nx(nx); t.NY = ny + 8; t.NZ = nz + 8;
    t.NC = (long)plane_doubles(nx, ny, nz);

    if (!fab_contains(Sborder, glo, ghi)) return CASTRO_AMD_ERR_ARG;
    if (!fab_contains(S_new, bxlo, bxhi)) return CASTRO_AMD_ERR_ARG;
    // the kernels address a component plane with a 32-bit byte offset: scratch planes and the caller's component
    // planes must stay below 4 GiB (a box of more than ~800^3 zones has to be tiled by the caller)
    {
        const double lim = 4294967296.0;
        auto plane_bytes = [](const castro_amd_fab* f) {
            return 8.0 * (double)(f->hi[0] - f->lo[0] + 1) * (double)(f->hi[1] - f->lo[1] + 1) * (double)(f->hi[2] - f->lo[2] + 1);
        };
        if (8.0 * (double)t.NC >= lim || plane_bytes(Sborder) >= lim || plane_bytes(S_new) >= lim) return CASTRO_AMD_ERR_UNSUPPORTED;
        for (int d = 0; d < 3; ++d) {
            if (flux_out && flux_out[d].p && plane_bytes(&flux_out[d]) >= lim) return CASTRO_AMD_ERR_UNSUPPORTED;
            if (mass_flux_out && mass_flux_out[d].p && plane_bytes(&mass_flux_out[d]) >= lim) return CASTRO_AMD_ERR_UNSUPPORTED;
            if (qe_out && qe_out[d].p && plane_bytes(&qe_out[d]) >= lim) return CASTRO_AMD_ERR_UNSUPPORTED;
        }
    }
    if (src && src->p) {
        // old_source: NSRC = 7 components, NUM_GROW_SRC = 3 ghost zones (Castro_setup.cpp:317-327)
        int s3lo[3], s3hi[3];
        for (int d = 0; d < 3; ++d) { s3lo[d] = bxlo[d] - 3; s3hi[d] = bxhi[d] + 3; }
        if (src->ncomp < 6 || !fab_contains(src, s3lo, s3hi)) return CASTRO_AMD_ERR_ARG;
    }
    B.reset_rhoe = params->transverse_reset_rhoe == 1;
    B.dS = to_dfab(Sborder); B.dN = to_dfab(S_new); B.dSrc = to_dfab(src);
    for (int d = 0; d < 3; ++d) {
        int flo[3] = { bxlo[0], bxlo[1], bxlo[2] }, fhi[3] = { bxhi[0], bxhi[1], bxhi[2] };
        fhi[d] += 1;
        // mfi.nodaltilebox(d): the high face belongs to this tile only at the valid box's high end
        B.acc_hi[d] = bxhi[d] + 1;
        if (vbxhi && B.acc_hi[d] <= vbxhi[d]) B.acc_hi[d] -= 1;
        fhi[d] = B.acc_hi[d];
        const castro_amd_fab* f = flux_out ? &flux_out[d] : nullptr;
        const castro_amd_fab* m = mass_flux_out ? &mass_flux_out[d] : nullptr;
        const castro_amd_fab* q = qe_out ? &qe_out[d] : nullptr;
        if (f && f->p && (f->ncomp != NUM_STATE || !fab_contains(f, flo, fhi))) return CASTRO_AMD_ERR_ARG;
        if (m && m->p && (m->ncomp != 1 || !fab_contains(m, flo, fhi))) return CASTRO_AMD_ERR_ARG;
        if (q && q->p && (q->ncomp != NGDNV || !fab_contains(q, flo, fhi))) return CASTRO_AMD_ERR_ARG;
        B.dF[d] = to_dfab(f); B.dM[d] = to_dfab(m); B.dQ[d] = to_dfab(q);
    }
    B.dCorr = to_dfab(nullptr);
    // the reference always has source_corrector defined when the predictor is on (Castro_advance_ctu.cpp:60-62):
    // running without it would silently drop the predictor
    if (params->source_term_predictor == 1 && src && src->p && !c->src_corr.p) return CASTRO_AMD_ERR_ARG;
    if (params->source_term_predictor == 1 && c->src_corr.p && src && src->p) {
        int s3lo[3], s3hi[3];
        for (int d = 0; d < 3; ++d) { s3lo[d] = bxlo[d] - 3; s3hi[d] = bxhi[d] + 3; }
        if (!fab_contains(&c->src_corr, s3lo, s3hi)) return CASTRO_AMD_ERR_ARG;
        B.dCorr = to_dfab(&c->src_corr);
    }
    return CASTRO_AMD_OK;
}

// the scratch arrays of one box, carved from p; returns the first double behind them
static double* carve_scratch(double* p, const Tile& t, bool reset_rhoe, DevScratch& S)
{
    const size_t NC = (size_t)t.NC;
    S.Q = p; p += NC * NPRIM;
    S.DIV = p; p += NC;
    S.SHK = p; p += NC;
    S.SRCQ = p; p += NC * 6;
    for (int d = 0; d < 3; ++d) { S.QM[d] = p; p += NC * NEDGE; S.QP[d] = p; p += NC * NEDGE; }
    for (int d = 0; d < 3; ++d) { S.F1[d] = p; p += NC * NF1; }
    for (int d = 0; d < 6; ++d) { S.F2[d] = p; p += NC * NF1; }
    for (int d = 0; d < 3; ++d) { S.FL[d] = p; p += NC * NFIN; }
    for (int d = 0; d < 3; ++d) { S.F1E[d] = reset_rhoe ? p : nullptr; if (reset_rhoe) p += NC; }
    for (int d = 0; d < 6; ++d) { S.F2E[d] = reset_rhoe ? p : nullptr; if (reset_rhoe) p += NC; }
    return p;
}

int castro_amd_ctu_hydro_fab_ex(castro_amd_ctx* c, const int bxlo[3], const int bxhi[3],
                                const int vbxlo[3], const int vbxhi[3],
                                const castro_amd_fab* Sborder, const castro_amd_fab* src,
                                const castro_amd_fab* S_new, const castro_amd_fab flux_out[3],
                                const castro_amd_fab mass_flux_out[3], const castro_amd_fab qe_out[3],
                                const castro_amd_geom* geom, const castro_amd_params* params,
                                double time, double dt, const castro_amd_hydro_opts* opts, void* stream)
{
    (void)time;
    PreparedBox B;
    int rc = prepare_box(c, bxlo, bxhi, vbxlo, vbxhi, Sborder, src, S_new, flux_out, mass_flux_out, qe_out, geom, params, opts, B);
    if (rc != CASTRO_AMD_OK) return rc;
    hipSetDevice(c->device);
    rc = reserve_planes(c, B.nx, B.ny, B.nz, kPlanes + (B.reset_rhoe ? kPlanesResetRhoe : 0));
    if (rc != CASTRO_AMD_OK) return rc;
    DevScratch S;
    carve_scratch(c->arena, B.t, B.reset_rhoe, S);
    LaunchAux aux;
    aux.sb_clean = opts->sborder_clean_ntimes;
    aux.side = c->side; aux.ev_fork = c->ev_fork; aux.ev_join = c->ev_join;
    for (int d = 0; d < 3; ++d) { aux.bc_lo[d] = geom->lo_bc[d]; aux.bc_hi[d] = geom->hi_bc[d]; }
    DevParams devP = to_devparams(params);
    devP.dtp = opts->d_dt;
    return launch_ctu_hydro(B.t, S, B.dS, B.dSrc, B.dN, B.dF, B.dM, B.dQ, to_devgeom(geom), devP, dt, opts->flags,
                            B.acc_hi, c->d_status, (hipStream_t)stream, &c->prof, opts->clean_ntimes, opts->d_out, B.dCorr, aux);
}

int castro_amd_step_control(castro_amd_ctx* c, double* d_red, double* d_ctl, const castro_amd_params* params,
                            double max_dt, double fixed_dt, double stop_time, int use_retry, void* stream)
{
    if (!c || !d_red || !d_ctl || !params) return CASTRO_AMD_ERR_ARG;
    hipSetDevice(c->device);
    return launch_step_control(d_red, d_ctl, params->cfl, params->change_max, params->small_dens, max_dt, fixed_dt, stop_time,
                               use_retry ? 1 : 0, (hipStream_t)stream, &c->prof);
}

int castro_amd_clean_state_fab(castro_amd_ctx* c, const castro_amd_fab* state, const int lo[3], const int hi[3],
                               const castro_amd_params* params, int ntimes, void* stream)
{
    if (!c || !state || !state->p || !params || state->ncomp != NUM_STATE || ntimes < 1) return CASTRO_AMD_ERR_ARG;
    if (!fab_contains(state, lo, hi)) return CASTRO_AMD_ERR_ARG;
    hipSetDevice(c->device);
    return launch_clean_state(to_dfab(state), lo, hi, to_devparams(params), ntimes, (hipStream_t)stream, &c->prof);
}

int castro_amd_clean_state_reduce_fab(castro_amd_ctx* c, const castro_amd_fab* state, const int lo[3], const int hi[3],
                                      const castro_amd_geom* geom, const castro_amd_params* params, int ntimes,
                                      double* d_out, void* stream)
{
    if (!c || !state || !state->p || !geom || !params || !d_out || state->ncomp != NUM_STATE || ntimes < 1)
        return CASTRO_AMD_ERR_ARG;
    if (!fab_contains(state, lo, hi)) return CASTRO_AMD_ERR_ARG;
    hipSetDevice(c->device);
    return launch_clean_state_reduce(to_dfab(state), lo, hi, to_devgeom(geom), to_devparams(params), ntimes, d_out,
                                     (hipStream_t)stream, &c->prof);
}

int castro_amd_estdt_fab(castro_amd_ctx* c, const castro_amd_fab* state, const int lo[3], const int hi[3],
                         const castro_amd_geom* geom, const castro_amd_params* params, double* d_out, void* stream)
{
    if (!c || !state || !state->p || !geom || !params || !d_out || state->ncomp != NUM_STATE) return CASTRO_AMD_ERR_ARG;
    if (!fab_contains(state, lo, hi)) return CASTRO_AMD_ERR_ARG;
    hipSetDevice(c->device);
    return launch_estdt(to_dfab(state), lo, hi, to_devgeom(geom), to_devparams(params), d_out, (hipStream_t)stream, &c->prof);
}

int castro_amd_old_gravity_source_fab(castro_amd_ctx* c, const castro_amd_fab* state, const castro_amd_fab* source,
                                      const int lo[3], const int hi[3], const double grav[3], int grav_source_type,
                                      double dt, void* stream)
{
    if (!c || !state || !state->p || !source || !source->p || !grav) return CASTRO_AMD_ERR_ARG;
    if (state->ncomp != NUM_STATE || source->ncomp < 7 || grav_source_type < 1 || grav_source_type > 4) return CASTRO_AMD_ERR_ARG;
    if (!fab_contains(state, lo, hi) || !fab_contains(source, lo, hi)) return CASTRO_AMD_ERR_ARG;
    hipSetDevice(c->device);
    return launch_old_grav_source(to_dfab(state), to_dfab(source), lo, hi, grav, grav_source_type, dt, (hipStream_t)stream, &c->prof);
}

int castro_amd_new_gravity_source_fab(castro_amd_ctx* c, const castro_amd_fab* state_old, const castro_amd_fab* state_new,
                                      const castro_amd_fab* source, const castro_amd_fab mass_fluxes[3],
                                      const int lo[3], const int hi[3], const double grav[3], int grav_source_type,
                                      double dt, const castro_amd_geom* geom, void* stream)
{
    if (!c || !state_old || !state_old->p || !state_new || !state_new->p || !source || !source->p || !mass_fluxes || !grav || !geom)
        return CASTRO_AMD_ERR_ARG;
    if (state_old->ncomp != NUM_STATE || state_new->ncomp != NUM_STATE || source->ncomp < 7) return CASTRO_AMD_ERR_ARG;
    if (grav_source_type < 1 || grav_source_type > 4 || geom->coord != 0) return CASTRO_AMD_ERR_ARG;
    if (!fab_contains(state_old, lo, hi) || !fab_contains(state_new, lo, hi) || !fab_contains(source, lo, hi)) return CASTRO_AMD_ERR_ARG;
    DFab M[3];
    for (int d = 0; d < 3; ++d) {
        int fhi[3] = { hi[0], hi[1], hi[2] };
        fhi[d] += 1;
        if (!mass_fluxes[d].p || mass_fluxes[d].ncomp != 1 || !fab_contains(&mass_fluxes[d], lo, fhi)) return CASTRO_AMD_ERR_ARG;
        M[d] = to_dfab(&mass_fluxes[d]);
    }
    hipSetDevice(c->device);
    return launch_new_grav_source(to_dfab(state_old), to_dfab(state_new), to_dfab(source), M, lo, hi, grav, grav_source_type,
                                  dt, geom->dx, (hipStream_t)stream, &c->prof);
}

int castro_amd_old_rotation_source_fab(castro_amd_ctx* c, const castro_amd_fab* state, const castro_amd_fab* source,
                                       const int lo[3], const int hi[3], const castro_amd_rotation* rot,
                                       const castro_amd_geom* geom, double dt, void* stream)
{
    if (!c || !state || !state->p || !source || !source->p || !rot || !geom) return CASTRO_AMD_ERR_ARG;
    if (state->ncomp != NUM_STATE || source->ncomp < 7 || rot->rot_source_type < 1 || rot->rot_source_type > 4 || geom->coord != 0)
        return CASTRO_AMD_ERR_ARG;
    if (!fab_contains(state, lo, hi) || !fab_contains(source, lo, hi)) return CASTRO_AMD_ERR_ARG;
    hipSetDevice(c->device);
    return launch_old_rot_source(to_dfab(state), to_dfab(source), lo, hi, rot, geom, dt, (hipStream_t)stream, &c->prof);
}

int castro_amd_new_rotation_source_fab(castro_amd_ctx* c, const castro_amd_fab* state_old, const castro_amd_fab* state_new,
                                       const castro_amd_fab* source, const castro_amd_fab mass_fluxes[3],
                                       const int lo[3], const int hi[3], const castro_amd_rotation* rot,
                                       const castro_amd_geom* geom, double dt, void* stream)
{
    if (!c || !state_old || !state_old->p || !state_new || !state_new->p || !source || !source->p || !mass_fluxes || !rot || !geom)
        return CASTRO_AMD_ERR_ARG;
    if (state_old->ncomp != NUM_STATE || state_new->ncomp != NUM_STATE || source->ncomp < 7) return CASTRO_AMD_ERR_ARG;
    if (rot->rot_source_type < 1 || rot->rot_source_type > 4 || geom->coord != 0 || !(dt > 0.0)) return CASTRO_AMD_ERR_ARG;
    if (!fab_contains(state_old, lo, hi) || !fab_contains(state_new, lo, hi) || !fab_contains(source, lo, hi)) return CASTRO_AMD_ERR_ARG;
    DFab M[3];
    for (int d = 0; d < 3; ++d) {
        int fhi[3] = { hi[0], hi[1], hi[2] };
        fhi[d] += 1;
        if (!mass_fluxes[d].p || mass_fluxes[d].ncomp != 1 || !fab_contains(&mass_fluxes[d], lo, fhi)) return CASTRO_AMD_ERR_ARG;
        M[d] = to_dfab(&mass_fluxes[d]);
    }
    hipSetDevice(c->device);
    return launch_new_rot_source(to_dfab(state_old), to_dfab(state_new), to_dfab(source), M, lo, hi, rot, geom, dt,
                                 (hipStream_t)stream, &c->prof);
}

// ---- the source stages and the level reductions for every box of a level in one call (include/castro_hydro_amd.h) ----
int castro_amd_sources_mf(castro_amd_ctx* c, int stage, int nboxes, const castro_amd_source_box* boxes,
                          const double* grav, int grav_source_type, const castro_amd_rotation* rot,
                          const castro_amd_geom* geom, const castro_amd_params* params, double dt, int clean_ntimes, void* stream)
{
    if (!c || (stage != 0 && stage != 1) || nboxes < 0 || (nboxes > 0 && !boxes) || !geom || !params || clean_ntimes < 0) return CASTRO_AMD_ERR_ARG;
    // the checks of the single-box entry points (castro_amd_old/new_gravity_source_fab, _rotation_source_fab, _apply_source_fab)
    if (grav && (grav_source_type < 1 || grav_source_type > 4)) return CASTRO_AMD_ERR_ARG;
    if (rot && (rot->rot_source_type < 1 || rot->rot_source_type > 4)) return CASTRO_AMD_ERR_ARG;
    if ((rot || (grav && stage == 1)) && geom->coord != 0) return CASTRO_AMD_ERR_ARG;
    if (rot && stage == 1 && !(dt > 0.0)) return CASTRO_AMD_ERR_ARG;
    if (nboxes == 0) return CASTRO_AMD_OK;
    std::vector<SrcBoxDev> tab((size_t)nboxes);
    for (int i = 0; i < nboxes; ++i) {
        const castro_amd_source_box& b = boxes[i];
        if (!b.S_old.p || !b.S_new.p || !b.source.p || b.S_old.ncomp != NUM_STATE || b.S_new.ncomp != NUM_STATE || b.source.ncomp < 7)
            return CASTRO_AMD_ERR_ARG;
        if (!fab_contains(&b.S_old, b.lo, b.hi) || !fab_contains(&b.S_new, b.lo, b.hi) || !fab_contains(&b.source, b.lo, b.hi)) return CASTRO_AMD_ERR_ARG;
        SrcBoxDev& T = tab[(size_t)i];
        T.So = to_dfab(&b.S_old); T.Sn = to_dfab(&b.S_new); T.Src = to_dfab(&b.source);
        T.M0 = T.M1 = T.M2 = to_dfab(nullptr);
        if (stage == 1 && (grav || rot)) {
            DFab M[3];
            for (int d = 0; d < 3; ++d) {
                int fhi[3] = { b.hi[0], b.hi[1], b.hi[2] };
                fhi[d] += 1;
                if (!b.mass_flux[d].p || b.mass_flux[d].ncomp != 1 || !fab_contains(&b.mass_flux[d], b.lo, fhi)) return CASTRO_AMD_ERR_ARG;
                M[d] = to_dfab(&b.mass_flux[d]);
            }
            T.M0 = M[0]; T.M1 = M[1]; T.M2 = M[2];
        }
        for (int d = 0; d < 3; ++d) {
            if (b.source.hi[d] < b.source.lo[d]) return CASTRO_AMD_ERR_ARG;
            T.lo[d] = b.source.lo[d]; T.n[d] = b.source.hi[d] - b.source.lo[d] + 1;
            T.vlo[d] = b.lo[d]; T.vhi[d] = b.hi[d];
        }
        T.nsc = b.source.ncomp;
    }
    hipSetDevice(c->device);
    const int rc = launch_sources_apply(stage, nboxes, tab.data(), grav, grav_source_type, rot, geom, to_devparams(params), dt, clean_ntimes,
                                        &c->ops_arena, (hipStream_t)stream, &c->prof);
    return rc == 0 ? CASTRO_AMD_OK : (rc < 0 ? CASTRO_AMD_ERR_HIP : rc);
}

int castro_amd_clean_state_reduce_mf(castro_amd_ctx* c, int nboxes, const castro_amd_state_box* boxes, const castro_amd_geom* geom,
                                     const castro_amd_params* params, int ntimes, double* d_out, void* stream)
{
    if (!c || nboxes < 0 || (nboxes > 0 && !boxes)) return CASTRO_AMD_ERR_ARG;
    for (int i = 0; i < nboxes; ++i) {
        const int rc = castro_amd_clean_state_reduce_fab(c, &boxes[i].state, boxes[i].lo, boxes[i].hi, geom, params, ntimes, d_out, stream);
        if (rc != CASTRO_AMD_OK) return rc;
    }
    return CASTRO_AMD_OK;
}

int castro_amd_estdt_mf(castro_amd_ctx* c, int nboxes, const castro_amd_state_box* boxes, const castro_amd_geom* geom,
                        const castro_amd_params* params, double* d_out, void* stream)
{
    if (!c || nboxes < 0 || (nboxes > 0 && !boxes)) return CASTRO_AMD_ERR_ARG;
    for (int i = 0; i < nboxes; ++i) {
        const int rc = castro_amd_estdt_fab(c, &boxes[i].state, boxes[i].lo, boxes[i].hi, geom, params, d_out, stream);
        if (rc != CASTRO_AMD_OK) return rc;
    }
    return CASTRO_AMD_OK;
}

int castro_amd_saxpy_fab(castro_amd_ctx* c, const castro_amd_fab* dst, double a, const castro_amd_fab* src, int ncomp,
                         const int lo[3], const int hi[3], void* stream)
{
    if (!c || !dst || !dst->p || !src || !src->p || ncomp < 1 || ncomp > dst->ncomp || ncomp > src->ncomp) return CASTRO_AMD_ERR_ARG;
    if (!fab_contains(dst, lo, hi) || !fab_contains(src, lo, hi)) return CASTRO_AMD_ERR_ARG;
    hipSetDevice(c->device);
    return launch_saxpy(to_dfab(dst), to_dfab(src), lo, hi, a, ncomp, (hipStream_t)stream, &c->prof);
}

int castro_amd_apply_source_fab(castro_amd_ctx* c, const castro_amd_fab* dst, const castro_amd_fab* base, double a,
                                const castro_amd_fab* src, int nsrc, const int lo[3], const int hi[3],
                                const castro_amd_params* params, int clean_ntimes, void* stream)
{
    if (!c || !dst || !dst->p || !base || !base->p || !src || !src->p || !params || clean_ntimes < 0) return CASTRO_AMD_ERR_ARG;
    if (dst->ncomp != NUM_STATE || base->ncomp != NUM_STATE || nsrc < 0 || nsrc > NUM_STATE || nsrc > src->ncomp) return CASTRO_AMD_ERR_ARG;
    if (!fab_contains(dst, lo, hi) || !fab_contains(base, lo, hi) || !fab_contains(src, lo, hi)) return CASTRO_AMD_ERR_ARG;
    hipSetDevice(c->device);
    return launch_apply_source(to_dfab(dst), to_dfab(base), to_dfab(src), lo, hi, a, nsrc, to_devparams(params), clean_ntimes,
                               (hipStream_t)stream, &c->prof);
}

static bool fab_ok(const castro_amd_fab* f, int ncomp) { return f && f->p && f->ncomp >= ncomp; }

int castro_amd_cc_interp_fab(castro_amd_ctx* c, const castro_amd_fab* crse, const castro_amd_fab* fine,
                             const int lo[3], const int hi[3], int ncomp, void* stream)
{
    if (!c || ncomp < 1 || !fab_ok(crse, ncomp) || !fab_ok(fine, ncomp) || !fab_contains(fine, lo, hi)) return CASTRO_AMD_ERR_ARG;
    int clo[3], chi[3];
    for (int d = 0; d < 3; ++d) {
        clo[d] = (lo[d] >= 0 ? lo[d] / 2 : -((-lo[d] + 1) / 2)) - 1;
        chi[d] = (hi[d] >= 0 ? hi[d] / 2 : -((-hi[d] + 1) / 2)) + 1;
    }
    if (!fab_contains(crse, clo, chi)) return CASTRO_AMD_ERR_ARG;
    hipSetDevice(c->device);
    return launch_cc_interp(to_dfab(crse), to_dfab(fine), lo, hi, ncomp, (hipStream_t)stream, &c->prof);
}

static int fab_ops_impl(castro_amd_ctx* c, int nops, const castro_amd_fab_op* ops, const castro_amd_params* params, void* stream);

int castro_amd_fab_ops(castro_amd_ctx* c, int nops, const castro_amd_fab_op* ops, void* stream)
{
    return fab_ops_impl(c, nops, ops, nullptr, stream);
}

int castro_amd_fab_ops_p(castro_amd_ctx* c, int nops, const castro_amd_fab_op* ops, const castro_amd_params* params, void* stream)
{
    if (!params) return CASTRO_AMD_ERR_ARG;
    return fab_ops_impl(c, nops, ops, params, stream);
}

static int fab_ops_impl(castro_amd_ctx* c, int nops, const castro_amd_fab_op* ops, const castro_amd_params* params, void* stream)
{
    if (!c || nops < 0 || (nops > 0 && !ops)) return CASTRO_AMD_ERR_ARG;
    if (nops == 0) return CASTRO_AMD_OK;
    std::vector<DFab> D(nops), X(nops), Y(nops);
    std::vector<int> lo(3 * nops), hi(3 * nops), kind(nops), dir(nops), side(nops), ncomp(nops);
    std::vector<double> a(nops), b(nops);
    for (int r = 0; r < nops; ++r) {
        const castro_amd_fab_op& o = ops[r];
        if (o.ncomp < 1 || !fab_ok(&o.dst, o.ncomp) || !fab_ok(&o.src, o.ncomp)) return CASTRO_AMD_ERR_ARG;
        if (o.kind != CASTRO_AMD_OP_REFLUX && !fab_contains(&o.dst, o.lo, o.hi)) return CASTRO_AMD_ERR_ARG;
        switch (o.kind) {
        case CASTRO_AMD_OP_CLEAN:
            if (!params || o.ncomp != NUM_STATE || o.dst.ncomp != NUM_STATE || o.a < 0.0) return CASTRO_AMD_ERR_ARG;
            break;
        case CASTRO_AMD_OP_INTERP_CLEAN: {
            if (!params || o.ncomp != NUM_STATE || o.dst.ncomp != NUM_STATE || o.src.ncomp != NUM_STATE || o.a < 0.0) return CASTRO_AMD_ERR_ARG;
            // the coarse zones under the region, grown by one (the slopes of cell_cons_interp)
            int clo[3], chi[3];
            for (int d = 0; d < 3; ++d) { clo[d] = (o.lo[d] >= 0 ? o.lo[d] / 2 : -((-o.lo[d] + 1) / 2)) - 1; chi[d] = (o.hi[d] >= 0 ? o.hi[d] / 2 : -((-o.hi[d] + 1) / 2)) + 1; }
            if (!fab_contains(&o.src, clo, chi)) return CASTRO_AMD_ERR_ARG;
            break; }
        case CASTRO_AMD_OP_INTERP: {
            if (o.ncomp > NUM_STATE) return CASTRO_AMD_ERR_ARG;
            int clo[3], chi[3];
            for (int d = 0; d < 3; ++d) { clo[d] = (o.lo[d] >= 0 ? o.lo[d] / 2 : -((-o.lo[d] + 1) / 2)) - 1; chi[d] = (o.hi[d] >= 0 ? o.hi[d] / 2 : -((-o.hi[d] + 1) / 2)) + 1; }
            if (!fab_contains(&o.src, clo, chi)) return CASTRO_AMD_ERR_ARG;
            break; }
        case CASTRO_AMD_OP_AVGDOWN: {
            int flo[3], fhi[3];
            for (int d = 0; d < 3; ++d) { flo[d] = 2 * o.lo[d]; fhi[d] = 2 * o.hi[d] + 1; }
            if (!fab_contains(&o.src, flo, fhi)) return CASTRO_AMD_ERR_ARG;
            break; }
        case CASTRO_AMD_OP_REFLUX: {
            if (o.dir < 0 || o.dir > 2 || o.side < 0 || o.side > 1 || !fab_contains(&o.src, o.lo, o.hi)) return CASTRO_AMD_ERR_ARG;
            int zlo[3] = { o.lo[0], o.lo[1], o.lo[2] }, zhi[3] = { o.hi[0], o.hi[1], o.hi[2] };
            if (o.side == 0) { zlo[o.dir] -= 1; zhi[o.dir] -= 1; }
            if (!fab_contains(&o.dst, zlo, zhi)) return CASTRO_AMD_ERR_ARG;
            break; }
        case CASTRO_AMD_OP_LINCOMB:
            if (!fab_ok(&o.src2, o.ncomp) || !fab_contains(&o.src2, o.lo, o.hi)) return CASTRO_AMD_ERR_ARG;
            /* fall through */
        case CASTRO_AMD_OP_COPY:
        case CASTRO_AMD_OP_FLUXREG_CRSE_INIT:
            if (!fab_contains(&o.src, o.lo, o.hi)) return CASTRO_AMD_ERR_ARG;
            break;
        case CASTRO_AMD_OP_FLUXREG_FINE_ADD: {
            if (o.dir < 0 || o.dir > 2) return CASTRO_AMD_ERR_ARG;
            int flo[3], fhi[3];
            for (int d = 0; d < 3; ++d) { flo[d] = 2 * o.lo[d]; fhi[d] = (d == o.dir) ? 2 * o.hi[d] : 2 * o.hi[d] + 1; }
            if (!fab_contains(&o.src, flo, fhi)) return CASTRO_AMD_ERR_ARG;
            break; }
        default:
            return CASTRO_AMD_ERR_ARG;
        }
        D[r] = to_dfab(&o.dst); X[r] = to_dfab(&o.src);
        Y[r] = (o.kind == CASTRO_AMD_OP_LINCOMB) ? to_dfab(&o.src2) : X[r];
        for (int d = 0; d < 3; ++d) { lo[3 * r + d] = o.lo[d]; hi[3 * r + d] = o.hi[d]; }
        kind[r] = o.kind; dir[r] = o.dir; side[r] = o.side; ncomp[r] = o.ncomp; a[r] = o.a; b[r] = o.b;
    }
    hipSetDevice(c->device);
    DevParams P;
    if (params) P = to_devparams(params);
    return launch_fab_ops(nops, D.data(), X.data(), Y.data(), lo.data(), hi.data(), kind.data(), dir.data(), side.data(), ncomp.data(),
                          a.data(), b.data(), (hipStream_t)stream, &c->prof, params ? &P : nullptr, &c->ops_arena);
}

int castro_amd_ctu_hydro_mf(castro_amd_ctx* const* ctxs, void* const* streams, int nctx,
                            const castro_amd_hydro_box* boxes, int nboxes,
                            const castro_amd_geom* geom, const castro_amd_params* params,
                            double time, double dt, const castro_amd_hydro_opts* opts, void* stream)
{
    if (!ctxs || !streams || nctx < 1 || nboxes < 0 || (nboxes > 0 && !boxes) || !geom || !params || !opts) return CASTRO_AMD_ERR_ARG;
    for (int k = 0; k < nctx; ++k) if (!ctxs[k]) return CASTRO_AMD_ERR_ARG;
    if (nboxes == 0) return CASTRO_AMD_OK;
    hipStream_t main_s = (hipStream_t)stream;
    // The whole level as ONE grid per kernel (default options, no source terms: launch_ctu_hydro_level): every box gets
    // its own scratch inside the first context's arena, the per-box arguments travel in a device table, 8 launches on the
    // caller's stream whatever the number of boxes.  CASTRO_AMD_LEVEL_GRID=0 keeps the per-box launches on nctx streams.
    {
        static const bool level_grid = [] { const char* e = std::getenv("CASTRO_AMD_LEVEL_GRID"); return !e || std::atoi(e) != 0; }();
        bool ok = level_grid && nboxes >= 2;
        // traced source terms (round 6): every box with its source FAB or none; a context that carries a source corrector
        // (castro.source_term_predictor = 1: one per box and context) stays box by box
        const bool with_src = boxes[0].src.p != nullptr;
        for (int i = 0; i < nboxes && ok; ++i) ok = (boxes[i].src.p != nullptr) == with_src;
        if (with_src && (params->source_term_predictor == 1 || ctxs[0]->src_corr.p || opts->sborder_clean_ntimes > 0 || opts->clean_ntimes > 0)) ok = false;
        DevParams devP = to_devparams(params);
        devP.dtp = opts->d_dt;
        if (ok && level_launch_supported(devP, opts->flags, with_src)) {
            castro_amd_ctx* c0 = ctxs[0];
            hipSetDevice(c0->device);
            std::vector<PreparedBox> pb((size_t)nboxes);
            std::vector<size_t> need((size_t)nboxes);
            for (int i = 0; i < nboxes; ++i) {
                const castro_amd_hydro_box& b = boxes[i];
                int rc = prepare_box(c0, b.bxlo, b.bxhi, b.vbxlo, b.vbxhi, &b.Sborder, &b.src, &b.S_new, b.flux, b.mass_flux, b.qe,
                                     geom, params, opts, pb[(size_t)i]);
                if (rc != CASTRO_AMD_OK) return rc;
                need[(size_t)i] = (size_t)pb[(size_t)i].t.NC * (size_t)kPlanes;
            }
            // Scratch of a level-wide launch: kPlanes planes (1.26 KB) per ghosted zone of EVERY box of the launch at once, where
            // the box-by-box path needs the largest box only.  So the level goes out in chunks of consecutive boxes whose scratch
            // fits a byte budget (CASTRO_AMD_LEVEL_SCRATCH_GB, default 32; a single box larger than that is a chunk of its own):
            // 8 launches per chunk, the chunks one after the other on the caller's stream in the same arena.  If the arena cannot
            // be had at all the call falls through to the box-by-box path below, which needs the largest box only.
            const double budget_gb = [] { const char* e = std::getenv("CASTRO_AMD_LEVEL_SCRATCH_GB"); const double v = e ? std::atof(e) : 32.0; return v > 0.0 ? v : 32.0; }();   // read per call: a host may change it between levels
            const size_t budget = (size_t)(budget_gb * 1073741824.0 / sizeof(double));
            std::vector<int> first;                       // first box of every chunk
            size_t cur = 0, largest = 0;
            for (int i = 0; i < nboxes; ++i) {
                if (i == 0 || cur + need[(size_t)i] > budget) {
                    first.push_back(i);
                    cur = 0;
                }
                cur += need[(size_t)i];
                if (cur > largest) largest = cur;
            }
            first.push_back(nboxes);
            bool have_arena = largest <= c0->arena_doubles;
            if (!have_arena) {
                if (c0->arena) { hipDeviceSynchronize(); hipFree(c0->arena); c0->arena = nullptr; c0->arena_doubles = 0; }
                if (hipMalloc(&c0->arena, largest * sizeof(double)) == hipSuccess) { c0->arena_doubles = largest; have_arena = true; }
                else { (void)hipGetLastError(); c0->arena = nullptr; }
            }
            if (have_arena) {
                const DevGeom dg = to_devgeom(geom);
                for (size_t ch = 0; ch + 1 < first.size(); ++ch) {
                    const int i0 = first[ch], i1 = first[ch + 1];
                    if (i1 - i0 == 1 && first.size() > 2) {
                        // a chunk of one box: the ordinary call (its own kernels need no table), same arena, same stream
                        const castro_amd_hydro_box& b = boxes[i0];
                        const int rc = castro_amd_ctu_hydro_fab_ex(c0, b.bxlo, b.bxhi, b.vbxlo, b.vbxhi, &b.Sborder, &b.src, &b.S_new,
                                                                   b.flux, b.mass_flux, b.qe, geom, params, time, dt, opts, main_s);
                        if (rc != CASTRO_AMD_OK) return rc;
                        continue;
                    }
                    std::vector<LevelBoxDesc> lb((size_t)(i1 - i0));
                    double* p = c0->arena;
                    for (int i = i0; i < i1; ++i) {
                        const PreparedBox& B = pb[(size_t)i];
                        LevelBoxDesc& L = lb[(size_t)(i - i0)];
                        L.t = B.t;
                        p = carve_scratch(p, B.t, false, L.S);
                        L.U = B.dS; L.Unew = B.dN; L.Src = B.dSrc;
                        for (int d = 0; d < 3; ++d) { L.fl[d] = B.dF[d]; L.mass[d] = B.dM[d]; L.qe[d] = B.dQ[d]; L.acc_hi[d] = B.acc_hi[d]; }
                    }
                    const int rc = launch_ctu_hydro_level(i1 - i0, lb.data(), &c0->level_arena, dg, devP, dt, opts->flags, c0->d_status,
                                                          main_s, &c0->prof, opts->clean_ntimes, opts->d_out, opts->sborder_clean_ntimes);
                    if (rc != 0) return rc;
                }
                return CASTRO_AMD_OK;
            }
            // no arena: box by box
        }
    }
    const int used = nboxes < nctx ? nboxes : nctx;
    const bool forked = !(used == 1 && (hipStream_t)streams[0] == main_s);
    castro_amd_ctx* c0 = ctxs[0];
    hipSetDevice(c0->device);
    if (forked) {
        if (!c0->mf_fork && hipEventCreateWithFlags(&c0->mf_fork, hipEventDisableTiming) != hipSuccess) return CASTRO_AMD_ERR_HIP;
        hipEventRecord(c0->mf_fork, main_s);
        for (int k = 0; k < used; ++k)
            if ((hipStream_t)streams[k] != main_s) hipStreamWaitEvent((hipStream_t)streams[k], c0->mf_fork, 0);
    }
    int rc = CASTRO_AMD_OK;
    for (int i = 0; i < nboxes && rc == CASTRO_AMD_OK; ++i) {
        const castro_amd_hydro_box& b = boxes[i];
        rc = castro_amd_ctu_hydro_fab_ex(ctxs[i % nctx], b.bxlo, b.bxhi, b.vbxlo, b.vbxhi, &b.Sborder, &b.src, &b.S_new,
                                         b.flux, b.mass_flux, b.qe, geom, params, time, dt, opts, streams[i % nctx]);
    }
    if (forked) {
        // join even after an error: the caller's stream must not run ahead of launches already made
        for (int k = 0; k < used; ++k) {
            if ((hipStream_t)streams[k] == main_s) continue;
            castro_amd_ctx* ck = ctxs[k];
            if (!ck->mf_join && hipEventCreateWithFlags(&ck->mf_join, hipEventDisableTiming) != hipSuccess) {
                // no event for this stream: join it the slow way (never under capture: the events exist after the first
                // eager call), record the error and keep joining the others
                hipStreamSynchronize((hipStream_t)streams[k]);
                ck->mf_join = nullptr;
                if (rc == CASTRO_AMD_OK) rc = CASTRO_AMD_ERR_HIP;
                continue;
            }
            hipEventRecord(ck->mf_join, (hipStream_t)streams[k]);
            hipStreamWaitEvent(main_s, ck->mf_join, 0);
        }
    }
    return rc;
}

int castro_amd_fillpatch_shell_fab(castro_amd_ctx* c, const castro_amd_fab* crse, const castro_amd_fab* fine,
                                   const int vlo[3], const int vhi[3], int ngrow, const castro_amd_params* params,
                                   int clean_ntimes, void* stream)
{
    if (!c || !params || !fab_ok(crse, NUM_STATE) || !fab_ok(fine, NUM_STATE) || ngrow < 1 || clean_ntimes < 0) return CASTRO_AMD_ERR_ARG;
    if (crse->ncomp != NUM_STATE || fine->ncomp != NUM_STATE) return CASTRO_AMD_ERR_ARG;
    int glo[3], ghi[3], clo[3], chi[3];
    for (int d = 0; d < 3; ++d) {
        if (vhi[d] < vlo[d]) return CASTRO_AMD_ERR_ARG;
        glo[d] = vlo[d] - ngrow; ghi[d] = vhi[d] + ngrow;
        clo[d] = (glo[d] >= 0 ? glo[d] / 2 : -((-glo[d] + 1) / 2)) - 1;
        chi[d] = (ghi[d] >= 0 ? ghi[d] / 2 : -((-ghi[d] + 1) / 2)) + 1;
    }
    if (!fab_contains(fine, glo, ghi) || !fab_contains(crse, clo, chi)) return CASTRO_AMD_ERR_ARG;
    hipSetDevice(c->device);
    return launch_fillpatch_shell(to_dfab(crse), to_dfab(fine), vlo, vhi, ngrow, to_devparams(params), clean_ntimes,
                                  (hipStream_t)stream, &c->prof);
}

int castro_amd_avgdown_fab(castro_amd_ctx* c, const castro_amd_fab* fine, const castro_amd_fab* crse,
                           const int lo[3], const int hi[3], int ncomp, void* stream)
{
    if (!c || ncomp < 1 || !fab_ok(crse, ncomp) || !fab_ok(fine, ncomp) || !fab_contains(crse, lo, hi)) return CASTRO_AMD_ERR_ARG;
    int flo[3], fhi[3];
    for (int d = 0; d < 3; ++d) { flo[d] = 2 * lo[d]; fhi[d] = 2 * hi[d] + 1; }
    if (!fab_contains(fine, flo, fhi)) return CASTRO_AMD_ERR_ARG;
    hipSetDevice(c->device);
    return launch_avgdown(to_dfab(fine), to_dfab(crse), lo, hi, ncomp, (hipStream_t)stream, &c->prof);
}

int castro_amd_fluxreg_crse_init_fab(castro_amd_ctx* c, const castro_amd_fab* reg, const castro_amd_fab* crse_flux,
                                     const int lo[3], const int hi[3], int ncomp, double mult, void* stream)
{
    if (!c || ncomp < 1 || !fab_ok(reg, ncomp) || !fab_ok(crse_flux, ncomp)) return CASTRO_AMD_ERR_ARG;
    if (!fab_contains(reg, lo, hi) || !fab_contains(crse_flux, lo, hi)) return CASTRO_AMD_ERR_ARG;
    hipSetDevice(c->device);
    return launch_fluxreg(to_dfab(reg), to_dfab(crse_flux), lo, hi, 0, ncomp, mult, 0, (hipStream_t)stream, &c->prof);
}

int castro_amd_fluxreg_fine_add_fab(castro_amd_ctx* c, const castro_amd_fab* reg, const castro_amd_fab* fine_flux,
                                    const int lo[3], const int hi[3], int dir, int ncomp, double mult, void* stream)
{
    if (!c || ncomp < 1 || dir < 0 || dir > 2 || !fab_ok(reg, ncomp) || !fab_ok(fine_flux, ncomp) || !fab_contains(reg, lo, hi))
        return CASTRO_AMD_ERR_ARG;
    int flo[3], fhi[3];
    for (int d = 0; d < 3; ++d) { flo[d] = 2 * lo[d]; fhi[d] = (d == dir) ? 2 * hi[d] : 2 * hi[d] + 1; }
    if (!fab_contains(fine_flux, flo, fhi)) return CASTRO_AMD_ERR_ARG;
    hipSetDevice(c->device);
    return launch_fluxreg(to_dfab(reg), to_dfab(fine_flux), lo, hi, dir, ncomp, mult, 1, (hipStream_t)stream, &c->prof);
}

int castro_amd_reflux_fab(castro_amd_ctx* c, const castro_amd_fab* state, const castro_amd_fab* reg,
                          const int lo[3], const int hi[3], int dir, int side, int ncomp, double vol, void* stream)
{
    if (!c || ncomp < 1 || dir < 0 || dir > 2 || side < 0 || side > 1 || !fab_ok(reg, ncomp) || !fab_ok(state, ncomp)) return CASTRO_AMD_ERR_ARG;
    if (!fab_contains(reg, lo, hi)) return CASTRO_AMD_ERR_ARG;
    int zlo[3] = { lo[0], lo[1], lo[2] }, zhi[3] = { hi[0], hi[1], hi[2] };
    if (side == 0) { zlo[dir] -= 1; zhi[dir] -= 1; }
    if (!fab_contains(state, zlo, zhi)) return CASTRO_AMD_ERR_ARG;
    hipSetDevice(c->device);
    return launch_reflux(to_dfab(state), to_dfab(reg), lo, hi, dir, side, ncomp, vol, (hipStream_t)stream, &c->prof);
}

int castro_amd_error_tag_fab(castro_amd_ctx* c, const castro_amd_fab* field, int comp, const castro_amd_fab* tags,
                             const int lo[3], const int hi[3], int kind, double value, void* stream)
{
    if (!c || !field || !field->p || !tags || !tags->p || comp < 0 || comp >= field->ncomp || kind < 0 || kind > 3) return CASTRO_AMD_ERR_ARG;
    int glo[3], ghi[3];
    const int g1 = kind >= 2 ? 1 : 0;
    for (int d = 0; d < 3; ++d) { glo[d] = lo[d] - g1; ghi[d] = hi[d] + g1; }
    if (!fab_contains(field, glo, ghi) || !fab_contains(tags, lo, hi)) return CASTRO_AMD_ERR_ARG;
    hipSetDevice(c->device);
    return launch_error_tag(to_dfab(field), comp, to_dfab(tags), lo, hi, kind, value, (hipStream_t)stream, &c->prof);
}

int castro_amd_lincomb_fab(castro_amd_ctx* c, const castro_amd_fab* dst, double a, const castro_amd_fab* x, double b,
                           const castro_amd_fab* y, int ncomp, const int lo[3], const int hi[3], void* stream)
{
    if (!c || ncomp < 1 || !fab_ok(dst, ncomp) || !fab_ok(x, ncomp) || !fab_ok(y, ncomp)) return CASTRO_AMD_ERR_ARG;
    if (!fab_contains(dst, lo, hi) || !fab_contains(x, lo, hi) || !fab_contains(y, lo, hi)) return CASTRO_AMD_ERR_ARG;
    hipSetDevice(c->device);
    return launch_lincomb(to_dfab(dst), to_dfab(x), to_dfab(y), lo, hi, a, b, ncomp, (hipStream_t)stream, &c->prof);
}

int castro_amd_derive_fab(castro_amd_ctx* c, int which, const castro_amd_fab* state, const castro_amd_fab* der, int dcomp,
                          const int lo[3], const int hi[3], const castro_amd_geom* geom, const castro_amd_params* params,
                          const double center[3], void* stream)
{
    if (!c || !state || !state->p || !der || !der->p || !geom || !params || !center) return CASTRO_AMD_ERR_ARG;
    if (which < 0 || which >= CASTRO_AMD_DER_COUNT || state->ncomp != NUM_STATE) return CASTRO_AMD_ERR_ARG;
    if (dcomp < 0 || dcomp >= der->ncomp || !fab_contains(der, lo, hi)) return CASTRO_AMD_ERR_ARG;
    if (geom->coord != 0) return CASTRO_AMD_ERR_UNSUPPORTED;
    int slo[3], shi[3];
    const int g1 = (which == CASTRO_AMD_DER_MAGVORT || which == CASTRO_AMD_DER_DIVU) ? 1 : 0;   // grow_box_by_one
    for (int d = 0; d < 3; ++d) { slo[d] = lo[d] - g1; shi[d] = hi[d] + g1; }
    if (!fab_contains(state, slo, shi)) return CASTRO_AMD_ERR_ARG;
    hipSetDevice(c->device);
    return launch_derive(which, to_dfab(state), to_dfab(der), dcomp, lo, hi, geom->dx, geom->problo,
                         to_devparams(params), center, (hipStream_t)stream, &c->prof);
}

int castro_amd_bc_fill_fab(castro_amd_ctx* c, const castro_amd_fab* state, const castro_amd_geom* geom, void* stream)
{
    if (!c || !state || !state->p || !geom) return CASTRO_AMD_ERR_ARG;
    hipSetDevice(c->device);
    return launch_bc_fill(to_dfab(state), state->lo, state->hi, state->ncomp, to_devgeom(geom),
                          geom->lo_bc, geom->hi_bc, (hipStream_t)stream, &c->prof);
}

int castro_amd_copy_fab(castro_amd_ctx* c, const castro_amd_fab* dst, const castro_amd_fab* src,
                        const int lo[3], const int hi[3], void* stream)
{
    if (!c || !dst || !src || !dst->p || !src->p || dst->ncomp != src->ncomp) return CASTRO_AMD_ERR_ARG;
    if (!fab_contains(dst, lo, hi) || !fab_contains(src, lo, hi)) return CASTRO_AMD_ERR_ARG;
    hipSetDevice(c->device);
    return launch_copy(to_dfab(dst), to_dfab(src), lo, hi, src->ncomp, (hipStream_t)stream, &c->prof);
}

int castro_amd_pack_fab(castro_amd_ctx* c, const castro_amd_fab* fab, const int lo[3], const int hi[3],
                        double* buf, void* stream)
{
    if (!c || !fab || !fab->p || !buf || !fab_contains(fab, lo, hi)) return CASTRO_AMD_ERR_ARG;
    hipSetDevice(c->device);
    return launch_pack(to_dfab(fab), lo, hi, fab->ncomp, buf, 0, (hipStream_t)stream, &c->prof);
}

int castro_amd_unpack_fab(castro_amd_ctx* c, const castro_amd_fab* fab, const int lo[3], const int hi[3],
                          const double* buf, void* stream)
{
    if (!c || !fab || !fab->p || !buf || !fab_contains(fab, lo, hi)) return CASTRO_AMD_ERR_ARG;
    hipSetDevice(c->device);
    return launch_pack(to_dfab(fab), lo, hi, fab->ncomp, const_cast<double*>(buf), 1, (hipStream_t)stream, &c->prof);
}

static int pack_regions(castro_amd_ctx* c, const castro_amd_fab* fab, int nregions, const int* lo, const int* hi,
                        const long long* offsets, double* buf, int unpack, void* stream)
{
    if (!c || !fab || !fab->p || !buf || !lo || !hi || !offsets || nregions < 0 || nregions > CASTRO_AMD_MAX_REGIONS)
        return CASTRO_AMD_ERR_ARG;
    for (int r = 0; r < nregions; ++r)
        if (!fab_contains(fab, lo + 3 * r, hi + 3 * r) || offsets[r] < 0) return CASTRO_AMD_ERR_ARG;
    hipSetDevice(c->device);
    return launch_pack_regions(to_dfab(fab), nregions, lo, hi, offsets, fab->ncomp, buf, unpack, (hipStream_t)stream, &c->prof);
}

int castro_amd_pack_regions_fab(castro_amd_ctx* c, const castro_amd_fab* fab, int nregions, const int* lo, const int* hi,
                                const long long* offsets, double* buf, void* stream)
{ return pack_regions(c, fab, nregions, lo, hi, offsets, buf, 0, stream); }

int castro_amd_unpack_regions_fab(castro_amd_ctx* c, const castro_amd_fab* fab, int nregions, const int* lo, const int* hi,
                                  const long long* offsets, const double* buf, void* stream)
{ return pack_regions(c, fab, nregions, lo, hi, offsets, const_cast<double*>(buf), 1, stream); }

// Exec/hydro_tests/Sedov/problem_initialize.H:8-113 (host part) + the per-zone kernel
int castro_amd_sedov_init_fab(castro_amd_ctx* c, const castro_amd_fab* state, const int lo[3], const int hi[3],
                              const castro_amd_geom* geom, const castro_amd_params* params,
                              double r_init, double p_ambient, double exp_energy, double dens_ambient,
                              int nsub, void* stream)
{
    if (!c || !state || !state->p || !geom || !params || state->ncomp != NUM_STATE) return CASTRO_AMD_ERR_ARG;
    if (!fab_contains(state, lo, hi)) return CASTRO_AMD_ERR_ARG;
    double center[3];
    for (int n = 0; n < 3; ++n) center[n] = 0.5 * (geom->problo[n] + geom->probhi[n]);
    // eos(eos_input_rp)
    const double g = params->eos_gamma;
    const double e_ambient = p_ambient / ((g - 1.0) * dens_ambient);
    const double mu = 1.0 / (1.0 * (1.0 / params->abar));          // xn = 1 (problem_initialize.H:25-29)
    const double temp_ambient = (g - 1.0) * e_ambient * (mu * M_U) / K_B;
    const double vctr = (4.0 / 3.0) * M_PI * r_init * r_init * r_init;
    const double e_exp = exp_energy / vctr / dens_ambient;
    hipSetDevice(c->device);
    return launch_sedov_init(to_dfab(state), lo, hi, to_devparams(params), geom->dx, geom->problo, center,
                             r_init, e_exp, e_ambient, temp_ambient, dens_ambient, nsub, (hipStream_t)stream, &c->prof);
}

int castro_amd_sod_init_fab(castro_amd_ctx* c, const castro_amd_fab* state, const int lo[3], const int hi[3],
                            const castro_amd_geom* geom, const castro_amd_params* params,
                            double rho_l, double u_l, double p_l, double rho_r, double u_r, double p_r,
                            int idir, double frac, void* stream)
{
    if (!c || !state || !state->p || !geom || !params || state->ncomp != NUM_STATE) return CASTRO_AMD_ERR_ARG;
    if (idir < 1 || idir > 3 || !fab_contains(state, lo, hi)) return CASTRO_AMD_ERR_ARG;
    const double g = params->eos_gamma;
    const double split = frac * (geom->problo[idir - 1] + geom->probhi[idir - 1]);
    const double e_l = p_l / ((g - 1.0) * rho_l), e_r = p_r / ((g - 1.0) * rho_r);
    const double mu = 1.0 / (1.0 * (1.0 / params->abar));          // xn = 1
    const double T_l = (g - 1.0) * e_l * (mu * M_U) / K_B;
    const double T_r = (g - 1.0) * e_r * (mu * M_U) / K_B;
    hipSetDevice(c->device);
    return launch_sod_init(to_dfab(state), lo, hi, geom->dx, geom->problo, split, idir - 1,
                           rho_l, u_l, rho_l * e_l, T_l, rho_r, u_r, rho_r * e_r, T_r, (hipStream_t)stream, &c->prof);
}

int castro_amd_ctx_set_source_corrector(castro_amd_ctx* c, const castro_amd_fab* corr)
{
    if (!c) return CASTRO_AMD_ERR_ARG;
    if (corr && corr->p) {
        if (corr->ncomp < 7) return CASTRO_AMD_ERR_ARG;
        c->src_corr = *corr;
    } else {
        c->src_corr.p = nullptr;
    }
    return CASTRO_AMD_OK;
}

int castro_amd_ctx_poison_scratch(castro_amd_ctx* c, void* stream)
{
    if (!c) return CASTRO_AMD_ERR_ARG;
    if (!c->arena) return CASTRO_AMD_OK;
    hipSetDevice(c->device);
    // all-ones bytes are a NaN in every double
    return hipMemsetAsync(c->arena, 0xFF, c->arena_doubles * sizeof(double), (hipStream_t)stream) == hipSuccess
        ? CASTRO_AMD_OK : CASTRO_AMD_ERR_HIP;
}

int castro_amd_ctx_profile(castro_amd_ctx* c, int enable)
{
    if (!c) return CASTRO_AMD_ERR_ARG;
    c->prof.enabled = enable != 0;
    return CASTRO_AMD_OK;
}

int castro_amd_ctx_profile_count(castro_amd_ctx* c)
{
    if (!c) return 0;
    prof_collect(&c->prof);
    return (int)c->prof.recs.size();
}

int castro_amd_ctx_profile_get(castro_amd_ctx* c, int idx, char* name, int name_len, double* total_ms, long long* launches)
{
    if (!c || idx < 0 || idx >= (int)c->prof.recs.size()) return CASTRO_AMD_ERR_ARG;
    const auto& r = c->prof.recs[idx];
    if (name && name_len > 0) { std::strncpy(name, r.name.c_str(), name_len - 1); name[name_len - 1] = 0; }
    if (total_ms) *total_ms = r.total_ms;
    if (launches) *launches = r.launches;
    return CASTRO_AMD_OK;
}

void castro_amd_ctx_profile_reset(castro_amd_ctx* c)
{
    if (!c) return;
    prof_collect(&c->prof);
    for (auto& r : c->prof.recs) { r.total_ms = 0.0; r.launches = 0; }
}

} // extern "C"
