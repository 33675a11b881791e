// ctu_kernels.h -- host/device shared descriptors of the HIP CTU hydro path.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "hydro_device.h"

struct castro_amd_rotation;      // include/castro_hydro_amd.h
struct castro_amd_geom;

namespace cad {

// one tile of work: bx = [lo,hi]; every scratch array is indexed on grow(bx,4)
struct Tile {
    int lo[3], hi[3];
    int glo[3];
    int NX, NY, NZ;
    long NC;       // doubles per component plane (>= NX*NY*NZ, padded for alignment)
};

// caller-owned FArrayBox on the device
struct DFab {
    double* p;
    int lo[3];
    long sy, sz, sn;
};

struct DevGeom {
    double dx[3];
    int domlo[3], domhi[3];
    int wall_lo[3], wall_hi[3];   // Symmetry / SlipWall / NoSlipWall: zero normal flux (riemann.cpp:53-59)
    int sym_lo[3], sym_hi[3];     // Symmetry only: PLM reflecting treatment (trace_plm.cpp:39-40, Castro_ctu.cpp:293-294)
};

struct DevScratch {
    double* Q;        // NPRIM planes
    double* DIV;      // 1 plane
    double* SHK;      // 1 plane (hybrid Riemann only)
    double* SRCQ;     // 6 planes (rho,u,v,w,p,rhoe primitive sources; only with a source FAB)
    double* QM[3];    // NEDGE planes each
    double* QP[3];
    double* F1[3];    // NF1 planes each
    double* F2[6];    // NF1 planes each, slot f2_slot(N,T)
    double* FL[3];    // NFIN planes each
    double* F1E[3];   // (rho e) flux of the first solves, 1 plane each -- written and read only with transverse_reset_rhoe = 1
    double* F2E[6];   // the same for the transverse-stage solves
};

// hipEvent-based per-kernel timing (enabled on request only: it serialises nothing, the
// events are recorded on the same stream as the kernels)
struct Profiler {
    struct Rec { std::string name; double total_ms = 0.0; long long launches = 0; };
    bool enabled = false;
    std::vector<Rec> recs;
    struct Pending { int rec; hipEvent_t e0, e1; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
    int cur = -1;
    hipEvent_t cur_e0{}, cur_e1{};
};
void prof_begin(Profiler* p, const char* name, hipStream_t s);
void prof_end(Profiler* p, hipStream_t s);
void prof_collect(Profiler* p);

// what a call carries besides its arrays: in-place cleaning of Sborder inside k_ctoprim, the context's side stream
struct LaunchAux {
    int sb_clean = 0;
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int bc_lo[3] = { 0, 0, 0 }, bc_hi[3] = { 0, 0, 0 };     // castro_amd_geom::lo_bc / hi_bc (CASTRO_AMD_BC_FILL)
};

int launch_ctu_hydro(const Tile& t, const DevScratch& S, const DFab& Sborder, const DFab& Src, const DFab& Snew,
                     const DFab fluxes[3], const DFab mass[3], const DFab qe[3],
                     const DevGeom& g, const DevParams& P, double dt, int flags, const int acc_hi[3],
                     int* d_status, hipStream_t stream, Profiler* prof, int clean_ntimes, double* red, const DFab& SrcCorr,
                     const LaunchAux& aux);

// device buffer for operation tables longer than a kernel argument holds (owned by the context)
struct FabOpsArena { void* p = nullptr; size_t bytes = 0; };

// One box of a level-wide launch (castro_amd_ctu_hydro_mf): its tile, its own scratch, the caller's arrays
struct LevelBoxDesc {
    Tile t;
    DevScratch S;
    DFab U, Unew, fl[3], mass[3], qe[3];
    int acc_hi[3];
    DFab Src;              // old-time source FAB to be traced (p == nullptr: none); all boxes of a launch alike
};
// default options only (PPM, CGF solver, no staging, the default kernel forms; traced source terms without a predictor): else box by box
bool level_launch_supported(const DevParams& P, int flags, bool with_src = false);
int launch_ctu_hydro_level(int nbox, const LevelBoxDesc* boxes, FabOpsArena* table, const DevGeom& g, const DevParams& P, double dt,
                           int flags, int* d_status, hipStream_t stream, Profiler* prof, int clean_ntimes, double* red, int sb_clean);

// auxiliary per-FAB kernels (aux_kernels.hip)
int launch_clean_state(const DFab& U, const int lo[3], const int hi[3], const DevParams& P, int ntimes,
                       hipStream_t stream, Profiler* prof);
int launch_clean_state_reduce(const DFab& U, const int lo[3], const int hi[3], const DevGeom& g, const DevParams& P,
                              int ntimes, double* d_out, hipStream_t stream, Profiler* prof);
int launch_step_control(double* red, double* ctl, double cfl, double change_max, double small_dens, double max_dt,
                        double fixed_dt, double stop_time, int retry_form, hipStream_t stream, Profiler* prof);
int launch_estdt(const DFab& U, const int lo[3], const int hi[3], const DevGeom& g, const DevParams& P,
                 double* d_out, hipStream_t stream, Profiler* prof);
int launch_old_grav_source(const DFab& U, const DFab& SRC, const int lo[3], const int hi[3], const double grav[3],
                           int type, double dt, hipStream_t stream, Profiler* prof);
int launch_new_grav_source(const DFab& UO, const DFab& UN, const DFab& SRC, const DFab M[3], const int lo[3], const int hi[3],
                           const double grav[3], int type, double dt, const double dx[3], hipStream_t stream, Profiler* prof);
int launch_old_rot_source(const DFab& U, const DFab& SRC, const int lo[3], const int hi[3], const ::castro_amd_rotation* r,
                          const ::castro_amd_geom* g, double dt, hipStream_t stream, Profiler* prof);
int launch_new_rot_source(const DFab& UO, const DFab& UN, const DFab& SRC, const DFab M[3], const int lo[3], const int hi[3],
                          const ::castro_amd_rotation* r, const ::castro_amd_geom* g, double dt, hipStream_t stream, Profiler* prof);
// one box of castro_amd_sources_mf: states, Source_Type FAB (nsc components), mass fluxes; [lo, lo + n): the zones of the source FAB
// (thread range), [vlo, vhi]: the valid zones
struct SrcBoxDev { DFab So, Sn, Src, M0, M1, M2; int lo[3], n[3]; int vlo[3], vhi[3]; int nsc; };
int launch_sources_apply(int stage, int nbox, const SrcBoxDev* boxes, const double* grav, int grav_type, const ::castro_amd_rotation* rot,
                         const ::castro_amd_geom* geom, const DevParams& P, double dt, int ntimes, FabOpsArena* arena,
                         hipStream_t stream, Profiler* prof);
int launch_saxpy(const DFab& D, const DFab& S, const int lo[3], const int hi[3], double a, int ncomp,
                 hipStream_t stream, Profiler* prof);
int launch_fab_ops(int nops, const DFab* D, const DFab* X, const DFab* Y, const int* lo, const int* hi, const int* kind,
                   const int* dir, const int* side, const int* ncomp, const double* a, const double* b, hipStream_t stream, Profiler* prof,
                   const DevParams* P = nullptr, FabOpsArena* arena = nullptr);
int launch_apply_source(const DFab& D, const DFab& B, const DFab& S, const int lo[3], const int hi[3], double a, int nsrc,
                        const DevParams& P, int ntimes, hipStream_t stream, Profiler* prof);
int launch_cc_interp(const DFab& C, const DFab& F, const int lo[3], const int hi[3], int ncomp, hipStream_t stream, Profiler* prof);
int launch_fillpatch_shell(const DFab& C, const DFab& F, const int vlo[3], const int vhi[3], int ng, const DevParams& P, int ntimes,
                           hipStream_t stream, Profiler* prof);
int launch_avgdown(const DFab& F, const DFab& C, const int lo[3], const int hi[3], int ncomp, hipStream_t stream, Profiler* prof);
int launch_fluxreg(const DFab& R, const DFab& X, const int lo[3], const int hi[3], int dir, int ncomp, double mult, int mode,
                   hipStream_t stream, Profiler* prof);
int launch_reflux(const DFab& U, const DFab& R, const int lo[3], const int hi[3], int dir, int side, int ncomp, double vol,
                  hipStream_t stream, Profiler* prof);
int launch_error_tag(const DFab& Q, int comp, const DFab& T, const int lo[3], const int hi[3], int kind, double value,
                     hipStream_t stream, Profiler* prof);
int launch_lincomb(const DFab& D, const DFab& X, const DFab& Y, const int lo[3], const int hi[3], double a, double bb, int ncomp,
                   hipStream_t stream, Profiler* prof);
int launch_derive(int which, const DFab& U, const DFab& D, int dcomp, const int lo[3], const int hi[3],
                  const double dx[3], const double problo[3], const DevParams& P, const double center[3],
                  hipStream_t stream, Profiler* prof);
int launch_bc_fill(const DFab& U, const int flo[3], const int fhi[3], int ncomp, const DevGeom& g,
                   const int lo_bc[3], const int hi_bc[3], hipStream_t stream, Profiler* prof);
int launch_copy(const DFab& dst, const DFab& src, const int lo[3], const int hi[3], int ncomp,
                hipStream_t stream, Profiler* prof);
int launch_pack(const DFab& f, const int lo[3], const int hi[3], int ncomp, double* buf, int unpack,
                hipStream_t stream, Profiler* prof);
int launch_pack_regions(const DFab& f, int nreg, const int* lo, const int* hi, const long long* off, int ncomp, double* buf,
                        int unpack, hipStream_t stream, Profiler* prof);
int launch_sedov_init(const DFab& U, const int lo[3], const int hi[3], const DevParams& P,
                      const double dx[3], const double problo[3], const double center[3],
                      double r_init, double e_exp, double e_ambient, double temp_ambient,
                      double dens_ambient, int nsub, hipStream_t stream, Profiler* prof);
int launch_sod_init(const DFab& U, const int lo[3], const int hi[3], const double dx[3],
                    const double problo[3], double split, int idir0,
                    double rho_l, double u_l, double rhoe_l, double T_l,
                    double rho_r, double u_r, double rhoe_r, double T_r,
                    hipStream_t stream, Profiler* prof);

} // namespace cad
