/*
 * castro_hydro_amd.h -- C ABI of the MI355X-native CTU hydro path for Castro.
 *
 * Drop-in boundary.  The reference has no FFI for this
 * path: the seam is the C++ member
 *     void Castro::construct_ctu_hydro_source(amrex::Real time, amrex::Real dt)
 *     (Source/hydro/Castro_hydro.H:44, body Source/hydro/Castro_ctu_hydro.cpp:16-1528),
 * called from Castro::do_advance_ctu (Source/driver/Castro_advance_ctu.cpp:156).
 * castro_amd_ctu_hydro_fab() replaces the body of its MFIter loop
 * (Castro_ctu_hydro.cpp:130-1480) for ONE FArrayBox / tile; the other entry
 * points replace the per-FAB sweeps the driver runs either side of it.
 *
 * Conventions
 *  - Arrays are AMReX FArrayBox memory: Fortran order, i fastest, component
 *    slowest, described by (pointer, lo[3], hi[3], ncomp) exactly like the
 *    reference's legacy BL_FORT_FAB_ARG_3D convention
 *    (Source/driver/Castro_F.H:19-34).  All pointers are DEVICE pointers
 *    (hipMalloc / AMReX The_Arena() on a HIP build).  FP64 only.
 *  - State layout is the Sedov build's: NUM_STATE = 8
 *    (URHO UMX UMY UMZ UEDEN UEINT UTEMP UFS), one species.
 *  - The caller owns every array for the duration of the call AND until the
 *    stream is synchronised (the AMReX analogue is Elixir,
 *    Castro_ctu_hydro.cpp:79-82).  The callee owns its scratch (the context).
 *  - Every call is asynchronous on `stream` (a hipStream_t passed as void*;
 *    NULL = the default stream).  No call allocates or synchronises once the
 *    context has been reserved for the tile size.
 *  - Error convention: the reference aborts on CPU and is silent on GPU
 *    (advection_util.cpp:56-68); here every entry point returns an int status
 *    (0 = ok, <0 = bad argument / unsupported option) and never aborts.
 *    Data-dependent failures (rho <= 0 met in ctoprim) are latched in a device
 *    flag readable with castro_amd_ctx_status().
 *  - A context is bound to one device and must be used from one stream at a
 *    time (re-entrant per (device, stream): create one context per stream).
 *  - Thread safety: different host threads may drive different contexts
 *    concurrently (the reference calls the path from inside `#pragma omp
 *    parallel`, Castro_ctu_hydro.cpp:66-72); ONE context must not be entered
 *    by two threads at once.  The CASTRO_AMD_* environment knobs are read by
 *    castro_amd_ctx_create() into process-wide settings: create the contexts
 *    before the threads start using them, not while another thread is
 *    inside a call.
 */
#ifndef CASTRO_HYDRO_AMD_H
#define CASTRO_HYDRO_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version of this header: bumped whenever the meaning or the SIZE of anything a caller allocates changes.
 *   3: reduction vectors (`d_out`) are THREE device doubles (0.2 had two): castro_amd_ctu_hydro_clean_fab,
 *      castro_amd_ctu_hydro_fab_ex / _mf (opts.d_out), castro_amd_clean_state_reduce_fab.
 *   4: castro_amd_numerics(), castro_amd_fill_boundary*(), castro_amd_abi_version().
 *   5: CASTRO_AMD_STAGE_VALID / _REST / CASTRO_AMD_BC_FILL, castro_amd_fill_boundary_ex, castro_amd_halo_plan_wait_packed,
 *      castro_amd_halo_group_* / castro_amd_fill_boundary_group (several boxes per rank), castro_amd_berger_rigoutsos;
 *      castro_amd_hydro_opts.sborder_clean_ntimes is accepted by the staged calls; CASTRO_AMD_OP_INTERP, castro_amd_sources_mf,
 *      castro_amd_clean_state_reduce_mf, castro_amd_estdt_mf.
 * A caller checks `castro_amd_abi_version() == CASTRO_AMD_ABI_VERSION` once after loading the library; a mismatch means
 * the library was built from another revision of this header (a 0.2 caller with 2-double vectors would be written 8 bytes
 * out of bounds by a 0.3 library). */
#define CASTRO_AMD_ABI_VERSION 5
/* release of the library: castro_amd_version() is "castro_hydro_amd <this> (gfx950, numerics=...)" -- the one place it is written */
#define CASTRO_AMD_RELEASE "0.6"

#define CASTRO_AMD_NUM_STATE 8
#define CASTRO_AMD_NGDNV 4
#define CASTRO_AMD_NUM_GROW 4

#define CASTRO_AMD_OK 0
#define CASTRO_AMD_ERR_ARG (-1)
#define CASTRO_AMD_ERR_UNSUPPORTED (-2)
#define CASTRO_AMD_ERR_NOMEM (-3)
#define CASTRO_AMD_ERR_HIP (-4)

/* FArrayBox descriptor: fab.dataPtr(), fab.loVect(), fab.hiVect(), fab.nComp() [AMReX] */
typedef struct castro_amd_fab {
    double *p;
    int lo[3];
    int hi[3];
    int ncomp;
} castro_amd_fab;

/* geom.data() + phys_bc (Castro::geom, Castro::phys_bc) */
typedef struct castro_amd_geom {
    double dx[3];        /* geom.CellSize() */
    double problo[3];    /* geom.ProbLo() */
    double probhi[3];    /* geom.ProbHi() */
    int domlo[3];        /* geom.Domain().loVect() */
    int domhi[3];        /* geom.Domain().hiVect() */
    int lo_bc[3];        /* phys_bc.lo(): 0 Interior 1 Inflow 2 Outflow 3 Symmetry 4 SlipWall 5 NoSlipWall */
    int hi_bc[3];        /* phys_bc.hi() */
    int coord;           /* geom.Coord(); only 0 (Cartesian) is supported */
} castro_amd_geom;

/* the castro:: runtime parameters the path reads (Source/driver/_cpp_parameters) */
typedef struct castro_amd_params {
    int ppm_type;                 /* 1 = PPM (default), 0 = PLM */
    int riemann_solver;           /* 0 CGF (default), 1 CG, 2 HLLC */
    int use_flattening;
    int hybrid_riemann;
    int first_order_hydro;
    int cg_maxiter;
    int cg_blend;                 /* 2 degrades to "keep last iterate" like the reference's GPU build */
    int transverse_use_eos;
    int transverse_reset_density;
    int transverse_reset_rhoe;
    int ppm_temp_fix;             /* 0 (default); 2 = EOS fix of the Riemann input states; 1 is a no-op in the CTU path */
    int plm_iorder;               /* 2 (default) or 1 */
    int plm_limiter;              /* 2 = 4th-order MC (default), 1 = 2nd-order MC */
    int use_pslope;               /* 1 (default): well-balanced pressure slope in PLM */
    double difmag;
    double small_dens, small_temp, small_pres, small_ener;
    double cg_tol;
    double dual_energy_eta1, dual_energy_eta2;
    double cfl, init_shrink, change_max;
    double eos_gamma;             /* gamma-law EOS (Microphysics EOS/gamma_law) */
    double small_x;
    double T_guess;
    double abar;                  /* mean molecular weight with eos_assume_neutral = 1 */
    double pslope_cutoff_density;
    int limit_fluxes_on_small_dens;   /* 0 (default); limit_hydro_fluxes_on_small_dens, advection_util.cpp:657-903 */
    int limit_fluxes_on_large_vel;    /* 0 (default); limit_hydro_fluxes_on_large_vel, :907-1075 (needs speed_limit > 0) */
    double speed_limit;               /* 0 (default: off); also enforce_speed_limit in clean_state, Castro.cpp:3049-3092 */
    int source_term_predictor;        /* 0 (default); 1: src_to_prim adds dt/2 x source_corrector to the momentum sources
                                       * (Castro_ctu.cpp:493-497); see castro_amd_ctx_set_source_corrector */
} castro_amd_params;

typedef struct castro_amd_ctx castro_amd_ctx;

/* Fill `p` with the reference defaults for the Sedov setup (see _cpp_parameters
 * and Exec/hydro_tests/Sedov/inputs.3d.sph*) and derive small_pres/small_ener
 * like Castro_setup.cpp:222-288. */
void castro_amd_default_params(castro_amd_params *p);
void castro_amd_finalize_params(castro_amd_params *p);

/* Context = scratch arena + device status word for one (device, stream). */
int castro_amd_ctx_create(castro_amd_ctx **ctx, int device);
void castro_amd_ctx_destroy(castro_amd_ctx *ctx);
/* Pre-allocate scratch for tiles up to nx*ny*nz valid cells (hipMalloc happens
 * here, never inside the calls below once reserved). */
int castro_amd_ctx_reserve(castro_amd_ctx *ctx, int nx, int ny, int nz);
/* Bytes of scratch currently held. */
long long castro_amd_ctx_scratch_bytes(const castro_amd_ctx *ctx);
/* Synchronises `stream`, returns and clears the latched device status bits:
 * bit0 = rho <= 0 or rho < small_dens met in ctoprim (advection_util.cpp:56-68). */
int castro_amd_ctx_status(castro_amd_ctx *ctx, void *stream);
/* Castro::source_corrector (a member MultiFab of the reference, Castro_advance.cpp:292-293, filled by
 * create_source_corrector, Castro.cpp:3780-3818): NSRC = 7 components on a box containing grow(bx, 3) of the tiles that
 * follow; the hydro calls on this context read it when params->source_term_predictor == 1.  NULL (or p == NULL)
 * clears it.  The caller owns the memory. */
int castro_amd_ctx_set_source_corrector(castro_amd_ctx *ctx, const castro_amd_fab *source_corrector);
/* Fill the scratch arena with NaNs (asynchronous on `stream`): lets a caller or a test prove that a call reads
 * nothing a previous call left behind. */
int castro_amd_ctx_poison_scratch(castro_amd_ctx *ctx, void *stream);

/* flags for castro_amd_ctu_hydro_fab */
#define CASTRO_AMD_UPDATE_ADD 0      /* S_new += dt*div(F)...   (reference semantics: S_new holds a copy of Sborder) */
#define CASTRO_AMD_UPDATE_FROM_SBORDER 1 /* S_new = Sborder + ... (elides MultiFab::Copy, Castro_advance_ctu.cpp:94) */
#define CASTRO_AMD_STAGE_A 4             /* only the part that reads no ghost zone of Sborder (ctoprim on bx, tracing on
                                          * grow(bx,-3)): may run while the halo exchange fills the ghost zones */
#define CASTRO_AMD_STAGE_B 8             /* the rest, after CASTRO_AMD_STAGE_A on the same context, tile and arguments */
/* The light split for the overlap of the halo exchange (round 6; not together with CASTRO_AMD_STAGE_A / _B):
 *   CASTRO_AMD_STAGE_VALID  only ctoprim -- with the sborder_clean_ntimes applications of clean_state -- on the zones of bx:
 *                           reads and writes NO ghost zone of Sborder, so it may run on one stream while the exchange fills
 *                           the ghost zones on another (after the exchange has packed: castro_amd_halo_plan_wait_packed);
 *   CASTRO_AMD_STAGE_REST   the rest, after CASTRO_AMD_STAGE_VALID on the same context, tile and arguments: ctoprim (+ the
 *                           cleans) on the ghost shell in ONE launch, then the whole un-split update;
 *   CASTRO_AMD_BC_FILL      the call fills the zones of grow(bx, 4) outside the problem domain in a non-periodic direction
 *                           itself (what castro_amd_bc_fill_fab does: FOEXTRAP / REFLECT_ODD per geom->lo_bc / hi_bc), from
 *                           in-domain zones it has cleaned already, and writes their primitive record in the same pass:
 *                           the caller passes Sborder with its same-level ghost zones exchanged and NO physical-boundary
 *                           fill.  clean_state is zone-local and even in the momenta, so "fill, then clean" (the
 *                           reference: Castro_advance.cpp:186) equals "clean, then fill"; every zone is cleaned
 *                           sborder_clean_ntimes times exactly.  Whole-box calls only.  CASTRO_AMD_ERR_UNSUPPORTED when a
 *                           mirrored ghost layer reaches past the in-domain zones of this FAB (a box thinner than its ghost
 *                           depth at a wall: fill with castro_amd_bc_fill_fab instead).
 * CASTRO_AMD_STAGE_REST with sborder_clean_ntimes > 0 needs CASTRO_AMD_BC_FILL (a boundary fill between the two stages
 * would copy zones that are clean already into zones the shell pass cleans again). */
#define CASTRO_AMD_STAGE_VALID 16
#define CASTRO_AMD_STAGE_REST 32
#define CASTRO_AMD_BC_FILL 64
#define CASTRO_AMD_FLUX_ASSIGN 2         /* flux_out[d] = 0 + dt*area*flux instead of +=: for callers that would zero
                                          * fluxes[d] just before this (one) hydro call of the step
                                          * (Castro_advance.cpp:391-394); elides that fill and the read of the RMW */

/*
 * The hot path: replaces the MFIter-loop body of Castro::construct_ctu_hydro_source
 * (Source/hydro/Castro_ctu_hydro.cpp:130-1480) for the tile bx = [bxlo, bxhi].
 *   vbxlo/vbxhi : valid box of the FAB the tile belongs to (mfi.validbox()); decides
 *                 mfi.nodaltilebox(d) (which shared faces this tile stores). Pass bx for whole-box calls.
 *   Sborder     : in, ncomp 8, box must contain grow(bx, 4)           (Castro::Sborder)
 *   src         : in, old_source (ncomp 7, box contains grow(bx,3)); p == NULL means identically zero
 *   S_new       : in/out on bx, ncomp 8                               (get_new_data(State_Type))
 *   flux_out[d] : in/out, += dt*area*flux on nodaltilebox(d), ncomp 8 (Castro::fluxes[d]); p NULL to skip
 *   mass_flux_out[d] : out, = scaled density flux, ncomp 1            (Castro::mass_fluxes[d]); p NULL to skip
 *   qe_out[d]   : out (optional, p NULL to skip), Godunov state u,v,w,p on the same faces, ncomp 4 (qe[d])
 * Returns CASTRO_AMD_ERR_ARG for boxes / component counts that do not fit, CASTRO_AMD_ERR_UNSUPPORTED for
 * non-Cartesian coordinates and for a tile or FAB whose component plane reaches 4 GiB (the
 * kernels address a plane with 32-bit byte offsets: tile boxes beyond ~800^3 zones).
 */
int castro_amd_ctu_hydro_fab(castro_amd_ctx *ctx,
                             const int bxlo[3], const int bxhi[3],
                             const int vbxlo[3], const int vbxhi[3],
                             const castro_amd_fab *Sborder,
                             const castro_amd_fab *src,
                             const castro_amd_fab *S_new,
                             const castro_amd_fab flux_out[3],
                             const castro_amd_fab mass_flux_out[3],
                             const castro_amd_fab qe_out[3],
                             const castro_amd_geom *geom,
                             const castro_amd_params *params,
                             double time, double dt, int flags, void *stream);

/*
 * castro_amd_ctu_hydro_fab followed, zone by zone and before S_new is written, by the sequence
 * do_advance_ctu runs on the updated state when no new-time source intervenes
 * (Source/driver/Castro_advance_ctu.cpp:168-225, 386-392):
 *   d_out[1] = min(d_out[1], density of the updated zone)         (S_new.min(URHO))
 *   clean_state applied `clean_ntimes` (>= 1) times               (Castro::clean_state)
 *   d_out[2] = min(d_out[2], dx/(c+|u|) of the zone cleaned ONCE) (Castro::estdt_cfl as the validity check of
 *                                                                  do_advance_ctu sees it, :386-392)
 *   d_out[0] = min(d_out[0], dx/(c+|u|) of the zone after the LAST clean_state)   (what estTimeStep of the next coarse
 *                                                                  step sees when post_timestep's clean_state rides along)
 * The result equals castro_amd_ctu_hydro_fab + castro_amd_clean_state_reduce_fab on bx, without the
 * second pass over S_new.  clean_ntimes == 0 is exactly castro_amd_ctu_hydro_fab; d_out may be NULL
 * (no reduction) and otherwise points to 3 device doubles initialised by the caller.
 */
int castro_amd_ctu_hydro_clean_fab(castro_amd_ctx *ctx,
                                   const int bxlo[3], const int bxhi[3],
                                   const int vbxlo[3], const int vbxhi[3],
                                   const castro_amd_fab *Sborder,
                                   const castro_amd_fab *src,
                                   const castro_amd_fab *S_new,
                                   const castro_amd_fab flux_out[3],
                                   const castro_amd_fab mass_flux_out[3],
                                   const castro_amd_fab qe_out[3],
                                   const castro_amd_geom *geom,
                                   const castro_amd_params *params,
                                   double time, double dt, int flags,
                                   int clean_ntimes, double *d_out, void *stream);

/*
 * The general form of the two calls above.  Fields beyond those of castro_amd_ctu_hydro_clean_fab:
 *   sborder_clean_ntimes  > 0: Castro::clean_state is applied that many times, IN PLACE, to every zone of
 *                         grow(bx, 4) of Sborder inside the pass that reads it (k_ctoprim) -- the clean_state(S_old) of
 *                         initialize_advance plus the clean_state(Sborder) after FillPatch
 *                         (Castro_advance.cpp:311, :186) without their own sweeps.  clean_state is zone-local, and
 *                         the ghost zones of a single level are copies (or mirror images) of valid zones, so
 *                         "FillPatch the uncleaned state, then clean everything" equals the reference's "clean, FillPatch,
 *                         clean".  Only for whole-box calls (bx == vbx, one tile per FAB: overlapping tiles would clean
 *                         shared ghost zones twice) without CASTRO_AMD_STAGE_A/B; Sborder is written.  With
 *                         CASTRO_AMD_STAGE_VALID the zones of bx are cleaned, with CASTRO_AMD_STAGE_REST the ghost shell
 *                         (pass the same count to both calls).
 *                         In the `contract` build (castro_amd_numerics()), on the default-solver path and together with
 *                         clean_ntimes > 0, the temperature and species components of Sborder are neither read nor
 *                         written back: nothing downstream depends on them (one species, gamma-law gas; the fused update
 *                         recomputes both for S_new) -- the other six components are cleaned in place as described.
 */
typedef struct castro_amd_hydro_opts {
    int flags;                  /* CASTRO_AMD_UPDATE_* | CASTRO_AMD_FLUX_ASSIGN | CASTRO_AMD_STAGE_* | CASTRO_AMD_BC_FILL */
    int clean_ntimes;           /* as in castro_amd_ctu_hydro_clean_fab */
    double *d_out;              /* as in castro_amd_ctu_hydro_clean_fab (device, THREE doubles) or NULL */
    int sborder_clean_ntimes;   /* see above; 0 = Sborder is read only */
    const double *d_dt;         /* not NULL: castro_amd_step_control's DEVICE vector `ctl` -- the kernels read the time step
                                 * from ctl[CASTRO_AMD_CTL_DTHYDRO] (the `dt` argument is ignored), and when
                                 * ctl[CASTRO_AMD_CTL_STATUS] != 0 (an earlier step of the batch was rejected) the call
                                 * leaves Sborder, S_new and the flux arrays untouched */
} castro_amd_hydro_opts;
int castro_amd_ctu_hydro_fab_ex(castro_amd_ctx *ctx,
                                const int bxlo[3], const int bxhi[3],
                                const int vbxlo[3], const int vbxhi[3],
                                const castro_amd_fab *Sborder,
                                const castro_amd_fab *src,
                                const castro_amd_fab *S_new,
                                const castro_amd_fab flux_out[3],
                                const castro_amd_fab mass_flux_out[3],
                                const castro_amd_fab qe_out[3],
                                const castro_amd_geom *geom,
                                const castro_amd_params *params,
                                double time, double dt, const castro_amd_hydro_opts *opts, void *stream);

/*
 * Host-free time stepping of a single level (no retries, no sources): one thread on the device does what the host does
 * between two hydro updates -- the tail of Castro::do_advance_ctu (Castro_advance_ctu.cpp:168-216, 386-392), the
 * time / step bookkeeping of Amr::coarseTimeStep and Castro::computeNewDt (Castro.cpp:1629-1819) for the next step:
 *   red (3 doubles, as written by castro_amd_ctu_hydro_clean_fab; reset to 1e200 on exit)
 *   ctl[CASTRO_AMD_CTL_DT]     in: the step just taken; out: the next step's dt
 *                              = min(cfl * red[0], change_max * dt) clipped to stop_time (fixed_dt > 0: fixed_dt)
 *   ctl[CASTRO_AMD_CTL_DTHYDRO] the dt the hydro kernels of that step use: dt, or with use_retry != 0 the single
 *                              subcycle of Castro::subcycle_advance_ctu, (time + dt) - time (Castro_advance_ctu.cpp:463-471),
 *                              which differs from dt in the last bit now and then; the validity check uses it too
 *   ctl[CASTRO_AMD_CTL_TIME]   += dt;   ctl[CASTRO_AMD_CTL_NSTEP] += 1
 *   ctl[CASTRO_AMD_CTL_STATUS] |= 1 if red[1] < small_dens (the step would be rejected),
 *                              |= 2 if change_max * cfl * red[2] < dt (timestep validity check); sticky: once set,
 *                              time, step count and dt stop changing, so that the host finds the first failure
 *   ctl[CASTRO_AMD_CTL_HIST + n % CASTRO_AMD_CTL_NHIST] = dt of step n (for hosts that want the sequence)
 * The hydro calls of the next step take their dt from ctl[CASTRO_AMD_CTL_DTHYDRO] (castro_amd_hydro_opts.d_dt = ctl).  The host
 * reads ctl when it wants to (one synchronisation per batch of steps instead of one per step).
 */
#define CASTRO_AMD_CTL_DT 0
#define CASTRO_AMD_CTL_TIME 1
#define CASTRO_AMD_CTL_NSTEP 2
#define CASTRO_AMD_CTL_STATUS 3
#define CASTRO_AMD_CTL_RHOMIN 4
#define CASTRO_AMD_CTL_EST 5
#define CASTRO_AMD_CTL_DTHYDRO 6
#define CASTRO_AMD_CTL_HIST 8
#define CASTRO_AMD_CTL_NHIST 56
#define CASTRO_AMD_CTL_SIZE 64
int castro_amd_step_control(castro_amd_ctx *ctx, double *d_red, double *d_ctl, const castro_amd_params *params,
                            double max_dt, double fixed_dt, double stop_time, int use_retry, void *stream);

/* Castro::clean_state on one FAB region (Source/driver/Castro.cpp:4238-4278):
 * enforce_min_density, normalize_species, reset_internal_energy, computeTemp,
 * applied `ntimes` times in a row to every zone of [lo,hi] (the reference runs it
 * up to three times back to back on the same zones between two hydro updates). */
int castro_amd_clean_state_fab(castro_amd_ctx *ctx, const castro_amd_fab *state,
                               const int lo[3], const int hi[3],
                               const castro_amd_params *params, int ntimes, void *stream);

/* The post-hydro sequence of Castro::do_advance_ctu fused into one pass over the zones of [lo,hi]:
 *   d_out[1] = min(d_out[1], min density of the state AS GIVEN)   (S_new.min(URHO), Castro_advance_ctu.cpp:168)
 *   clean_state applied `ntimes` times                            (Castro_advance_ctu.cpp:221-225)
 *   d_out[2] = min(d_out[2], min dx/(c+|u|) after the first application), d_out[0] = ... after the last
 *                                                                 (estTimeStep, Castro_advance_ctu.cpp:386)
 * d_out: device pointer to 3 doubles initialised by the caller (castro_amd_estdt_fab writes [0] and [1] only). */
int castro_amd_clean_state_reduce_fab(castro_amd_ctx *ctx, const castro_amd_fab *state,
                                      const int lo[3], const int hi[3], const castro_amd_geom *geom,
                                      const castro_amd_params *params, int ntimes, double *d_out, void *stream);

/* Gravity source terms for gravity.gravity_type = "ConstantGrav" (grav[] is the same in every zone,
 * Source/gravity/Gravity.cpp:860-866), castro.grav_source_type 1..4 (default 4):
 *   castro_amd_old_gravity_source_fab  Castro::construct_old_gravity_source, Source/gravity/Castro_gravity.cpp:234-362
 *   castro_amd_new_gravity_source_fab  Castro::construct_new_gravity_source, :384-596 (needs mass_fluxes[d] of the hydro call)
 *   castro_amd_saxpy_fab               Castro::apply_source_to_state (MultiFab::Saxpy), Source/sources/Castro_sources.cpp:10-19
 * `source` has NSRC = 7 components and is accumulated into (+=), as in the reference. */
int castro_amd_old_gravity_source_fab(castro_amd_ctx *ctx, const castro_amd_fab *state, const castro_amd_fab *source,
                                      const int lo[3], const int hi[3], const double grav[3], int grav_source_type,
                                      double dt, void *stream);
int castro_amd_new_gravity_source_fab(castro_amd_ctx *ctx, const castro_amd_fab *state_old, const castro_amd_fab *state_new,
                                      const castro_amd_fab *source, const castro_amd_fab mass_fluxes[3],
                                      const int lo[3], const int hi[3], const double grav[3], int grav_source_type,
                                      double dt, const castro_amd_geom *geom, void *stream);
/* dst = base + a * src[0:nsrc] (other components copied; base may be dst) followed by clean_state x clean_ntimes, one
 * pass: MultiFab::Copy(S_new, Sborder) + apply_source_to_state + clean_state of do_advance_ctu
 * (Castro_advance_ctu.cpp:94, 127-131, 262-268).  dst, base: ncomp 8. */
int castro_amd_apply_source_fab(castro_amd_ctx *ctx, const castro_amd_fab *dst, const castro_amd_fab *base, double a,
                                const castro_amd_fab *src, int nsrc, const int lo[3], const int hi[3],
                                const castro_amd_params *params, int clean_ntimes, void *stream);
int castro_amd_saxpy_fab(castro_amd_ctx *ctx, const castro_amd_fab *dst, double a, const castro_amd_fab *src, int ncomp,
                         const int lo[3], const int hi[3], void *stream);

/* Rotation source terms in the rotating frame (castro.do_rotation, state_in_rotating_frame = 1):
 *   castro_amd_old_rotation_source_fab  Castro::rsrc,     Source/rotation/rotation_sources.cpp:9-137
 *   castro_amd_new_rotation_source_fab  Castro::corrrsrc, :140-500 (implicit Coriolis update :186-237,318-355;
 *                                       rot_source_type 4 uses mass_fluxes[d] and the rotational potential,
 *                                       Source/rotation/Rotation.H:77-95)
 * omega = 2 pi / castro.rotational_period along castro.rot_axis (Rotation.H:10-22); center = problem::center. */
typedef struct castro_amd_rotation {
    double omega[3];
    double center[3];
    int include_centrifugal;        /* castro.rotation_include_centrifugal (default 1) */
    int include_coriolis;           /* castro.rotation_include_coriolis (1) */
    int rot_source_type;            /* castro.rot_source_type 1..4 (4) */
    int implicit_rotation_update;   /* castro.implicit_rotation_update (1) */
} castro_amd_rotation;
int castro_amd_old_rotation_source_fab(castro_amd_ctx *ctx, const castro_amd_fab *state, const castro_amd_fab *source,
                                       const int lo[3], const int hi[3], const castro_amd_rotation *rot,
                                       const castro_amd_geom *geom, double dt, void *stream);
int castro_amd_new_rotation_source_fab(castro_amd_ctx *ctx, const castro_amd_fab *state_old, const castro_amd_fab *state_new,
                                       const castro_amd_fab *source, const castro_amd_fab mass_fluxes[3],
                                       const int lo[3], const int hi[3], const castro_amd_rotation *rot,
                                       const castro_amd_geom *geom, double dt, void *stream);

/* The source stages of do_advance_ctu for EVERY box of a level in one call (the MFIter loops of do_old_sources /
 * do_new_sources, Source/sources/Castro_sources.cpp:230-349, around construct_ctu_hydro_source): a level of many small
 * boxes is bound by the host's launch rate, and these stages are three to four launches per box.  Per box, in this order:
 *   stage 0 (old time, Castro_advance_ctu.cpp:94, 127-131):  source = 0 (the whole FAB, ghost zones too);
 *            += old gravity source (grav != NULL); += old rotation source (rot != NULL);
 *            S_new = S_old + dt * source on [lo, hi], then clean_state x clean_ntimes      (castro_amd_apply_source_fab)
 *   stage 1 (new time, :262-268):  source = 0; += new gravity source; += new rotation source (both read S_old, S_new and
 *            mass_flux[d]);  S_new += dt * source on [lo, hi], then clean_state x clean_ntimes
 * with the arithmetic -- and the kernels -- of the single-box entry points above: the same bits.  stage 0: `source` is the
 * old-time Source_Type FAB with its ghost zones; stage 1: the new-time one (valid zones); mass_flux is read by stage 1 only. */
typedef struct castro_amd_source_box {
    int lo[3], hi[3];
    castro_amd_fab S_old, S_new, source;
    castro_amd_fab mass_flux[3];
} castro_amd_source_box;
int castro_amd_sources_mf(castro_amd_ctx *ctx, int stage, int nboxes, const castro_amd_source_box *boxes,
                          const double *grav /* [3] or NULL */, int grav_source_type, const castro_amd_rotation *rot /* or NULL */,
                          const castro_amd_geom *geom, const castro_amd_params *params, double dt, int clean_ntimes, void *stream);
/* castro_amd_clean_state_reduce_fab / castro_amd_estdt_fab on the valid zones [lo, hi] of every box of a level, reduced
 * into ONE d_out (3 device doubles initialised by the caller): Castro_advance_ctu.cpp:168-225 and Castro::estTimeStep
 * (Castro.cpp:1507-1626) as one call per level. */
typedef struct castro_amd_state_box {
    int lo[3], hi[3];
    castro_amd_fab state;
} castro_amd_state_box;
int castro_amd_clean_state_reduce_mf(castro_amd_ctx *ctx, int nboxes, const castro_amd_state_box *boxes,
                                     const castro_amd_geom *geom, const castro_amd_params *params, int ntimes,
                                     double *d_out, void *stream);
int castro_amd_estdt_mf(castro_amd_ctx *ctx, int nboxes, const castro_amd_state_box *boxes,
                        const castro_amd_geom *geom, const castro_amd_params *params, double *d_out, void *stream);

/* Two-level AMR building blocks, refinement ratio 2 (first slice).  The reference calls AMReX for
 * all of these [3P, not in the reference tree]; the arithmetic is restated from the published descriptions and is
 * NOT pinned against an AMReX build:
 *   castro_amd_cc_interp_fab          FillPatch coarse->fine with cell_cons_interp (Castro_setup.cpp:352-364): MC-limited
 *                                     slopes, one scaling factor per coarse zone keeping the children inside the range of
 *                                     the 27 neighbours; fills the fine zones [lo,hi]; crse must cover them grown by one
 *   castro_amd_lincomb_fab            dst = a x + b y  (time interpolation of the coarse data in FillPatch)
 *   castro_amd_avgdown_fab            amrex::average_down of Castro::avgDown (Castro.cpp:3096-3113) on coarse zones [lo,hi]
 *   castro_amd_fluxreg_crse_init_fab  FluxRegister::CrseInit of FluxRegCrseInit (Castro.cpp:2487-2512): reg = mult * flux
 *   castro_amd_fluxreg_fine_add_fab   FluxRegister::FineAdd of FluxRegFineAdd (:2516-2545): reg += mult * sum of 4 fine faces
 *   castro_amd_reflux_fab             FluxRegister::Reflux of Castro::reflux (:2549-2700): zones outside the faces [lo,hi]
 *                                     get -reg/vol (side 0, low face of the fine region) or +reg/vol (side 1)
 *   castro_amd_error_tag_fab          amrex::AMRErrorTag of Castro::errorEst (Castro.cpp:3131-3164): kind 0 value_greater,
 *                                     1 value_less, 2 gradient, 3 relative_gradient on component `comp` of `field`
 *                                     (one ghost zone for the gradient kinds); tags (1 comp, 1.0 = tagged) are OR-ed
 *   castro_amd_fillpatch_shell_fab    the coarse-level part of a fine box's FillPatch in one launch: cc_interp into every
 *                                     zone of grow([vlo,vhi], ngrow) outside [vlo,vhi], followed by clean_state x
 *                                     clean_ntimes there (Castro_advance.cpp:186); ncomp 8
 * Register planes are ordinary FABs, one coarse face thick, in coarse face index space. */
/* Several independent FAB-to-FAB region operations in one launch (a level of many small boxes is launch-bound
 * otherwise: ~60 launches per box and advance).  The operations of one call must not write a zone that another one of
 * the same call reads or writes.  Kinds and their single-operation equivalents:
 *   CASTRO_AMD_OP_COPY               castro_amd_copy_fab              dst = src                (ncomp components)
 *   CASTRO_AMD_OP_LINCOMB            castro_amd_lincomb_fab           dst = a src + b src2
 *   CASTRO_AMD_OP_FLUXREG_CRSE_INIT  castro_amd_fluxreg_crse_init_fab dst = a src
 *   CASTRO_AMD_OP_FLUXREG_FINE_ADD   castro_amd_fluxreg_fine_add_fab  dst += a (sum of the 4 fine faces of src), dir
 *   CASTRO_AMD_OP_REFLUX             castro_amd_reflux_fab            dst zones outside the faces [lo,hi] -= / += src / a
 *   CASTRO_AMD_OP_AVGDOWN            castro_amd_avgdown_fab           dst = mean of the 8 fine zones of src (region in dst's index space)
 *                                                                      (side 0 / 1), a = zone volume */
#define CASTRO_AMD_OP_COPY 0
#define CASTRO_AMD_OP_LINCOMB 1
#define CASTRO_AMD_OP_FLUXREG_CRSE_INIT 2
#define CASTRO_AMD_OP_FLUXREG_FINE_ADD 3
#define CASTRO_AMD_OP_REFLUX 4
#define CASTRO_AMD_OP_CLEAN 5           /* Castro::clean_state x (int)a on the region of dst (ncomp 8); needs castro_amd_fab_ops_p */
#define CASTRO_AMD_OP_INTERP_CLEAN 6    /* dst (fine, ncomp 8) = cell_cons_interp of src (coarse data under it) on the region,
                                         * then clean_state x (int)a: one slab of a FillPatch ghost shell
                                         * (castro_amd_fillpatch_shell_fab does the six slabs of one box); castro_amd_fab_ops_p */
#define CASTRO_AMD_OP_AVGDOWN 7         /* dst (coarse) = mean of the 8 fine zones of src under each zone of the region (castro_amd_avgdown_fab):
                                         * Castro::avgDown of a whole level in one call */
#define CASTRO_AMD_OP_INTERP 8          /* dst (fine) = cell_cons_interp of src on the region, the first ncomp (<= 8) components, nothing else
                                         * (castro_amd_cc_interp_fab): the ghost shells of the Source_Type FillPatch of a level in one call */
typedef struct castro_amd_fab_op {
    int kind;
    int dir;                     /* FLUXREG_FINE_ADD, REFLUX */
    int ncomp;
    int lo[3], hi[3];            /* region, in the index space of dst (REFLUX: the faces, in the index space of src) */
    int side;                    /* REFLUX only */
    double a, b;
    castro_amd_fab dst, src, src2;   /* src2: LINCOMB only */
} castro_amd_fab_op;
int castro_amd_fab_ops(castro_amd_ctx *ctx, int nops, const castro_amd_fab_op *ops, void *stream);
/* The same with the runtime parameters the CLEAN kinds need.  Any number of operations: up to sixteen travel as a kernel
 * argument, longer tables through a device buffer of the context, ONE launch either way -- the per-box sweeps of a level
 * of many small boxes (clean_state of every box, the ghost shells of every box) become one launch per level. */
int castro_amd_fab_ops_p(castro_amd_ctx *ctx, int nops, const castro_amd_fab_op *ops, const castro_amd_params *params,
                         void *stream);

/*
 * The MFIter loop itself (Source/hydro/Castro_ctu_hydro.cpp:130-1480): castro_amd_ctu_hydro_fab_ex for every box of a level
 * in one call.  Box i runs on ctxs[i % nctx] / streams[i % nctx] (each context has its own scratch; small boxes fill a
 * fraction of the chip, so several are kept in flight); the streams are forked from `stream` and joined to it with
 * events inside the call, so the caller sees one asynchronous operation on `stream`.  nctx == 1 with streams[0] == stream
 * is a plain loop.  opts applies to every box (d_out: one reduction vector for the level).
 */
typedef struct castro_amd_hydro_box {
    int bxlo[3], bxhi[3], vbxlo[3], vbxhi[3];
    castro_amd_fab Sborder, src, S_new;
    castro_amd_fab flux[3], mass_flux[3], qe[3];
} castro_amd_hydro_box;
int castro_amd_ctu_hydro_mf(castro_amd_ctx *const *ctxs, void *const *streams, int nctx,
                            const castro_amd_hydro_box *boxes, int nboxes,
                            const castro_amd_geom *geom, const castro_amd_params *params,
                            double time, double dt, const castro_amd_hydro_opts *opts, void *stream);

int castro_amd_fillpatch_shell_fab(castro_amd_ctx *ctx, const castro_amd_fab *crse, const castro_amd_fab *fine,
                                   const int vlo[3], const int vhi[3], int ngrow, const castro_amd_params *params,
                                   int clean_ntimes, void *stream);
int castro_amd_cc_interp_fab(castro_amd_ctx *ctx, const castro_amd_fab *crse, const castro_amd_fab *fine,
                             const int lo[3], const int hi[3], int ncomp, void *stream);
int castro_amd_error_tag_fab(castro_amd_ctx *ctx, const castro_amd_fab *field, int comp, const castro_amd_fab *tags,
                             const int lo[3], const int hi[3], int kind, double value, void *stream);
int castro_amd_lincomb_fab(castro_amd_ctx *ctx, const castro_amd_fab *dst, double a, const castro_amd_fab *x, double b,
                           const castro_amd_fab *y, int ncomp, const int lo[3], const int hi[3], void *stream);
int castro_amd_avgdown_fab(castro_amd_ctx *ctx, const castro_amd_fab *fine, const castro_amd_fab *crse,
                           const int lo[3], const int hi[3], int ncomp, void *stream);
int castro_amd_fluxreg_crse_init_fab(castro_amd_ctx *ctx, const castro_amd_fab *reg, const castro_amd_fab *crse_flux,
                                     const int lo[3], const int hi[3], int ncomp, double mult, void *stream);
int castro_amd_fluxreg_fine_add_fab(castro_amd_ctx *ctx, const castro_amd_fab *reg, const castro_amd_fab *fine_flux,
                                    const int lo[3], const int hi[3], int dir, int ncomp, double mult, void *stream);
int castro_amd_reflux_fab(castro_amd_ctx *ctx, const castro_amd_fab *state, const castro_amd_fab *reg,
                          const int lo[3], const int hi[3], int dir, int side, int ncomp, double vol, void *stream);

/* Derived plotfile fields (Source/driver/Derive.cpp, registered in Castro_setup.cpp:756-960) for the
 * 3-D Cartesian gamma-law build.  Not provided: entropy (needs the Microphysics entropy formula),
 * StateErr, circvel, angular_momentum_{x,y,z}. */
enum {
    CASTRO_AMD_DER_PRESSURE = 0,   /* ca_derpres         Derive.cpp:24   */
    CASTRO_AMD_DER_KINENG,         /* ca_derkineng       :860 */
    CASTRO_AMD_DER_SOUNDSPEED,     /* ca_dersoundspeed   :180 */
    CASTRO_AMD_DER_GAMMA_1,        /* ca_dergamma1       :216 */
    CASTRO_AMD_DER_MACHNUMBER,     /* ca_dermachnumber   :251 */
    CASTRO_AMD_DER_MAGVORT,        /* ca_dermagvort      :929  (needs 1 ghost zone of state) */
    CASTRO_AMD_DER_DIVU,           /* ca_derdivu         :1021 (needs 1 ghost zone of state) */
    CASTRO_AMD_DER_EINT_E1,        /* eint_E, ca_dereint1 :57 */
    CASTRO_AMD_DER_EINT_E2,        /* eint_e, ca_dereint2 :79 */
    CASTRO_AMD_DER_LOGDEN,         /* ca_derlogden       :95 */
    CASTRO_AMD_DER_SPEC,           /* X(spec), ca_derspec :891 */
    CASTRO_AMD_DER_ABAR,           /* ca_derabar         :907 */
    CASTRO_AMD_DER_X_VELOCITY,     /* ca_dervel          :516 */
    CASTRO_AMD_DER_Y_VELOCITY,
    CASTRO_AMD_DER_Z_VELOCITY,
    CASTRO_AMD_DER_MAGVEL,         /* ca_dermagvel       :532 */
    CASTRO_AMD_DER_RADVEL,         /* ca_derradialvel    :572 (about `center`) */
    CASTRO_AMD_DER_MAGMOM,         /* ca_dermagmom       :692 */
    CASTRO_AMD_DER_STATEERR_0,     /* ca_derstate        :1087 (three components: density, Temp, X) */
    CASTRO_AMD_DER_STATEERR_1,
    CASTRO_AMD_DER_STATEERR_2,
    CASTRO_AMD_DER_CIRCVEL,        /* ca_dercircvel      :627 (about `center`, domain_is_plane_parallel = 0) */
    CASTRO_AMD_DER_ANGMOM_X,       /* ca_derangmomx/y/z  :711-870 (about `center`) */
    CASTRO_AMD_DER_ANGMOM_Y,
    CASTRO_AMD_DER_ANGMOM_Z,
    CASTRO_AMD_DER_COUNT
};
/* der(:,:,:,dcomp) = derived field `which` of `state` on [lo,hi]; `center` = problem::center (radvel only). */
int castro_amd_derive_fab(castro_amd_ctx *ctx, int which, const castro_amd_fab *state,
                          const castro_amd_fab *der, int dcomp, const int lo[3], const int hi[3],
                          const castro_amd_geom *geom, const castro_amd_params *params,
                          const double center[3], void *stream);

/* Castro::estdt_cfl (Source/driver/timestep.cpp:31-140) and S_new.min(URHO)
 * (Castro_advance_ctu.cpp:168) fused: d_out[0] = min over [lo,hi] of dx/(c+|u|)
 * (NOT yet multiplied by cfl), d_out[1] = min density.  d_out is a device
 * pointer to 2 doubles that the CALLER initialises (e.g. to +huge) so several
 * FABs can reduce into the same words. */
int castro_amd_estdt_fab(castro_amd_ctx *ctx, const castro_amd_fab *state,
                         const int lo[3], const int hi[3], const castro_amd_geom *geom,
                         const castro_amd_params *params, double *d_out, void *stream);

/* Physical-boundary ghost fill of a grown state FAB (the GpuBndryFuncFab /
 * ca_statefill part of AmrLevel::FillPatch: Source/problems/Castro_bc_fill_nd.cpp:11-125,
 * BC tables Source/driver/Castro_setup.cpp:40-53). Fills every zone of the FAB
 * outside the problem domain; x, then y, then z. */
int castro_amd_bc_fill_fab(castro_amd_ctx *ctx, const castro_amd_fab *state,
                           const castro_amd_geom *geom, void *stream);

/* dst(lo:hi, 0:ncomp) = src(lo:hi, 0:ncomp) between two FABs (MultiFab::Copy /
 * the same-level copy part of FillPatch). */
int castro_amd_copy_fab(castro_amd_ctx *ctx, const castro_amd_fab *dst, const castro_amd_fab *src,
                        const int lo[3], const int hi[3], void *stream);

/* Pack / unpack a sub-box of a FAB to / from a contiguous buffer (same FAB
 * ordering restricted to the sub-box): the halo staging used by the RCCL
 * FillBoundary replacement. */
int castro_amd_pack_fab(castro_amd_ctx *ctx, const castro_amd_fab *fab, const int lo[3], const int hi[3],
                        double *buf, void *stream);
int castro_amd_unpack_fab(castro_amd_ctx *ctx, const castro_amd_fab *fab, const int lo[3], const int hi[3],
                          const double *buf, void *stream);

/* The same for all regions of one FillBoundary in ONE launch (a rank has up to 26 neighbours): region r is the box
 * lo[3r..3r+2] : hi[3r..3r+2] and lives at buf + offsets[r] (in doubles), r < nregions <= CASTRO_AMD_MAX_REGIONS. */
#define CASTRO_AMD_MAX_REGIONS 32
int castro_amd_pack_regions_fab(castro_amd_ctx *ctx, const castro_amd_fab *fab, int nregions, const int *lo, const int *hi,
                                const long long *offsets, double *buf, void *stream);
int castro_amd_unpack_regions_fab(castro_amd_ctx *ctx, const castro_amd_fab *fab, int nregions, const int *lo, const int *hi,
                                  const long long *offsets, const double *buf, void *stream);

/* Problem initial data on [lo,hi] of `state` (ncomp 8):
 * Exec/hydro_tests/Sedov/problem_initialize.H:8-113 + problem_initialize_state_data.H:8-148 */
int castro_amd_sedov_init_fab(castro_amd_ctx *ctx, const castro_amd_fab *state, const int lo[3], const int hi[3],
                              const castro_amd_geom *geom, const castro_amd_params *params,
                              double r_init, double p_ambient, double exp_energy, double dens_ambient,
                              int nsub, void *stream);
/* Exec/hydro_tests/Sod/problem_initialize*.H (use_Tinit = 0); idir is 1-based */
int castro_amd_sod_init_fab(castro_amd_ctx *ctx, const castro_amd_fab *state, const int lo[3], const int hi[3],
                            const castro_amd_geom *geom, const castro_amd_params *params,
                            double rho_l, double u_l, double p_l, double rho_r, double u_r, double p_r,
                            int idir, double frac, void *stream);

/*
 * FillBoundary on RCCL: the same-level ghost exchange of AmrLevel::FillPatch as Castro::expand_state uses it
 * (Source/driver/Castro.cpp:4201-4209) followed by the physical-boundary fill
 * (Source/problems/Castro_bc_fill_nd.cpp:11-125), for hosts without torch.distributed (castro_amd/csrc/halo_rccl.hip).
 * One rank per GPU; librccl is bound at run time (dlopen), so a single-GPU host never needs it.
 *
 *   castro_amd_comm        an RCCL communicator: created here from a 128-byte unique id that the host distributes with
 *                          whatever it has (MPI_Bcast, a file, torch.distributed), or adopted from the host's own ncclComm_t;
 *   castro_amd_halo_plan   the regions of ONE FAB shape (<= 26 neighbours), with the packed send / receive buffers: built
 *                          once per level and box, reused every step;
 *   castro_amd_fill_boundary   pack (1 launch) -> ncclGroupStart / ncclRecv.. / ncclSend.. / ncclGroupEnd -> unpack
 *                          (<= 2 launches) -> physical BC fill, all enqueued on `stream`: nothing synchronises, nothing allocates.
 *
 * A region sends the zones `sbox` of this rank's FAB to `peer` and receives the ghost zones `rbox` from the same peer
 * (equal shapes).  send_tag / recv_tag order the messages between a pair of ranks: the recv_tag of the region that
 * receives a message must equal the send_tag of the region that sent it (e.g. tag = 1 + ox + 3 (1 + oy) + 9 (1 + oz) of the
 * direction the message TRAVELS in).  peer == own rank (a periodic wrap onto this rank) is a local copy.
 *
 * A plan owns its two packed buffers (device memory outside any allocator of the host) until castro_amd_halo_plan_destroy,
 * which the owner calls once nothing of the plan is in flight; a plan is SINGLE-STREAM (two calls with one plan on two
 * streams would race on its buffers: one plan per stream and FAB shape).  castro_amd_fill_boundary selects the
 * communicator's device itself.
 */
typedef struct castro_amd_comm castro_amd_comm;
typedef struct castro_amd_halo_plan castro_amd_halo_plan;
#define CASTRO_AMD_UNIQUE_ID_BYTES 128
typedef struct castro_amd_halo_region {
    int peer;
    int sbox_lo[3], sbox_hi[3];
    int rbox_lo[3], rbox_hi[3];
    int send_tag, recv_tag;
} castro_amd_halo_region;
const char *castro_amd_comm_version(void);                       /* "RCCL x.y.z" of the library bound at run time */
int castro_amd_comm_unique_id(void *id);                         /* ncclGetUniqueId: id = CASTRO_AMD_UNIQUE_ID_BYTES host bytes */
int castro_amd_comm_create(castro_amd_comm **out, int nranks, int rank, const void *id, int device);   /* ncclCommInitRank (collective) */
int castro_amd_comm_adopt(castro_amd_comm **out, void *nccl_comm, int device);   /* wrap the host's ncclComm_t; not destroyed here */
int castro_amd_comm_rank(const castro_amd_comm *comm);
int castro_amd_comm_size(const castro_amd_comm *comm);
int castro_amd_comm_destroy(castro_amd_comm *comm);
int castro_amd_halo_plan_create(castro_amd_halo_plan **out, castro_amd_comm *comm, int nregions,
                                const castro_amd_halo_region *regions, int ncomp);
int castro_amd_halo_plan_destroy(castro_amd_halo_plan *plan);
long long castro_amd_halo_plan_bytes_sent(const castro_amd_halo_plan *plan);   /* bytes this rank sends to OTHER ranks per exchange */
/* geom == NULL: no physical-boundary fill (interior boxes of a level, or a caller that fills them itself) */
int castro_amd_fill_boundary(castro_amd_ctx *ctx, castro_amd_halo_plan *plan, const castro_amd_fab *state,
                             const castro_amd_geom *geom, void *stream);
/* castro_amd_fill_boundary for a caller that overlaps the exchange with work on another stream: the plan's "packed" event is
 * recorded on `stream` right behind the pack launch -- the LAST read of the valid zones of `state` by this call; everything after
 * it touches ghost zones only.  castro_amd_halo_plan_wait_packed makes `other_stream` wait for that event, after which work that
 * WRITES valid zones of `state` (CASTRO_AMD_STAGE_VALID with sborder_clean_ntimes > 0) may run on it beside the exchange.  Both
 * calls are capturable (the event becomes a dependency between the two streams of the graph).  flags: reserved, 0. */
int castro_amd_fill_boundary_ex(castro_amd_ctx *ctx, castro_amd_halo_plan *plan, const castro_amd_fab *state,
                                const castro_amd_geom *geom, int flags, void *stream);
int castro_amd_halo_plan_wait_packed(castro_amd_halo_plan *plan, void *other_stream);
/*
 * Several boxes per rank (round 6): ONE grouped exchange for all FABs of a level that this rank owns -- an AMR level, or two
 * boxes per GPU (Source/driver/Castro.cpp:4201-4209 over a BoxArray of any shape).  Sends and receives are separate lists:
 * with boxes of unequal size the zones a box sends to a neighbour and those it receives from it differ in shape.
 *   fab   index of the local FAB (0 .. nfabs-1) the zones [lo, hi] belong to
 *   peer  rank of the other end (own rank: a copy between two local boxes, or a periodic wrap; no RCCL call)
 *   tag   identifies the message between the two ranks: both ends must derive the same number for it, e.g.
 *         (source box * nboxes + destination box) * 27 + code of the periodic shift, box = index in the level's BoxArray
 * The k-th send to a peer in tag order is matched with the peer's k-th receive from this rank in tag order; two messages of
 * one pair of ranks with one tag are refused.  A local receive needs a local send of the same tag and size.
 * The group owns its two packed buffers; single-stream like a plan.  castro_amd_fill_boundary_group: states[f] is the FAB of
 * local box f; geom == NULL: no physical-boundary fill.
 */
typedef struct castro_amd_halo_group castro_amd_halo_group;
typedef struct castro_amd_halo_msg {
    int fab;
    int peer;
    int lo[3], hi[3];
    int tag;
} castro_amd_halo_msg;
int castro_amd_halo_group_create(castro_amd_halo_group **out, castro_amd_comm *comm, int nfabs,
                                 int nsends, const castro_amd_halo_msg *sends, int nrecvs, const castro_amd_halo_msg *recvs, int ncomp);
int castro_amd_halo_group_destroy(castro_amd_halo_group *group);
long long castro_amd_halo_group_bytes_sent(const castro_amd_halo_group *group);   /* bytes to OTHER ranks per exchange */
int castro_amd_fill_boundary_group(castro_amd_ctx *ctx, castro_amd_halo_group *group, const castro_amd_fab *states,
                                   const castro_amd_geom *geom, void *stream);
/* the overlap hooks of castro_amd_fill_boundary_ex / castro_amd_halo_plan_wait_packed for a group: the event is recorded behind
 * the LAST pack launch (the valid zones of every local FAB have been read by then).  flags: reserved, 0. */
int castro_amd_fill_boundary_group_ex(castro_amd_ctx *ctx, castro_amd_halo_group *group, const castro_amd_fab *states,
                                      const castro_amd_geom *geom, int flags, void *stream);
int castro_amd_halo_group_wait_packed(castro_amd_halo_group *group, void *other_stream);
/* ncclAllReduce(MIN) in place on n device doubles: the [dt estimate, min density, ...] reduction of a step */
int castro_amd_allreduce_min(castro_amd_comm *comm, double *d_buf, int n, void *stream);

/*
 * Grid generation of a regrid on the HOST (no device work, no context): the point clustering of Berger & Rigoutsos (1991) that
 * Amr::grid_places applies to the tags of Castro::errorEst (Source/driver/Castro.cpp:3131-3164) [3P: AMReX's ClusterList, restated
 * from the paper, not pinned against an AMReX build].  tags / mask: nz*ny*nx bytes, [z][y][x], non-zero = tagged / allowed
 * (mask NULL: everything allowed).  Every tagged, allowed cell ends up in exactly one box; a box lies inside the mask and is
 * filled to at least grid_eff with tags or is no longer than min_cells a side.  boxes: 6 ints per box (z0, y0, x0, z1, y1, x1),
 * inclusive.  Returns the number of boxes (only the first max_boxes are written: call again with a larger array if it is more)
 * or a negative error.  Same boxes as castro_amd/cluster.py::berger_rigoutsos (tests/test_cluster_cpu.py).
 */
int castro_amd_berger_rigoutsos(const unsigned char *tags, const unsigned char *mask, int nz, int ny, int nx,
                                double grid_eff, int min_cells, int *boxes, int max_boxes);

/* Library/version introspection */
const char *castro_amd_version(void);
/* CASTRO_AMD_ABI_VERSION the library was built with (see the macro above) */
int castro_amd_abi_version(void);
/* Numerics mode of this build of the library:
 *   "exact"    -ffp-contract=off, IEEE division / sqrt: bit-identical to the reference's CPU expression order ;
 *   "contract" FMA contraction, reciprocal-based division and rsq-based sqrt (<= 1 ulp each): agrees with `exact` to the
 *              north-star tolerance (rtol 1e-10 on every plotfile field, tests/test_gpu_parity.py) and is faster.
 * Both builds export the same ABI; castro_amd/_lib.py picks by CASTRO_AMD_NUMERICS. */
const char *castro_amd_numerics(void);
/* Name and average device time (ms) of the most recent launch of each hot-path
 * kernel when profiling is enabled with castro_amd_ctx_profile(ctx, 1): the
 * library brackets every kernel with hipEvents on `stream`. */
int castro_amd_ctx_profile(castro_amd_ctx *ctx, int enable);
int castro_amd_ctx_profile_count(castro_amd_ctx *ctx);
int castro_amd_ctx_profile_get(castro_amd_ctx *ctx, int idx, char *name, int name_len,
                               double *total_ms, long long *launches);
void castro_amd_ctx_profile_reset(castro_amd_ctx *ctx);

/*
 * Pointwise forms of the per-interface / per-zone functions of the path, for known-answer vectors recorded at the level
 * of the reference's own functions.  Flat lists of n points, device pointers, component-major: a[comp * n + point].
 * No context: nothing is allocated.
 *   castro_amd_cmpflx_points   body of Castro::cmpflx_plus_godunov for one interface (Source/hydro/riemann.cpp:62-203):
 *                              load_input_states (riemann.H:66-246), riemannus / riemanncg / HLLC (riemann_solvers.H),
 *                              compute_flux_q (:14-211), passive upwinding, HLL in shocked zones.
 *                              qm, qp: 7 comps (rho,u,v,w,p,rhoe,X); cl, cr: sound speed of the zones either side;
 *                              bnd_fac (NULL = 1) and is_shock (NULL = 0) per interface;
 *                              out: 11 comps (F_rho, F_mom normal/t/tt, F_E, F_eint, F_X, Godunov un, ut, utt, p)
 *   castro_amd_ppm_points      ppm_reconstruct (ppm.H:54-139) + ppm_int_profile (:157-252) of a five-point stencil
 *                              s (5 comps): out = (sm, sp, Ip[3], Im[3]) under the waves u-c, u, u+c
 *   castro_amd_flatten_points  the one-direction flattening coefficient of Castro::uflatten (flatten.cpp:12-166):
 *                              p7 = pressure at i-3..i+3, u5 = normal velocity at i-2..i+2
 *   castro_amd_trans_points    actual_trans_single (trans.cpp:66-437, ntrans = 1, tdir = transverse direction) or
 *                              actual_trans_final (:498-862, ntrans = 2): q 7 comps, flux records 8 comps
 *                              (rho, mx, my, mz, E, X fluxes, Godunov un, Godunov p) at the high (r) / low (l) face;
 *                              fe (NULL unless transverse_reset_rhoe = 1): the (rho e) fluxes at the faces 1r, 1l
 *                              (, 2r, 2l), rows of n
 */
int castro_amd_cmpflx_points(long long n, int idir, const double *qm, const double *qp, const double *cl, const double *cr,
                             const double *bnd_fac, const int *is_shock, const castro_amd_params *params, double *out,
                             void *stream);
int castro_amd_ppm_points(long long n, const double *s, const double *flatn, const double *u, const double *c, double dtdx,
                          double *out, void *stream);
int castro_amd_flatten_points(long long n, const double *p7, const double *u5, double *out, void *stream);
int castro_amd_trans_points(long long n, int ntrans, int tdir, const double *q, const double *f1r, const double *f1l,
                            const double *f2r, const double *f2l, const double *fe, double cdtdx1, double cdtdx2,
                            const castro_amd_params *params, double *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif
