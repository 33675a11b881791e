/*
 * castro_oracle.h -- CPU oracle for the Castro CTU hydro path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it.  The shipped
 * library (castro_amd/libcastro_hydro_amd.so) never links or calls it.
 *
 * It is a plain-C restatement of the algorithm in BoxLib-Codes/Castro 21.07
 * (Source/hydro, Source/driver), organised like the reference's CPU path
 * (one array sweep per stage, thread-private scratch per tile, OpenMP over
 * tiles).  Every function cites the reference file:line it follows.
 *
 * PARITY STATUS: the reference cannot be compiled in this image (AMReX and
 * Microphysics submodules are empty), and the reference ships
 * no bitwise golden vectors for this path.  The oracle is pinned at
 * known-answer level against the reference's own Verification tables
 * (Exec/hydro_tests/Sedov/Verification/spherical_sedov.dat,
 * Exec/hydro_tests/Sod/Verification/{sod,test2,test3}-exact.out) and against
 * the two reference outputs recorded in SURVEY.md section 8c.  Bitwise parity
 * with a real reference binary is UNPINNED ("parity unpinned").
 *
 * Third-party arithmetic restated (not in the reference tree):
 *   Microphysics EOS/gamma_law (release paired with Castro 21.07, i.e. 21.07;
 *   no pinned hash is recorded in the empty submodule): p=(gamma-1) rho e, etc.
 */
#ifndef CASTRO_ORACLE_H
#define CASTRO_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- state indices for the Sedov build (output of
 *      Source/driver/set_variables.py on Source/driver/_variables) ---- */
enum { URHO = 0, UMX = 1, UMY = 2, UMZ = 3, UEDEN = 4, UEINT = 5, UTEMP = 6, UFS = 7 };
enum { QRHO = 0, QU = 1, QV = 2, QW = 3, QPRES = 4, QREINT = 5, QTEMP = 6, QFS = 7 };
enum { QGAMC = 0, QC = 1 };
enum { GDU = 0, GDV = 1, GDW = 2, GDPRES = 3 };
enum { NUMSPEC = 1, NUM_STATE = 8, NQ = 8, NQAUX = 2, NGDNV = 4, NQSRC = 7, NSRC = 7,
       NPASSIVE = 1, NUM_GROW = 4, NUM_GROW_SRC = 3 };

/* physical boundary types, Source/driver/Castro.H (enum order used by
 * castro.lo_bc / hi_bc) */
enum { BC_INTERIOR = 0, BC_INFLOW = 1, BC_OUTFLOW = 2, BC_SYMMETRY = 3, BC_SLIPWALL = 4,
       BC_NOSLIPWALL = 5 };

/* Array4 analogue: i fastest, component slowest */
typedef struct {
    double *p;
    int lo[3], hi[3];
    int nc;
    long sy, sz, sn;
} ora_a4;

/* runtime parameters read by the hot path (Source/driver/_cpp_parameters) */
typedef struct {
    int ppm_type;                 /* :88  */
    int riemann_solver;           /* :113 */
    int use_flattening;           /* :132 */
    int hybrid_riemann;           /* :107 */
    int first_order_hydro;        /* :192 */
    int cg_maxiter;               /* :117 */
    int cg_blend;                 /* :128 */
    int transverse_use_eos;       /* :136 */
    int transverse_reset_density; /* :141 */
    int transverse_reset_rhoe;    /* :146 */
    int ppm_temp_fix;
    int plm_iorder;
    int plm_limiter;
    int use_pslope;               /* :161 */
    double difmag;                /* :40  */
    double small_dens, small_temp, small_pres, small_ener; /* :43-54 */
    double cg_tol;                /* :121 */
    double dual_energy_eta1;      /* :151 */
    double dual_energy_eta2;      /* :156 */
    double cfl, init_shrink, change_max; /* :318,322,326 */
    double eos_gamma;             /* Microphysics gamma_law: eos_gamma */
    double small_x;               /* network small_x */
    double T_guess;
    double abar;                  /* species A (eos_assume_neutral=1 => mu = abar) */
    double pslope_cutoff_density; /* :165 */
    int limit_fluxes_on_small_dens;   /* :168 */
    int limit_fluxes_on_large_vel;    /* :171 */
    double speed_limit;               /* :175 */
    int source_term_predictor;        /* :188  0 (default); 1: lagged predictor of the momentum sources, Castro.cpp:3780-3818 */
} ora_params;

typedef struct {
    double dx[3];
    double problo[3];
    double probhi[3];
    int domlo[3], domhi[3];
    int lo_bc[3], hi_bc[3];
    int coord;
} ora_geom;

/* ---------------- helpers ---------------- */
void ora_default_params(ora_params *p);
void ora_finalize_params(ora_params *p);   /* small_pres/small_ener floors, Castro_setup.cpp:222-288 */
ora_a4 ora_make_a4(double *p, const int lo[3], const int hi[3], int nc);

/* ---------------- EOS (Microphysics gamma_law restated) -------- */
/* xn: mass fraction of the (one) species -- the mean molecular weight of the gamma-law EOS with eos_assume_neutral = 1 is
 * mu = abar = 1 / sum_k(X_k / A_k), so T(e) and e(T) depend on the composition the caller passes */
typedef struct { double rho, T, e, p, gam1, cs, dpde, dpdr_e, xn; } ora_eos_t;
void ora_eos_re(const ora_params *P, ora_eos_t *s);
/* constant-gravity source terms (Source/gravity/Castro_gravity.cpp:234-614) and Saxpy */
void ora_old_gravity_source(const int lo[3], const int hi[3], ora_a4 uold, ora_a4 source, const double grav[3],
                            int grav_source_type, double dt);
void ora_new_gravity_source(const int lo[3], const int hi[3], ora_a4 uold, ora_a4 unew, ora_a4 source,
                            const ora_a4 mflux[3], const double grav[3], int grav_source_type, double dt,
                            const double dx[3]);
void ora_saxpy(const int lo[3], const int hi[3], ora_a4 dst, double a, ora_a4 src, int ncomp);
/* castro.do_rotation with state_in_rotating_frame = 1 (Source/rotation): omega = 2 pi / rotational_period along rot_axis */
typedef struct {
    double omega[3];
    double center[3];              /* problem::center */
    int include_centrifugal;       /* castro.rotation_include_centrifugal (1) */
    int include_coriolis;          /* castro.rotation_include_coriolis (1) */
    int rot_source_type;           /* castro.rot_source_type 1..4 (4) */
    int implicit_rotation_update;  /* castro.implicit_rotation_update (1) */
} ora_rotation;
void ora_old_rotation_source(const int lo[3], const int hi[3], ora_a4 uold, ora_a4 source, const ora_rotation *R,
                             const ora_geom *G, double dt);
void ora_new_rotation_source(const int lo[3], const int hi[3], ora_a4 uold, ora_a4 unew, ora_a4 source,
                             const ora_a4 mflux[3], const ora_rotation *R, const ora_geom *G, double dt);
/* two-level AMR building blocks (AMReX arithmetic restated, see ora_amr.c) */
void ora_cc_interp(const int lo[3], const int hi[3], ora_a4 crse, ora_a4 fine, int ncomp);
void ora_avgdown(const int lo[3], const int hi[3], ora_a4 fine, ora_a4 crse, int ncomp);
void ora_reg_crse_init(const int lo[3], const int hi[3], ora_a4 reg, ora_a4 cflux, int ncomp, double mult);
void ora_reg_fine_add(const int lo[3], const int hi[3], ora_a4 reg, ora_a4 fflux, int dir, int ncomp, double mult);
void ora_reflux(const int lo[3], const int hi[3], ora_a4 state, ora_a4 reg, int dir, int side, int ncomp, double vol);
void ora_error_tag(const int lo[3], const int hi[3], ora_a4 q, int comp, ora_a4 tags, int kind, double value);
void ora_lincomb(const int lo[3], const int hi[3], ora_a4 dst, double a, ora_a4 x, double b, ora_a4 y, int ncomp);
/* Source/driver/Derive.cpp; `which` uses the CASTRO_AMD_DER_* numbering */
int ora_derive(int which, const int lo[3], const int hi[3], ora_a4 dat, ora_a4 der, const ora_geom *G,
               const ora_params *P, const double center[3]);
void ora_eos_rt(const ora_params *P, ora_eos_t *s);
void ora_eos_rp(const ora_params *P, ora_eos_t *s);

/* ---------------- per-stage kernels (reference names) ---------------- */
int  ora_ctoprim(const int lo[3], const int hi[3], ora_a4 uin, ora_a4 q, ora_a4 qaux, const ora_params *P);
void ora_uflatten(const int lo[3], const int hi[3], ora_a4 q, ora_a4 flatn, int pres_comp);
void ora_shock(const int lo[3], const int hi[3], ora_a4 q, ora_a4 shk, const ora_geom *G);
void ora_src_to_prim(const int lo[3], const int hi[3], ora_a4 q, ora_a4 old_src, ora_a4 srcQ, const ora_params *P, double dt);
/* Castro::source_corrector is a class member, not an argument of construct_ctu_hydro_source: the oracle keeps it the same
 * way.  NSRC comps on a box containing grow(bx, 3); p == NULL: none.  Read by ora_src_to_prim when
 * P->source_term_predictor == 1. */
void ora_set_source_corrector(const ora_a4 *corr);
void ora_divu(const int lo[3], const int hi[3], ora_a4 q, ora_a4 div, const ora_geom *G);
void ora_trace_ppm(const int lo[3], const int hi[3], int idir, ora_a4 q, ora_a4 qaux, ora_a4 srcQ,
                   ora_a4 flatn, ora_a4 qm, ora_a4 qp, const int vlo[3], const int vhi[3],
                   double dt, const ora_geom *G, const ora_params *P);
void ora_trace_plm(const int lo[3], const int hi[3], int idir, ora_a4 q, ora_a4 qaux, ora_a4 srcQ,
                   ora_a4 flatn, ora_a4 qm, ora_a4 qp, const int vlo[3], const int vhi[3],
                   double dt, const ora_geom *G, const ora_params *P);
void ora_plm_reflect_fix(const int lo[3], const int hi[3], int idir, ora_a4 qm, ora_a4 qp, const ora_geom *G);
void ora_cmpflx_plus_godunov(const int lo[3], const int hi[3], ora_a4 qm, ora_a4 qp, ora_a4 flx,
                             ora_a4 qgdnv, ora_a4 qaux, ora_a4 shk, int idir,
                             const ora_geom *G, const ora_params *P);
void ora_cmpflx_points(long n, int idir, const double *qm, const double *qp, const double *cl, const double *cr,
                       const double *bnd_fac, const int *is_shock, const ora_params *P, double *out);
void ora_trans_single(const int lo[3], const int hi[3], int idir_t, int idir_n, ora_a4 qm, ora_a4 qmo,
                      ora_a4 qp, ora_a4 qpo, ora_a4 qaux, ora_a4 flux_t, ora_a4 q_t,
                      double hdt, double cdtdx, const ora_params *P);
void ora_trans_final(const int lo[3], const int hi[3], int idir_n, int idir_t1, int idir_t2,
                     ora_a4 qm, ora_a4 qmo, ora_a4 qp, ora_a4 qpo, ora_a4 qaux,
                     ora_a4 flux_t1, ora_a4 flux_t2, ora_a4 q_t1, ora_a4 q_t2,
                     double cdtdx_t1, double cdtdx_t2, const ora_params *P);
void ora_reset_edge_state_thermo(const int lo[3], const int hi[3], ora_a4 qedge, const ora_params *P);
void ora_apply_av(const int lo[3], const int hi[3], int idir, ora_a4 div, ora_a4 uin, ora_a4 flux,
                  const ora_geom *G, const ora_params *P);
void ora_normalize_species_fluxes(const int lo[3], const int hi[3], ora_a4 flux);
void ora_limit_hydro_fluxes_on_small_dens(const int lo[3], const int hi[3], int idir, ora_a4 u, ora_a4 q, ora_a4 flux,
                                          const ora_geom *G, const ora_params *P, double dt);
void ora_limit_hydro_fluxes_on_large_vel(const int lo[3], const int hi[3], int idir, ora_a4 u, ora_a4 q, ora_a4 flux,
                                         const ora_geom *G, const ora_params *P, double dt);
void ora_scale_flux(const int lo[3], const int hi[3], ora_a4 flux, double area, double dt);
void ora_consup_hydro(const int lo[3], const int hi[3], ora_a4 U_new, ora_a4 flux0, ora_a4 qx,
                      ora_a4 flux1, ora_a4 qy, ora_a4 flux2, ora_a4 qz, double dt, const ora_geom *G);

/* single-interface Riemann entry points (for known-answer tests) */
void ora_riemann_single(int solver, const double ql[7], const double qr[7], double csmall, double cavg,
                        double bnd_fac, const ora_params *P, double qint[7]);
void ora_ppm_reconstruct(const double s[5], double flatn, double *sm, double *sp);
void ora_ppm_int_profile(double sm, double sp, double sc, double u, double c, double dtdx,
                         double Ip[3], double Im[3]);

/* ---------------- orchestrator: Castro::construct_ctu_hydro_source ------- */
/* One FAB/box.  Sborder: box grown by >= 4; src may have p==NULL (zero);
 * S_new: RMW on bx; flux_out[d]: += on nodal box d (fbx), mass_flux_out: =.
 * tile[3] <= 0 means "whole box" (GPU style); CPU reference default is
 * hydro_tile_size = (1024,16,16) (Source/driver/Castro.cpp:133).
 * Returns 0 on success, 1 if a non-positive/small density was met in ctoprim. */
int ora_construct_ctu_hydro_source(const int bxlo[3], const int bxhi[3], ora_a4 Sborder, ora_a4 src,
                                   ora_a4 S_new, ora_a4 flux_out[3], ora_a4 mass_flux_out[3],
                                   ora_a4 qe_out[3], const ora_geom *G, const ora_params *P,
                                   double time, double dt, const int tile[3], int nthreads);

int ora_ctu_hydro_tile(const int bxlo[3], const int bxhi[3], const int vlo[3], const int vhi[3],
                       ora_a4 Sborder, ora_a4 src, ora_a4 S_new, ora_a4 flux_out[3], ora_a4 mass_flux_out[3],
                       ora_a4 qe_out[3], const ora_geom *G, const ora_params *P, double dt);

/* ---------------- state maintenance (Source/driver) ---------------- */
void ora_clean_state(const int lo[3], const int hi[3], ora_a4 u, const ora_params *P);
void ora_set_state_threads(int n);
double ora_estdt_cfl(const int lo[3], const int hi[3], ora_a4 u, const ora_geom *G, const ora_params *P);
/* the same with a NaN zone entering the minimum as -1e300: the form of the checks inside an advance */
double ora_estdt_cfl_guarded(const int lo[3], const int hi[3], ora_a4 u, const ora_geom *G, const ora_params *P);
double ora_min_density(const int lo[3], const int hi[3], ora_a4 u);
void ora_bc_fill(ora_a4 u, const ora_geom *G);   /* physical-BC ghost fill, SURVEY D.2 */
void ora_fill_interior_copy(ora_a4 dst, ora_a4 src, const int lo[3], const int hi[3]);

/* ---------------- problems ---------------- */
void ora_sedov_init(const int lo[3], const int hi[3], ora_a4 state, const ora_geom *G, const ora_params *P,
                    double r_init, double p_ambient, double exp_energy, double dens_ambient, int nsub);
void ora_sod_init(const int lo[3], const int hi[3], ora_a4 state, const ora_geom *G, const ora_params *P,
                  double rho_l, double u_l, double p_l, double rho_r, double u_r, double p_r,
                  int idir, double frac);

/* ---------------- single-box level driver (do_advance_ctu mirror) -------- */
typedef struct ora_level ora_level;
ora_level *ora_level_create(const int n[3], const ora_geom *G, const ora_params *P, int nthreads);
void ora_level_destroy(ora_level *L);
double *ora_level_state(ora_level *L);          /* S_new, valid box, FAB layout, NUM_STATE comps */
void ora_level_set_last_dt(ora_level *L, double last_dt);   /* restart: see Level.set_state of oracle_lib.py */
double *ora_level_flux(ora_level *L, int dir);  /* fluxes[dir], nodal box, NUM_STATE comps */
double *ora_level_mass_flux(ora_level *L, int dir);
void ora_level_set_tile(ora_level *L, const int tile[3]);
void ora_level_post_timestep(ora_level *L);     /* Castro::post_timestep on one level: clean_state(S_new) */
void ora_level_post_init(ora_level *L);         /* clean_state after initData (Castro.cpp:1100-1160) */
double ora_level_est_time_step(ora_level *L);
double ora_level_initial_dt(ora_level *L, double stop_time);
double ora_level_new_dt(ora_level *L, double dt_old, double cur_time, double stop_time);
int  ora_level_advance(ora_level *L, double time, double dt);   /* 0 ok, 1 density failure, 2 dt check failed */
/* Castro::advance with castro.use_retry = 1 (Castro_advance_ctu.cpp:403-768): 0 ok, -1 / -2 = the reference's aborts */
int  ora_level_advance_retry(ora_level *L, double time, double dt, double retry_subcycle_factor,
                             int max_subcycles, double dt_cutoff);
int  ora_level_nsubcycles(ora_level *L);
int  ora_level_nretries(ora_level *L);
double *ora_level_old_state(ora_level *L);
/* castro.do_grav with gravity.gravity_type = ConstantGrav: sources in do_advance_ctu (Castro_advance_ctu.cpp:113-143,256-274) */
void ora_level_set_gravity(ora_level *L, int do_grav, double const_grav, int grav_source_type);
void ora_level_set_rotation(ora_level *L, int do_rot, const ora_rotation *R);
double ora_level_last_hydro_seconds(ora_level *L);

/* ---------------- subcycled AMR driver, one nested box per level (ora_amr_level.c) -------- */
typedef struct ora_amr ora_amr;
ora_amr *ora_amr_create(int nlev, const int *boxes /* nlev x (lo[3], hi[3]), each in its own level's index space */,
                        const ora_geom *G0, const ora_params *P, int nthreads);
void ora_amr_destroy(ora_amr *A);
void ora_amr_init_sedov(ora_amr *A, double r_init, double p_ambient, double exp_energy, double dens_ambient, int nsub);
void ora_amr_init_sod(ora_amr *A, double rho_l, double u_l, double p_l, double rho_r, double u_r, double p_r, int idir, double frac);
void ora_amr_set_state(ora_amr *A, int level, const double *data);
void ora_amr_post_init(ora_amr *A, int clean_first);   /* (clean_state on every level,) average down from the finest level */
double ora_amr_step(ora_amr *A, double stop_time);      /* coarse dt taken, or a negative status */
double *ora_amr_state(ora_amr *A, int level);            /* new-time state on the level's valid box */
double ora_amr_time(ora_amr *A);
int ora_amr_status(ora_amr *A);

#ifdef __cplusplus
}
#endif
#endif
