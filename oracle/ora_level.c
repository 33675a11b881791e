/*
 * ora_level.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see castro_oracle.h).
 * Single-box, single-level mirror of the reference's time-step driver around
 * the hot path:
 *   Castro::advance / initialize_advance    Source/driver/Castro_advance.cpp:19-121,232-410
 *   Castro::do_advance_ctu                  Source/driver/Castro_advance_ctu.cpp:15-397
 *   retry_advance_ctu / subcycle_advance_ctu Source/driver/Castro_advance_ctu.cpp:403-768
 *   initialize_do_advance (FillPatch+clean) Source/driver/Castro_advance.cpp:124-209
 *   estTimeStep / computeNewDt / computeInitialDt / initialTimeStep
 *                                           Source/driver/Castro.cpp:1490-1866
 * plus the Sod initial data (Exec/hydro_tests/Sod/problem_initialize*.H).
 */
#include <stdio.h>
#include <time.h>
#include "ora_internal.h"

struct ora_level {
    int n[3];
    int lo[3], hi[3];       /* valid box = domain */
    int glo[3], ghi[3];     /* grown by NUM_GROW */
    ora_geom G;
    ora_params P;
    int nthreads;
    int tile[3];
    double *S_new, *S_old, *Sborder;
    double *fluxes[3], *mass_fluxes[3];
    double hydro_seconds;
    double *prev_old;       /* prev_state old data kept across a retry (Castro_advance_ctu.cpp:430-451) */
    int nsubcycles, nretries;
    int do_grav, grav_source_type;   /* castro.do_grav, castro.grav_source_type; gravity.const_grav along z */
    double grav[3];
    double *old_source, *new_source; /* Source_Type old (NUM_GROW_SRC ghosts) / new data */
    double *source_corrector;        /* Castro::source_corrector (NSRC comps, NUM_GROW_SRC ghosts), source_term_predictor = 1 */
    double lastDt;                   /* Castro.cpp:906, Castro_advance_ctu.cpp:715 */
    int in_retry;
    int do_rot;                      /* castro.do_rotation */
    ora_rotation rot;
};

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

ora_level *ora_level_create(const int n[3], const ora_geom *G, const ora_params *P, int nthreads)
{
    ora_level *L = (ora_level *)calloc(1, sizeof(ora_level));
    L->G = *G;
    L->P = *P;
    L->nthreads = nthreads;
    /* hydro_tile_size default on CPU, Source/driver/Castro.cpp:133 */
    L->tile[0] = 1024; L->tile[1] = 16; L->tile[2] = 16;
    L->lastDt = 1.e200;
    size_t nv = 1, ng = 1;
    for (int d = 0; d < 3; ++d) {
        L->n[d] = n[d];
        L->lo[d] = G->domlo[d];
        L->hi[d] = G->domlo[d] + n[d] - 1;
        L->glo[d] = L->lo[d] - NUM_GROW;
        L->ghi[d] = L->hi[d] + NUM_GROW;
        nv *= (size_t)n[d];
        ng *= (size_t)(n[d] + 2 * NUM_GROW);
    }
    L->S_new = (double *)calloc(nv * NUM_STATE, sizeof(double));
    L->S_old = (double *)calloc(nv * NUM_STATE, sizeof(double));
    L->Sborder = (double *)calloc(ng * NUM_STATE, sizeof(double));
    for (int d = 0; d < 3; ++d) {
        size_t nf = nv / n[d] * (n[d] + 1);
        L->fluxes[d] = (double *)calloc(nf * NUM_STATE, sizeof(double));
        L->mass_fluxes[d] = (double *)calloc(nf, sizeof(double));
    }
    return L;
}

void ora_level_destroy(ora_level *L)
{
    if (!L) return;
    free(L->S_new); free(L->S_old); free(L->Sborder); free(L->prev_old); free(L->old_source); free(L->new_source);
    free(L->source_corrector);
    for (int d = 0; d < 3; ++d) { free(L->fluxes[d]); free(L->mass_fluxes[d]); }
    free(L);
}

double *ora_level_state(ora_level *L) { return L->S_new; }
/* restart from a state handed in from outside (a checkpoint in the reference's terms, Castro_io.cpp:restart): the data have
 * been copied into ora_level_state() by the caller as they are -- no clean_state, a checkpoint holds the state post_timestep left --
 * and the level remembers the last time step it took (Castro.cpp:906) */
void ora_level_set_last_dt(ora_level *L, double last_dt) { L->lastDt = last_dt; }
double *ora_level_flux(ora_level *L, int dir) { return L->fluxes[dir]; }
double *ora_level_mass_flux(ora_level *L, int dir) { return L->mass_fluxes[dir]; }
void ora_level_set_tile(ora_level *L, const int tile[3]) { for (int d = 0; d < 3; ++d) L->tile[d] = tile[d]; }
double ora_level_last_hydro_seconds(ora_level *L) { return L->hydro_seconds; }

/* Castro::initData tail: clean_state on the fresh data (Castro.cpp:1100-1160) */
void ora_level_post_init(ora_level *L)
{
    ora_a4 S = ora_make_a4(L->S_new, L->lo, L->hi, NUM_STATE);
    ora_clean_state(L->lo, L->hi, S, &L->P);
}

/* Castro::estTimeStep, Castro.cpp:1507-1626 (hydro limiter only; max_dt = 1e200) */
double ora_level_est_time_step(ora_level *L)
{
    ora_set_state_threads(L->nthreads);
    ora_a4 S = ora_make_a4(L->S_new, L->lo, L->hi, NUM_STATE);
    double estdt = 1.e200;
    double estdt_hydro = ora_estdt_cfl(L->lo, L->hi, S, &L->G, &L->P);
    estdt_hydro *= L->P.cfl;
    if (estdt_hydro < estdt) estdt = estdt_hydro;
    return estdt;
}

/* computeInitialDt (Castro.cpp:1822-1866) with initialTimeStep (:1490-1504) */
double ora_level_initial_dt(ora_level *L, double stop_time)
{
    double dt_0 = L->P.init_shrink * ora_level_est_time_step(L);
    const double eps = 0.001 * dt_0;
    double cur_time = 0.0;
    if (stop_time >= 0.0) {
        if ((cur_time + dt_0) > (stop_time - eps)) dt_0 = stop_time - cur_time;
    }
    return dt_0;
}

/* computeNewDt (Castro.cpp:1629-1819), single level, no plot_per limiting */
double ora_level_new_dt(ora_level *L, double dt_old, double cur_time, double stop_time)
{
    double dt_0 = ora_level_est_time_step(L);
    dt_0 = (dt_0 < L->P.change_max * dt_old) ? dt_0 : L->P.change_max * dt_old;   /* std::min */
    const double eps = 2.220446049250313e-16;
    if (stop_time >= 0.0) {
        if ((cur_time + dt_0) >= (stop_time - eps)) dt_0 = stop_time - cur_time;
    }
    return dt_0;
}

/* One level advance.  0 ok; 1 = density failure; 2 = dt validity check failed */
static void level_fabs(ora_level *L, ora_a4 fl[3], ora_a4 mf[3], ora_a4 qe[3])
{
    for (int d = 0; d < 3; ++d) {
        int fhi[3] = { L->hi[0], L->hi[1], L->hi[2] };
        fhi[d] += 1;
        fl[d] = ora_make_a4(L->fluxes[d], L->lo, fhi, NUM_STATE);
        mf[d] = ora_make_a4(L->mass_fluxes[d], L->lo, fhi, 1);
        qe[d].p = NULL;
    }
}

static void level_swap(ora_level *L) { double *t = L->S_old; L->S_old = L->S_new; L->S_new = t; }

/* zero the flux registers (Castro_advance.cpp:391-394; retry: Castro_advance_ctu.cpp:455-461) */
static void level_zero_fluxes(ora_level *L)
{
    ora_a4 fl[3], mf[3], qe[3];
    level_fabs(L, fl, mf, qe);
    for (int d = 0; d < 3; ++d) {
        memset(L->fluxes[d], 0, sizeof(double) * (size_t)fl[d].sn * NUM_STATE);
        memset(L->mass_fluxes[d], 0, sizeof(double) * (size_t)mf[d].sn);
    }
}

/* Castro::do_advance_ctu (Castro_advance_ctu.cpp:15-397) on the current time levels.
 * Returns 0, 1 (small/negative density) or 2 (timestep validity check failed). */
static int level_do_advance(ora_level *L, double time, double dt)
{
    const ora_params *P = &L->P;
    ora_set_state_threads(L->nthreads);
    ora_a4 S_old = ora_make_a4(L->S_old, L->lo, L->hi, NUM_STATE);
    ora_a4 S_new = ora_make_a4(L->S_new, L->lo, L->hi, NUM_STATE);
    ora_a4 fl[3], mf[3], qe[3];
    level_fabs(L, fl, mf, qe);

    /* initialize_do_advance: Sborder = FillPatch(S_old, 4 ghosts); clean_state(Sborder, 4)
     * (clean_state(S_old) is initialize_advance's: once per step, not once per attempt) */
    ora_a4 Sb = ora_make_a4(L->Sborder, L->glo, L->ghi, NUM_STATE);
    ora_fill_interior_copy(Sb, S_old, L->lo, L->hi);
    ora_bc_fill(Sb, &L->G);
    ora_clean_state(L->glo, L->ghi, Sb, P);

    /* MultiFab::Copy(S_new, Sborder) (Castro_advance_ctu.cpp:94) */
    ora_fill_interior_copy(S_new, Sb, L->lo, L->hi);

    /* old-time sources (:113-143): do_old_sources = construct, apply to S_new with full dt, clean_state;
     * then FillPatch the ghost zones of old_source for the tracing */
    ora_a4 src; memset(&src, 0, sizeof(src));
    ora_a4 osrc, nsrc;
    int slo[3], shi[3];
    const int have_src = L->do_grav || L->do_rot;
    if (have_src) {
        size_t ns = 1, nv1 = 1;
        for (int d = 0; d < 3; ++d) {
            slo[d] = L->lo[d] - NUM_GROW_SRC; shi[d] = L->hi[d] + NUM_GROW_SRC;
            ns *= (size_t)(L->n[d] + 2 * NUM_GROW_SRC); nv1 *= (size_t)L->n[d];
        }
        if (!L->old_source) L->old_source = (double *)malloc(sizeof(double) * ns * NSRC);
        if (!L->new_source) L->new_source = (double *)calloc(nv1 * NSRC, sizeof(double));
        memset(L->old_source, 0, sizeof(double) * ns * NSRC);
        osrc = ora_make_a4(L->old_source, slo, shi, NSRC);
        nsrc = ora_make_a4(L->new_source, L->lo, L->hi, NSRC);
        if (P->source_term_predictor == 1) {
            /* create_source_corrector (Castro_advance_ctu.cpp:60-62, Castro.cpp:3780-3818): FillPatch of the old
             * Source_Type data (the corrector of the last advance after the swap), momentum components, x 2 / lastDt */
            if (!L->source_corrector) L->source_corrector = (double *)calloc(ns * NSRC, sizeof(double));
            ora_a4 corr = ora_make_a4(L->source_corrector, slo, shi, NSRC);
            if (!L->in_retry) {
                memset(L->source_corrector, 0, sizeof(double) * ns * NSRC);
                for (int n = UMX; n <= UMZ; ++n)
                for (int k = L->lo[2]; k <= L->hi[2]; ++k)
                for (int j = L->lo[1]; j <= L->hi[1]; ++j)
                for (int i = L->lo[0]; i <= L->hi[0]; ++i) A4(corr,i,j,k,n) = A4(nsrc,i,j,k,n);
                ora_bc_fill(corr, &L->G);
                const double f = 2.0 / L->lastDt;
                for (size_t m = 0; m < ns * NSRC; ++m) L->source_corrector[m] *= f;
            }
            ora_set_source_corrector(&corr);
        }
        if (L->do_grav) ora_old_gravity_source(L->lo, L->hi, Sb, osrc, L->grav, L->grav_source_type, dt);
        if (L->do_rot) ora_old_rotation_source(L->lo, L->hi, Sb, osrc, &L->rot, &L->G, dt);
        ora_saxpy(L->lo, L->hi, S_new, dt, osrc, NSRC);
        ora_clean_state(L->lo, L->hi, S_new, P);
        ora_bc_fill(osrc, &L->G);
        src = osrc;
    }

    /* construct_ctu_hydro_source (:156) */
    double t0 = now_s();
    int bad = ora_construct_ctu_hydro_source(L->lo, L->hi, Sb, src, S_new, fl, mf, qe, &L->G, P,
                                             time, dt, L->tile, L->nthreads);
    L->hydro_seconds = now_s() - t0;
    (void)bad;
    ora_set_source_corrector(NULL);

    /* small/negative density check (:168-216); retry_small_density_cutoff keeps its default (-1e200) */
    if (ora_min_density(L->lo, L->hi, S_new) < P->small_dens) return 1;

    /* clean_state(S_new) (:221-225) */
    ora_clean_state(L->lo, L->hi, S_new, P);

    /* new-time sources (:256-274): do_new_sources = construct the corrector, apply, clean_state */
    if (have_src) {
        memset(L->new_source, 0, sizeof(double) * (size_t)nsrc.sn * NSRC);
        if (L->do_grav) ora_new_gravity_source(L->lo, L->hi, Sb, S_new, nsrc, mf, L->grav, L->grav_source_type, dt, L->G.dx);
        if (L->do_rot) ora_new_rotation_source(L->lo, L->hi, Sb, S_new, nsrc, mf, &L->rot, &L->G, dt);
        ora_saxpy(L->lo, L->hi, S_new, dt, nsrc, NSRC);
        ora_clean_state(L->lo, L->hi, S_new, P);
    }

    /* timestep validity check (:386-392); the guarded minimum: a NaN zone rejects the step */
    double new_dt = 1.e200;
    {
        double e = ora_estdt_cfl_guarded(L->lo, L->hi, ora_make_a4(L->S_new, L->lo, L->hi, NUM_STATE), &L->G, &L->P) * L->P.cfl;
        if (e < new_dt) new_dt = e;
    }
    if (P->change_max * new_dt < dt) return 2;
    return 0;
}

/* Castro::advance without retries: one do_advance_ctu over the whole step */
/* initialize_advance (Castro_advance.cpp:232-410): swap the time levels, clean_state(S_old) on the state data, zero
 * the fluxes */
static void level_initialize_advance(ora_level *L)
{
    level_swap(L);
    ora_set_state_threads(L->nthreads);
    ora_clean_state(L->lo, L->hi, ora_make_a4(L->S_old, L->lo, L->hi, NUM_STATE), &L->P);
    level_zero_fluxes(L);
}

/* Castro::post_timestep (Castro.cpp:1871-1916) on a single level: clean_state(S_new) */
void ora_level_post_timestep(ora_level *L)
{
    ora_set_state_threads(L->nthreads);
    ora_clean_state(L->lo, L->hi, ora_make_a4(L->S_new, L->lo, L->hi, NUM_STATE), &L->P);
}

int ora_level_advance(ora_level *L, double time, double dt)
{
    level_initialize_advance(L);
    L->in_retry = 0;
    const int st = level_do_advance(L, time, dt);
    if (st == 0) L->lastDt = dt;
    return st;
}

/* Castro::advance with castro.use_retry = 1: initialize_advance + subcycle_advance_ctu
 * (Castro_advance_ctu.cpp:507-768) + retry_advance_ctu (:403-503).
 * Returns 0, or -1 "subcycled timesteps too short", -2 "too many subcycles" (amrex::Abort in the reference). */
int ora_level_advance_retry(ora_level *L, double time, double dt, double retry_subcycle_factor,
                            int max_subcycles, double dt_cutoff)
{
    const size_t nv = (size_t)L->n[0] * L->n[1] * L->n[2] * NUM_STATE;
    level_initialize_advance(L);
    double dt_subcycle = 1.e200;
    int have_prev = 0;

    if (dt_subcycle == 1.e200) dt_subcycle = dt;
    double subcycle_time = time;
    int sub_iteration = 0;
    const double eps = 1.0e-14;
    int do_swap = 0;
    L->nsubcycles = 0; L->nretries = 0;

    while (subcycle_time < (1.0 - eps) * (time + dt)) {
        if (subcycle_time + dt_subcycle > (1.0 - dt_cutoff) * (time + dt)) {
            dt_subcycle = (time + dt) - subcycle_time;
        }
        if (dt_subcycle <= dt_cutoff * time) return -1;
        int num_subcycles_remaining = (int)round(((time + dt) - subcycle_time) / dt_subcycle);
        if (num_subcycles_remaining > max_subcycles) return -2;

        if (do_swap) level_swap(L); else do_swap = 1;

        int status = level_do_advance(L, subcycle_time, dt_subcycle);
        L->in_retry = 0;

        if (status != 0) {
            /* retry_advance_ctu */
            dt_subcycle = amin(dt_subcycle, dt_subcycle) * retry_subcycle_factor;
            if (!have_prev) {
                if (!L->prev_old) L->prev_old = (double *)malloc(sizeof(double) * nv);
                memcpy(L->prev_old, L->S_old, sizeof(double) * nv);
                have_prev = 1;
            }
            level_zero_fluxes(L);
            do_swap = 0;
            L->nretries += 1;
            L->in_retry = 1;
            continue;
        }
        subcycle_time += dt_subcycle;
        sub_iteration += 1;
        L->lastDt = dt_subcycle;
    }
    if (sub_iteration > 1 && have_prev) {
        /* state[k].replaceOldData(*prev_state[k]) (:716-726) */
        memcpy(L->S_old, L->prev_old, sizeof(double) * nv);
    }
    L->nsubcycles = sub_iteration;
    return 0;
}

void ora_level_set_gravity(ora_level *L, int do_grav, double const_grav, int grav_source_type)
{
    L->do_grav = do_grav; L->grav_source_type = grav_source_type;
    L->grav[0] = 0.0; L->grav[1] = 0.0; L->grav[2] = const_grav;     /* Gravity.cpp:860-866 */
}

void ora_level_set_rotation(ora_level *L, int do_rot, const ora_rotation *R)
{
    L->do_rot = do_rot;
    if (R) L->rot = *R;
}

int ora_level_nsubcycles(ora_level *L) { return L->nsubcycles; }
int ora_level_nretries(ora_level *L) { return L->nretries; }
double *ora_level_old_state(ora_level *L) { return L->S_old; }

/* ------------------------------------------------------------------ */
/* Exec/hydro_tests/Sod/problem_initialize.H + problem_initialize_state_data.H
 * (use_Tinit = 0).  idir is 1-based as in the reference's probin. */
void ora_sod_init(const int lo[3], const int hi[3], ora_a4 state, const ora_geom *G, const ora_params *P,
                  double rho_l, double u_l, double p_l, double rho_r, double u_r, double p_r,
                  int idir, double frac)
{
    double split[3];
    for (int d = 0; d < 3; ++d) split[d] = frac * (G->problo[d] + G->probhi[d]);

    ora_eos_t es;
    es.xn = 1.0;                                       /* Sod problem_initialize.H:29-30 */
    es.rho = rho_l; es.p = p_l; es.T = 100000.0;
    ora_eos_rp(P, &es);
    const double rhoe_l = rho_l * es.e;
    const double T_l = es.T;

    es.rho = rho_r; es.p = p_r; es.T = 100000.0;
    ora_eos_rp(P, &es);
    const double rhoe_r = rho_r * es.e;
    const double T_r = es.T;

    const double *dx = G->dx;
    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        double xyz[3] = { G->problo[0] + dx[0] * ((double)i + 0.5),
                          G->problo[1] + dx[1] * ((double)j + 0.5),
                          G->problo[2] + dx[2] * ((double)k + 0.5) };
        int left = xyz[idir - 1] <= split[idir - 1];
        double rho = left ? rho_l : rho_r;
        double u = left ? u_l : u_r;
        double rhoe = left ? rhoe_l : rhoe_r;
        A4(state,i,j,k,URHO) = rho;
        A4(state,i,j,k,UMX) = 0.0;
        A4(state,i,j,k,UMY) = 0.0;
        A4(state,i,j,k,UMZ) = 0.0;
        A4(state,i,j,k,UMX + idir - 1) = rho * u;
        A4(state,i,j,k,UEDEN) = rhoe + 0.5 * rho * u * u;
        A4(state,i,j,k,UEINT) = rhoe;
        A4(state,i,j,k,UTEMP) = left ? T_l : T_r;
        A4(state,i,j,k,UFS) = A4(state,i,j,k,URHO);
    }
}
