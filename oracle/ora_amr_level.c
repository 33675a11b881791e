/*
 * ora_amr_level.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see castro_oracle.h): an independent restatement of the
 * subcycled AMR time step for a hierarchy of nested boxes, one box per level, refinement ratio 2.
 *
 * What is restated here is the ORDER of operations, from the reference where it is in the tree and from AMReX's
 * published behaviour [3P] where it is not:
 *   Amr::coarseTimeStep / Amr::timeStep [3P]   compute dt for all levels, advance level l, then twice level l+1
 *                                              (n_cycle = 2), then post_timestep(l)
 *   Castro::computeNewDt / computeInitialDt    Source/driver/Castro.cpp:1629-1866 (all levels, dt_0 / n_factor)
 *   Castro::advance                            Source/driver/Castro_advance.cpp:19-121:
 *       initialize_advance (:232-410)          swap time levels, clean_state(S_old) ON THE STATE DATA, zero the fluxes
 *       subcycle_advance_ctu                   Castro_advance_ctu.cpp:507-768 -- one subcycle of (time + dt) - time;
 *                                              a rejected step makes this oracle give up (status < 0), it does not retry
 *       initialize_do_advance (:124-209)       Sborder = FillPatch(S_old, prev_time, 4 ghosts); clean_state(Sborder, 4)
 *       do_advance_ctu                         Castro_advance_ctu.cpp:15-397
 *       finalize_advance (:413-470)            FluxRegCrseInit (-1 x this level's fluxes into the finer level's
 *                                              registers, COPY), FluxRegFineAdd (+ this level's fluxes into its own)
 *   Castro::post_timestep                      Castro.cpp:1871-1920: reflux(level, level+1), avgDown, clean_state(S_new)
 *                                              -- clean_state on EVERY level, the finest included
 *   Castro::reflux / avgDown                   Castro.cpp:2549-2700, 3096-3113
 *   AmrLevel::FillPatch [3P]                   fine ghost zones = cell_cons_interp (Castro_setup.cpp:352-364) of the
 *                                              coarse STATE DATA interpolated in time between its old and new time
 *                                              levels (the coarse zones outside the domain by the coarse level's
 *                                              physical-boundary fill), then the fine level's physical-boundary fill.
 *                                              The coarse data are NOT cleaned on the way: clean_state reaches the fine
 *                                              ghost zones through clean_state(Sborder).
 *                                              Time interpolation: (1 - a) old + a new with a = 0 for the first fine
 *                                              step of a coarse step and a = 1/2 for the second (the exact fractions;
 *                                              AMReX computes a from the time stamps [3P]).
 * The per-zone arithmetic is the oracle's own (ora_ctu_hydro.c, ora_state.c, ora_amr.c).  The data structures are
 * deliberately not castro_amd/amr.py's: whole-level arrays, no overlap tables, no batching.
 */
#include <stdio.h>
#include "ora_internal.h"

#define AMR_MAXLEV 6

typedef struct {
    int lo[3], hi[3];            /* valid box in this level's index space */
    int glo[3], ghi[3];          /* grown by NUM_GROW */
    int plo[3], phi[3];          /* the box in the parent's index space (levels >= 1) */
    ora_geom G;
    double *S_old, *S_new;       /* state data on the valid box */
    double *Sborder;             /* grown by NUM_GROW */
    double *flux[3], *mflux[3];
    double *reg[3][2];           /* flux registers: parent-level faces on the low / high side of the box (levels >= 1) */
    int rlo[3][2][3], rhi[3][2][3];
} amr_lev;

struct ora_amr {
    int nlev, nthreads;
    ora_params P;
    amr_lev L[AMR_MAXLEV];
    double dt_level[AMR_MAXLEV];
    double time;
    int nstep, status;
};

static size_t box_zones(const int lo[3], const int hi[3])
{
    return (size_t)(hi[0] - lo[0] + 1) * (size_t)(hi[1] - lo[1] + 1) * (size_t)(hi[2] - lo[2] + 1);
}

static int fdiv2(int i) { return (i >= 0) ? i / 2 : -((-i + 1) / 2); }

/* boxes[l] = (lo, hi) of level l in ITS OWN index space; level 0 is the domain (domlo = 0) */
ora_amr *ora_amr_create(int nlev, const int *boxes, const ora_geom *G0, const ora_params *P, int nthreads)
{
    if (nlev < 1 || nlev > AMR_MAXLEV) return NULL;
    ora_amr *A = (ora_amr *)calloc(1, sizeof(ora_amr));
    A->nlev = nlev; A->nthreads = nthreads; A->P = *P;
    for (int l = 0; l < nlev; ++l) {
        amr_lev *L = &A->L[l];
        L->G = *G0;
        for (int d = 0; d < 3; ++d) {
            L->lo[d] = boxes[6 * l + d]; L->hi[d] = boxes[6 * l + 3 + d];
            L->glo[d] = L->lo[d] - NUM_GROW; L->ghi[d] = L->hi[d] + NUM_GROW;
            L->G.dx[d] = G0->dx[d] / (double)(1 << l);
            L->G.domlo[d] = G0->domlo[d] * (1 << l);
            L->G.domhi[d] = (G0->domhi[d] + 1) * (1 << l) - 1;
            if (l > 0) {
                if ((L->lo[d] & 1) || !((L->hi[d] + 1) % 2 == 0)) { free(A); return NULL; }   /* boxes of a refined level are coarsenable */
                L->plo[d] = fdiv2(L->lo[d]); L->phi[d] = fdiv2(L->hi[d]);
            }
        }
        const size_t nv = box_zones(L->lo, L->hi), ng = box_zones(L->glo, L->ghi);
        L->S_old = (double *)calloc(nv * NUM_STATE, sizeof(double));
        L->S_new = (double *)calloc(nv * NUM_STATE, sizeof(double));
        L->Sborder = (double *)calloc(ng * NUM_STATE, sizeof(double));
        for (int d = 0; d < 3; ++d) {
            int fhi[3] = { L->hi[0], L->hi[1], L->hi[2] };
            fhi[d] += 1;
            L->flux[d] = (double *)calloc(box_zones(L->lo, fhi) * NUM_STATE, sizeof(double));
            L->mflux[d] = (double *)calloc(box_zones(L->lo, fhi), sizeof(double));
            if (l > 0)
                for (int side = 0; side < 2; ++side) {
                    for (int e = 0; e < 3; ++e) { L->rlo[d][side][e] = L->plo[e]; L->rhi[d][side][e] = L->phi[e]; }
                    L->rlo[d][side][d] = L->rhi[d][side][d] = side == 0 ? L->plo[d] : L->phi[d] + 1;
                    L->reg[d][side] = (double *)calloc(box_zones(L->rlo[d][side], L->rhi[d][side]) * NUM_STATE, sizeof(double));
                }
        }
    }
    return A;
}

void ora_amr_destroy(ora_amr *A)
{
    if (!A) return;
    for (int l = 0; l < A->nlev; ++l) {
        amr_lev *L = &A->L[l];
        free(L->S_old); free(L->S_new); free(L->Sborder);
        for (int d = 0; d < 3; ++d) { free(L->flux[d]); free(L->mflux[d]); free(L->reg[d][0]); free(L->reg[d][1]); }
    }
    free(A);
}

double *ora_amr_state(ora_amr *A, int l) { return A->L[l].S_new; }
double ora_amr_time(ora_amr *A) { return A->time; }
int ora_amr_status(ora_amr *A) { return A->status; }

static ora_a4 a4_new(amr_lev *L) { return ora_make_a4(L->S_new, L->lo, L->hi, NUM_STATE); }
static ora_a4 a4_old(amr_lev *L) { return ora_make_a4(L->S_old, L->lo, L->hi, NUM_STATE); }

/* Castro::avgDown + the clean_state of post_timestep / post_init on the coarser level */
static void avg_down(ora_amr *A, int l)      /* level l+1 onto level l */
{
    amr_lev *F = &A->L[l + 1], *C = &A->L[l];
    ora_avgdown(F->plo, F->phi, a4_new(F), a4_new(C), NUM_STATE);
}

/* Castro::initData ends with clean_state on the level's fresh data (Castro.cpp:1100-1160); Castro::post_init
 * (Castro.cpp:2220-2235) then averages down from the finest level -- nothing else: the first clean_state the averaged
 * zones see is initialize_advance's */
void ora_amr_post_init(ora_amr *A, int clean_first)
{
    ora_set_state_threads(A->nthreads);
    if (clean_first)
        for (int l = 0; l < A->nlev; ++l) ora_clean_state(A->L[l].lo, A->L[l].hi, a4_new(&A->L[l]), &A->P);
    for (int l = A->nlev - 2; l >= 0; --l) avg_down(A, l);
    A->time = 0.0; A->nstep = 0; A->status = 0;
}

/* overwrite the new-time state of a level (tests: data on which clean_state is not idempotent) */
void ora_amr_set_state(ora_amr *A, int l, const double *data)
{
    memcpy(A->L[l].S_new, data, sizeof(double) * box_zones(A->L[l].lo, A->L[l].hi) * NUM_STATE);
}

void ora_amr_init_sedov(ora_amr *A, double r_init, double p_ambient, double exp_energy, double dens_ambient, int nsub)
{
    for (int l = 0; l < A->nlev; ++l)
        ora_sedov_init(A->L[l].lo, A->L[l].hi, a4_new(&A->L[l]), &A->L[l].G, &A->P, r_init, p_ambient, exp_energy, dens_ambient, nsub);
    ora_amr_post_init(A, 1);
}

void ora_amr_init_sod(ora_amr *A, double rho_l, double u_l, double p_l, double rho_r, double u_r, double p_r, int idir, double frac)
{
    for (int l = 0; l < A->nlev; ++l)
        ora_sod_init(A->L[l].lo, A->L[l].hi, a4_new(&A->L[l]), &A->L[l].G, &A->P, rho_l, u_l, p_l, rho_r, u_r, p_r, idir, frac);
    ora_amr_post_init(A, 1);
}

/* AmrLevel::FillPatch of the state of level l at its OLD time into Sborder; a = position of that time inside the
 * parent's [old, new] interval.  Returns 0, or -3 when a coarse zone the interpolation needs lies inside the domain but
 * outside the parent's box (not properly nested). */
static int fill_patch(ora_amr *A, int l, double a)
{
    amr_lev *L = &A->L[l];
    ora_a4 Sb = ora_make_a4(L->Sborder, L->glo, L->ghi, NUM_STATE);
    if (l > 0) {
        amr_lev *C = &A->L[l - 1];
        /* the parent's state at this time on the coarse zones under the grown box, grown by one for the slopes */
        int clo[3], chi[3];
        for (int d = 0; d < 3; ++d) { clo[d] = fdiv2(L->glo[d]) - 1; chi[d] = fdiv2(L->ghi[d]) + 1; }
        const size_t nc = box_zones(clo, chi);
        double *ct = (double *)malloc(sizeof(double) * nc * NUM_STATE);
        ora_a4 CT = ora_make_a4(ct, clo, chi, NUM_STATE);
        for (int d = 0; d < 3; ++d) {
            const int inlo = clo[d] < C->G.domlo[d] ? C->G.domlo[d] : clo[d];
            const int inhi = chi[d] > C->G.domhi[d] ? C->G.domhi[d] : chi[d];
            if (inlo < C->lo[d] || inhi > C->hi[d]) { free(ct); return -3; }
        }
        int ilo[3], ihi[3];
        for (int d = 0; d < 3; ++d) { ilo[d] = clo[d] < C->lo[d] ? C->lo[d] : clo[d]; ihi[d] = chi[d] > C->hi[d] ? C->hi[d] : chi[d]; }
        ora_lincomb(ilo, ihi, CT, 1.0 - a, a4_old(C), a, a4_new(C), NUM_STATE);
        ora_bc_fill(CT, &C->G);                       /* coarse zones outside the domain */
        /* every zone of the grown box (the valid zones are overwritten next) */
        ora_cc_interp(L->glo, L->ghi, CT, Sb, NUM_STATE);
        free(ct);
    }
    ora_fill_interior_copy(Sb, a4_old(L), L->lo, L->hi);
    ora_bc_fill(Sb, &L->G);
    return 0;
}

static void lev_fabs(amr_lev *L, ora_a4 fl[3], ora_a4 mf[3], ora_a4 qe[3])
{
    for (int d = 0; d < 3; ++d) {
        int fhi[3] = { L->hi[0], L->hi[1], L->hi[2] };
        fhi[d] += 1;
        fl[d] = ora_make_a4(L->flux[d], L->lo, fhi, NUM_STATE);
        mf[d] = ora_make_a4(L->mflux[d], L->lo, fhi, 1);
        qe[d].p = NULL;
    }
}

/* Castro::estTimeStep (hydro limiter) */
static double est_time_step(ora_amr *A, int l)
{
    amr_lev *L = &A->L[l];
    double e = ora_estdt_cfl(L->lo, L->hi, a4_new(L), &L->G, &A->P) * A->P.cfl;
    return e < 1.e200 ? e : 1.e200;
}

/* Castro::advance for level l.  0, or < 0: the step was rejected (this oracle does not retry) / bad nesting */
static int advance_level(ora_amr *A, int l, double time, double dt, double a)
{
    amr_lev *L = &A->L[l];
    const ora_params *P = &A->P;
    ora_set_state_threads(A->nthreads);
    /* initialize_advance: swap, clean_state(S_old), zero the fluxes */
    { double *t = L->S_old; L->S_old = L->S_new; L->S_new = t; }
    ora_clean_state(L->lo, L->hi, a4_old(L), P);
    ora_a4 fl[3], mf[3], qe[3];
    lev_fabs(L, fl, mf, qe);
    for (int d = 0; d < 3; ++d) {
        memset(L->flux[d], 0, sizeof(double) * (size_t)fl[d].sn * NUM_STATE);
        memset(L->mflux[d], 0, sizeof(double) * (size_t)mf[d].sn);
    }
    /* subcycle_advance_ctu with one subcycle: do_advance_ctu(time, (time + dt) - time) */
    const double dts = (time + dt) - time;
    /* initialize_do_advance */
    int rc = fill_patch(A, l, a);
    if (rc) return rc;
    ora_a4 Sb = ora_make_a4(L->Sborder, L->glo, L->ghi, NUM_STATE);
    ora_clean_state(L->glo, L->ghi, Sb, P);
    /* do_advance_ctu */
    ora_a4 Sn = a4_new(L);
    ora_fill_interior_copy(Sn, Sb, L->lo, L->hi);
    ora_a4 nosrc; memset(&nosrc, 0, sizeof(nosrc));
    const int tile[3] = { 1024, 16, 16 };
    ora_construct_ctu_hydro_source(L->lo, L->hi, Sb, nosrc, Sn, fl, mf, qe, &L->G, P, time, dts, tile, A->nthreads);
    if (ora_min_density(L->lo, L->hi, Sn) < P->small_dens) return -1;
    ora_clean_state(L->lo, L->hi, Sn, P);
    {   /* the check inside the advance: guarded minimum (a NaN zone rejects the step) */
        double e = ora_estdt_cfl_guarded(L->lo, L->hi, a4_new(L), &L->G, &A->P) * A->P.cfl;
        if (P->change_max * (e < 1.e200 ? e : 1.e200) < dts) return -2;
    }
    /* finalize_advance: FluxRegCrseInit, FluxRegFineAdd */
    if (l + 1 < A->nlev) {
        amr_lev *F = &A->L[l + 1];
        for (int d = 0; d < 3; ++d)
            for (int side = 0; side < 2; ++side)
                ora_reg_crse_init(F->rlo[d][side], F->rhi[d][side], ora_make_a4(F->reg[d][side], F->rlo[d][side], F->rhi[d][side], NUM_STATE),
                                  fl[d], NUM_STATE, -1.0);
    }
    if (l > 0)
        for (int d = 0; d < 3; ++d)
            for (int side = 0; side < 2; ++side)
                ora_reg_fine_add(L->rlo[d][side], L->rhi[d][side], ora_make_a4(L->reg[d][side], L->rlo[d][side], L->rhi[d][side], NUM_STATE),
                                 fl[d], d, NUM_STATE, 1.0);
    return 0;
}

/* Castro::post_timestep for level l */
static void post_timestep(ora_amr *A, int l)
{
    amr_lev *L = &A->L[l];
    if (l + 1 < A->nlev) {
        amr_lev *F = &A->L[l + 1];
        const double vol = L->G.dx[0] * L->G.dx[1] * L->G.dx[2];
        /* FluxRegister::Reflux, orientation by orientation; a face of the fine box on the domain boundary has no
         * coarse zone outside */
        for (int d = 0; d < 3; ++d)
            for (int side = 0; side < 2; ++side) {
                const int zone = side == 0 ? F->plo[d] - 1 : F->phi[d] + 1;
                if (zone < L->lo[d] || zone > L->hi[d]) continue;
                ora_reflux(F->rlo[d][side], F->rhi[d][side], a4_new(L),
                           ora_make_a4(F->reg[d][side], F->rlo[d][side], F->rhi[d][side], NUM_STATE), d, side, NUM_STATE, vol);
            }
        avg_down(A, l);
    }
    ora_set_state_threads(A->nthreads);
    ora_clean_state(L->lo, L->hi, a4_new(L), &A->P);
}

/* Amr::timeStep */
static int time_step(ora_amr *A, int l, double time, double dt, double a)
{
    int rc = advance_level(A, l, time, dt, a);
    if (rc) return rc;
    if (l + 1 < A->nlev)
        for (int it = 0; it < 2; ++it) {
            rc = time_step(A, l + 1, time + it * (dt / 2), dt / 2, 0.5 * it);
            if (rc) return rc;
        }
    post_timestep(A, l);
    return 0;
}

/* Amr::coarseTimeStep: returns the coarse dt taken, or a negative status */
double ora_amr_step(ora_amr *A, double stop_time)
{
    const ora_params *P = &A->P;
    double dt_0 = 1.0e+100;
    int n_factor = 1;
    if (A->nstep == 0) {
        /* computeInitialDt with initialTimeStep = init_shrink * estTimeStep (Castro.cpp:1490-1504, 1822-1866) */
        for (int i = 0; i < A->nlev; ++i) {
            A->dt_level[i] = P->init_shrink * est_time_step(A, i);
            n_factor *= (i == 0 ? 1 : 2);
            dt_0 = amin(dt_0, n_factor * A->dt_level[i]);
        }
        const double eps = 0.001 * dt_0;
        if (stop_time >= 0.0 && (A->time + dt_0) > (stop_time - eps)) dt_0 = stop_time - A->time;
    } else {
        /* computeNewDt (Castro.cpp:1629-1819) */
        double dt_min[AMR_MAXLEV];
        for (int i = 0; i < A->nlev; ++i) dt_min[i] = est_time_step(A, i);
        for (int i = 0; i < A->nlev; ++i) dt_min[i] = amin(dt_min[i], P->change_max * A->dt_level[i]);
        for (int i = 0; i < A->nlev; ++i) {
            n_factor *= (i == 0 ? 1 : 2);
            dt_0 = amin(dt_0, n_factor * dt_min[i]);
        }
        const double eps = 2.220446049250313e-16;
        if (stop_time >= 0.0 && (A->time + dt_0) >= (stop_time - eps)) dt_0 = stop_time - A->time;
    }
    n_factor = 1;
    for (int i = 0; i < A->nlev; ++i) {
        n_factor *= (i == 0 ? 1 : 2);
        A->dt_level[i] = dt_0 / n_factor;
    }
    const int rc = time_step(A, 0, A->time, dt_0, 0.0);
    if (rc) { A->status = rc; return (double)rc; }
    A->time += dt_0;
    A->nstep += 1;
    return dt_0;
}
