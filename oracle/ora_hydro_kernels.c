/*
 * ora_hydro_kernels.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see castro_oracle.h).
 * Restates: ctoprim, uflatten, shock, src_to_prim, divu, apply_av,
 * normalize_species_fluxes, scale_flux (Source/hydro/advection_util.cpp,
 * flatten.cpp), consup_hydro (Source/hydro/Castro_ctu.cpp), the gamma-law EOS
 * (Microphysics EOS/gamma_law, not in tree: SURVEY.md D.3) and the parameter
 * defaults (Source/driver/_cpp_parameters, Castro_setup.cpp:222-288).
 */
#include "ora_internal.h"

/* ------------------------------------------------------------------ */
/* parameters                                                          */
/* ------------------------------------------------------------------ */

/* Source/driver/_cpp_parameters:36-400 defaults; cfl/init_shrink/change_max as
 * set by Exec/hydro_tests/Sedov/inputs.3d.sph:23-28; eos_gamma=1.4 from
 * inputs.3d.sph.testsuite:80 */
void ora_default_params(ora_params *p)
{
    memset(p, 0, sizeof(*p));
    p->ppm_type = 1;
    p->riemann_solver = 0;
    p->use_flattening = 1;
    p->hybrid_riemann = 0;
    p->first_order_hydro = 0;
    p->cg_maxiter = 12;
    p->cg_blend = 2;
    p->transverse_use_eos = 0;
    p->transverse_reset_density = 1;
    p->transverse_reset_rhoe = 0;
    p->limit_fluxes_on_small_dens = 0;
    p->limit_fluxes_on_large_vel = 0;
    p->speed_limit = 0.0;
    p->source_term_predictor = 0;
    p->ppm_temp_fix = 0;
    p->plm_iorder = 2;
    p->plm_limiter = 2;
    p->use_pslope = 1;
    p->pslope_cutoff_density = -1.e20;
    p->difmag = 0.1;
    p->small_dens = -1.e200;
    p->small_temp = -1.e200;
    p->small_pres = -1.e200;
    p->small_ener = -1.e200;
    p->cg_tol = 1.0e-5;
    p->dual_energy_eta1 = 1.0e0;
    p->dual_energy_eta2 = 1.0e-4;
    p->cfl = 0.5;
    p->init_shrink = 0.01;
    p->change_max = 1.1;
    p->eos_gamma = 1.4;
    p->small_x = 1.e-30;
    p->T_guess = 1.e8;
    p->abar = 1.0;
    ora_finalize_params(p);
}

/* Castro_setup.cpp:222-236 (negative => 1e-100) and :259-288 (raise
 * small_pres/small_ener to the EOS values at (small_dens, small_temp)) */
void ora_finalize_params(ora_params *p)
{
    if (p->small_dens < 0.0) p->small_dens = 1.e-100;
    if (p->small_temp < 0.0) p->small_temp = 1.e-100;
    if (p->small_pres < 0.0) p->small_pres = 1.e-100;
    if (p->small_ener < 0.0) p->small_ener = 1.e-100;
    ora_eos_t s;
    s.rho = p->small_dens;
    s.T = p->small_temp;
    s.xn = 1.0;                 /* Castro_setup.cpp:279: xn = 1 / NumSpec */
    ora_eos_rt(p, &s);
    p->small_pres = amax(p->small_pres, s.p);
    p->small_ener = amax(p->small_ener, s.e);
}

ora_a4 ora_make_a4(double *p, const int lo[3], const int hi[3], int nc)
{
    ora_a4 a;
    a.p = p;
    for (int d = 0; d < 3; ++d) { a.lo[d] = lo[d]; a.hi[d] = hi[d]; }
    a.nc = nc;
    long nx = hi[0] - lo[0] + 1, ny = hi[1] - lo[1] + 1, nz = hi[2] - lo[2] + 1;
    a.sy = nx; a.sz = nx * ny; a.sn = nx * ny * nz;
    return a;
}

/* ------------------------------------------------------------------ */
/* gamma-law EOS (Microphysics EOS/gamma_law, restated per SURVEY D.3) */
/* ------------------------------------------------------------------ */
/* CODATA-2010 cgs constants as used by Microphysics' fundamental_constants
 * (unverifiable here: only the Temp field depends on them). */
#define ORA_K_B 1.3806488e-16
#define ORA_M_U 1.660538921e-24

/* composition(): abar = 1 / sum_k(xn_k / A_k), one species of mass number P->abar; mu = abar (eos_assume_neutral) */
static inline double eos_mu(const ora_params *P, const ora_eos_t *s)
{
    double sum = s->xn * (1.0 / P->abar);
    return 1.0 / sum;
}

static inline void eos_finish(const ora_params *P, ora_eos_t *s)
{
    s->gam1 = P->eos_gamma;
    s->cs = sqrt(P->eos_gamma * s->p / s->rho);
    s->dpde = (P->eos_gamma - 1.0) * s->rho;
    s->dpdr_e = (P->eos_gamma - 1.0) * s->e;
}

/* eos_input_re: rho, e given */
void ora_eos_re(const ora_params *P, ora_eos_t *s)
{
    s->p = (P->eos_gamma - 1.0) * s->rho * s->e;
    s->T = (P->eos_gamma - 1.0) * s->e * (eos_mu(P, s) * ORA_M_U) / ORA_K_B;
    eos_finish(P, s);
}

/* eos_input_rt: rho, T given */
void ora_eos_rt(const ora_params *P, ora_eos_t *s)
{
    s->e = ORA_K_B * s->T / ((P->eos_gamma - 1.0) * (eos_mu(P, s) * ORA_M_U));
    s->p = (P->eos_gamma - 1.0) * s->rho * s->e;
    eos_finish(P, s);
}

/* eos_input_rp: rho, p given */
void ora_eos_rp(const ora_params *P, ora_eos_t *s)
{
    s->e = s->p / ((P->eos_gamma - 1.0) * s->rho);
    s->T = (P->eos_gamma - 1.0) * s->e * (eos_mu(P, s) * ORA_M_U) / ORA_K_B;
    eos_finish(P, s);
}

/* ------------------------------------------------------------------ */
/* Castro::ctoprim  (Source/hydro/advection_util.cpp:26-200)           */
/* ------------------------------------------------------------------ */
int ora_ctoprim(const int lo[3], const int hi[3], ora_a4 uin, ora_a4 q, ora_a4 qaux, const ora_params *P)
{
    int bad = 0;
    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        /* :56-68 -- the reference aborts here on CPU; we flag instead */
        if (A4(uin,i,j,k,URHO) <= 0.0 || A4(uin,i,j,k,URHO) < P->small_dens) bad = 1;

        A4(q,i,j,k,QRHO) = A4(uin,i,j,k,URHO);
        double rhoinv = 1.0 / A4(q,i,j,k,QRHO);

        A4(q,i,j,k,QU) = A4(uin,i,j,k,UMX) * rhoinv;
        A4(q,i,j,k,QV) = A4(uin,i,j,k,UMY) * rhoinv;
        A4(q,i,j,k,QW) = A4(uin,i,j,k,UMZ) * rhoinv;

        /* :91-99 dual energy */
        double kineng = 0.5 * A4(q,i,j,k,QRHO) * (A4(q,i,j,k,QU)*A4(q,i,j,k,QU) +
                                                  A4(q,i,j,k,QV)*A4(q,i,j,k,QV) +
                                                  A4(q,i,j,k,QW)*A4(q,i,j,k,QW));

        if ((A4(uin,i,j,k,UEDEN) - kineng) > P->dual_energy_eta1 * A4(uin,i,j,k,UEDEN)) {
            A4(q,i,j,k,QREINT) = (A4(uin,i,j,k,UEDEN) - kineng) * rhoinv;
        } else {
            A4(q,i,j,k,QREINT) = A4(uin,i,j,k,UEINT) * rhoinv;
        }

        A4(q,i,j,k,QTEMP) = A4(uin,i,j,k,UTEMP);

        /* :124-128 passives */
        for (int ip = 0; ip < NPASSIVE; ++ip) {
            A4(q,i,j,k,qpassmap(ip)) = A4(uin,i,j,k,upassmap(ip)) * rhoinv;
        }

        /* :131-147 EOS call */
        ora_eos_t es;
        es.T = A4(q,i,j,k,QTEMP);
        es.rho = A4(q,i,j,k,QRHO);
        es.e = A4(q,i,j,k,QREINT);
        es.xn = A4(q,i,j,k,QFS);          /* advection_util.cpp:139 */
        ora_eos_re(P, &es);

        A4(q,i,j,k,QTEMP) = es.T;
        A4(q,i,j,k,QREINT) = es.e * A4(q,i,j,k,QRHO);
        A4(q,i,j,k,QPRES) = es.p;

        /* :194-195 */
        A4(qaux,i,j,k,QGAMC) = es.gam1;
        A4(qaux,i,j,k,QC) = es.cs;
    }
    return bad;
}

/* ------------------------------------------------------------------ */
/* Castro::uflatten  (Source/hydro/flatten.cpp:12-166)                 */
/* ------------------------------------------------------------------ */
static inline double flatten_1d(ora_a4 q, int i, int j, int k, int di, int dj, int dk, int pc, int uc)
{
    const double small_pres = 1.e-200;     /* :16 */
    const double shktst = 0.33;            /* :19 */
    const double zcut1 = 0.75;             /* :20 */
    const double zcut2 = 0.85;             /* :21 */
    const double dzcut = 1.0 / (zcut2 - zcut1);

#define QS(s, n) A4(q, i + (s)*di, j + (s)*dj, k + (s)*dk, n)
    double dp = QS(1, pc) - QS(-1, pc);
    int ishft = dp > 0.0 ? 1 : -1;

    double denom = amax(small_pres, fabs(QS(2, pc) - QS(-2, pc)));
    double zeta = fabs(dp) / denom;
    double z = amin(1.0, amax(0.0, dzcut * (zeta - zcut1)));

    double tst = 0.0;
    if (QS(-1, uc) - QS(1, uc) >= 0.0) tst = 1.0;

    double tmp = amin(QS(1, pc), QS(-1, pc));

    double chi = 0.0;
    if (fabs(dp) > shktst * tmp) chi = tst;

    dp = QS(1 - ishft, pc) - QS(-1 - ishft, pc);

    denom = amax(small_pres, fabs(QS(2 - ishft, pc) - QS(-2 - ishft, pc)));
    zeta = fabs(dp) / denom;
    double z2 = amin(1.0, amax(0.0, dzcut * (zeta - zcut1)));

    tst = 0.0;
    if (QS(-1 - ishft, uc) - QS(1 - ishft, uc) >= 0.0) tst = 1.0;

    tmp = amin(QS(1 - ishft, pc), QS(-1 - ishft, pc));

    double chi2 = 0.0;
    if (fabs(dp) > shktst * tmp) chi2 = tst;
#undef QS
    return 1.0 - amax(chi2 * z2, chi * z);
}

void ora_uflatten(const int lo[3], const int hi[3], ora_a4 q, ora_a4 flatn, int pres_comp)
{
    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        A4(flatn,i,j,k,0) = flatten_1d(q, i, j, k, 1, 0, 0, pres_comp, QU);                               /* :29-70  */
        A4(flatn,i,j,k,0) = amin(A4(flatn,i,j,k,0), flatten_1d(q, i, j, k, 0, 1, 0, pres_comp, QV));       /* :73-117 */
        A4(flatn,i,j,k,0) = amin(A4(flatn,i,j,k,0), flatten_1d(q, i, j, k, 0, 0, 1, pres_comp, QW));       /* :120-162 */
    }
}

/* ------------------------------------------------------------------ */
/* Castro::shock  (Source/hydro/advection_util.cpp:203-363), 3-D Cartesian */
/* ------------------------------------------------------------------ */
void ora_shock(const int lo[3], const int hi[3], ora_a4 q, ora_a4 shk, const ora_geom *G)
{
    const double small = 1.e-10;
    const double eps = 0.33e0;
    double dxinv = 1.0 / G->dx[0];
    double dyinv = 1.0 / G->dx[1];
    double dzinv = 1.0 / G->dx[2];

    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        double div_u = 0.0;
        div_u += 0.5 * (A4(q,i+1,j,k,QU) - A4(q,i-1,j,k,QU)) * dxinv;
        div_u += 0.5 * (A4(q,i,j+1,k,QV) - A4(q,i,j-1,k,QV)) * dyinv;
        div_u += 0.5 * (A4(q,i,j,k+1,QW) - A4(q,i,j,k-1,QW)) * dzinv;

        double px_pre, px_post, e_x;
        if (A4(q,i+1,j,k,QPRES) - A4(q,i-1,j,k,QPRES) < 0.0) {
            px_pre = A4(q,i+1,j,k,QPRES); px_post = A4(q,i-1,j,k,QPRES);
        } else {
            px_pre = A4(q,i-1,j,k,QPRES); px_post = A4(q,i+1,j,k,QPRES);
        }
        /* std::pow(x, 2) == x*x (correctly rounded in glibc) */
        e_x = (A4(q,i+1,j,k,QU) - A4(q,i-1,j,k,QU)) * (A4(q,i+1,j,k,QU) - A4(q,i-1,j,k,QU));

        double py_pre, py_post, e_y;
        if (A4(q,i,j+1,k,QPRES) - A4(q,i,j-1,k,QPRES) < 0.0) {
            py_pre = A4(q,i,j+1,k,QPRES); py_post = A4(q,i,j-1,k,QPRES);
        } else {
            py_pre = A4(q,i,j-1,k,QPRES); py_post = A4(q,i,j+1,k,QPRES);
        }
        e_y = (A4(q,i,j+1,k,QV) - A4(q,i,j-1,k,QV)) * (A4(q,i,j+1,k,QV) - A4(q,i,j-1,k,QV));

        double pz_pre, pz_post, e_z;
        if (A4(q,i,j,k+1,QPRES) - A4(q,i,j,k-1,QPRES) < 0.0) {
            pz_pre = A4(q,i,j,k+1,QPRES); pz_post = A4(q,i,j,k-1,QPRES);
        } else {
            pz_pre = A4(q,i,j,k-1,QPRES); pz_post = A4(q,i,j,k+1,QPRES);
        }
        e_z = (A4(q,i,j,k+1,QW) - A4(q,i,j,k-1,QW)) * (A4(q,i,j,k+1,QW) - A4(q,i,j,k-1,QW));

        double denom = 1.0 / (e_x + e_y + e_z + small);
        e_x = e_x * denom;
        e_y = e_y * denom;
        e_z = e_z * denom;

        double p_pre = e_x * px_pre + e_y * py_pre + e_z * pz_pre;
        double p_post = e_x * px_post + e_y * py_post + e_z * pz_post;

        double pjump = (p_pre == 0) ? 0.0 : eps - (p_post - p_pre) / p_pre;

        if (pjump < 0.0 && div_u < 0.0) {
            A4(shk,i,j,k,0) = 1.0;
        } else {
            A4(shk,i,j,k,0) = 0.0;
        }
    }
}

/* ------------------------------------------------------------------ */
/* Castro::src_to_prim  (Source/hydro/Castro_ctu.cpp:468-545)          */
/* CTU with source_term_predictor = 0: srcU = old_src                   */
/* ------------------------------------------------------------------ */
static ora_a4 g_source_corrector;      /* p == NULL: none */
void ora_set_source_corrector(const ora_a4 *corr)
{
    if (corr) g_source_corrector = *corr; else g_source_corrector.p = NULL;
}

void ora_src_to_prim(const int lo[3], const int hi[3], ora_a4 q, ora_a4 old_src, ora_a4 srcQ, const ora_params *P, double dt)
{
    const ora_a4 src_corr = g_source_corrector;
    const int predict = P->source_term_predictor == 1 && src_corr.p != NULL;
    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        for (int n = 0; n < NQSRC; ++n) A4(srcQ,i,j,k,n) = 0.0;

        double srcU[NSRC];
        for (int n = 0; n < NSRC; ++n) {
            srcU[n] = 0.0;
            /* Castro_ctu.cpp:493-497: the lagged predictor time-centres the momentum sources */
            if (predict && (n == UMX || n == UMY || n == UMZ)) srcU[n] += 0.5 * dt * A4(src_corr,i,j,k,n);
            if (old_src.p) srcU[n] += A4(old_src,i,j,k,n);
        }

        double rhoinv = 1.0 / A4(q,i,j,k,QRHO);

        ora_eos_t es;
        es.T = A4(q,i,j,k,QTEMP);
        es.rho = A4(q,i,j,k,QRHO);
        es.e = A4(q,i,j,k,QREINT) * rhoinv;
        es.xn = A4(q,i,j,k,QFS);          /* Castro_ctu.cpp:513 */
        ora_eos_re(P, &es);

        A4(srcQ,i,j,k,QRHO) = srcU[URHO];
        A4(srcQ,i,j,k,QU) = (srcU[UMX] - A4(q,i,j,k,QU) * A4(srcQ,i,j,k,QRHO)) * rhoinv;
        A4(srcQ,i,j,k,QV) = (srcU[UMY] - A4(q,i,j,k,QV) * A4(srcQ,i,j,k,QRHO)) * rhoinv;
        A4(srcQ,i,j,k,QW) = (srcU[UMZ] - A4(q,i,j,k,QW) * A4(srcQ,i,j,k,QRHO)) * rhoinv;
        A4(srcQ,i,j,k,QREINT) = srcU[UEINT];
        A4(srcQ,i,j,k,QPRES) = es.dpde *
            (A4(srcQ,i,j,k,QREINT) - A4(q,i,j,k,QREINT) * A4(srcQ,i,j,k,QRHO) * rhoinv) *
            rhoinv + es.dpdr_e * A4(srcQ,i,j,k,QRHO);
    }
}

/* ------------------------------------------------------------------ */
/* Castro::divu  (Source/hydro/advection_util.cpp:366-479, 3-D :458-475) */
/* ------------------------------------------------------------------ */
void ora_divu(const int lo[3], const int hi[3], ora_a4 q, ora_a4 div, const ora_geom *G)
{
    double dxinv = 1.0 / G->dx[0];
    double dyinv = 1.0 / G->dx[1];
    double dzinv = 1.0 / G->dx[2];

    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        double ux = 0.25 * (A4(q,i,j,k,QU) - A4(q,i-1,j,k,QU) +
                            A4(q,i,j,k-1,QU) - A4(q,i-1,j,k-1,QU) +
                            A4(q,i,j-1,k,QU) - A4(q,i-1,j-1,k,QU) +
                            A4(q,i,j-1,k-1,QU) - A4(q,i-1,j-1,k-1,QU)) * dxinv;

        double vy = 0.25 * (A4(q,i,j,k,QV) - A4(q,i,j-1,k,QV) +
                            A4(q,i,j,k-1,QV) - A4(q,i,j-1,k-1,QV) +
                            A4(q,i-1,j,k,QV) - A4(q,i-1,j-1,k,QV) +
                            A4(q,i-1,j,k-1,QV) - A4(q,i-1,j-1,k-1,QV)) * dyinv;

        double wz = 0.25 * (A4(q,i,j,k,QW) - A4(q,i,j,k-1,QW) +
                            A4(q,i,j-1,k,QW) - A4(q,i,j-1,k-1,QW) +
                            A4(q,i-1,j,k,QW) - A4(q,i-1,j,k-1,QW) +
                            A4(q,i-1,j-1,k,QW) - A4(q,i-1,j-1,k-1,QW)) * dzinv;

        A4(div,i,j,k,0) = ux + vy + wz;
    }
}

/* ------------------------------------------------------------------ */
/* Castro::apply_av  (Source/hydro/advection_util.cpp:482-528)         */
/* ------------------------------------------------------------------ */
void ora_apply_av(const int lo[3], const int hi[3], int idir, ora_a4 div, ora_a4 uin, ora_a4 flux,
                  const ora_geom *G, const ora_params *P)
{
    double diff_coeff = P->difmag;
    for (int n = 0; n < NUM_STATE; ++n) {
        if (n == UTEMP) continue;
        for (int k = lo[2]; k <= hi[2]; ++k)
        for (int j = lo[1]; j <= hi[1]; ++j)
        for (int i = lo[0]; i <= hi[0]; ++i) {
            double div1;
            if (idir == 0) {
                div1 = 0.25 * (A4(div,i,j,k,0) + A4(div,i,j+1,k,0) +
                               A4(div,i,j,k+1,0) + A4(div,i,j+1,k+1,0));
                div1 = diff_coeff * amin(0.0, div1);
                div1 = div1 * (A4(uin,i,j,k,n) - A4(uin,i-1,j,k,n));
            } else if (idir == 1) {
                div1 = 0.25 * (A4(div,i,j,k,0) + A4(div,i+1,j,k,0) +
                               A4(div,i,j,k+1,0) + A4(div,i+1,j,k+1,0));
                div1 = diff_coeff * amin(0.0, div1);
                div1 = div1 * (A4(uin,i,j,k,n) - A4(uin,i,j-1,k,n));
            } else {
                div1 = 0.25 * (A4(div,i,j,k,0) + A4(div,i+1,j,k,0) +
                               A4(div,i,j+1,k,0) + A4(div,i+1,j+1,k,0));
                div1 = diff_coeff * amin(0.0, div1);
                div1 = div1 * (A4(uin,i,j,k,n) - A4(uin,i,j,k-1,n));
            }
            A4(flux,i,j,k,n) += G->dx[idir] * div1;
        }
    }
}

/* ------------------------------------------------------------------ */
/* dflux (Source/hydro/advection_util.H:10-79): the flux of a cell-centred state, 3-D Cartesian
 * (mom_flux_has_p is always true there), no hybrid momentum                                    */
/* ------------------------------------------------------------------ */
static void dflux(const double u[NUM_STATE], const double q[NQ], int dir, double flux[NUM_STATE])
{
    for (int n = 0; n < NUM_STATE; ++n) flux[n] = 0.0;
    double v_adv = q[QU + dir];
    flux[URHO] = u[URHO] * v_adv;
    flux[UMX] = u[UMX] * v_adv;
    flux[UMY] = u[UMY] * v_adv;
    flux[UMZ] = u[UMZ] * v_adv;
    flux[UEDEN] = (u[UEDEN] + q[QPRES]) * v_adv;
    flux[UEINT] = u[UEINT] * v_adv;
    flux[UMX + dir] = flux[UMX + dir] + q[QPRES];
    for (int ip = 0; ip < NPASSIVE; ++ip) {
        int n = upassmap(ip);
        flux[n] = u[n] * v_adv;
    }
}

static void limiter_states(ora_a4 u, ora_a4 q, int idir, int i, int j, int k,
                           double uL[NUM_STATE], double qL[NQ], double uR[NUM_STATE], double qR[NQ])
{
    const int il = i - (idir == 0), jl = j - (idir == 1), kl = k - (idir == 2);
    for (int n = 0; n < NUM_STATE; ++n) { uR[n] = A4(u,i,j,k,n); uL[n] = A4(u,il,jl,kl,n); }
    for (int n = 0; n < NQ; ++n) { qR[n] = A4(q,i,j,k,n); qL[n] = A4(q,il,jl,kl,n); }
}

/* ------------------------------------------------------------------ */
/* Castro::limit_hydro_fluxes_on_small_dens (advection_util.cpp:657-903), Hu, Adams & Shu (2013)  */
/* ------------------------------------------------------------------ */
void ora_limit_hydro_fluxes_on_small_dens(const int lo[3], const int hi[3], int idir, ora_a4 u, ora_a4 q, ora_a4 flux,
                                          const ora_geom *G, const ora_params *P, double dt)
{
    const double density_floor_tolerance = 1.1;
    double density_floor = P->small_dens * density_floor_tolerance;
    density_floor *= 3 * 2;
    const double dtdx = dt / G->dx[idir];
    const double lcfl = P->cfl;
    const double alpha = 1.0 / 3;
    const double vol = G->dx[0] * G->dx[1] * G->dx[2];
    const double area = (idir == 0) ? G->dx[1] * G->dx[2] : (idir == 1) ? G->dx[0] * G->dx[2] : G->dx[0] * G->dx[1];

    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        double uL[NUM_STATE], uR[NUM_STATE], qL[NQ], qR[NQ];
        limiter_states(u, q, idir, i, j, k, uL, qL, uR, qR);
        const double volR = vol, volL = vol;

        if (uR[URHO] < density_floor || uL[URHO] < density_floor) {
            for (int n = 0; n < NUM_STATE; ++n) A4(flux,i,j,k,n) = 0.0;
            continue;
        }

        double fluxL[NUM_STATE], fluxR[NUM_STATE], fluxLF[NUM_STATE];
        dflux(uL, qL, idir, fluxL);
        dflux(uR, qR, idir, fluxR);
        for (int n = 0; n < NUM_STATE; ++n)
            fluxLF[n] = 0.5 * (fluxL[n] + fluxR[n] + (lcfl / dtdx / alpha) * (uL[n] - uR[n]));

        double flux_coefR = 2.0 * (dt / alpha) * area / volR;
        double flux_coefL = 2.0 * (dt / alpha) * area / volL;

        double drhoL = flux_coefL * A4(flux,i,j,k,URHO);
        double rhoL = uL[URHO] - drhoL;
        double drhoR = flux_coefR * A4(flux,i,j,k,URHO);
        double rhoR = uR[URHO] + drhoR;

        double theta = 1.0;
        if (rhoL < density_floor) {
            double drhoLF = flux_coefL * fluxLF[URHO];
            double rhoLF = uL[URHO] - drhoLF;
            theta = amin(theta, (density_floor - rhoLF) / (rhoL - rhoLF));
        } else if (rhoR < density_floor) {
            double drhoLF = flux_coefR * fluxLF[URHO];
            double rhoLF = uR[URHO] + drhoLF;
            theta = amin(theta, (density_floor - rhoLF) / (rhoR - rhoLF));
        }
        theta = amin(1.0, amax(theta, 0.0));

        for (int n = 0; n < NUM_STATE; ++n)
            A4(flux,i,j,k,n) = (1.0 - theta) * fluxLF[n] + theta * A4(flux,i,j,k,n);
        A4(flux,i,j,k,UTEMP) = 0.0;

        drhoR = flux_coefR * A4(flux,i,j,k,URHO);
        drhoL = flux_coefL * A4(flux,i,j,k,URHO);
        if (uR[URHO] + drhoR < density_floor) {
            for (int n = 0; n < NUM_STATE; ++n)
                A4(flux,i,j,k,n) = A4(flux,i,j,k,n) * fabs((density_floor - uR[URHO]) / drhoR);
        } else if (uL[URHO] - drhoL < density_floor) {
            for (int n = 0; n < NUM_STATE; ++n)
                A4(flux,i,j,k,n) = A4(flux,i,j,k,n) * fabs((density_floor - uL[URHO]) / drhoL);
        }
    }
}

/* ------------------------------------------------------------------ */
/* Castro::limit_hydro_fluxes_on_large_vel (advection_util.cpp:907-1075)                          */
/* ------------------------------------------------------------------ */
void ora_limit_hydro_fluxes_on_large_vel(const int lo[3], const int hi[3], int idir, ora_a4 u, ora_a4 q, ora_a4 flux,
                                         const ora_geom *G, const ora_params *P, double dt)
{
    if (P->speed_limit <= 0.0) return;
    const double dtdx = dt / G->dx[idir];
    const double lcfl = P->cfl;
    const double alpha = 1.0 / 3;
    const double vol = G->dx[0] * G->dx[1] * G->dx[2];
    const double area = (idir == 0) ? G->dx[1] * G->dx[2] : (idir == 1) ? G->dx[0] * G->dx[2] : G->dx[0] * G->dx[1];
    const double lspeed_limit = P->speed_limit / (2 * 3);

    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        double uL[NUM_STATE], uR[NUM_STATE], qL[NQ], qR[NQ];
        limiter_states(u, q, idir, i, j, k, uL, qL, uR, qR);
        const double volR = vol, volL = vol;

        double fluxL[NUM_STATE], fluxR[NUM_STATE], fluxLF[NUM_STATE];
        dflux(uL, qL, idir, fluxL);
        dflux(uR, qR, idir, fluxR);
        for (int n = 0; n < NUM_STATE; ++n)
            fluxLF[n] = 0.5 * (fluxL[n] + fluxR[n] + (lcfl / dtdx / alpha) * (uL[n] - uR[n]));

        double flux_coefR = 2.0 * (dt / alpha) * area / volR;
        double flux_coefL = 2.0 * (dt / alpha) * area / volL;

        double theta = 1.0;
        for (int n = 0; n < 3; ++n) {
            int UMOM = UMX + n;
            double drhouL = flux_coefL * A4(flux,i,j,k,UMOM);
            double rhouL = fabs(uL[UMOM] - drhouL);
            double drhoL = flux_coefL * A4(flux,i,j,k,URHO);
            double rhoL = uL[URHO] - drhoL;
            double drhouR = flux_coefR * A4(flux,i,j,k,UMOM);
            double rhouR = fabs(uR[UMOM] + drhouR);
            double drhoR = flux_coefR * A4(flux,i,j,k,URHO);
            double rhoR = uR[URHO] + drhoR;

            if (fabs(rhouL) > rhoL * lspeed_limit) {
                double drhouLF = flux_coefL * fluxLF[UMOM];
                double rhouLF = fabs(uL[UMOM] - drhouLF);
                theta = amin(theta, fabs(rhoL * lspeed_limit - rhouLF) / fabs(rhouL - rhouLF));
            } else if (fabs(rhouR) > rhoR * lspeed_limit) {
                double drhouLF = flux_coefR * fluxLF[UMOM];
                double rhouLF = fabs(uR[UMOM] + drhouLF);
                theta = amin(theta, fabs(rhoR * lspeed_limit - rhouLF) / fabs(rhouR - rhouLF));
            }
        }
        theta = amin(1.0, amax(theta, 0.0));

        for (int n = 0; n < NUM_STATE; ++n)
            A4(flux,i,j,k,n) = (1.0 - theta) * fluxLF[n] + theta * A4(flux,i,j,k,n);
        A4(flux,i,j,k,UTEMP) = 0.0;
    }
}

/* ------------------------------------------------------------------ */
/* Castro::normalize_species_fluxes (advection_util.cpp:577-613)       */
/* ------------------------------------------------------------------ */
void ora_normalize_species_fluxes(const int lo[3], const int hi[3], ora_a4 flux)
{
    const double eps = 2.220446049250313e-16; /* numeric_limits<double>::epsilon() */
    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        double sum = 0.0;
        for (int n = UFS; n < UFS + NUMSPEC; ++n) sum += A4(flux,i,j,k,n);

        double fac = 1.0;
        if (fabs(sum) > eps * fabs(A4(flux,i,j,k,URHO))) {
            fac = A4(flux,i,j,k,URHO) / sum;
        }
        for (int n = UFS; n < UFS + NUMSPEC; ++n) A4(flux,i,j,k,n) = A4(flux,i,j,k,n) * fac;
    }
}

/* ------------------------------------------------------------------ */
/* Castro::scale_flux (advection_util.cpp:616-641), 3-D                */
/* ------------------------------------------------------------------ */
void ora_scale_flux(const int lo[3], const int hi[3], ora_a4 flux, double area, double dt)
{
    for (int n = 0; n < NUM_STATE; ++n)
    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        A4(flux,i,j,k,n) = dt * A4(flux,i,j,k,n) * area;
    }
}

/* ------------------------------------------------------------------ */
/* Castro::consup_hydro (Source/hydro/Castro_ctu.cpp:11-86), 3-D Cartesian;
 * geometry_util::area/volume from Source/driver/Castro_util.H:147-295   */
/* ------------------------------------------------------------------ */
void ora_consup_hydro(const int lo[3], const int hi[3], ora_a4 U_new, ora_a4 flux0, ora_a4 qx,
                      ora_a4 flux1, ora_a4 qy, ora_a4 flux2, ora_a4 qz, double dt, const ora_geom *G)
{
    const double *dx = G->dx;
    const double area0 = dx[1] * dx[2];
    const double area1 = dx[0] * dx[2];
    const double area2 = dx[0] * dx[1];
    const double vol = dx[0] * dx[1] * dx[2];

    for (int n = 0; n < NUM_STATE; ++n)
    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        double volinv = 1.0 / vol;

        A4(U_new,i,j,k,n) = A4(U_new,i,j,k,n) + dt *
            ( A4(flux0,i,  j,k,n) * area0
            - A4(flux0,i+1,j,k,n) * area0
            + A4(flux1,i,j,  k,n) * area1
            - A4(flux1,i,j+1,k,n) * area1
            + A4(flux2,i,j,k,  n) * area2
            - A4(flux2,i,j,k+1,n) * area2
            ) * volinv;

        if (n == UEINT) {
            double pdu = (A4(qx,i+1,j,k,GDPRES) + A4(qx,i,j,k,GDPRES)) *
                (A4(qx,i+1,j,k,GDU) * area0 - A4(qx,i,j,k,GDU) * area0);

            pdu += (A4(qy,i,j+1,k,GDPRES) + A4(qy,i,j,k,GDPRES)) *
                (A4(qy,i,j+1,k,GDV) * area1 - A4(qy,i,j,k,GDV) * area1);

            pdu += (A4(qz,i,j,k+1,GDPRES) + A4(qz,i,j,k,GDPRES)) *
                (A4(qz,i,j,k+1,GDW) * area2 - A4(qz,i,j,k,GDW) * area2);

            pdu = 0.5 * pdu * volinv;

            A4(U_new,i,j,k,n) = A4(U_new,i,j,k,n) - dt * pdu;
        }
        /* UMX grad-p term only for non-Cartesian (mom_flux_has_p false): n/a in 3-D */
    }
}
