/*
 * ora_riemann.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see castro_oracle.h).
 * Restates Source/hydro/riemann.cpp, riemann.H and riemann_solvers.H (3-D, no
 * radiation): cmpflx_plus_godunov, load_input_states, riemannus (CGF, default),
 * riemanncg (CG) + wsqge + pstar_bisection, HLLC, HLL, compute_flux_q.
 */
#include <stdio.h>
#include <string.h>
#include "ora_internal.h"

/* riemann.H:66-246 */
static void load_input_states(int i, int j, int k, int idir, ora_a4 qleft, ora_a4 qright, ora_a4 qaux,
                              RiemannState *ql, RiemannState *qr, RiemannAux *raux, const ora_params *P)
{
    const double small = 1.e-8;

    ql->rho = amax(A4(qleft,i,j,k,QRHO), P->small_dens);
    qr->rho = amax(A4(qright,i,j,k,QRHO), P->small_dens);

    int im = i, jm = j, km = k;
    if (idir == 0) {
        ql->un = A4(qleft,i,j,k,QU); ql->ut = A4(qleft,i,j,k,QV); ql->utt = A4(qleft,i,j,k,QW);
        qr->un = A4(qright,i,j,k,QU); qr->ut = A4(qright,i,j,k,QV); qr->utt = A4(qright,i,j,k,QW);
        im = i - 1;
    } else if (idir == 1) {
        ql->un = A4(qleft,i,j,k,QV); ql->ut = A4(qleft,i,j,k,QU); ql->utt = A4(qleft,i,j,k,QW);
        qr->un = A4(qright,i,j,k,QV); qr->ut = A4(qright,i,j,k,QU); qr->utt = A4(qright,i,j,k,QW);
        jm = j - 1;
    } else {
        ql->un = A4(qleft,i,j,k,QW); ql->ut = A4(qleft,i,j,k,QU); ql->utt = A4(qleft,i,j,k,QV);
        qr->un = A4(qright,i,j,k,QW); qr->ut = A4(qright,i,j,k,QU); qr->utt = A4(qright,i,j,k,QV);
        km = k - 1;
    }

    raux->csmall = amax(small, small * amax(A4(qaux,i,j,k,QC), A4(qaux,im,jm,km,QC)));
    raux->cavg = 0.5 * (A4(qaux,i,j,k,QC) + A4(qaux,im,jm,km,QC));

    ql->gamc = A4(qaux,im,jm,km,QGAMC);
    qr->gamc = A4(qaux,i,j,k,QGAMC);

    ql->p = A4(qleft,i,j,k,QPRES);
    qr->p = A4(qright,i,j,k,QPRES);

    ql->rhoe = A4(qleft,i,j,k,QREINT);
    qr->rhoe = A4(qright,i,j,k,QREINT);

    /* :198-244 thermodynamic clean-up through the EOS at small_temp */
    if (ql->rhoe <= 0.0 || ql->p < P->small_pres) {
        ora_eos_t es;
        es.T = P->small_temp;
        es.rho = ql->rho;
        es.xn = A4(qleft,i,j,k,QFS);       /* riemann.H:207 */
        ora_eos_rt(P, &es);
        ql->rhoe = ql->rho * es.e;
        ql->p = es.p;
        ql->gamc = es.gam1;
    }

    if (qr->rhoe <= 0.0 || qr->p < P->small_pres) {
        ora_eos_t es;
        es.T = P->small_temp;
        es.rho = qr->rho;
        es.xn = A4(qright,i,j,k,QFS);      /* riemann.H:231 */
        ora_eos_rt(P, &es);
        qr->rhoe = qr->rho * es.e;
        qr->p = es.p;
        qr->gamc = es.gam1;
    }
}

/* riemann_solvers.H:597-820 -- Colella, Glaz & Ferguson two-shock solver */
void ora_riemannus(const RiemannState *ql, const RiemannState *qr, const RiemannAux *raux,
                   RiemannState *qint, const ora_params *P)
{
    double wsmall = P->small_dens * raux->csmall;

    double wl = amax(wsmall, sqrt(fabs(ql->gamc * ql->p * ql->rho)));
    double wr = amax(wsmall, sqrt(fabs(qr->gamc * qr->p * qr->rho)));

    double wwinv = 1.0 / (wl + wr);
    double pstar = ((wr * ql->p + wl * qr->p) + wl * wr * (ql->un - qr->un)) * wwinv;
    double ustar = ((wl * ql->un + wr * qr->un) + (ql->p - qr->p)) * wwinv;

    pstar = amax(pstar, P->small_pres);

    if (fabs(ustar) < RC_SMALLU * 0.5 * (fabs(ql->un) + fabs(qr->un))) {
        ustar = 0.0;
    }

    double sgnm = copysign(1.0, ustar);
    if (ustar == 0.0) sgnm = 0.0;

    double fp = 0.5 * (1.0 + sgnm);
    double fm = 0.5 * (1.0 - sgnm);

    double ro = fp * ql->rho + fm * qr->rho;
    double uo = fp * ql->un + fm * qr->un;
    double po = fp * ql->p + fm * qr->p;
    double reo = fp * ql->rhoe + fm * qr->rhoe;
    double gamco = fp * ql->gamc + fm * qr->gamc;

    ro = amax(P->small_dens, ro);

    double roinv = 1.0 / ro;

    double co = sqrt(fabs(gamco * po * roinv));
    co = amax(raux->csmall, co);
    double co2inv = 1.0 / (co * co);

    qint->ut = fp * ql->ut + fm * qr->ut;
    qint->utt = fp * ql->utt + fm * qr->utt;

    double drho = (pstar - po) * co2inv;
    double rstar = ro + drho;
    rstar = amax(P->small_dens, rstar);

    double entho = (reo + po) * roinv * co2inv;
    double estar = reo + (pstar - po) * entho;

    double cstar = sqrt(fabs(gamco * pstar / rstar));
    cstar = amax(cstar, raux->csmall);

    double spout = co - sgnm * uo;
    double spin = cstar - sgnm * ustar;

    double ushock = 0.5 * (spin + spout);

    if (pstar - po > 0.0) {
        spin = ushock;
        spout = ushock;
    }

    double scr = spout - spin;
    if (spout - spin == 0.0) {
        scr = RC_SMALL * raux->cavg;
    }

    double frac = (1.0 + (spout + spin) / scr) * 0.5;
    frac = amax(0.0, amin(1.0, frac));

    qint->rho = frac * rstar + (1.0 - frac) * ro;
    qint->un = frac * ustar + (1.0 - frac) * uo;

    qint->p = frac * pstar + (1.0 - frac) * po;
    double regdnv = frac * estar + (1.0 - frac) * reo;

    if (spout < 0.0) {
        qint->rho = ro;
        qint->un = uo;
        qint->p = po;
        regdnv = reo;
    }

    if (spin >= 0.0) {
        qint->rho = rstar;
        qint->un = ustar;
        qint->p = pstar;
        regdnv = estar;
    }

    qint->p = amax(qint->p, P->small_pres);
    qint->rhoe = regdnv;

    qint->un = qint->un * raux->bnd_fac;
    qint->gamc = 0.0; /* not part of the solution; keep defined */
}

/* riemann.H:248-282 */
static inline void wsqge(double p, double v, double gam, double gdot, double *gstar,
                         double gmin, double gmax, double csq, double pstar, double *wsq)
{
    *gstar = (pstar - p) * gdot / (pstar + p) + gam;
    *gstar = amax(gmin, amin(gmax, *gstar));

    double alpha = pstar - (*gstar - 1.0) * p / (gam - 1.0);
    if (alpha == 0.0) {
        alpha = RC_SMLP1 * (pstar + p);
    }

    double beta = pstar + 0.5 * (*gstar - 1.0) * (pstar + p);

    *wsq = (pstar - p) * beta / (v * alpha);

    if (fabs(pstar - p) < RC_SMLP1 * (pstar + p)) {
        *wsq = csq;
    }
    *wsq = amax(*wsq, (0.5 * (gam - 1.0) / gam) * csq);
}

/* riemann.H:285-376 */
static void pstar_bisection(double *pstar_lo, double *pstar_hi,
                            double ul, double pl, double taul, double gamel, double clsql,
                            double ur, double pr, double taur, double gamer, double clsqr,
                            double gdot, double gmin, double gmax,
                            int lcg_maxiter, double lcg_tol,
                            double *pstar, double *gamstar, int *converged)
{
    double wlsq = 0.0;
    wsqge(pl, taul, gamel, gdot, gamstar, gmin, gmax, clsql, *pstar_lo, &wlsq);
    double wrsq = 0.0;
    wsqge(pr, taur, gamer, gdot, gamstar, gmin, gmax, clsqr, *pstar_lo, &wrsq);

    double wl = 1.0 / sqrt(wlsq);
    double wr = 1.0 / sqrt(wrsq);

    double ustar_l = ul - (*pstar_lo - *pstar) * wl;
    double ustar_r = ur + (*pstar_lo - *pstar) * wr;

    double f_lo = ustar_l - ustar_r;

    wsqge(pl, taul, gamel, gdot, gamstar, gmin, gmax, clsql, *pstar_hi, &wlsq);
    wsqge(pr, taur, gamer, gdot, gamstar, gmin, gmax, clsqr, *pstar_hi, &wrsq);

    wl = 1.0 / sqrt(wlsq);
    wr = 1.0 / sqrt(wrsq);

    ustar_l = ul - (*pstar_hi - *pstar) * wl;
    ustar_r = ur + (*pstar_hi - *pstar) * wr;

    *converged = 0;
    double pstar_c = 0.0;

    for (int iter = 0; iter < PSTAR_BISECT_FACTOR * lcg_maxiter; iter++) {
        pstar_c = 0.5 * (*pstar_lo + *pstar_hi);

        wsqge(pl, taul, gamel, gdot, gamstar, gmin, gmax, clsql, pstar_c, &wlsq);
        wsqge(pr, taur, gamer, gdot, gamstar, gmin, gmax, clsqr, pstar_c, &wrsq);

        wl = 1.0 / sqrt(wlsq);
        wr = 1.0 / sqrt(wrsq);

        ustar_l = ul - (pstar_c - pl) * wl;
        ustar_r = ur - (pstar_c - pr) * wr;

        double f_c = ustar_l - ustar_r;

        if (0.5 * fabs(*pstar_lo - *pstar_hi) < lcg_tol * pstar_c) {
            *converged = 1;
            break;
        }

        if (f_lo * f_c < 0.0) {
            *pstar_hi = pstar_c;
        } else {
            *pstar_lo = pstar_c;
            f_lo = f_c;
        }
    }

    *pstar = pstar_c;
}

/* riemann_solvers.H:225-581 -- Colella & Glaz (1985), CPU path (history kept) */
/* faces on which the reference would have called amrex::Error since ora_cg_aborts_reset() (cg_blend = 0, no convergence) */
static long ora_cg_aborts = 0;
long ora_cg_aborts_count(void) { return ora_cg_aborts; }
void ora_cg_aborts_reset(void) { ora_cg_aborts = 0; }

void ora_riemanncg(const RiemannState *ql, const RiemannState *qr, const RiemannAux *raux,
                   RiemannState *qint, const ora_params *P)
{
    const double weakwv = 1.e-3;
    double pstar_hist[HISTORY_SIZE];

    double taul = 1.0 / ql->rho;
    double taur = 1.0 / qr->rho;

    double clsql = ql->gamc * ql->p * ql->rho;
    double clsqr = qr->gamc * qr->p * qr->rho;

    double gamel = ql->p / ql->rhoe + 1.0;
    double gamer = qr->p / qr->rhoe + 1.0;

    double gmin = amin(amin(gamel, gamer), 1.0);
    double gmax = amax(amax(gamel, gamer), 2.0);

    double game_bar = 0.5 * (gamel + gamer);
    double gamc_bar = 0.5 * (ql->gamc + qr->gamc);

    double gdot = 2.0 * (1.0 - game_bar / gamc_bar) * (game_bar - 1.0);

    double wsmall = P->small_dens * raux->csmall;
    double wl = amax(wsmall, sqrt(fabs(clsql)));
    double wr = amax(wsmall, sqrt(fabs(clsqr)));

    double pstar = ql->p + ((qr->p - ql->p) - wr * (qr->un - ql->un)) * wl / (wl + wr);
    pstar = amax(pstar, P->small_pres);

    double gamstar = 0.0;

    double wlsq = 0.0;
    wsqge(ql->p, taul, gamel, gdot, &gamstar, gmin, gmax, clsql, pstar, &wlsq);

    double wrsq = 0.0;
    wsqge(qr->p, taur, gamer, gdot, &gamstar, gmin, gmax, clsqr, pstar, &wrsq);

    double pstar_old = pstar;

    wl = sqrt(wlsq);
    wr = sqrt(wrsq);

    double ustar_l = ql->un - (pstar - ql->p) / wl;
    double ustar_r = qr->un + (pstar - qr->p) / wr;

    pstar = ql->p + ((qr->p - ql->p) - wr * (qr->un - ql->un)) * wl / (wl + wr);
    pstar = amax(pstar, P->small_pres);

    int converged = 0;
    int iter = 0;
    while ((iter < P->cg_maxiter && !converged) || iter < 2) {
        wsqge(ql->p, taul, gamel, gdot, &gamstar, gmin, gmax, clsql, pstar, &wlsq);
        wsqge(qr->p, taur, gamer, gdot, &gamstar, gmin, gmax, clsqr, pstar, &wrsq);

        /* NOTE: these are really the inverses of the wave speeds! */
        wl = 1.0 / sqrt(wlsq);
        wr = 1.0 / sqrt(wrsq);

        double ustar_r_old = ustar_r;
        double ustar_l_old = ustar_l;

        ustar_r = qr->un - (qr->p - pstar) * wr;
        ustar_l = ql->un + (ql->p - pstar) * wl;

        double dpditer = fabs(pstar_old - pstar);

        double zp = fabs(ustar_l - ustar_l_old);
        if (zp - weakwv * raux->cavg <= 0.0) {
            zp = dpditer * wl;
        }

        double zm = fabs(ustar_r - ustar_r_old);
        if (zm - weakwv * raux->cavg <= 0.0) {
            zm = dpditer * wr;
        }

        double denom = dpditer / amax(zp + zm, RC_SMALL * raux->cavg);
        pstar_old = pstar;
        pstar = pstar - denom * (ustar_r - ustar_l);
        pstar = amax(pstar, P->small_pres);

        double err = fabs(pstar - pstar_old);
        if (err < P->cg_tol * pstar) {
            converged = 1;
        }

        if (iter < HISTORY_SIZE) pstar_hist[iter] = pstar;
        iter++;
    }

    if (!converged) {
        if (P->cg_blend == 0) {
            /* riemann_solvers.H:392-394: the reference aborts here (amrex::Error) -- the input has no defined result */
            ora_cg_aborts++;
            if (ora_cg_aborts == 1) fprintf(stderr, "oracle: non-convergence in the Riemann solver (cg_blend=0): the reference aborts\n");
        } else if (P->cg_blend == 1) {
            pstar = ql->p + ((qr->p - ql->p) - wr * (qr->un - ql->un)) * wl / (wl + wr);
        } else if (P->cg_blend == 2) {
            double pstarl = 1.e200;
            double pstaru = -1.e200;
            for (int n = P->cg_maxiter - 6; n < P->cg_maxiter; n++) {
                pstarl = amin(pstarl, pstar_hist[n]);
                pstaru = amax(pstaru, pstar_hist[n]);
            }
            pstarl = amax(pstarl, P->small_pres);
            pstaru = amax(pstaru, P->small_pres);

            pstar_bisection(&pstarl, &pstaru,
                            ql->un, ql->p, taul, gamel, clsql,
                            qr->un, qr->p, taur, gamer, clsqr,
                            gdot, gmin, gmax, P->cg_maxiter, P->cg_tol,
                            &pstar, &gamstar, &converged);
            if (!converged) {
                fprintf(stderr, "oracle: non-convergence in the Riemann solver (bisection)\n");
            }
        }
    }

    ustar_r = qr->un - (qr->p - pstar) * wr; /* careful -- here wl, wr are 1/W */
    ustar_l = ql->un + (ql->p - pstar) * wl;

    double ustar = 0.5 * (ustar_l + ustar_r);

    if (fabs(ustar) < RC_SMALLU * 0.5 * (fabs(ql->un) + fabs(qr->un))) {
        ustar = 0.0;
    }

    double ro, uo, po, tauo, gamco, gameo;

    if (ustar > 0.0) {
        ro = ql->rho; uo = ql->un; po = ql->p; tauo = taul; gamco = ql->gamc; gameo = gamel;
    } else if (ustar < 0.0) {
        ro = qr->rho; uo = qr->un; po = qr->p; tauo = taur; gamco = qr->gamc; gameo = gamer;
    } else {
        ro = 0.5 * (ql->rho + qr->rho);
        uo = 0.5 * (ql->un + qr->un);
        po = 0.5 * (ql->p + qr->p);
        tauo = 0.5 * (taul + taur);
        gamco = 0.5 * (ql->gamc + qr->gamc);
        gameo = 0.5 * (gamel + gamer);
    }

    ro = amax(P->small_dens, 1.0 / tauo);
    tauo = 1.0 / ro;

    double co = sqrt(fabs(gamco * po * tauo));
    co = amax(raux->csmall, co);
    double clsq = (co * ro) * (co * ro); /* std::pow(co*ro, 2) */

    double wosq = 0.0;
    wsqge(po, tauo, gameo, gdot, &gamstar, gmin, gmax, clsq, pstar, &wosq);

    double sgnm = copysign(1.0, ustar);

    double wo = sqrt(wosq);
    double dpjmp = pstar - po;

    double rstar = 1.0 - ro * dpjmp / wosq;
    rstar = ro / rstar;
    rstar = amax(P->small_dens, rstar);

    double cstar = sqrt(fabs(gamco * pstar / rstar));
    cstar = amax(cstar, raux->csmall);

    double spout = co - sgnm * uo;
    double spin = cstar - sgnm * ustar;

    double ushock = wo * tauo - sgnm * uo;

    if (pstar - po >= 0.0) {
        spin = ushock;
        spout = ushock;
    }

    double frac = 0.5 * (1.0 + (spin + spout) / amax(amax(spout - spin, spin + spout),
                                                     RC_SMALL * raux->cavg));

    if (ustar > 0.0) {
        qint->ut = ql->ut; qint->utt = ql->utt;
    } else if (ustar < 0.0) {
        qint->ut = qr->ut; qint->utt = qr->utt;
    } else {
        qint->ut = 0.5 * (ql->ut + qr->ut);
        qint->utt = 0.5 * (ql->utt + qr->utt);
    }

    qint->rho = frac * rstar + (1.0 - frac) * ro;
    qint->un = frac * ustar + (1.0 - frac) * uo;
    qint->p = frac * pstar + (1.0 - frac) * po;
    double game_int = frac * gamstar + (1.0 - frac) * gameo;

    if (spout < 0.0) {
        qint->rho = ro; qint->un = uo; qint->p = po; game_int = gameo;
    }

    if (spin >= 0.0) {
        qint->rho = rstar; qint->un = ustar; qint->p = pstar; game_int = gamstar;
    }

    qint->p = amax(qint->p, P->small_pres);

    qint->un = qint->un * raux->bnd_fac;

    qint->rhoe = qint->p / (game_int - 1.0);
    qint->gamc = 0.0;
}

/* riemann_solvers.H:14-211 -- 3-D Cartesian: mom_flux_has_p(idir,idir) is true */
static inline void compute_flux_q(int i, int j, int k, int idir, const RiemannState *qint,
                                  ora_a4 F, ora_a4 qgdnv)
{
    int m1, m2, m3;
    if (idir == 0) { m1 = UMX; m2 = UMY; m3 = UMZ; }
    else if (idir == 1) { m1 = UMY; m2 = UMX; m3 = UMZ; }
    else { m1 = UMZ; m2 = UMX; m3 = UMY; }

    A4(F,i,j,k,URHO) = qint->rho * qint->un;

    A4(F,i,j,k,m1) = A4(F,i,j,k,URHO) * qint->un;
    A4(F,i,j,k,m2) = A4(F,i,j,k,URHO) * qint->ut;
    A4(F,i,j,k,m3) = A4(F,i,j,k,URHO) * qint->utt;

    A4(F,i,j,k,m1) += qint->p;

    double rhoetot = qint->rhoe + 0.5 * qint->rho *
        (qint->un * qint->un + qint->ut * qint->ut + qint->utt * qint->utt);

    A4(F,i,j,k,UEDEN) = qint->un * (rhoetot + qint->p);
    A4(F,i,j,k,UEINT) = qint->un * qint->rhoe;

    A4(F,i,j,k,UTEMP) = 0.0;

    /* reduced Godunov state (store_full_state == false) */
    if (idir == 0) {
        A4(qgdnv,i,j,k,GDU) = qint->un; A4(qgdnv,i,j,k,GDV) = qint->ut; A4(qgdnv,i,j,k,GDW) = qint->utt;
    } else if (idir == 1) {
        A4(qgdnv,i,j,k,GDV) = qint->un; A4(qgdnv,i,j,k,GDU) = qint->ut; A4(qgdnv,i,j,k,GDW) = qint->utt;
    } else {
        A4(qgdnv,i,j,k,GDW) = qint->un; A4(qgdnv,i,j,k,GDU) = qint->ut; A4(qgdnv,i,j,k,GDV) = qint->utt;
    }
    A4(qgdnv,i,j,k,GDPRES) = qint->p;
}

/* riemann.H:379-501 helpers for HLLC */
static inline void cons_state(const double *q, double *U)
{
    U[URHO] = q[QRHO];
    U[UMX] = q[QRHO] * q[QU];
    U[UMY] = q[QRHO] * q[QV];
    U[UMZ] = q[QRHO] * q[QW];
    U[UEDEN] = q[QREINT] + 0.5 * q[QRHO] * (q[QU] * q[QU] + q[QV] * q[QV] + q[QW] * q[QW]);
    U[UEINT] = q[QREINT];
    U[UTEMP] = 0.0;
    for (int ip = 0; ip < NPASSIVE; ++ip) U[upassmap(ip)] = q[QRHO] * q[qpassmap(ip)];
}

static inline void HLLC_state(int idir, double S_k, double S_c, const double *q, double *U)
{
    double u_k = 0.0;
    if (idir == 0) u_k = q[QU];
    else if (idir == 1) u_k = q[QV];
    else if (idir == 2) u_k = q[QW];

    double hllc_factor = q[QRHO] * (S_k - u_k) / (S_k - S_c);
    U[URHO] = hllc_factor;

    if (idir == 0) {
        U[UMX] = hllc_factor * S_c; U[UMY] = hllc_factor * q[QV]; U[UMZ] = hllc_factor * q[QW];
    } else if (idir == 1) {
        U[UMX] = hllc_factor * q[QU]; U[UMY] = hllc_factor * S_c; U[UMZ] = hllc_factor * q[QW];
    } else {
        U[UMX] = hllc_factor * q[QU]; U[UMY] = hllc_factor * q[QV]; U[UMZ] = hllc_factor * S_c;
    }

    U[UEDEN] = hllc_factor * (q[QREINT] / q[QRHO] +
                              0.5 * (q[QU] * q[QU] + q[QV] * q[QV] + q[QW] * q[QW]) +
                              (S_c - u_k) * (S_c + q[QPRES] / (q[QRHO] * (S_k - u_k))));
    U[UEINT] = hllc_factor * q[QREINT] / q[QRHO];

    U[UTEMP] = 0.0;

    for (int ip = 0; ip < NPASSIVE; ++ip) U[upassmap(ip)] = hllc_factor * q[qpassmap(ip)];
}

static inline void compute_flux(int idir, double bnd_fac, const double *U, double p, double *F)
{
    double u_flx = U[UMX + idir] / U[URHO];
    if (bnd_fac == 0) u_flx = 0.0;

    F[URHO] = U[URHO] * u_flx;
    F[UMX] = U[UMX] * u_flx;
    F[UMY] = U[UMY] * u_flx;
    F[UMZ] = U[UMZ] * u_flx;

    F[UMX + idir] = F[UMX + idir] + p;

    F[UEINT] = U[UEINT] * u_flx;
    F[UEDEN] = (U[UEDEN] + p) * u_flx;

    F[UTEMP] = 0.0;

    for (int ip = 0; ip < NPASSIVE; ++ip) { int n = upassmap(ip); F[n] = U[n] * u_flx; }
}

/* riemann_solvers.H:991-1258 */
static void HLLC(int i, int j, int k, int idir, ora_a4 ql, ora_a4 qr, ora_a4 qaux, ora_a4 uflx,
                 ora_a4 qgdnv, double bnd_fac, const ora_params *P)
{
    const double small = 1.e-8;
    int iu, sx = 0, sy = 0, sz = 0;
    if (idir == 0) { iu = QU; sx = 1; } else if (idir == 1) { iu = QV; sy = 1; } else { iu = QW; sz = 1; }

    double rl = amax(A4(ql,i,j,k,QRHO), P->small_dens);
    double ul = A4(ql,i,j,k,iu);
    double pl = amax(A4(ql,i,j,k,QPRES), P->small_pres);

    double rr = amax(A4(qr,i,j,k,QRHO), P->small_dens);
    double ur = A4(qr,i,j,k,iu);
    double pr = amax(A4(qr,i,j,k,QPRES), P->small_pres);

    double csmall = amax(small, amax(small * A4(qaux,i,j,k,QC), small * A4(qaux,i-sx,j-sy,k-sz,QC)));
    double cavg = 0.5 * (A4(qaux,i,j,k,QC) + A4(qaux,i-sx,j-sy,k-sz,QC));

    double gamcl = A4(qaux,i-sx,j-sy,k-sz,QGAMC);
    double gamcr = A4(qaux,i,j,k,QGAMC);

    double wsmall = P->small_dens * csmall;
    double wl = amax(wsmall, sqrt(fabs(gamcl * pl * rl)));
    double wr = amax(wsmall, sqrt(fabs(gamcr * pr * rr)));

    double wwinv = 1.0 / (wl + wr);
    double pstar = ((wr * pl + wl * pr) + wl * wr * (ul - ur)) * wwinv;
    double ustar = ((wl * ul + wr * ur) + (pl - pr)) * wwinv;

    pstar = amax(pstar, P->small_pres);

    if (fabs(ustar) < RC_SMALLU * 0.5 * (fabs(ul) + fabs(ur))) ustar = 0.0;

    double ro, uo, po, gamco;
    if (ustar > 0.0) { ro = rl; uo = ul; po = pl; gamco = gamcl; }
    else if (ustar < 0.0) { ro = rr; uo = ur; po = pr; gamco = gamcr; }
    else { ro = 0.5 * (rl + rr); uo = 0.5 * (ul + ur); po = 0.5 * (pl + pr); gamco = 0.5 * (gamcl + gamcr); }

    ro = amax(P->small_dens, ro);

    double roinv = 1.0 / ro;
    double co = sqrt(fabs(gamco * po * roinv));
    co = amax(csmall, co);
    double co2inv = 1.0 / (co * co);

    double rstar = ro + (pstar - po) * co2inv;
    rstar = amax(P->small_dens, rstar);

    double cstar = sqrt(fabs(gamco * pstar / rstar));
    cstar = amax(cstar, csmall);

    double sgnm = copysign(1.0, ustar);
    double spout = co - sgnm * uo;
    double spin = cstar - sgnm * ustar;
    double ushock = 0.5 * (spin + spout);

    if (pstar - po > 0.0) { spin = ushock; spout = ushock; }

    double scr = spout - spin;
    if (spout - spin == 0.0) scr = small * cavg;

    double frac = (1.0 + (spout + spin) / scr) * 0.5;
    frac = amax(0.0, amin(1.0, frac));

    double qint[NQ];
    for (int n = 0; n < NQ; ++n) qint[n] = 0.0;
    qint[QRHO] = frac * rstar + (1.0 - frac) * ro;
    qint[iu] = frac * ustar + (1.0 - frac) * uo;
    qint[QPRES] = frac * pstar + (1.0 - frac) * po;

    double S_l = amin(ul - sqrt(gamcl * pl / rl), ur - sqrt(gamcr * pr / rr));
    double S_r = amax(ul + sqrt(gamcl * pl / rl), ur + sqrt(gamcr * pr / rr));

    double S_c = (pr - pl + rl * ul * (S_l - ul) - rr * ur * (S_r - ur)) /
        (rl * (S_l - ul) - rr * (S_r - ur));

    double q_zone[NQ], U_state[NUM_STATE], U_hllc_state[NUM_STATE], F_state[NUM_STATE];

    if (S_r <= 0.0) {
        for (int n = 0; n < NQ; n++) q_zone[n] = A4(qr,i,j,k,n);
        cons_state(q_zone, U_state);
        compute_flux(idir, bnd_fac, U_state, pr, F_state);
    } else if (S_r > 0.0 && S_c <= 0.0) {
        for (int n = 0; n < NQ; n++) q_zone[n] = A4(qr,i,j,k,n);
        cons_state(q_zone, U_state);
        compute_flux(idir, bnd_fac, U_state, pr, F_state);
        HLLC_state(idir, S_r, S_c, q_zone, U_hllc_state);
        for (int n = 0; n < NUM_STATE; n++) F_state[n] = F_state[n] + S_r * (U_hllc_state[n] - U_state[n]);
    } else if (S_c > 0.0 && S_l < 0.0) {
        for (int n = 0; n < NQ; n++) q_zone[n] = A4(ql,i,j,k,n);
        cons_state(q_zone, U_state);
        compute_flux(idir, bnd_fac, U_state, pl, F_state);
        HLLC_state(idir, S_l, S_c, q_zone, U_hllc_state);
        for (int n = 0; n < NUM_STATE; n++) F_state[n] = F_state[n] + S_l * (U_hllc_state[n] - U_state[n]);
    } else {
        for (int n = 0; n < NQ; n++) q_zone[n] = A4(ql,i,j,k,n);
        cons_state(q_zone, U_state);
        compute_flux(idir, bnd_fac, U_state, pl, F_state);
    }

    for (int n = 0; n < NUM_STATE; n++) A4(uflx,i,j,k,n) = F_state[n];

    A4(qgdnv,i,j,k,GDU) = qint[QU];
    A4(qgdnv,i,j,k,GDV) = qint[QV];
    A4(qgdnv,i,j,k,GDW) = qint[QW];
    A4(qgdnv,i,j,k,GDPRES) = qint[QPRES];
}

/* riemann_solvers.H:834-978 -- HLLE flux (hybrid_riemann only) */
static void HLL(const double *ql, const double *qr, double cl, double cr, int idir, double *flux_hll)
{
    const double small_hll = 1.e-10;
    int ivel, ivelt, iveltt, imom, imomt, imomtt;
    if (idir == 0) { ivel = QU; ivelt = QV; iveltt = QW; imom = UMX; imomt = UMY; imomtt = UMZ; }
    else if (idir == 1) { ivel = QV; ivelt = QU; iveltt = QW; imom = UMY; imomt = UMX; imomtt = UMZ; }
    else { ivel = QW; ivelt = QU; iveltt = QV; imom = UMZ; imomt = UMX; imomtt = UMY; }

    double rhol_sqrt = sqrt(ql[QRHO]);
    double rhor_sqrt = sqrt(qr[QRHO]);
    double rhod = 1.0 / (rhol_sqrt + rhor_sqrt);

    double dv = qr[ivel] - ql[ivel];
    double cavg = sqrt((rhol_sqrt * cl * cl + rhor_sqrt * cr * cr) * rhod +
                       0.5 * rhol_sqrt * rhor_sqrt * rhod * rhod * (dv * dv));

    double uavg = (rhol_sqrt * ql[ivel] + rhor_sqrt * qr[ivel]) * rhod;
    double a1 = uavg - cavg;
    double a4 = uavg + cavg;

    double bl = amin(a1, ql[ivel] - cl);
    double br = amax(a4, qr[ivel] + cr);

    double bm = amin(0.0, bl);
    double bp = amax(0.0, br);

    double bd = bp - bm;
    if (fabs(bd) < small_hll * amax(fabs(bm), fabs(bp))) return;

    bd = 1.0 / bd;

    double fl_tmp = ql[QRHO] * ql[ivel];
    double fr_tmp = qr[QRHO] * qr[ivel];
    flux_hll[URHO] = (bp * fl_tmp - bm * fr_tmp) * bd + bp * bm * bd * (qr[QRHO] - ql[QRHO]);

    fl_tmp = ql[QRHO] * ql[ivel] * ql[ivel];
    fr_tmp = qr[QRHO] * qr[ivel] * qr[ivel];
    fl_tmp = fl_tmp + ql[QPRES];
    fr_tmp = fr_tmp + qr[QPRES];
    flux_hll[imom] = (bp * fl_tmp - bm * fr_tmp) * bd + bp * bm * bd * (qr[QRHO] * qr[ivel] - ql[QRHO] * ql[ivel]);

    fl_tmp = ql[QRHO] * ql[ivel] * ql[ivelt];
    fr_tmp = qr[QRHO] * qr[ivel] * qr[ivelt];
    flux_hll[imomt] = (bp * fl_tmp - bm * fr_tmp) * bd + bp * bm * bd * (qr[QRHO] * qr[ivelt] - ql[QRHO] * ql[ivelt]);

    fl_tmp = ql[QRHO] * ql[ivel] * ql[iveltt];
    fr_tmp = qr[QRHO] * qr[ivel] * qr[iveltt];
    flux_hll[imomtt] = (bp * fl_tmp - bm * fr_tmp) * bd + bp * bm * bd * (qr[QRHO] * qr[iveltt] - ql[QRHO] * ql[iveltt]);

    double rhoEl = ql[QREINT] + 0.5 * ql[QRHO] * (ql[ivel] * ql[ivel] + ql[ivelt] * ql[ivelt] + ql[iveltt] * ql[iveltt]);
    fl_tmp = ql[ivel] * (rhoEl + ql[QPRES]);
    double rhoEr = qr[QREINT] + 0.5 * qr[QRHO] * (qr[ivel] * qr[ivel] + qr[ivelt] * qr[ivelt] + qr[iveltt] * qr[iveltt]);
    fr_tmp = qr[ivel] * (rhoEr + qr[QPRES]);
    flux_hll[UEDEN] = (bp * fl_tmp - bm * fr_tmp) * bd + bp * bm * bd * (rhoEr - rhoEl);

    fl_tmp = ql[QREINT] * ql[ivel];
    fr_tmp = qr[QREINT] * qr[ivel];
    flux_hll[UEINT] = (bp * fl_tmp - bm * fr_tmp) * bd + bp * bm * bd * (qr[QREINT] - ql[QREINT]);

    for (int ip = 0; ip < NPASSIVE; ip++) {
        int n = upassmap(ip);
        int nqs = qpassmap(ip);
        fl_tmp = ql[QRHO] * ql[nqs] * ql[ivel];
        fr_tmp = qr[QRHO] * qr[nqs] * qr[ivel];
        flux_hll[n] = (bp * fl_tmp - bm * fr_tmp) * bd + bp * bm * bd * (qr[QRHO] * qr[nqs] - ql[QRHO] * ql[nqs]);
    }
}

/* Castro::cmpflx_plus_godunov (riemann.cpp:15-206) with riemann_state
 * (riemann_solvers.H:1262-1388) inlined; store_full_state = false */
/* debugging aid for the parity campaigns (tools/fuzz_case_faces.py): called at the start of every box-level Riemann solve with
 * its input states, so that the solve can be repeated face by face on another implementation */
typedef void (*ora_riemann_hook_t)(int idir, const int lo[3], const int hi[3], ora_a4 qm, ora_a4 qp, ora_a4 qaux, ora_a4 shk);
static ora_riemann_hook_t ora_riemann_hook = 0;
void ora_set_debug_riemann_hook(ora_riemann_hook_t f) { ora_riemann_hook = f; }

void ora_cmpflx_plus_godunov(const int lo[3], const int hi[3], ora_a4 qm, ora_a4 qp, ora_a4 flx,
                             ora_a4 qgdnv, ora_a4 qaux, ora_a4 shk, int idir,
                             const ora_geom *G, const ora_params *P)
{
    if (ora_riemann_hook) ora_riemann_hook(idir, lo, hi, qm, qp, qaux, shk);
    const int special_bnd_lo = (G->lo_bc[idir] == BC_SYMMETRY || G->lo_bc[idir] == BC_SLIPWALL ||
                                G->lo_bc[idir] == BC_NOSLIPWALL);
    const int special_bnd_hi = (G->hi_bc[idir] == BC_SYMMETRY || G->hi_bc[idir] == BC_SLIPWALL ||
                                G->hi_bc[idir] == BC_NOSLIPWALL);

    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        const int idx[3] = {i, j, k};
        double bnd_fac = 1.0;
        if ((idx[idir] == G->domlo[idir] && special_bnd_lo) ||
            (idx[idir] == G->domhi[idir] + 1 && special_bnd_hi)) {
            bnd_fac = 0.0;
        }

        if (P->riemann_solver == 0 || P->riemann_solver == 1) {
            RiemannState ql, qr, qint;
            RiemannAux raux;

            if (P->ppm_temp_fix == 2) {
                /* riemann_solvers.H:1281-1330: recompute p on the edges from (rho, e, X) */
                ora_eos_t es;
                es.T = P->T_guess;
                es.rho = A4(qm,i,j,k,QRHO);
                es.e = A4(qm,i,j,k,QREINT) / A4(qm,i,j,k,QRHO);
                es.xn = A4(qm,i,j,k,QFS);
                ora_eos_re(P, &es);
                A4(qm,i,j,k,QREINT) = es.e * es.rho;
                A4(qm,i,j,k,QPRES) = es.p;

                es.rho = A4(qp,i,j,k,QRHO);
                es.e = A4(qp,i,j,k,QREINT) / A4(qp,i,j,k,QRHO);
                es.xn = A4(qp,i,j,k,QFS);
                ora_eos_re(P, &es);
                A4(qp,i,j,k,QREINT) = es.e * es.rho;
                A4(qp,i,j,k,QPRES) = es.p;
            }

            load_input_states(i, j, k, idir, qm, qp, qaux, &ql, &qr, &raux, P);
            raux.bnd_fac = bnd_fac;

            if (P->riemann_solver == 0) {
                ora_riemannus(&ql, &qr, &raux, &qint, P);
            } else {
                ora_riemanncg(&ql, &qr, &raux, &qint, P);
            }

            compute_flux_q(i, j, k, idir, &qint, flx, qgdnv);

            /* passives: riemann.cpp:107-131 */
            double sgnm = copysign(1.0, qint.un);
            if (qint.un == 0.0) sgnm = 0.0;

            double fp = 0.5 * (1.0 + sgnm);
            double fm = 0.5 * (1.0 - sgnm);

            for (int ip = 0; ip < NPASSIVE; ip++) {
                int nqp = qpassmap(ip);
                int n = upassmap(ip);
                double X_int = fp * A4(qm,i,j,k,nqp) + fm * A4(qp,i,j,k,nqp);
                A4(flx,i,j,k,n) = A4(flx,i,j,k,URHO) * X_int;
            }
        } else {
            HLLC(i, j, k, idir, qm, qp, qaux, flx, qgdnv, bnd_fac, P);
        }

        if (P->hybrid_riemann == 1) {
            /* riemann.cpp:150-203 */
            int im = i - (idir == 0), jm = j - (idir == 1), km = k - (idir == 2);
            int is_shock = (int)(A4(shk,im,jm,km,0) + A4(shk,i,j,k,0));
            if (is_shock >= 1) {
                double cl = A4(qaux,im,jm,km,QC);
                double cr = A4(qaux,i,j,k,QC);
                double ql_zone[NQ], qr_zone[NQ], flx_zone[NUM_STATE];
                for (int n = 0; n < NQ; n++) { ql_zone[n] = A4(qm,i,j,k,n); qr_zone[n] = A4(qp,i,j,k,n); }
                for (int n = 0; n < NUM_STATE; n++) flx_zone[n] = A4(flx,i,j,k,n);
                HLL(ql_zone, qr_zone, cl, cr, idir, flx_zone);
                for (int n = 0; n < NUM_STATE; n++) A4(flx,i,j,k,n) = flx_zone[n];
            }
        }
    }
}

/* single-interface entry point for known-answer tests.
 * q = {rho, un, ut, utt, p, rhoe, gamc} */
void ora_riemann_single(int solver, const double qlv[7], const double qrv[7], double csmall, double cavg,
                        double bnd_fac, const ora_params *P, double out[7])
{
    RiemannState ql = {qlv[0], qlv[4], qlv[5], qlv[6], qlv[1], qlv[2], qlv[3]};
    RiemannState qr = {qrv[0], qrv[4], qrv[5], qrv[6], qrv[1], qrv[2], qrv[3]};
    RiemannAux raux = {csmall, cavg, bnd_fac};
    RiemannState qi;
    if (solver == 0) ora_riemannus(&ql, &qr, &raux, &qi, P);
    else ora_riemanncg(&ql, &qr, &raux, &qi, P);
    out[0] = qi.rho; out[1] = qi.un; out[2] = qi.ut; out[3] = qi.utt; out[4] = qi.p; out[5] = qi.rhoe;
    out[6] = 0.0;
}

/* Pointwise driver of ora_cmpflx_plus_godunov for known-answer vectors: interface `p` of n lies between two zones of
 * its own (qaux QC = cl, cr; QGAMC = eos_gamma), arrays component-major a[comp * n + p]; qm, qp: 7 comps
 * (rho,u,v,w,p,rhoe,X); bnd_fac 0 puts the interface on a SlipWall; out: 11 comps (flux rho, normal / first / second
 * transverse momentum, E, eint, X, Godunov un, ut, utt, p) -- the layout of castro_amd_cmpflx_points. */
void ora_cmpflx_points(long n, int idir, const double *qm, const double *qp, const double *cl, const double *cr,
                       const double *bnd_fac, const int *is_shock, const ora_params *P, double *out)
{
    const int it = (idir == 0) ? 1 : 0, itt = (idir == 2) ? 1 : 2;
    for (long p = 0; p < n; ++p) {
        int zlo[3] = {0, 0, 0}, zhi[3] = {0, 0, 0}, flo[3] = {0, 0, 0};
        zlo[idir] = -1;
        double qmv[NQ * 2] = {0}, qpv[NQ * 2] = {0}, aux[NQAUX * 2], shk[2] = {0, 0}, flx[NUM_STATE * 2] = {0}, gd[NGDNV * 2] = {0};
        /* the face 0 of direction idir sits between zones -1 and 0; face arrays cover [-1,0] too, index 1 = face 0 */
        const int src[7] = {QRHO, QU, QV, QW, QPRES, QREINT, QFS};
        for (int m = 0; m < 7; ++m) { qmv[src[m] * 2 + 1] = qm[m * n + p]; qpv[src[m] * 2 + 1] = qp[m * n + p]; }
        aux[QGAMC * 2 + 0] = P->eos_gamma; aux[QGAMC * 2 + 1] = P->eos_gamma;
        aux[QC * 2 + 0] = cl[p]; aux[QC * 2 + 1] = cr[p];
        if (is_shock && is_shock[p]) shk[1] = 1.0;
        ora_geom G;
        memset(&G, 0, sizeof(G));
        for (int d = 0; d < 3; ++d) { G.dx[d] = 1.0; G.domlo[d] = -100; G.domhi[d] = 100; G.lo_bc[d] = BC_OUTFLOW; G.hi_bc[d] = BC_OUTFLOW; }
        if (bnd_fac && bnd_fac[p] == 0.0) { G.domlo[idir] = 0; G.lo_bc[idir] = BC_SLIPWALL; }
        ora_cmpflx_plus_godunov(flo, flo, ora_make_a4(qmv, zlo, zhi, NQ), ora_make_a4(qpv, zlo, zhi, NQ),
                                ora_make_a4(flx, zlo, zhi, NUM_STATE), ora_make_a4(gd, zlo, zhi, NGDNV),
                                ora_make_a4(aux, zlo, zhi, NQAUX), ora_make_a4(shk, zlo, zhi, 1), idir, &G, P);
        out[0 * n + p] = flx[URHO * 2 + 1];
        out[1 * n + p] = flx[(UMX + idir) * 2 + 1];
        out[2 * n + p] = flx[(UMX + it) * 2 + 1];
        out[3 * n + p] = flx[(UMX + itt) * 2 + 1];
        out[4 * n + p] = flx[UEDEN * 2 + 1];
        out[5 * n + p] = flx[UEINT * 2 + 1];
        out[6 * n + p] = flx[UFS * 2 + 1];
        out[7 * n + p] = gd[(GDU + idir) * 2 + 1];
        out[8 * n + p] = gd[(GDU + it) * 2 + 1];
        out[9 * n + p] = gd[(GDU + itt) * 2 + 1];
        out[10 * n + p] = gd[GDPRES * 2 + 1];
    }
}
