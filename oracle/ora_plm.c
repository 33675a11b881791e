/*
 * ora_plm.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see castro_oracle.h).
 * Restates the piecewise-linear predictor (ppm_type = 0): Source/hydro/slope.H (uslope, pslope),
 * Source/hydro/trace_plm.cpp (3-D) and the reflecting-boundary fix-up of
 * Castro::ctu_plm_states (Source/hydro/Castro_ctu.cpp:287-433).
 */
#include "ora_internal.h"

static inline void load_stencil(ora_a4 q, int idir, int i, int j, int k, int n, double *s)
{
    const int di = (idir == 0), dj = (idir == 1), dk = (idir == 2);
    for (int m = -2; m <= 2; ++m) s[m + 2] = A4(q, i + m * di, j + m * dj, k + m * dk, n);
}

/* slope.H:27-121 */
static double uslope(const double *q, double flatn, int bnd_lo_reflect, int bnd_hi_reflect, const ora_params *P)
{
    double dq;
    if (P->plm_iorder == 1) {
        dq = 0.0;
    } else {
        if (P->plm_limiter == 1) {
            double dlft = 2.0 * (q[i0] - q[im1]);
            double drgt = 2.0 * (q[ip1] - q[i0]);
            double dcen = 0.25 * (dlft + drgt);
            double dsgn = copysign(1.0, dcen);
            double slop = amin(fabs(dlft), fabs(drgt));
            double dlim = dlft * drgt >= 0.0 ? slop : 0.0;
            dq = flatn * dsgn * amin(dlim, fabs(dcen));
        } else {
            double qm2 = q[im2], qm1 = q[im1], q0 = q[i0], qp1 = q[ip1], qp2 = q[ip2];

            if (bnd_lo_reflect) {
                qm2 = -qp1;
                qm1 = -3.0 * q0 + qp1 - 0.125 * (qp2 + qp1);
            }
            if (bnd_hi_reflect) {
                qp2 = -qm1;
                qp1 = -3.0 * q0 + qm1 - 0.125 * (qm2 + qm1);
            }

            double dlftp1 = 2.0 * (qp1 - q0);
            double drgtp1 = 2.0 * (qp2 - qp1);
            double dcen = 0.25 * (dlftp1 + drgtp1);
            double dsgn = copysign(1.0, dcen);
            double slop = amin(fabs(dlftp1), fabs(drgtp1));
            double dlim = dlftp1 * drgtp1 >= 0.0 ? slop : 0.0;
            double dfp1 = dsgn * amin(dlim, fabs(dcen));

            double dlftm1 = 2.0 * (qm1 - qm2);
            double drgtm1 = 2.0 * (q0 - qm1);
            dcen = 0.25 * (dlftm1 + drgtm1);
            dsgn = copysign(1.0, dcen);
            slop = amin(fabs(dlftm1), fabs(drgtm1));
            dlim = dlftm1 * drgtm1 >= 0.0 ? slop : 0.0;
            double dfm1 = dsgn * amin(dlim, fabs(dcen));

            double dlft = drgtm1;
            double drgt = dlftp1;
            dcen = 0.25 * (dlft + drgt);
            dsgn = copysign(1.0, dcen);
            slop = amin(fabs(dlft), fabs(drgt));
            dlim = dlft * drgt >= 0.0 ? slop : 0.0;

            double dq1 = (4.0 / 3.0) * dcen - (1.0 / 6.0) * (dfp1 + dfm1);
            dq = flatn * dsgn * amin(dlim, fabs(dq1));
        }
    }
    return dq;
}

/* slope.H:137-241 */
static void pslope(const double *rho, const double *p, const double *src, double flatn,
                   int lo_bc_test, int hi_bc_test, double dx, double *dp, const ora_params *P)
{
    if (P->plm_iorder == 1) {
        *dp = 0.0;
    } else {
        if (rho[i0] < P->pslope_cutoff_density) return;

        double p0_hse = p[i0];
        double pp1_hse = p0_hse + 0.25 * dx * (rho[i0] + rho[ip1]) * (src[i0] + src[ip1]);
        double pp2_hse = pp1_hse + 0.25 * dx * (rho[ip1] + rho[ip2]) * (src[ip1] + src[ip2]);
        double pm1_hse = p0_hse - 0.25 * dx * (rho[i0] + rho[im1]) * (src[i0] + src[im1]);
        double pm2_hse = pm1_hse - 0.25 * dx * (rho[im1] + rho[im2]) * (src[im1] + src[im2]);

        double p0 = 0.0;
        double pp1 = p[ip1] - pp1_hse;
        double pp2 = p[ip2] - pp2_hse;
        double pm1 = p[im1] - pm1_hse;
        double pm2 = p[im2] - pm2_hse;

        if (lo_bc_test) { pm1 = 0.0; pm2 = 0.0; }
        if (hi_bc_test) { pp1 = 0.0; pp2 = 0.0; }

        double dlftp1 = pp1 - p0;
        double drgtp1 = pp2 - pp1;
        double dcen = 0.5 * (dlftp1 + drgtp1);
        double dsgn = copysign(1.0, dcen);
        double dlim = dlftp1 * drgtp1 >= 0.0 ? 2.0 * amin(fabs(dlftp1), fabs(drgtp1)) : 0.0;
        double dfp1 = dsgn * amin(dlim, fabs(dcen));

        double dlftm1 = pm1 - pm2;
        double drgtm1 = p0 - pm1;
        dcen = 0.5 * (dlftm1 + drgtm1);
        dsgn = copysign(1.0, dcen);
        dlim = dlftm1 * drgtm1 >= 0.0 ? 2.0 * amin(fabs(dlftm1), fabs(drgtm1)) : 0.0;
        double dfm1 = dsgn * amin(dlim, fabs(dcen));

        double dlft = drgtm1;
        double drgt = dlftp1;
        dcen = 0.5 * (dlft + drgt);
        dsgn = copysign(1.0, dcen);
        dlim = dlft * drgt >= 0.0 ? 2.0 * amin(fabs(dlft), fabs(drgt)) : 0.0;

        double dp1 = (4.0 / 3.0) * dcen - (1.0 / 6.0) * (dfp1 + dfm1);
        *dp = flatn * dsgn * amin(dlim, fabs(dp1));
        *dp += rho[i0] * src[i0] * dx;
    }
}

/* Castro::trace_plm (Source/hydro/trace_plm.cpp:17-339), 3-D */
void ora_trace_plm(const int lo[3], const int hi[3], int idir, ora_a4 q_arr, ora_a4 qaux_arr, ora_a4 srcQ,
                   ora_a4 flatn_arr, ora_a4 qm, ora_a4 qp, const int vlo[3], const int vhi[3],
                   double dt, const ora_geom *G, const ora_params *P)
{
    const int lo_symm = G->lo_bc[idir] == BC_SYMMETRY;
    const int hi_symm = G->hi_bc[idir] == BC_SYMMETRY;
    const double dtdx = dt / G->dx[idir];

    int QUN, QUT, QUTT;
    if (idir == 0) { QUN = QU; QUT = QV; QUTT = QW; }
    else if (idir == 1) { QUN = QV; QUT = QW; QUTT = QU; }
    else { QUN = QW; QUT = QU; QUTT = QV; }

    const double lsmall_dens = P->small_dens;
    const double lsmall_pres = P->small_pres;

    enum { IEIGN_RHO = 0, IEIGN_UN = 1, IEIGN_UT = 2, IEIGN_UTT = 3, IEIGN_P = 4, IEIGN_RE = 5, NEIGN = 6 };
    int cvars[NEIGN];
    cvars[IEIGN_RHO] = QRHO; cvars[IEIGN_UN] = QUN; cvars[IEIGN_UT] = QUT;
    cvars[IEIGN_UTT] = QUTT; cvars[IEIGN_P] = QPRES; cvars[IEIGN_RE] = QREINT;

    const int di = (idir == 0), dj = (idir == 1), dk = (idir == 2);

    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        const int idx[3] = {i, j, k};
        const int lo_bc_test = lo_symm && idx[idir] == G->domlo[idir];
        const int hi_bc_test = hi_symm && idx[idir] == G->domhi[idir];

        double cc = A4(qaux_arr,i,j,k,QC);
        double csq = cc * cc;

        double rho = A4(q_arr,i,j,k,QRHO);
        double un = A4(q_arr,i,j,k,QUN);
        double ut = A4(q_arr,i,j,k,QUT);
        double utt = A4(q_arr,i,j,k,QUTT);
        double p = A4(q_arr,i,j,k,QPRES);
        double rhoe = A4(q_arr,i,j,k,QREINT);

        double enth = (rhoe + p) / (rho * csq);

        double dq[NEIGN];
        double s[5];
        double flat = A4(flatn_arr,i,j,k,0);

        for (int n = 0; n < NEIGN; n++) {
            int v = cvars[n];
            load_stencil(q_arr, idir, i, j, k, v, s);
            int vtest = v == QUN;
            dq[n] = uslope(s, flat, lo_bc_test && vtest, hi_bc_test && vtest, P);
        }

        if (P->use_pslope == 1) {
            double trho[5], src[5];
            load_stencil(q_arr, idir, i, j, k, QPRES, s);
            load_stencil(q_arr, idir, i, j, k, QRHO, trho);
            load_stencil(srcQ, idir, i, j, k, QUN, src);
            double dp = dq[IEIGN_P];
            pslope(trho, s, src, flat, lo_bc_test, hi_bc_test, G->dx[idir], &dp, P);
            dq[IEIGN_P] = dp;
        }

        double alpham = 0.5 * (dq[IEIGN_P] / (rho * cc) - dq[IEIGN_UN]) * (rho / cc);
        double alphap = 0.5 * (dq[IEIGN_P] / (rho * cc) + dq[IEIGN_UN]) * (rho / cc);
        double alpha0r = dq[IEIGN_RHO] - dq[IEIGN_P] / csq;
        double alpha0e = dq[IEIGN_RE] - dq[IEIGN_P] * enth;
        double alpha0ut = dq[IEIGN_UT];
        double alpha0utt = dq[IEIGN_UTT];

        double e[3];
        e[0] = un - cc; e[1] = un; e[2] = un + cc;

        /* right state on the i interface */
        double ref_fac = 0.5 * (1.0 + dtdx * amin(e[0], 0.0));
        double rho_ref = rho - ref_fac * dq[IEIGN_RHO];
        double un_ref = un - ref_fac * dq[IEIGN_UN];
        double ut_ref = ut - ref_fac * dq[IEIGN_UT];
        double utt_ref = utt - ref_fac * dq[IEIGN_UTT];
        double p_ref = p - ref_fac * dq[IEIGN_P];
        double rhoe_ref = rhoe - ref_fac * dq[IEIGN_RE];

        double trace_fac0 = 0.0;
        double trace_fac1 = 0.25 * dtdx * (e[0] - e[1]) * (1.0 - copysign(1.0, e[1]));
        double trace_fac2 = 0.25 * dtdx * (e[0] - e[2]) * (1.0 - copysign(1.0, e[2]));

        double apright = trace_fac2 * alphap;
        double amright = trace_fac0 * alpham;

        double azrright = trace_fac1 * alpha0r;
        double azeright = trace_fac1 * alpha0e;
        double azut1rght = trace_fac1 * alpha0ut;
        double azutt1rght = trace_fac1 * alpha0utt;

        if (idx[idir] >= vlo[idir]) {
            A4(qp,i,j,k,QRHO) = amax(lsmall_dens, rho_ref + apright + amright + azrright);
            A4(qp,i,j,k,QUN) = un_ref + (apright - amright) * cc / rho;
            A4(qp,i,j,k,QUT) = ut_ref + azut1rght;
            A4(qp,i,j,k,QUTT) = utt_ref + azutt1rght;
            A4(qp,i,j,k,QPRES) = amax(lsmall_pres, p_ref + (apright + amright) * csq);
            A4(qp,i,j,k,QREINT) = rhoe_ref + (apright + amright) * enth * csq + azeright;

            A4(qp,i,j,k,QRHO) += 0.5 * dt * A4(srcQ,i,j,k,QRHO);
            A4(qp,i,j,k,QRHO) = amax(lsmall_dens, A4(qp,i,j,k,QRHO));
            A4(qp,i,j,k,QUN) += 0.5 * dt * A4(srcQ,i,j,k,QUN);
            A4(qp,i,j,k,QUT) += 0.5 * dt * A4(srcQ,i,j,k,QUT);
            A4(qp,i,j,k,QUTT) += 0.5 * dt * A4(srcQ,i,j,k,QUTT);
            A4(qp,i,j,k,QREINT) += 0.5 * dt * A4(srcQ,i,j,k,QREINT);
            A4(qp,i,j,k,QPRES) += 0.5 * dt * A4(srcQ,i,j,k,QPRES);
        }

        /* left state on the i+1 interface */
        ref_fac = 0.5 * (1.0 - dtdx * amax(e[2], 0.0));
        rho_ref = rho + ref_fac * dq[IEIGN_RHO];
        un_ref = un + ref_fac * dq[IEIGN_UN];
        ut_ref = ut + ref_fac * dq[IEIGN_UT];
        utt_ref = utt + ref_fac * dq[IEIGN_UTT];
        p_ref = p + ref_fac * dq[IEIGN_P];
        rhoe_ref = rhoe + ref_fac * dq[IEIGN_RE];

        trace_fac0 = 0.25 * dtdx * (e[2] - e[0]) * (1.0 + copysign(1.0, e[0]));
        trace_fac1 = 0.25 * dtdx * (e[2] - e[1]) * (1.0 + copysign(1.0, e[1]));
        trace_fac2 = 0.0;

        double apleft = trace_fac2 * alphap;
        double amleft = trace_fac0 * alpham;

        double azrleft = trace_fac1 * alpha0r;
        double azeleft = trace_fac1 * alpha0e;
        double azut1left = trace_fac1 * alpha0ut;
        double azutt1left = trace_fac1 * alpha0utt;

        if (idx[idir] <= vhi[idir]) {
            const int ii = i + di, jj = j + dj, kk = k + dk;
            A4(qm,ii,jj,kk,QRHO) = amax(lsmall_dens, rho_ref + apleft + amleft + azrleft);
            A4(qm,ii,jj,kk,QUN) = un_ref + (apleft - amleft) * cc / rho;
            A4(qm,ii,jj,kk,QUT) = ut_ref + azut1left;
            A4(qm,ii,jj,kk,QUTT) = utt_ref + azutt1left;
            A4(qm,ii,jj,kk,QPRES) = amax(lsmall_pres, p_ref + (apleft + amleft) * csq);
            A4(qm,ii,jj,kk,QREINT) = rhoe_ref + (apleft + amleft) * enth * csq + azeleft;

            A4(qm,ii,jj,kk,QRHO) = amax(lsmall_dens, A4(qm,ii,jj,kk,QRHO) + 0.5 * dt * A4(srcQ,i,j,k,QRHO));
            A4(qm,ii,jj,kk,QUN) += 0.5 * dt * A4(srcQ,i,j,k,QUN);
            A4(qm,ii,jj,kk,QUT) += 0.5 * dt * A4(srcQ,i,j,k,QUT);
            A4(qm,ii,jj,kk,QUTT) += 0.5 * dt * A4(srcQ,i,j,k,QUTT);
            A4(qm,ii,jj,kk,QREINT) += 0.5 * dt * A4(srcQ,i,j,k,QREINT);
            A4(qm,ii,jj,kk,QPRES) += 0.5 * dt * A4(srcQ,i,j,k,QPRES);
        }

        /* passives :305-336 */
        for (int ipassive = 0; ipassive < NPASSIVE; ipassive++) {
            int n = qpassmap(ipassive);
            load_stencil(q_arr, idir, i, j, k, n, s);
            double dX = uslope(s, flat, 0, 0, P);

            if (idx[idir] >= vlo[idir]) {
                double spzero = un >= 0.0 ? -1.0 : un * dtdx;
                A4(qp,i,j,k,n) = A4(q_arr,i,j,k,n) + 0.5 * (-1.0 - spzero) * dX;
            }

            double spzero = un >= 0.0 ? un * dtdx : 1.0;
            double acmpleft = 0.5 * (1.0 - spzero) * dX;
            if (idx[idir] <= vhi[idir]) {
                A4(qm,i+di,j+dj,k+dk,n) = A4(q_arr,i,j,k,n) + acmpleft;
            }
        }
    }
}

/* Castro::ctu_plm_states fix-up, Castro_ctu.cpp:287-433: at a Symmetry boundary the state outside the
 * domain is the reflection of the inside one */
void ora_plm_reflect_fix(const int lo[3], const int hi[3], int idir, ora_a4 qm, ora_a4 qp, const ora_geom *G)
{
    const int lo_bc_test = G->lo_bc[idir] == BC_SYMMETRY;
    const int hi_bc_test = G->hi_bc[idir] == BC_SYMMETRY;
    const int QUN = QU + idir;
    if (!lo_bc_test && !hi_bc_test) return;
    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        const int idx[3] = {i, j, k};
        if (lo_bc_test && idx[idir] == G->domlo[idir]) {
            for (int n = 0; n < NQ; n++) {
                if (n == QUN) A4(qm,i,j,k,n) = -A4(qp,i,j,k,n);
                else A4(qm,i,j,k,n) = A4(qp,i,j,k,n);
            }
        }
        if (hi_bc_test && idx[idir] == G->domhi[idir] + 1) {
            for (int n = 0; n < NQ; n++) {
                if (n == QUN) A4(qp,i,j,k,n) = -A4(qm,i,j,k,n);
                else A4(qp,i,j,k,n) = A4(qm,i,j,k,n);
            }
        }
    }
}
