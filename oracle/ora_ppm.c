/*
 * ora_ppm.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see castro_oracle.h).
 * Restates Source/hydro/ppm.H, reconstruction.H and trace_ppm.cpp (3-D, no
 * radiation): PPM reconstruction + characteristic tracing.
 */
#include "ora_internal.h"

/* reconstruction.H:10-39 */
static inline void load_stencil(ora_a4 q, int idir, int i, int j, int k, int n, double *s)
{
    if (idir == 0) {
        s[im2] = A4(q,i-2,j,k,n); s[im1] = A4(q,i-1,j,k,n); s[i0] = A4(q,i,j,k,n);
        s[ip1] = A4(q,i+1,j,k,n); s[ip2] = A4(q,i+2,j,k,n);
    } else if (idir == 1) {
        s[im2] = A4(q,i,j-2,k,n); s[im1] = A4(q,i,j-1,k,n); s[i0] = A4(q,i,j,k,n);
        s[ip1] = A4(q,i,j+1,k,n); s[ip2] = A4(q,i,j+2,k,n);
    } else {
        s[im2] = A4(q,i,j,k-2,n); s[im1] = A4(q,i,j,k-1,n); s[i0] = A4(q,i,j,k,n);
        s[ip1] = A4(q,i,j,k+1,n); s[ip2] = A4(q,i,j,k+2,n);
    }
}

/* ppm.H:54-139 */
void ora_ppm_reconstruct(const double *s, double flatn, double *sm_out, double *sp_out)
{
    double sm, sp;

    /* van Leer slopes */
    double dsl = 2.0 * (s[im1] - s[im2]);
    double dsr = 2.0 * (s[i0] - s[im1]);

    double dsvl_l = 0.0;
    if (dsl * dsr > 0.0) {
        double dsc = 0.5 * (s[i0] - s[im2]);
        dsvl_l = copysign(1.0, dsc) * amin(fabs(dsc), amin(fabs(dsl), fabs(dsr)));
    }

    dsl = 2.0 * (s[i0] - s[im1]);
    dsr = 2.0 * (s[ip1] - s[i0]);

    double dsvl_r = 0.0;
    if (dsl * dsr > 0.0) {
        double dsc = 0.5 * (s[ip1] - s[im1]);
        dsvl_r = copysign(1.0, dsc) * amin(fabs(dsc), amin(fabs(dsl), fabs(dsr)));
    }

    /* interpolate s to edges */
    sm = 0.5 * (s[i0] + s[im1]) - (1.0 / 6.0) * (dsvl_r - dsvl_l);

    /* make sure sedge lies in between adjacent cell-centered values */
    sm = amax(sm, amin(s[i0], s[im1]));
    sm = amin(sm, amax(s[i0], s[im1]));

    dsl = 2.0 * (s[i0] - s[im1]);
    dsr = 2.0 * (s[ip1] - s[i0]);

    dsvl_l = 0.0;
    if (dsl * dsr > 0.0) {
        double dsc = 0.5 * (s[ip1] - s[im1]);
        dsvl_l = copysign(1.0, dsc) * amin(fabs(dsc), amin(fabs(dsl), fabs(dsr)));
    }

    dsl = 2.0 * (s[ip1] - s[i0]);
    dsr = 2.0 * (s[ip2] - s[ip1]);

    dsvl_r = 0.0;
    if (dsl * dsr > 0.0) {
        double dsc = 0.5 * (s[ip2] - s[i0]);
        dsvl_r = copysign(1.0, dsc) * amin(fabs(dsc), amin(fabs(dsl), fabs(dsr)));
    }

    sp = 0.5 * (s[ip1] + s[i0]) - (1.0 / 6.0) * (dsvl_r - dsvl_l);

    sp = amax(sp, amin(s[ip1], s[i0]));
    sp = amin(sp, amax(s[ip1], s[i0]));

    /* flatten the parabola */
    sm = flatn * sm + (1.0 - flatn) * s[i0];
    sp = flatn * sp + (1.0 - flatn) * s[i0];

    /* Colella & Sekora (2008) quadratic limiter, ppm.H:128-137 */
    if ((sp - s[i0]) * (s[i0] - sm) <= 0.0) {
        sp = s[i0];
        sm = s[i0];
    } else if (fabs(sp - s[i0]) >= 2.0 * fabs(sm - s[i0])) {
        sp = 3.0 * s[i0] - 2.0 * sm;
    } else if (fabs(sm - s[i0]) >= 2.0 * fabs(sp - s[i0])) {
        sm = 3.0 * s[i0] - 2.0 * sp;
    }

    *sm_out = sm;
    *sp_out = sp;
}

/* ppm.H:225-252 */
static inline void ppm_int_profile_single(double sm, double sp, double sc, double lam, double dtdx,
                                          double *Ip, double *Im)
{
    double s6 = 6.0 * sc - 3.0 * (sm + sp);
    double sigma = fabs(lam) * dtdx;

    if (lam <= 0.0) {
        *Ip = sp;
        *Im = sm + 0.5 * sigma * (sp - sm + (1.0 - (2.0 / 3.0) * sigma) * s6);
    } else {
        *Ip = sp - 0.5 * sigma * (sp - sm - (1.0 - (2.0 / 3.0) * sigma) * s6);
        *Im = sm;
    }
}

/* ppm.H:157-211 */
void ora_ppm_int_profile(double sm, double sp, double sc, double u, double c, double dtdx,
                         double *Ip, double *Im)
{
    double s6 = 6.0 * sc - 3.0 * (sm + sp);

    /* u-c wave */
    double speed = u - c;
    double sigma = fabs(speed) * dtdx;
    if (speed <= 0.0) {
        Ip[0] = sp;
        Im[0] = sm + 0.5 * sigma * (sp - sm + (1.0 - (2.0 / 3.0) * sigma) * s6);
    } else {
        Ip[0] = sp - 0.5 * sigma * (sp - sm - (1.0 - (2.0 / 3.0) * sigma) * s6);
        Im[0] = sm;
    }

    /* u wave */
    speed = u;
    sigma = fabs(speed) * dtdx;
    if (speed <= 0.0) {
        Ip[1] = sp;
        Im[1] = sm + 0.5 * sigma * (sp - sm + (1.0 - (2.0 / 3.0) * sigma) * s6);
    } else {
        Ip[1] = sp - 0.5 * sigma * (sp - sm - (1.0 - (2.0 / 3.0) * sigma) * s6);
        Im[1] = sm;
    }

    /* u+c wave */
    speed = u + c;
    sigma = fabs(speed) * dtdx;
    if (speed <= 0.0) {
        Ip[2] = sp;
        Im[2] = sm + 0.5 * sigma * (sp - sm + (1.0 - (2.0 / 3.0) * sigma) * s6);
    } else {
        Ip[2] = sp - 0.5 * sigma * (sp - sm - (1.0 - (2.0 / 3.0) * sigma) * s6);
        Im[2] = sm;
    }
}

/* Castro::trace_ppm  (Source/hydro/trace_ppm.cpp:15-594), 3-D, no radiation */
void ora_trace_ppm(const int lo[3], const int hi[3], int idir, ora_a4 q_arr, ora_a4 qaux_arr, ora_a4 srcQ,
                   ora_a4 flatn, ora_a4 qm, ora_a4 qp, const int vlo[3], const int vhi[3],
                   double dt, const ora_geom *G, const ora_params *P)
{
    double hdt = 0.5 * dt;
    double dtdx = dt / G->dx[idir];

    /* :66-93 CPU pre-scan: do we trace under the sources at all? */
    int do_source_trace[NQSRC];
    for (int n = 0; n < NQSRC; ++n) {
        do_source_trace[n] = 0;
        if (!srcQ.p) continue;
        for (int k = lo[2] - 2; k <= hi[2] + 2 && !do_source_trace[n]; ++k)
        for (int j = lo[1] - 2; j <= hi[1] + 2 && !do_source_trace[n]; ++j)
        for (int i = lo[0] - 2; i <= hi[0] + 2; ++i) {
            if (fabs(A4(srcQ,i,j,k,n)) > 0.0) { do_source_trace[n] = 1; break; }
        }
    }

    int QUN, QUT, QUTT;
    if (idir == 0) { QUN = QU; QUT = QV; QUTT = QW; }
    else if (idir == 1) { QUN = QV; QUT = QW; QUTT = QU; }
    else { QUN = QW; QUT = QU; QUTT = QV; }

    double lsmall_dens = P->small_dens;
    double lsmall_pres = P->small_pres;

    const int di = (idir == 0), dj = (idir == 1), dk = (idir == 2);

    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {

        double cc = A4(qaux_arr,i,j,k,QC);
        double un = A4(q_arr,i,j,k,QUN);

        double s[5];
        double flat = A4(flatn,i,j,k,0);
        double sm, sp;

        /* density */
        double Ip_rho[3], Im_rho[3];
        load_stencil(q_arr, idir, i, j, k, QRHO, s);
        ora_ppm_reconstruct(s, flat, &sm, &sp);
        ora_ppm_int_profile(sm, sp, s[i0], un, cc, dtdx, Ip_rho, Im_rho);

        /* normal velocity */
        double Ip_un_0, Im_un_0, Ip_un_2, Im_un_2;
        load_stencil(q_arr, idir, i, j, k, QUN, s);
        ora_ppm_reconstruct(s, flat, &sm, &sp);
        ppm_int_profile_single(sm, sp, s[i0], un - cc, dtdx, &Ip_un_0, &Im_un_0);
        ppm_int_profile_single(sm, sp, s[i0], un + cc, dtdx, &Ip_un_2, &Im_un_2);

        /* pressure */
        double Ip_p[3], Im_p[3];
        load_stencil(q_arr, idir, i, j, k, QPRES, s);
        ora_ppm_reconstruct(s, flat, &sm, &sp);
        ora_ppm_int_profile(sm, sp, s[i0], un, cc, dtdx, Ip_p, Im_p);

        /* rho e */
        double Ip_rhoe[3], Im_rhoe[3];
        load_stencil(q_arr, idir, i, j, k, QREINT, s);
        ora_ppm_reconstruct(s, flat, &sm, &sp);
        ora_ppm_int_profile(sm, sp, s[i0], un, cc, dtdx, Ip_rhoe, Im_rhoe);

        /* transverse velocities */
        double Ip_ut_1, Im_ut_1, Ip_utt_1, Im_utt_1;
        load_stencil(q_arr, idir, i, j, k, QUT, s);
        ora_ppm_reconstruct(s, flat, &sm, &sp);
        ppm_int_profile_single(sm, sp, s[i0], un, dtdx, &Ip_ut_1, &Im_ut_1);

        load_stencil(q_arr, idir, i, j, k, QUTT, s);
        ora_ppm_reconstruct(s, flat, &sm, &sp);
        ppm_int_profile_single(sm, sp, s[i0], un, dtdx, &Ip_utt_1, &Im_utt_1);

        /* gamma_c */
        double Ip_gc_0, Im_gc_0, Ip_gc_2, Im_gc_2;
        load_stencil(qaux_arr, idir, i, j, k, QGAMC, s);
        ora_ppm_reconstruct(s, flat, &sm, &sp);
        ppm_int_profile_single(sm, sp, s[i0], un - cc, dtdx, &Ip_gc_0, &Im_gc_0);
        ppm_int_profile_single(sm, sp, s[i0], un + cc, dtdx, &Ip_gc_2, &Im_gc_2);

        /* source terms :226-330 */
        double Ip_src_rho[3] = {0.0, 0.0, 0.0}, Im_src_rho[3] = {0.0, 0.0, 0.0};
        if (do_source_trace[QRHO]) {
            load_stencil(srcQ, idir, i, j, k, QRHO, s);
            ora_ppm_reconstruct(s, flat, &sm, &sp);
            ora_ppm_int_profile(sm, sp, s[i0], un, cc, dtdx, Ip_src_rho, Im_src_rho);
        }

        double Ip_src_un_0 = 0.0, Im_src_un_0 = 0.0, Ip_src_un_2 = 0.0, Im_src_un_2 = 0.0;
        if (do_source_trace[QUN]) {
            load_stencil(srcQ, idir, i, j, k, QUN, s);
            ora_ppm_reconstruct(s, flat, &sm, &sp);
            ppm_int_profile_single(sm, sp, s[i0], un - cc, dtdx, &Ip_src_un_0, &Im_src_un_0);
            ppm_int_profile_single(sm, sp, s[i0], un + cc, dtdx, &Ip_src_un_2, &Im_src_un_2);
        }

        double Ip_src_p[3] = {0.0, 0.0, 0.0}, Im_src_p[3] = {0.0, 0.0, 0.0};
        if (do_source_trace[QPRES]) {
            load_stencil(srcQ, idir, i, j, k, QPRES, s);
            ora_ppm_reconstruct(s, flat, &sm, &sp);
            ora_ppm_int_profile(sm, sp, s[i0], un, cc, dtdx, Ip_src_p, Im_src_p);
        }

        double Ip_src_rhoe[3] = {0.0, 0.0, 0.0}, Im_src_rhoe[3] = {0.0, 0.0, 0.0};
        if (do_source_trace[QREINT]) {
            load_stencil(srcQ, idir, i, j, k, QREINT, s);
            ora_ppm_reconstruct(s, flat, &sm, &sp);
            ora_ppm_int_profile(sm, sp, s[i0], un, cc, dtdx, Ip_src_rhoe, Im_src_rhoe);
        }

        double Ip_src_ut_1 = 0.0, Im_src_ut_1 = 0.0;
        if (do_source_trace[QUT]) {
            load_stencil(srcQ, idir, i, j, k, QUT, s);
            ora_ppm_reconstruct(s, flat, &sm, &sp);
            ppm_int_profile_single(sm, sp, s[i0], un, dtdx, &Ip_src_ut_1, &Im_src_ut_1);
        }

        double Ip_src_utt_1 = 0.0, Im_src_utt_1 = 0.0;
        if (do_source_trace[QUTT]) {
            load_stencil(srcQ, idir, i, j, k, QUTT, s);
            ora_ppm_reconstruct(s, flat, &sm, &sp);
            ppm_int_profile_single(sm, sp, s[i0], un, dtdx, &Ip_src_utt_1, &Im_src_utt_1);
        }

        const int do_plus  = (idir == 0 && i >= vlo[0]) || (idir == 1 && j >= vlo[1]) || (idir == 2 && k >= vlo[2]);
        const int do_minus = (idir == 0 && i <= vhi[0]) || (idir == 1 && j <= vhi[1]) || (idir == 2 && k <= vhi[2]);

        /* passives :340-377 */
        for (int ipassive = 0; ipassive < NPASSIVE; ++ipassive) {
            int n = qpassmap(ipassive);
            double Ip_passive, Im_passive;
            load_stencil(q_arr, idir, i, j, k, n, s);
            ora_ppm_reconstruct(s, flat, &sm, &sp);
            ppm_int_profile_single(sm, sp, s[i0], un, dtdx, &Ip_passive, &Im_passive);

            if (do_plus) A4(qp,i,j,k,n) = Im_passive;
            if (do_minus) A4(qm,i+di,j+dj,k+dk,n) = Ip_passive;
        }

        /* plus state on face i  :382-466 */
        if (do_plus) {
            double rho_ref = Im_rho[0] + hdt * Im_src_rho[0];
            double un_ref = Im_un_0 + hdt * Im_src_un_0;

            double p_ref = Im_p[0] + hdt * Im_src_p[0];
            double rhoe_g_ref = Im_rhoe[0] + hdt * Im_src_rhoe[0];

            double gam_g_ref = Im_gc_0;

            rho_ref = amax(rho_ref, lsmall_dens);

            double rho_ref_inv = 1.0 / rho_ref;
            p_ref = amax(p_ref, lsmall_pres);

            double csq_ref = gam_g_ref * p_ref * rho_ref_inv;
            double cc_ref = sqrt(csq_ref);
            double cc_ref_inv = 1.0 / cc_ref;
            double h_g_ref = (p_ref + rhoe_g_ref) * rho_ref_inv;

            double dum = un_ref - Im_un_0 - hdt * Im_src_un_0;
            double dptotm = p_ref - Im_p[0] - hdt * Im_src_p[0];

            double drho = rho_ref - Im_rho[1] - hdt * Im_src_rho[1];
            double dptot = p_ref - Im_p[1] - hdt * Im_src_p[1];
            double drhoe_g = rhoe_g_ref - Im_rhoe[1] - hdt * Im_src_rhoe[1];

            double dup = un_ref - Im_un_2 - hdt * Im_src_un_2;
            double dptotp = p_ref - Im_p[2] - hdt * Im_src_p[2];

            double alpham = 0.5 * (dptotm * rho_ref_inv * cc_ref_inv - dum) * rho_ref * cc_ref_inv;
            double alphap = 0.5 * (dptotp * rho_ref_inv * cc_ref_inv + dup) * rho_ref * cc_ref_inv;
            double alpha0r = drho - dptot / csq_ref;
            double alpha0e_g = drhoe_g - dptot * h_g_ref / csq_ref;

            alpham = un - cc > 0.0 ? 0.0 : -alpham;
            alphap = un + cc > 0.0 ? 0.0 : -alphap;
            alpha0r = un > 0.0 ? 0.0 : -alpha0r;
            alpha0e_g = un > 0.0 ? 0.0 : -alpha0e_g;

            A4(qp,i,j,k,QRHO) = amax(lsmall_dens, rho_ref + alphap + alpham + alpha0r);
            A4(qp,i,j,k,QUN) = un_ref + (alphap - alpham) * cc_ref * rho_ref_inv;
            A4(qp,i,j,k,QREINT) = amax(P->small_dens * P->small_ener,
                                       rhoe_g_ref + (alphap + alpham) * h_g_ref + alpha0e_g);
            A4(qp,i,j,k,QPRES) = amax(lsmall_pres, p_ref + (alphap + alpham) * csq_ref);

            A4(qp,i,j,k,QUT) = Im_ut_1 + hdt * Im_src_ut_1;
            A4(qp,i,j,k,QUTT) = Im_utt_1 + hdt * Im_src_utt_1;
        }

        /* minus state on face i+1  :470-561 */
        if (do_minus) {
            double rho_ref = Ip_rho[2] + hdt * Ip_src_rho[2];
            double un_ref = Ip_un_2 + hdt * Ip_src_un_2;

            double p_ref = Ip_p[2] + hdt * Ip_src_p[2];
            double rhoe_g_ref = Ip_rhoe[2] + hdt * Ip_src_rhoe[2];

            double gam_g_ref = Ip_gc_2;

            rho_ref = amax(rho_ref, lsmall_dens);
            double rho_ref_inv = 1.0 / rho_ref;
            p_ref = amax(p_ref, lsmall_pres);

            double csq_ref = gam_g_ref * p_ref * rho_ref_inv;
            double cc_ref = sqrt(csq_ref);
            double cc_ref_inv = 1.0 / cc_ref;
            double h_g_ref = (p_ref + rhoe_g_ref) * rho_ref_inv;

            double dum = un_ref - Ip_un_0 - hdt * Ip_src_un_0;
            double dptotm = p_ref - Ip_p[0] - hdt * Ip_src_p[0];

            double drho = rho_ref - Ip_rho[1] - hdt * Ip_src_rho[1];
            double dptot = p_ref - Ip_p[1] - hdt * Ip_src_p[1];
            double drhoe_g = rhoe_g_ref - Ip_rhoe[1] - hdt * Ip_src_rhoe[1];

            double dup = un_ref - Ip_un_2 - hdt * Ip_src_un_2;
            double dptotp = p_ref - Ip_p[2] - hdt * Ip_src_p[2];

            double alpham = 0.5 * (dptotm * rho_ref_inv * cc_ref_inv - dum) * rho_ref * cc_ref_inv;
            double alphap = 0.5 * (dptotp * rho_ref_inv * cc_ref_inv + dup) * rho_ref * cc_ref_inv;
            double alpha0r = drho - dptot / csq_ref;
            double alpha0e_g = drhoe_g - dptot * h_g_ref / csq_ref;

            alpham = un - cc > 0.0 ? -alpham : 0.0;
            alphap = un + cc > 0.0 ? -alphap : 0.0;
            alpha0r = un > 0.0 ? -alpha0r : 0.0;
            alpha0e_g = un > 0.0 ? -alpha0e_g : 0.0;

            const int ii = i + di, jj = j + dj, kk = k + dk;
            A4(qm,ii,jj,kk,QRHO) = amax(lsmall_dens, rho_ref + alphap + alpham + alpha0r);
            A4(qm,ii,jj,kk,QUN) = un_ref + (alphap - alpham) * cc_ref * rho_ref_inv;
            A4(qm,ii,jj,kk,QREINT) = amax(P->small_dens * P->small_ener,
                                          rhoe_g_ref + (alphap + alpham) * h_g_ref + alpha0e_g);
            A4(qm,ii,jj,kk,QPRES) = amax(lsmall_pres, p_ref + (alphap + alpham) * csq_ref);

            A4(qm,ii,jj,kk,QUT) = Ip_ut_1 + hdt * Ip_src_ut_1;
            A4(qm,ii,jj,kk,QUTT) = Ip_utt_1 + hdt * Ip_src_utt_1;
        }
    }
}
