/*
 * ora_ctu_hydro.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see castro_oracle.h).
 * Restates the orchestration of Castro::construct_ctu_hydro_source
 * (Source/hydro/Castro_ctu_hydro.cpp:16-1528, 3-D / no radiation branch) and
 * Castro::ctu_ppm_states (Source/hydro/Castro_ctu.cpp:89-150): per tile,
 * thread-private scratch FABs, the fixed sequence of ~75 sweeps (SURVEY.md
 * A.3), OpenMP over tiles (Castro_ctu_hydro.cpp:66-72,130).
 */
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "ora_internal.h"



static void box_grow(const int lo[3], const int hi[3], int gx, int gy, int gz, int olo[3], int ohi[3])
{
    olo[0] = lo[0] - gx; olo[1] = lo[1] - gy; olo[2] = lo[2] - gz;
    ohi[0] = hi[0] + gx; ohi[1] = hi[1] + gy; ohi[2] = hi[2] + gz;
}

static void surrounding_nodes(const int lo[3], const int hi[3], int d, int olo[3], int ohi[3])
{
    for (int n = 0; n < 3; ++n) { olo[n] = lo[n]; ohi[n] = hi[n]; }
    ohi[d] += 1;
}

typedef struct {
    ora_fab flatn, shk, q, qaux, src_q;
    ora_fab qxm, qxp, qym, qyp, qzm, qzp;
    ora_fab div, ftmp1, ftmp2, qgdnvtmp1, qgdnvtmp2, ql, qr;
    ora_fab flux[3], qe[3];
    ora_fab qmyx, qpyx, qmzx, qpzx, qmxy, qpxy, qmzy, qpzy, qmxz, qpxz, qmyz, qpyz;
} scratch_t;

static void scratch_free(scratch_t *s)
{
    ora_fab *f = (ora_fab *)s;
    for (size_t n = 0; n < sizeof(scratch_t) / sizeof(ora_fab); ++n) fab_free(&f[n]);
}

/* one tile: body of the MFIter loop, Castro_ctu_hydro.cpp:130-1480 */
static int ctu_tile(const int bxlo[3], const int bxhi[3], const int vlo[3], const int vhi[3],
                    ora_a4 Sborder, ora_a4 old_source, ora_a4 S_new, ora_a4 *fluxes, ora_a4 *mass_fluxes,
                    ora_a4 *qe_out, const ora_geom *G, const ora_params *P, double dt, scratch_t *S)
{
    const double *dx = G->dx;
    int bad = 0;

    int obxlo[3], obxhi[3];
    box_grow(bxlo, bxhi, 1, 1, 1, obxlo, obxhi);                       /* :137 */

    fab_resize(&S->flatn, obxlo, obxhi, 1);

    int qbxlo[3], qbxhi[3], qbx3lo[3], qbx3hi[3];
    box_grow(bxlo, bxhi, NUM_GROW, NUM_GROW, NUM_GROW, qbxlo, qbxhi);  /* :186 */
    box_grow(bxlo, bxhi, 3, 3, 3, qbx3lo, qbx3hi);                     /* :187 */

    fab_resize(&S->q, qbxlo, qbxhi, NQ);
    fab_resize(&S->qaux, qbxlo, qbxhi, NQAUX);

    bad |= ora_ctoprim(qbxlo, qbxhi, Sborder, S->q.a, S->qaux.a, P);   /* :199 */

    /* flattening coefficient :228-266 */
    if (P->first_order_hydro == 1) {
        for (long n = 0; n < S->flatn.a.sn; ++n) S->flatn.a.p[n] = 0.0;
    } else if (P->use_flattening == 1) {
        ora_uflatten(obxlo, obxhi, S->q.a, S->flatn.a, QPRES);
    } else {
        for (long n = 0; n < S->flatn.a.sn; ++n) S->flatn.a.p[n] = 1.0;
    }

    int xbxlo[3], xbxhi[3], ybxlo[3], ybxhi[3], zbxlo[3], zbxhi[3];
    surrounding_nodes(bxlo, bxhi, 0, xbxlo, xbxhi);                    /* :268-277 */
    surrounding_nodes(bxlo, bxhi, 1, ybxlo, ybxhi);
    surrounding_nodes(bxlo, bxhi, 2, zbxlo, zbxhi);
    int gxlo[3], gxhi[3], gylo[3], gyhi[3], gzlo[3], gzhi[3];
    box_grow(xbxlo, xbxhi, 1, 1, 1, gxlo, gxhi);
    box_grow(ybxlo, ybxhi, 1, 1, 1, gylo, gyhi);
    box_grow(zbxlo, zbxhi, 1, 1, 1, gzlo, gzhi);

    /* shock flag :279-303 */
    fab_resize(&S->shk, obxlo, obxhi, 1);
    if (P->hybrid_riemann == 1) {
        ora_shock(obxlo, obxhi, S->q.a, S->shk.a, G);
    } else {
        for (long n = 0; n < S->shk.a.sn; ++n) S->shk.a.p[n] = 0.0;
    }

    /* primitive-variable sources :307-315 */
    fab_resize(&S->src_q, qbx3lo, qbx3hi, NQSRC);
    ora_src_to_prim(qbx3lo, qbx3hi, S->q.a, old_source, S->src_q.a, P, dt);

    /* interface states :337-428 */
    fab_resize(&S->qxm, obxlo, obxhi, NQ); fab_resize(&S->qxp, obxlo, obxhi, NQ);
    fab_resize(&S->qym, obxlo, obxhi, NQ); fab_resize(&S->qyp, obxlo, obxhi, NQ);
    fab_resize(&S->qzm, obxlo, obxhi, NQ); fab_resize(&S->qzp, obxlo, obxhi, NQ);

    if (P->ppm_type == 0) {
        /* Castro::ctu_plm_states, Castro_ctu.cpp:216-434: trace, then the reflecting-BC fix-up */
        ora_trace_plm(obxlo, obxhi, 0, S->q.a, S->qaux.a, S->src_q.a, S->flatn.a, S->qxm.a, S->qxp.a, bxlo, bxhi, dt, G, P);
        ora_plm_reflect_fix(obxlo, obxhi, 0, S->qxm.a, S->qxp.a, G);
        ora_trace_plm(obxlo, obxhi, 1, S->q.a, S->qaux.a, S->src_q.a, S->flatn.a, S->qym.a, S->qyp.a, bxlo, bxhi, dt, G, P);
        ora_plm_reflect_fix(obxlo, obxhi, 1, S->qym.a, S->qyp.a, G);
        ora_trace_plm(obxlo, obxhi, 2, S->q.a, S->qaux.a, S->src_q.a, S->flatn.a, S->qzm.a, S->qzp.a, bxlo, bxhi, dt, G, P);
        ora_plm_reflect_fix(obxlo, obxhi, 2, S->qzm.a, S->qzp.a, G);
    } else {
        /* Castro::ctu_ppm_states, Castro_ctu.cpp:112-149 */
        ora_trace_ppm(obxlo, obxhi, 0, S->q.a, S->qaux.a, S->src_q.a, S->flatn.a, S->qxm.a, S->qxp.a, bxlo, bxhi, dt, G, P);
        ora_trace_ppm(obxlo, obxhi, 1, S->q.a, S->qaux.a, S->src_q.a, S->flatn.a, S->qym.a, S->qyp.a, bxlo, bxhi, dt, G, P);
        ora_trace_ppm(obxlo, obxhi, 2, S->q.a, S->qaux.a, S->src_q.a, S->flatn.a, S->qzm.a, S->qzp.a, bxlo, bxhi, dt, G, P);
    }

    /* node-centred div(u) :430-436 */
    fab_resize(&S->div, obxlo, obxhi, 1);
    ora_divu(obxlo, obxhi, S->q.a, S->div.a, G);

    fab_resize(&S->flux[0], gxlo, gxhi, NUM_STATE); fab_resize(&S->qe[0], gxlo, gxhi, NGDNV);
    fab_resize(&S->flux[1], gylo, gyhi, NUM_STATE); fab_resize(&S->qe[1], gylo, gyhi, NGDNV);
    fab_resize(&S->flux[2], gzlo, gzhi, NUM_STATE); fab_resize(&S->qe[2], gzlo, gzhi, NGDNV);

    /* scratch used by the transverse stages: sized on obx (largest box used) */
    fab_resize(&S->ftmp1, obxlo, obxhi, NUM_STATE);
    fab_resize(&S->ftmp2, obxlo, obxhi, NUM_STATE);
    fab_resize(&S->qgdnvtmp1, obxlo, obxhi, NGDNV);
    fab_resize(&S->qgdnvtmp2, obxlo, obxhi, NGDNV);
    fab_resize(&S->ql, obxlo, obxhi, NQ);
    fab_resize(&S->qr, obxlo, obxhi, NQ);

    const double hdt = 0.5 * dt;                                      /* :682-690 */

    const double hdtdx = 0.5 * dt / dx[0];
    const double hdtdy = 0.5 * dt / dx[1];
    const double hdtdz = 0.5 * dt / dx[2];

    const double cdtdx = dt / dx[0] / 3.0;
    const double cdtdy = dt / dx[1] / 3.0;
    const double cdtdz = dt / dx[2] / 3.0;

    int blo[3], bhi[3];

    /* ---- F^x :694-722 ---- */
    box_grow(xbxlo, xbxhi, 0, 1, 1, blo, bhi);
    ora_cmpflx_plus_godunov(blo, bhi, S->qxm.a, S->qxp.a, S->ftmp1.a, S->qgdnvtmp1.a, S->qaux.a, S->shk.a, 0, G, P);

    /* tyxbx :724-755 */
    box_grow(ybxlo, ybxhi, 0, 0, 1, blo, bhi);
    fab_resize(&S->qmyx, blo, bhi, NQ); fab_resize(&S->qpyx, blo, bhi, NQ);
    ora_trans_single(blo, bhi, 0, 1, S->qym.a, S->qmyx.a, S->qyp.a, S->qpyx.a, S->qaux.a, S->ftmp1.a, S->qgdnvtmp1.a, hdt, cdtdx, P);
    ora_reset_edge_state_thermo(blo, bhi, S->qmyx.a, P);
    ora_reset_edge_state_thermo(blo, bhi, S->qpyx.a, P);

    /* tzxbx :757-783 */
    box_grow(zbxlo, zbxhi, 0, 1, 0, blo, bhi);
    fab_resize(&S->qmzx, blo, bhi, NQ); fab_resize(&S->qpzx, blo, bhi, NQ);
    ora_trans_single(blo, bhi, 0, 2, S->qzm.a, S->qmzx.a, S->qzp.a, S->qpzx.a, S->qaux.a, S->ftmp1.a, S->qgdnvtmp1.a, hdt, cdtdx, P);
    ora_reset_edge_state_thermo(blo, bhi, S->qmzx.a, P);
    ora_reset_edge_state_thermo(blo, bhi, S->qpzx.a, P);

    /* ---- F^y :785-804 ---- */
    box_grow(ybxlo, ybxhi, 1, 0, 1, blo, bhi);
    ora_cmpflx_plus_godunov(blo, bhi, S->qym.a, S->qyp.a, S->ftmp1.a, S->qgdnvtmp1.a, S->qaux.a, S->shk.a, 1, G, P);

    /* txybx :806-835 */
    box_grow(xbxlo, xbxhi, 0, 0, 1, blo, bhi);
    fab_resize(&S->qmxy, blo, bhi, NQ); fab_resize(&S->qpxy, blo, bhi, NQ);
    ora_trans_single(blo, bhi, 1, 0, S->qxm.a, S->qmxy.a, S->qxp.a, S->qpxy.a, S->qaux.a, S->ftmp1.a, S->qgdnvtmp1.a, hdt, cdtdy, P);
    ora_reset_edge_state_thermo(blo, bhi, S->qmxy.a, P);
    ora_reset_edge_state_thermo(blo, bhi, S->qpxy.a, P);

    /* tzybx :837-866 */
    box_grow(zbxlo, zbxhi, 1, 0, 0, blo, bhi);
    fab_resize(&S->qmzy, blo, bhi, NQ); fab_resize(&S->qpzy, blo, bhi, NQ);
    ora_trans_single(blo, bhi, 1, 2, S->qzm.a, S->qmzy.a, S->qzp.a, S->qpzy.a, S->qaux.a, S->ftmp1.a, S->qgdnvtmp1.a, hdt, cdtdy, P);
    ora_reset_edge_state_thermo(blo, bhi, S->qmzy.a, P);
    ora_reset_edge_state_thermo(blo, bhi, S->qpzy.a, P);

    /* ---- F^z :868-883 ---- */
    box_grow(zbxlo, zbxhi, 1, 1, 0, blo, bhi);
    ora_cmpflx_plus_godunov(blo, bhi, S->qzm.a, S->qzp.a, S->ftmp1.a, S->qgdnvtmp1.a, S->qaux.a, S->shk.a, 2, G, P);

    /* txzbx :885-914 */
    box_grow(xbxlo, xbxhi, 0, 1, 0, blo, bhi);
    fab_resize(&S->qmxz, blo, bhi, NQ); fab_resize(&S->qpxz, blo, bhi, NQ);
    ora_trans_single(blo, bhi, 2, 0, S->qxm.a, S->qmxz.a, S->qxp.a, S->qpxz.a, S->qaux.a, S->ftmp1.a, S->qgdnvtmp1.a, hdt, cdtdz, P);
    ora_reset_edge_state_thermo(blo, bhi, S->qmxz.a, P);
    ora_reset_edge_state_thermo(blo, bhi, S->qpxz.a, P);

    /* tyzbx :916-945 */
    box_grow(ybxlo, ybxhi, 1, 0, 0, blo, bhi);
    fab_resize(&S->qmyz, blo, bhi, NQ); fab_resize(&S->qpyz, blo, bhi, NQ);
    ora_trans_single(blo, bhi, 2, 1, S->qym.a, S->qmyz.a, S->qyp.a, S->qpyz.a, S->qaux.a, S->ftmp1.a, S->qgdnvtmp1.a, hdt, cdtdz, P);
    ora_reset_edge_state_thermo(blo, bhi, S->qmyz.a, P);
    ora_reset_edge_state_thermo(blo, bhi, S->qpyz.a, P);

    /* ---- final x flux :949-1026 ---- */
    box_grow(ybxlo, ybxhi, 1, 0, 0, blo, bhi);   /* cyzbx: F^{y|z} */
    ora_cmpflx_plus_godunov(blo, bhi, S->qmyz.a, S->qpyz.a, S->ftmp1.a, S->qgdnvtmp1.a, S->qaux.a, S->shk.a, 1, G, P);
    box_grow(zbxlo, zbxhi, 1, 0, 0, blo, bhi);   /* czybx: F^{z|y} */
    ora_cmpflx_plus_godunov(blo, bhi, S->qmzy.a, S->qpzy.a, S->ftmp2.a, S->qgdnvtmp2.a, S->qaux.a, S->shk.a, 2, G, P);

    ora_trans_final(xbxlo, xbxhi, 0, 1, 2, S->qxm.a, S->ql.a, S->qxp.a, S->qr.a, S->qaux.a,
                    S->ftmp1.a, S->ftmp2.a, S->qgdnvtmp1.a, S->qgdnvtmp2.a, hdtdy, hdtdz, P);
    ora_reset_edge_state_thermo(xbxlo, xbxhi, S->ql.a, P);
    ora_reset_edge_state_thermo(xbxlo, xbxhi, S->qr.a, P);
    ora_cmpflx_plus_godunov(xbxlo, xbxhi, S->ql.a, S->qr.a, S->flux[0].a, S->qe[0].a, S->qaux.a, S->shk.a, 0, G, P);

    /* ---- final y flux :1028-1106 ---- */
    box_grow(zbxlo, zbxhi, 0, 1, 0, blo, bhi);   /* czxbx: F^{z|x} */
    ora_cmpflx_plus_godunov(blo, bhi, S->qmzx.a, S->qpzx.a, S->ftmp1.a, S->qgdnvtmp1.a, S->qaux.a, S->shk.a, 2, G, P);
    box_grow(xbxlo, xbxhi, 0, 1, 0, blo, bhi);   /* cxzbx: F^{x|z} */
    ora_cmpflx_plus_godunov(blo, bhi, S->qmxz.a, S->qpxz.a, S->ftmp2.a, S->qgdnvtmp2.a, S->qaux.a, S->shk.a, 0, G, P);

    ora_trans_final(ybxlo, ybxhi, 1, 0, 2, S->qym.a, S->ql.a, S->qyp.a, S->qr.a, S->qaux.a,
                    S->ftmp2.a, S->ftmp1.a, S->qgdnvtmp2.a, S->qgdnvtmp1.a, hdtdx, hdtdz, P);
    ora_reset_edge_state_thermo(ybxlo, ybxhi, S->ql.a, P);
    ora_reset_edge_state_thermo(ybxlo, ybxhi, S->qr.a, P);
    ora_cmpflx_plus_godunov(ybxlo, ybxhi, S->ql.a, S->qr.a, S->flux[1].a, S->qe[1].a, S->qaux.a, S->shk.a, 1, G, P);

    /* ---- final z flux :1108-1186 ---- */
    box_grow(xbxlo, xbxhi, 0, 0, 1, blo, bhi);   /* cxybx: F^{x|y} */
    ora_cmpflx_plus_godunov(blo, bhi, S->qmxy.a, S->qpxy.a, S->ftmp1.a, S->qgdnvtmp1.a, S->qaux.a, S->shk.a, 0, G, P);
    box_grow(ybxlo, ybxhi, 0, 0, 1, blo, bhi);   /* cyxbx: F^{y|x} */
    ora_cmpflx_plus_godunov(blo, bhi, S->qmyx.a, S->qpyx.a, S->ftmp2.a, S->qgdnvtmp2.a, S->qaux.a, S->shk.a, 1, G, P);

    ora_trans_final(zbxlo, zbxhi, 2, 0, 1, S->qzm.a, S->ql.a, S->qzp.a, S->qr.a, S->qaux.a,
                    S->ftmp1.a, S->ftmp2.a, S->qgdnvtmp1.a, S->qgdnvtmp2.a, hdtdx, hdtdy, P);
    ora_reset_edge_state_thermo(zbxlo, zbxhi, S->ql.a, P);
    ora_reset_edge_state_thermo(zbxlo, zbxhi, S->qr.a, P);
    ora_cmpflx_plus_godunov(zbxlo, zbxhi, S->ql.a, S->qr.a, S->flux[2].a, S->qe[2].a, S->qaux.a, S->shk.a, 2, G, P);

    /* ---- clean the fluxes :1192-1243 ---- */
    for (int idir = 0; idir < 3; ++idir) {
        int nlo[3], nhi[3];
        surrounding_nodes(bxlo, bxhi, idir, nlo, nhi);
        ora_a4 f = S->flux[idir].a;

        for (int k = nlo[2]; k <= nhi[2]; ++k)
        for (int j = nlo[1]; j <= nhi[1]; ++j)
        for (int i = nlo[0]; i <= nhi[0]; ++i) A4(f,i,j,k,UTEMP) = 0.e0;

        ora_apply_av(nlo, nhi, idir, S->div.a, Sborder, f, G, P);
        if (P->limit_fluxes_on_small_dens == 1)                      /* :1219-1228 */
            ora_limit_hydro_fluxes_on_small_dens(nlo, nhi, idir, Sborder, S->q.a, f, G, P, dt);
        if (P->limit_fluxes_on_large_vel == 1)                       /* :1230-1239 */
            ora_limit_hydro_fluxes_on_large_vel(nlo, nhi, idir, Sborder, S->q.a, f, G, P, dt);
        ora_normalize_species_fluxes(nlo, nhi, f);
    }

    /* ---- conservative update :1247-1275 ---- */
    ora_consup_hydro(bxlo, bxhi, S_new, S->flux[0].a, S->qe[0].a, S->flux[1].a, S->qe[1].a,
                     S->flux[2].a, S->qe[2].a, dt, G);

    /* ---- scale and store the fluxes :1322-1433 ---- */
    for (int idir = 0; idir < 3; ++idir) {
        int nlo[3], nhi[3];
        surrounding_nodes(bxlo, bxhi, idir, nlo, nhi);
        ora_a4 f = S->flux[idir].a;
        double area = (idir == 0) ? dx[1] * dx[2] : (idir == 1) ? dx[0] * dx[2] : dx[0] * dx[1];

        ora_scale_flux(nlo, nhi, f, area, dt);

        /* mfi.nodaltilebox(idir): shared faces belong to the lower tile unless
         * this tile touches the valid box's high end */
        int tlo[3], thi[3];
        surrounding_nodes(bxlo, bxhi, idir, tlo, thi);
        if (thi[idir] <= vhi[idir]) thi[idir] -= 1;
        (void)vlo;

        if (fluxes && fluxes[idir].p) {
            ora_a4 F = fluxes[idir];
            for (int n = 0; n < NUM_STATE; ++n)
            for (int k = tlo[2]; k <= thi[2]; ++k)
            for (int j = tlo[1]; j <= thi[1]; ++j)
            for (int i = tlo[0]; i <= thi[0]; ++i) A4(F,i,j,k,n) += A4(f,i,j,k,n);
        }
        if (mass_fluxes && mass_fluxes[idir].p) {
            ora_a4 M = mass_fluxes[idir];
            for (int k = tlo[2]; k <= thi[2]; ++k)
            for (int j = tlo[1]; j <= thi[1]; ++j)
            for (int i = tlo[0]; i <= thi[0]; ++i) A4(M,i,j,k,0) = A4(f,i,j,k,URHO);
        }
        if (qe_out && qe_out[idir].p) {
            ora_a4 Q = qe_out[idir];
            ora_a4 qe = S->qe[idir].a;
            for (int n = 0; n < NGDNV; ++n)
            for (int k = tlo[2]; k <= thi[2]; ++k)
            for (int j = tlo[1]; j <= thi[1]; ++j)
            for (int i = tlo[0]; i <= thi[0]; ++i) A4(Q,i,j,k,n) = A4(qe,i,j,k,n);
        }
    }
    return bad;
}

int ora_construct_ctu_hydro_source(const int bxlo[3], const int bxhi[3], ora_a4 Sborder, ora_a4 src,
                                   ora_a4 S_new, ora_a4 flux_out[3], ora_a4 mass_flux_out[3],
                                   ora_a4 qe_out[3], const ora_geom *G, const ora_params *P,
                                   double time, double dt, const int tile[3], int nthreads)
{
    (void)time;
    /* tile decomposition (AMReX MFIter tiling [3P]: ntiles = max(n/ts,1), sizes
     * as equal as possible, the first `nleft` tiles one cell longer) */
    int nt[3], ts[3], nleft[3], n[3];
    for (int d = 0; d < 3; ++d) {
        n[d] = bxhi[d] - bxlo[d] + 1;
        int t = (tile && tile[d] > 0) ? tile[d] : n[d];
        nt[d] = n[d] / t; if (nt[d] < 1) nt[d] = 1;
        ts[d] = n[d] / nt[d];
        nleft[d] = n[d] - nt[d] * ts[d];
    }
    const int ntiles = nt[0] * nt[1] * nt[2];
    int bad = 0;

#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif

#pragma omp parallel num_threads(nthreads) reduction(| : bad)
    {
        scratch_t S;
        memset(&S, 0, sizeof(S));
#pragma omp for schedule(dynamic)
        for (int it = 0; it < ntiles; ++it) {
            int t3[3] = { it % nt[0], (it / nt[0]) % nt[1], it / (nt[0] * nt[1]) };
            int tlo[3], thi[3];
            for (int d = 0; d < 3; ++d) {
                int off = t3[d] * ts[d] + (t3[d] < nleft[d] ? t3[d] : nleft[d]);
                int len = ts[d] + (t3[d] < nleft[d] ? 1 : 0);
                tlo[d] = bxlo[d] + off;
                thi[d] = tlo[d] + len - 1;
            }
            bad |= ctu_tile(tlo, thi, bxlo, bxhi, Sborder, src, S_new, flux_out, mass_flux_out, qe_out, G, P, dt, &S);
        }
        scratch_free(&S);
    }
    return bad;
}

/* One tile of a larger valid box (what one C-ABI call of the HIP path computes): bx = tile,
 * [vlo,vhi] = valid box of the FAB (decides mfi.nodaltilebox). */
int ora_ctu_hydro_tile(const int bxlo[3], const int bxhi[3], const int vlo[3], const int vhi[3],
                       ora_a4 Sborder, ora_a4 src, ora_a4 S_new, ora_a4 flux_out[3], ora_a4 mass_flux_out[3],
                       ora_a4 qe_out[3], const ora_geom *G, const ora_params *P, double dt)
{
    scratch_t S;
    memset(&S, 0, sizeof(S));
    int bad = ctu_tile(bxlo, bxhi, vlo, vhi, Sborder, src, S_new, flux_out, mass_flux_out, qe_out, G, P, dt, &S);
    scratch_free(&S);
    return bad;
}
