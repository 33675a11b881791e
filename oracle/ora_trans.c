/*
 * ora_trans.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see castro_oracle.h).
 * Restates Source/hydro/trans.cpp (3-D, no radiation): trans_single /
 * actual_trans_single, trans_final / actual_trans_final, and
 * Source/hydro/edge_util.cpp reset_edge_state_thermo.
 */
#include "ora_internal.h"

/* trans.cpp:66-437 (3-D branch) */
static void actual_trans_single(const int lo[3], const int hi[3], int idir_t, int idir_n, int d,
                                ora_a4 q_arr, ora_a4 qo_arr, ora_a4 qaux_arr, ora_a4 flux_t, ora_a4 q_t,
                                double cdtdx, const ora_params *P)
{
    const int reset_density = P->transverse_reset_density;
    const int reset_rhoe = P->transverse_reset_rhoe;
    const double small_p = P->small_pres;

    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        int il = i, jl = j, kl = k, ir = i, jr = j, kr = k;

        if (idir_t == 0) ir = i + 1;
        else if (idir_t == 1) jr = j + 1;
        else kr = k + 1;

        if (idir_n == 0) { il += d; ir += d; }
        else if (idir_n == 1) { jl += d; jr += d; }
        else { kl += d; kr += d; }

        /* passives :171-189 */
        for (int ip = 0; ip < NPASSIVE; ip++) {
            int n = upassmap(ip);
            int nqp = qpassmap(ip);
            double rrnew = A4(q_arr,i,j,k,QRHO) - cdtdx * (A4(flux_t,ir,jr,kr,URHO) - A4(flux_t,il,jl,kl,URHO));
            double compu = A4(q_arr,i,j,k,QRHO) * A4(q_arr,i,j,k,nqp) - cdtdx * (A4(flux_t,ir,jr,kr,n) - A4(flux_t,il,jl,kl,n));
            A4(qo_arr,i,j,k,nqp) = compu / rrnew;
        }

        double pgp = A4(q_t,ir,jr,kr,GDPRES);
        double pgm = A4(q_t,il,jl,kl,GDPRES);
        double ugp = A4(q_t,ir,jr,kr,GDU + idir_t);
        double ugm = A4(q_t,il,jl,kl,GDU + idir_t);

        double dup = pgp * ugp - pgm * ugm;
        double du = ugp - ugm;
        double pav = 0.5 * (pgp + pgm);

        double gamc = A4(qaux_arr,il,jl,kl,QGAMC);

        /* convert to conservation form :268-280 */
        double rrn = A4(q_arr,i,j,k,QRHO);
        double run = rrn * A4(q_arr,i,j,k,QU);
        double rvn = rrn * A4(q_arr,i,j,k,QV);
        double rwn = rrn * A4(q_arr,i,j,k,QW);
        double ekenn = 0.5 * rrn * (A4(q_arr,i,j,k,QU) * A4(q_arr,i,j,k,QU) + A4(q_arr,i,j,k,QV) * A4(q_arr,i,j,k,QV) + A4(q_arr,i,j,k,QW) * A4(q_arr,i,j,k,QW));
        double ren = A4(q_arr,i,j,k,QREINT) + ekenn;

        /* add transverse predictor :332-337 */
        double rrnewn = rrn - cdtdx * (A4(flux_t,ir,jr,kr,URHO) - A4(flux_t,il,jl,kl,URHO));
        double runewn = run - cdtdx * (A4(flux_t,ir,jr,kr,UMX) - A4(flux_t,il,jl,kl,UMX));
        double rvnewn = rvn - cdtdx * (A4(flux_t,ir,jr,kr,UMY) - A4(flux_t,il,jl,kl,UMY));
        double rwnewn = rwn - cdtdx * (A4(flux_t,ir,jr,kr,UMZ) - A4(flux_t,il,jl,kl,UMZ));
        double renewn = ren - cdtdx * (A4(flux_t,ir,jr,kr,UEDEN) - A4(flux_t,il,jl,kl,UEDEN));

        /* :351-366 */
        int reset_state = 0;
        if (reset_density == 1 && rrnewn < 0.0) {
            rrnewn = rrn;
            runewn = run;
            rvnewn = rvn;
            rwnewn = rwn;
            renewn = ren;
            reset_state = 1;
        }

        /* back to primitive :369-378 */
        A4(qo_arr,i,j,k,QRHO) = rrnewn;
        double rhoinv = 1.0 / rrnewn;
        A4(qo_arr,i,j,k,QU) = runewn * rhoinv;
        A4(qo_arr,i,j,k,QV) = rvnewn * rhoinv;
        A4(qo_arr,i,j,k,QW) = rwnewn * rhoinv;

        double rhoekenn = 0.5 * (runewn * runewn + rvnewn * rvnewn + rwnewn * rwnewn) * rhoinv;
        A4(qo_arr,i,j,k,QREINT) = renewn - rhoekenn;

        if (!reset_state) {
            if (reset_rhoe == 1 && A4(qo_arr,i,j,k,QREINT) <= 0.0) {
                A4(qo_arr,i,j,k,QREINT) = A4(q_arr,i,j,k,QREINT) - cdtdx * (A4(flux_t,ir,jr,kr,UEINT) - A4(flux_t,il,jl,kl,UEINT) + pav * du);
            }

            if (A4(qo_arr,i,j,k,QREINT) <= 0.0) {
                A4(qo_arr,i,j,k,QREINT) = A4(q_arr,i,j,k,QREINT);
            }

            double pnewn = A4(q_arr,i,j,k,QPRES) - cdtdx * (dup + pav * du * (gamc - 1.0));
            A4(qo_arr,i,j,k,QPRES) = amax(pnewn, small_p);
        } else {
            A4(qo_arr,i,j,k,QPRES) = A4(q_arr,i,j,k,QPRES);
            A4(qo_arr,i,j,k,QREINT) = A4(q_arr,i,j,k,QREINT);
        }
    }
}

/* trans.cpp:14-63 */
void ora_trans_single(const int lo[3], const int hi[3], int idir_t, int idir_n, ora_a4 qm, ora_a4 qmo,
                      ora_a4 qp, ora_a4 qpo, ora_a4 qaux, ora_a4 flux_t, ora_a4 q_t,
                      double hdt, double cdtdx, const ora_params *P)
{
    (void)hdt;
    actual_trans_single(lo, hi, idir_t, idir_n, -1, qm, qmo, qaux, flux_t, q_t, cdtdx, P);
    actual_trans_single(lo, hi, idir_t, idir_n, 0, qp, qpo, qaux, flux_t, q_t, cdtdx, P);
}

/* trans.cpp:498-862 (no radiation) */
static void actual_trans_final(const int lo[3], const int hi[3], int idir_n, int idir_t1, int idir_t2, int d,
                               ora_a4 q_arr, ora_a4 qo_arr, ora_a4 qaux_arr,
                               ora_a4 flux_t1, ora_a4 flux_t2, ora_a4 q_t1, ora_a4 q_t2,
                               double cdtdx_t1, double cdtdx_t2, const ora_params *P)
{
    const int reset_density = P->transverse_reset_density;
    const int reset_rhoe = P->transverse_reset_rhoe;
    const double small_p = P->small_pres;

    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        int iln = i, jln = j, kln = k;
        int il_t1 = i, jl_t1 = j, kl_t1 = k, ir_t1 = i, jr_t1 = j, kr_t1 = k;
        int il_t2 = i, jl_t2 = j, kl_t2 = k, ir_t2 = i, jr_t2 = j, kr_t2 = k;

        if (idir_n == 0) {
            ir_t1 += d; jr_t1 += 1;
            ir_t2 += d; kr_t2 += 1;
            iln += d; il_t1 += d; il_t2 += d;
        } else if (idir_n == 1) {
            ir_t1 += 1; jr_t1 += d;
            jr_t2 += d; kr_t2 += 1;
            jln += d; jl_t1 += d; jl_t2 += d;
        } else {
            ir_t1 += 1; kr_t1 += d;
            jr_t2 += 1; kr_t2 += d;
            kln += d; kl_t1 += d; kl_t2 += d;
        }

#define F1R(n) A4(flux_t1, ir_t1, jr_t1, kr_t1, n)
#define F1L(n) A4(flux_t1, il_t1, jl_t1, kl_t1, n)
#define F2R(n) A4(flux_t2, ir_t2, jr_t2, kr_t2, n)
#define F2L(n) A4(flux_t2, il_t2, jl_t2, kl_t2, n)

        /* passives :606-627 */
        for (int ip = 0; ip < NPASSIVE; ++ip) {
            int n = upassmap(ip);
            int nqp = qpassmap(ip);

            double rrn = A4(q_arr,i,j,k,QRHO);
            double compn = rrn * A4(q_arr,i,j,k,nqp);
            double rrnewn = rrn - cdtdx_t1 * (F1R(URHO) - F1L(URHO))
                                - cdtdx_t2 * (F2R(URHO) - F2L(URHO));
            double compnn = compn - cdtdx_t1 * (F1R(n) - F1L(n))
                                  - cdtdx_t2 * (F2R(n) - F2L(n));

            A4(qo_arr,i,j,k,nqp) = compnn / rrnewn;
        }

        double pgt1p = A4(q_t1,ir_t1,jr_t1,kr_t1,GDPRES);
        double pgt1m = A4(q_t1,il_t1,jl_t1,kl_t1,GDPRES);
        double ugt1p = A4(q_t1,ir_t1,jr_t1,kr_t1,GDU + idir_t1);
        double ugt1m = A4(q_t1,il_t1,jl_t1,kl_t1,GDU + idir_t1);

        double pgt2p = A4(q_t2,ir_t2,jr_t2,kr_t2,GDPRES);
        double pgt2m = A4(q_t2,il_t2,jl_t2,kl_t2,GDPRES);
        double ugt2p = A4(q_t2,ir_t2,jr_t2,kr_t2,GDU + idir_t2);
        double ugt2m = A4(q_t2,il_t2,jl_t2,kl_t2,GDU + idir_t2);

        double dupt1 = pgt1p * ugt1p - pgt1m * ugt1m;
        double pt1av = 0.5 * (pgt1p + pgt1m);
        double dut1 = ugt1p - ugt1m;
        /* :666 */
        double pt1new = cdtdx_t1 * (dupt1 + pt1av * dut1 * (A4(qaux_arr,iln,jln,kln,QGAMC) - 1.0));

        double dupt2 = pgt2p * ugt2p - pgt2m * ugt2m;
        double pt2av = 0.5 * (pgt2p + pgt2m);
        double dut2 = ugt2p - ugt2m;
        /* :675 */
        double pt2new = cdtdx_t2 * (dupt2 + pt2av * dut2 * (A4(qaux_arr,iln,jln,kln,QGAMC) - 1.0));

        /* convert to conservation form */
        double rrn = A4(q_arr,i,j,k,QRHO);
        double run = rrn * A4(q_arr,i,j,k,QU);
        double rvn = rrn * A4(q_arr,i,j,k,QV);
        double rwn = rrn * A4(q_arr,i,j,k,QW);
        double ekenn = 0.5 * rrn * (A4(q_arr,i,j,k,QU) * A4(q_arr,i,j,k,QU) + A4(q_arr,i,j,k,QV) * A4(q_arr,i,j,k,QV) + A4(q_arr,i,j,k,QW) * A4(q_arr,i,j,k,QW));
        double ren = A4(q_arr,i,j,k,QREINT) + ekenn;

        /* add transverse predictor :739-758 */
        double rrnewn = rrn - cdtdx_t1 * (F1R(URHO) - F1L(URHO))
                            - cdtdx_t2 * (F2R(URHO) - F2L(URHO));
        double runewn = run - cdtdx_t1 * (F1R(UMX) - F1L(UMX))
                            - cdtdx_t2 * (F2R(UMX) - F2L(UMX));
        double rvnewn = rvn - cdtdx_t1 * (F1R(UMY) - F1L(UMY))
                            - cdtdx_t2 * (F2R(UMY) - F2L(UMY));
        double rwnewn = rwn - cdtdx_t1 * (F1R(UMZ) - F1L(UMZ))
                            - cdtdx_t2 * (F2R(UMZ) - F2L(UMZ));
        double renewn = ren - cdtdx_t1 * (F1R(UEDEN) - F1L(UEDEN))
                            - cdtdx_t2 * (F2R(UEDEN) - F2L(UEDEN));

        int reset_state = 0;
        if (reset_density == 1 && rrnewn < 0.0) {
            rrnewn = rrn;
            runewn = run;
            rvnewn = rvn;
            rwnewn = rwn;
            renewn = ren;
            reset_state = 1;
        }

        A4(qo_arr,i,j,k,QRHO) = rrnewn;
        A4(qo_arr,i,j,k,QU) = runewn / rrnewn;
        A4(qo_arr,i,j,k,QV) = rvnewn / rrnewn;
        A4(qo_arr,i,j,k,QW) = rwnewn / rrnewn;

        double rhoekenn = 0.5 * (runewn * runewn + rvnewn * rvnewn + rwnewn * rwnewn) / rrnewn;
        A4(qo_arr,i,j,k,QREINT) = renewn - rhoekenn;

        if (!reset_state) {
            if (reset_rhoe == 1 && A4(qo_arr,i,j,k,QREINT) <= 0.0) {
                A4(qo_arr,i,j,k,QREINT) = A4(q_arr,i,j,k,QREINT)
                    - cdtdx_t1 * (F1R(UEINT) - F1L(UEINT) + pt1av * dut1)
                    - cdtdx_t2 * (F2R(UEINT) - F2L(UEINT) + pt2av * dut2);
            }

            if (A4(qo_arr,i,j,k,QREINT) <= 0.0) {
                A4(qo_arr,i,j,k,QREINT) = A4(q_arr,i,j,k,QREINT);
            }

            double pnewn = A4(q_arr,i,j,k,QPRES) - pt1new - pt2new;
            A4(qo_arr,i,j,k,QPRES) = pnewn;
        } else {
            A4(qo_arr,i,j,k,QPRES) = A4(q_arr,i,j,k,QPRES);
            A4(qo_arr,i,j,k,QREINT) = A4(q_arr,i,j,k,QREINT);
        }

        A4(qo_arr,i,j,k,QPRES) = amax(A4(qo_arr,i,j,k,QPRES), small_p);
#undef F1R
#undef F1L
#undef F2R
#undef F2L
    }
}

/* trans.cpp:441-494 */
void ora_trans_final(const int lo[3], const int hi[3], int idir_n, int idir_t1, int idir_t2,
                     ora_a4 qm, ora_a4 qmo, ora_a4 qp, ora_a4 qpo, ora_a4 qaux,
                     ora_a4 flux_t1, ora_a4 flux_t2, ora_a4 q_t1, ora_a4 q_t2,
                     double cdtdx_t1, double cdtdx_t2, const ora_params *P)
{
    actual_trans_final(lo, hi, idir_n, idir_t1, idir_t2, -1, qm, qmo, qaux, flux_t1, flux_t2, q_t1, q_t2, cdtdx_t1, cdtdx_t2, P);
    actual_trans_final(lo, hi, idir_n, idir_t1, idir_t2, 0, qp, qpo, qaux, flux_t1, flux_t2, q_t1, q_t2, cdtdx_t1, cdtdx_t2, P);
}

/* edge_util.cpp:6-76 */
void ora_reset_edge_state_thermo(const int lo[3], const int hi[3], ora_a4 qedge, const ora_params *P)
{
    if (P->transverse_reset_rhoe != 1 && P->transverse_use_eos != 1) return; /* no-op with defaults */

    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        ora_eos_t es;
        if (P->transverse_reset_rhoe == 1) {
            if (A4(qedge,i,j,k,QREINT) < 0.0) {
                es.rho = A4(qedge,i,j,k,QRHO);
                es.T = P->small_temp;
                es.xn = A4(qedge,i,j,k,QFS);           /* edge_util.cpp:34 */
                ora_eos_rt(P, &es);
                A4(qedge,i,j,k,QREINT) = A4(qedge,i,j,k,QRHO) * es.e;
                A4(qedge,i,j,k,QPRES) = es.p;
            }
        }
        if (P->transverse_use_eos == 1) {
            es.rho = A4(qedge,i,j,k,QRHO);
            es.e = A4(qedge,i,j,k,QREINT) / A4(qedge,i,j,k,QRHO);
            es.T = P->small_temp;
            es.xn = A4(qedge,i,j,k,QFS);               /* edge_util.cpp:54 */
            ora_eos_re(P, &es);
            A4(qedge,i,j,k,QREINT) = es.e * es.rho;
            A4(qedge,i,j,k,QPRES) = amax(es.p, P->small_pres);
        }
    }
}
