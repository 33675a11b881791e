/* ora_amr.c -- two-level AMR building blocks (TEST INFRASTRUCTURE, see castro_oracle.h).
 *
 * Reference call sites: FillPatch with cell_cons_interp (Source/driver/Castro_setup.cpp:352-364,
 * Castro.cpp:4201-4209), FluxRegCrseInit / FluxRegFineAdd (Castro.cpp:2487-2545), reflux (:2549-2700),
 * avgDown (:3096-3113).  The arithmetic lives in AMReX (CellConservativeLinear, FluxRegister,
 * average_down) [3P, release paired with Castro 21.07], which is NOT in /root/reference: restated from the
 * published algorithm descriptions, parity with an AMReX build is UNPINNED.  Refinement ratio 2, 3-D Cartesian. */
#include <math.h>
#include "castro_oracle.h"
#include "ora_internal.h"

/* monotonized-central limited slope of the coarse data along one direction (per coarse-cell width) */
static inline double mc_slope(double um, double u0, double up)
{
    double dl = u0 - um, dr = up - u0;
    double dc = 0.5 * (up - um);
    if (dl * dr <= 0.0) return 0.0;
    double lim = 2.0 * amin(fabs(dl), fabs(dr));
    return copysign(1.0, dc) * amin(fabs(dc), lim);
}

/* Cell-conservative linear interpolation: every fine zone of [lo,hi] (fine index space) gets
 *   u_c + sx*ox + sy*oy + sz*oz,  o = -1/4 or +1/4 (fine-zone centre relative to the coarse centre, in coarse widths)
 * with MC-limited slopes, scaled by one factor per coarse zone so that the eight sub-zone values stay inside the
 * range of the 27 surrounding coarse values.  The mean over the 8 children is u_c exactly (conservative).
 * `crse` must contain the coarse zones under [lo,hi] grown by one. */
void ora_cc_interp(const int lo[3], const int hi[3], ora_a4 crse, ora_a4 fine, int ncomp)
{
    for (int n = 0; n < ncomp; ++n)
    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        const int ic = (i >= 0) ? i / 2 : -((-i + 1) / 2);
        const int jc = (j >= 0) ? j / 2 : -((-j + 1) / 2);
        const int kc = (k >= 0) ? k / 2 : -((-k + 1) / 2);
        const double u0 = A4(crse,ic,jc,kc,n);
        double sx = mc_slope(A4(crse,ic-1,jc,kc,n), u0, A4(crse,ic+1,jc,kc,n));
        double sy = mc_slope(A4(crse,ic,jc-1,kc,n), u0, A4(crse,ic,jc+1,kc,n));
        double sz = mc_slope(A4(crse,ic,jc,kc-1,n), u0, A4(crse,ic,jc,kc+1,n));
        double umax = u0, umin = u0;
        for (int kk = -1; kk <= 1; ++kk)
        for (int jj = -1; jj <= 1; ++jj)
        for (int ii = -1; ii <= 1; ++ii) {
            double v = A4(crse,ic+ii,jc+jj,kc+kk,n);
            umax = amax(umax, v);
            umin = amin(umin, v);
        }
        const double dmax = 0.25 * (fabs(sx) + fabs(sy) + fabs(sz));      /* largest excursion among the children */
        double alpha = 1.0;
        if (dmax > umax - u0) alpha = amin(alpha, (umax - u0) / dmax);
        if (dmax > u0 - umin) alpha = amin(alpha, (u0 - umin) / dmax);
        const double ox = (i - 2 * ic == 0) ? -0.25 : 0.25;
        const double oy = (j - 2 * jc == 0) ? -0.25 : 0.25;
        const double oz = (k - 2 * kc == 0) ? -0.25 : 0.25;
        A4(fine,i,j,k,n) = u0 + alpha * (sx * ox + sy * oy + sz * oz);
    }
}

/* amrex::average_down (equal volumes): coarse zone of [lo,hi] (coarse index space) = mean of its 8 children */
void ora_avgdown(const int lo[3], const int hi[3], ora_a4 fine, ora_a4 crse, int ncomp)
{
    for (int n = 0; n < ncomp; ++n)
    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        double s = 0.0;
        for (int kk = 0; kk < 2; ++kk)
        for (int jj = 0; jj < 2; ++jj)
        for (int ii = 0; ii < 2; ++ii) s += A4(fine,2*i+ii,2*j+jj,2*k+kk,n);
        A4(crse,i,j,k,n) = 0.125 * s;
    }
}

/* FluxRegister::CrseInit on one face plane: reg = mult * crse_flux on the coarse faces [lo,hi] of direction dir */
void ora_reg_crse_init(const int lo[3], const int hi[3], ora_a4 reg, ora_a4 cflux, int ncomp, double mult)
{
    for (int n = 0; n < ncomp; ++n)
    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) A4(reg,i,j,k,n) = mult * A4(cflux,i,j,k,n);
}

/* FluxRegister::FineAdd: reg += mult * (sum of the 4 fine faces covering each coarse face); fluxes are already
 * scaled by dt * area, so the sum is the time- and area-integrated fine flux */
void ora_reg_fine_add(const int lo[3], const int hi[3], ora_a4 reg, ora_a4 fflux, int dir, int ncomp, double mult)
{
    const int t1 = (dir + 1) % 3, t2 = (dir + 2) % 3;
    for (int n = 0; n < ncomp; ++n)
    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        int c[3] = { i, j, k }, f[3];
        double s = 0.0;
        for (int b = 0; b < 2; ++b)
        for (int a = 0; a < 2; ++a) {
            f[dir] = 2 * c[dir];
            f[t1] = 2 * c[t1] + a;
            f[t2] = 2 * c[t2] + b;
            s += A4(fflux,f[0],f[1],f[2],n);
        }
        A4(reg,i,j,k,n) += mult * s;
    }
}

/* FluxRegister::Reflux on one face plane: the coarse zones on the outside of the faces [lo,hi] get -/+ reg / vol
 * (side 0 = low face of the fine region: zone at index-1, minus; side 1 = high face: zone at the face index, plus) */
void ora_reflux(const int lo[3], const int hi[3], ora_a4 state, ora_a4 reg, int dir, int side, int ncomp, double vol)
{
    for (int n = 0; n < ncomp; ++n)
    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        int c[3] = { i, j, k };
        if (side == 0) c[dir] -= 1;
        const double mult = side == 0 ? -1.0 : 1.0;
        A4(state,c[0],c[1],c[2],n) += mult * A4(reg,i,j,k,n) / vol;
    }
}

/* dst = a*x + b*y on [lo,hi] (StateData time interpolation of the coarse data in FillPatch) */
void ora_lincomb(const int lo[3], const int hi[3], ora_a4 dst, double a, ora_a4 x, double b, ora_a4 y, int ncomp)
{
    for (int n = 0; n < ncomp; ++n)
    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) A4(dst,i,j,k,n) = a * A4(x,i,j,k,n) + b * A4(y,i,j,k,n);
}

/* amrex::AMRErrorTag (the amr.refinement_indicators of Castro::errorEst, Castro.cpp:3131-3164) [3P, restated]:
 * kind 0 value_greater (q >= v), 1 value_less (q <= v), 2 gradient (largest one-sided difference to the six
 * neighbours >= v), 3 relative_gradient (... >= v |q|).  Tags are OR-ed into `tags` (1.0 = tagged). */
void ora_error_tag(const int lo[3], const int hi[3], ora_a4 q, int comp, ora_a4 tags, int kind, double value)
{
    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        const double u = A4(q,i,j,k,comp);
        int tag = 0;
        if (kind == 0) tag = u >= value;
        else if (kind == 1) tag = u <= value;
        else {
            double ax = fabs(A4(q,i+1,j,k,comp) - u); ax = amax(ax, fabs(u - A4(q,i-1,j,k,comp)));
            double ay = fabs(A4(q,i,j+1,k,comp) - u); ay = amax(ay, fabs(u - A4(q,i,j-1,k,comp)));
            double az = fabs(A4(q,i,j,k+1,comp) - u); az = amax(az, fabs(u - A4(q,i,j,k-1,comp)));
            double g = amax(amax(ax, ay), az);
            tag = (kind == 2) ? (g >= value) : (g >= value * fabs(u));
        }
        if (tag) A4(tags,i,j,k,0) = 1.0;
    }
}
