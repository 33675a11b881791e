/* ora_derive.c -- derived plotfile fields (TEST INFRASTRUCTURE, see castro_oracle.h).
 * Restates Source/driver/Derive.cpp for the 3-D Cartesian gamma-law build. */
#include <math.h>
#include "castro_oracle.h"
#include "ora_internal.h"

/* ids shared with include/castro_hydro_amd.h (CASTRO_AMD_DER_*) */
enum { DER_PRESSURE = 0, DER_KINENG, DER_SOUNDSPEED, DER_GAMMA_1, DER_MACHNUMBER, DER_MAGVORT, DER_DIVU,
       DER_EINT_E1, DER_EINT_E2, DER_LOGDEN, DER_SPEC, DER_ABAR, DER_X_VELOCITY, DER_Y_VELOCITY, DER_Z_VELOCITY,
       DER_MAGVEL, DER_RADVEL, DER_MAGMOM, DER_STATEERR_0, DER_STATEERR_1, DER_STATEERR_2, DER_CIRCVEL,
       DER_ANGMOM_X, DER_ANGMOM_Y, DER_ANGMOM_Z, DER_COUNT };

static void zone_eos(ora_a4 dat, int i, int j, int k, const ora_params *P, ora_eos_t *es)
{
    /* the eos_input_re call every EOS-based derive makes, Derive.cpp:37-50 */
    double rhoInv = 1.0 / A4(dat,i,j,k,URHO);
    es->rho = A4(dat,i,j,k,URHO);
    es->T = A4(dat,i,j,k,UTEMP);
    es->e = A4(dat,i,j,k,UEINT) * rhoInv;
    es->xn = A4(dat,i,j,k,UFS) * rhoInv;               /* Derive.cpp:43 */
    ora_eos_re(P, es);
}

/* `state` must hold one ghost zone around [lo,hi] for DER_MAGVORT / DER_DIVU (grow_box_by_one,
 * Castro_setup.cpp:843,854) */
int ora_derive(int which, const int lo[3], const int hi[3], ora_a4 dat, ora_a4 der, const ora_geom *G,
               const ora_params *P, const double center[3])
{
    if (which < 0 || which >= DER_COUNT) return -1;
    const double *dx = G->dx;
    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        ora_eos_t es;
        double v = 0.0;
        switch (which) {
        case DER_PRESSURE:      /* ca_derpres :24-55 */
            zone_eos(dat, i, j, k, P, &es); v = es.p; break;
        case DER_KINENG:        /* ca_derkineng :860-876 */
            v = 0.5 / A4(dat,i,j,k,URHO) * (A4(dat,i,j,k,UMX) * A4(dat,i,j,k,UMX) +
                                            A4(dat,i,j,k,UMY) * A4(dat,i,j,k,UMY) +
                                            A4(dat,i,j,k,UMZ) * A4(dat,i,j,k,UMZ));
            break;
        case DER_SOUNDSPEED:    /* ca_dersoundspeed :180-214 */
            zone_eos(dat, i, j, k, P, &es); v = es.cs; break;
        case DER_GAMMA_1:       /* ca_dergamma1 :216-249 */
            zone_eos(dat, i, j, k, P, &es); v = es.gam1; break;
        case DER_MACHNUMBER:    /* ca_dermachnumber :251-287 */
            zone_eos(dat, i, j, k, P, &es);
            v = sqrt(A4(dat,i,j,k,UMX) * A4(dat,i,j,k,UMX) +
                     A4(dat,i,j,k,UMY) * A4(dat,i,j,k,UMY) +
                     A4(dat,i,j,k,UMZ) * A4(dat,i,j,k,UMZ)) / A4(dat,i,j,k,URHO) / es.cs;
            break;
        case DER_MAGVORT: {     /* ca_dermagvort :929-1019, Cartesian */
            double vx = 0.5 * (A4(dat,i+1,j,k,UMY) / A4(dat,i+1,j,k,URHO) - A4(dat,i-1,j,k,UMY) / A4(dat,i-1,j,k,URHO)) / dx[0];
            double wx = 0.5 * (A4(dat,i+1,j,k,UMZ) / A4(dat,i+1,j,k,URHO) - A4(dat,i-1,j,k,UMZ) / A4(dat,i-1,j,k,URHO)) / dx[0];
            double uy = 0.5 * (A4(dat,i,j+1,k,UMX) / A4(dat,i,j+1,k,URHO) - A4(dat,i,j-1,k,UMX) / A4(dat,i,j-1,k,URHO)) / dx[1];
            double wy = 0.5 * (A4(dat,i,j+1,k,UMZ) / A4(dat,i,j+1,k,URHO) - A4(dat,i,j-1,k,UMZ) / A4(dat,i,j-1,k,URHO)) / dx[1];
            double uz = 0.5 * (A4(dat,i,j,k+1,UMX) / A4(dat,i,j,k+1,URHO) - A4(dat,i,j,k-1,UMX) / A4(dat,i,j,k-1,URHO)) / dx[2];
            double vz = 0.5 * (A4(dat,i,j,k+1,UMY) / A4(dat,i,j,k+1,URHO) - A4(dat,i,j,k-1,UMY) / A4(dat,i,j,k-1,URHO)) / dx[2];
            double v1 = wy - vz, v2 = uz - wx, v3 = vx - uy;
            v = sqrt(v1 * v1 + v2 * v2 + v3 * v3);
            break; }
        case DER_DIVU: {        /* ca_derdivu :1021-1085, Cartesian */
            double uhi = A4(dat,i+1,j,k,UMX) / A4(dat,i+1,j,k,URHO);
            double ulo = A4(dat,i-1,j,k,UMX) / A4(dat,i-1,j,k,URHO);
            double vhi = A4(dat,i,j+1,k,UMY) / A4(dat,i,j+1,k,URHO);
            double vlo = A4(dat,i,j-1,k,UMY) / A4(dat,i,j-1,k,URHO);
            double whi = A4(dat,i,j,k+1,UMZ) / A4(dat,i,j,k+1,URHO);
            double wlo = A4(dat,i,j,k-1,UMZ) / A4(dat,i,j,k-1,URHO);
            v = 0.5 * (uhi - ulo) / dx[0];
            v += 0.5 * (vhi - vlo) / dx[1];
            v += 0.5 * (whi - wlo) / dx[2];
            break; }
        case DER_EINT_E1: {     /* ca_dereint1 :57-77 */
            double rhoInv = 1.0 / A4(dat,i,j,k,URHO);
            double ux = A4(dat,i,j,k,UMX) * rhoInv, uy = A4(dat,i,j,k,UMY) * rhoInv, uz = A4(dat,i,j,k,UMZ) * rhoInv;
            v = A4(dat,i,j,k,UEDEN) * rhoInv - 0.5 * (ux * ux + uy * uy + uz * uz);
            break; }
        case DER_EINT_E2:       /* ca_dereint2 :79-93 */
            v = A4(dat,i,j,k,UEINT) / A4(dat,i,j,k,URHO); break;
        case DER_LOGDEN:        /* ca_derlogden :95-108 */
            v = log10(A4(dat,i,j,k,URHO)); break;
        case DER_SPEC:          /* ca_derspec :891-905 */
            v = A4(dat,i,j,k,UFS) / A4(dat,i,j,k,URHO); break;
        case DER_ABAR: {        /* ca_derabar :907-927, one species of mass number abar */
            double sum = 0.0;
            double xn = A4(dat,i,j,k,UFS) / A4(dat,i,j,k,URHO);
            sum += xn / P->abar;
            v = 1.0 / sum;
            break; }
        case DER_X_VELOCITY: v = A4(dat,i,j,k,UMX) / A4(dat,i,j,k,URHO); break;   /* ca_dervel :516-530 */
        case DER_Y_VELOCITY: v = A4(dat,i,j,k,UMY) / A4(dat,i,j,k,URHO); break;
        case DER_Z_VELOCITY: v = A4(dat,i,j,k,UMZ) / A4(dat,i,j,k,URHO); break;
        case DER_MAGVEL: {      /* ca_dermagvel :532-551 */
            double deninv = 1.0 / A4(dat,i,j,k,URHO);
            v = sqrt((A4(dat,i,j,k,UMX) * A4(dat,i,j,k,UMX) + A4(dat,i,j,k,UMY) * A4(dat,i,j,k,UMY) +
                      A4(dat,i,j,k,UMZ) * A4(dat,i,j,k,UMZ))) * deninv;
            break; }
        case DER_RADVEL: {      /* ca_derradialvel :572-625, not plane-parallel */
            double x = G->problo[0] + ((double)i + 0.5) * dx[0] - center[0];
            double y = G->problo[1] + ((double)j + 0.5) * dx[1] - center[1];
            double z = G->problo[2] + ((double)k + 0.5) * dx[2] - center[2];
            double r = sqrt(x * x + y * y + z * z);
            v = (A4(dat,i,j,k,UMX) * x + A4(dat,i,j,k,UMY) * y + A4(dat,i,j,k,UMZ) * z) / (A4(dat,i,j,k,URHO) * r);
            break; }
        case DER_MAGMOM:        /* ca_dermagmom :692-709 */
            v = sqrt(A4(dat,i,j,k,UMX) * A4(dat,i,j,k,UMX) + A4(dat,i,j,k,UMY) * A4(dat,i,j,k,UMY) +
                     A4(dat,i,j,k,UMZ) * A4(dat,i,j,k,UMZ));
            break;
        case DER_STATEERR_0: v = A4(dat,i,j,k,URHO); break;                        /* ca_derstate :1087-1110 */
        case DER_STATEERR_1: v = A4(dat,i,j,k,UTEMP); break;
        case DER_STATEERR_2: v = A4(dat,i,j,k,UFS) / A4(dat,i,j,k,URHO); break;
        case DER_CIRCVEL: {     /* ca_dercircvel :627-689, not plane-parallel */
            double x = G->problo[0] + ((double)i + 0.5) * dx[0] - center[0];
            double y = G->problo[1] + ((double)j + 0.5) * dx[1] - center[1];
            double z = G->problo[2] + ((double)k + 0.5) * dx[2] - center[2];
            double r = sqrt(x * x + y * y + z * z);
            double vtot2 = (A4(dat,i,j,k,UMX) * A4(dat,i,j,k,UMX) + A4(dat,i,j,k,UMY) * A4(dat,i,j,k,UMY) +
                            A4(dat,i,j,k,UMZ) * A4(dat,i,j,k,UMZ)) / (A4(dat,i,j,k,URHO) * A4(dat,i,j,k,URHO));
            double vr = (A4(dat,i,j,k,UMX) * x + A4(dat,i,j,k,UMY) * y + A4(dat,i,j,k,UMZ) * z) / (A4(dat,i,j,k,URHO) * r);
            v = sqrt(amax(vtot2 - vr * vr, 0.0));
            break; }
        case DER_ANGMOM_X: case DER_ANGMOM_Y: case DER_ANGMOM_Z: {   /* ca_derangmomx/y/z :711-870 */
            double loc[3];
            loc[0] = G->problo[0] + (0.5 + i) * dx[0];
            loc[1] = G->problo[1] + (0.5 + j) * dx[1];
            loc[2] = G->problo[2] + (0.5 + k) * dx[2];
            for (int dir = 0; dir < 3; ++dir) loc[dir] -= center[dir];
            if (which == DER_ANGMOM_X) v = loc[1] * A4(dat,i,j,k,UMZ) - loc[2] * A4(dat,i,j,k,UMY);
            else if (which == DER_ANGMOM_Y) v = loc[2] * A4(dat,i,j,k,UMX) - loc[0] * A4(dat,i,j,k,UMZ);
            else v = loc[0] * A4(dat,i,j,k,UMY) - loc[1] * A4(dat,i,j,k,UMX);
            break; }
        }
        A4(der,i,j,k,0) = v;
    }
    return 0;
}
