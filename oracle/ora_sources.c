/* ora_sources.c -- gravity source terms (TEST INFRASTRUCTURE, see castro_oracle.h).
 * Restates Source/gravity/Castro_gravity.cpp:234-614 for gravity.gravity_type = "ConstantGrav"
 * (Source/gravity/Gravity.cpp:860-866: grav = (0, 0, const_grav) everywhere, ghost zones included)
 * and Castro::apply_source_to_state (Source/sources/Castro_sources.cpp:10-19). */
#include <math.h>
#include "castro_oracle.h"
#include "ora_internal.h"

/* Castro::construct_old_gravity_source, Castro_gravity.cpp:234-362: source += gravity source at t^n */
void ora_old_gravity_source(const int lo[3], const int hi[3], ora_a4 uold, ora_a4 source, const double grav[3],
                            int grav_source_type, double dt)
{
    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        double snew[NUM_STATE], src[NSRC];
        for (int n = 0; n < NSRC; ++n) src[n] = 0.0;

        double rho = A4(uold,i,j,k,URHO);
        double rhoInv = 1.0 / rho;
        for (int n = 0; n < NUM_STATE; ++n) snew[n] = A4(uold,i,j,k,n);

        double old_ke = 0.5 * (snew[UMX] * snew[UMX] + snew[UMY] * snew[UMY] + snew[UMZ] * snew[UMZ]) * rhoInv;

        double Sr[3];
        for (int n = 0; n < 3; ++n) {
            Sr[n] = rho * grav[n];
            src[UMX+n] = Sr[n];
            snew[UMX+n] += dt * src[UMX+n];
        }

        double SrE;
        if (grav_source_type == 1 || grav_source_type == 2) {
            SrE = (A4(uold,i,j,k,UMX) * Sr[0] + A4(uold,i,j,k,UMY) * Sr[1] + A4(uold,i,j,k,UMZ) * Sr[2]) * rhoInv;
        } else if (grav_source_type == 3) {
            double new_ke = 0.5 * (snew[UMX] * snew[UMX] + snew[UMY] * snew[UMY] + snew[UMZ] * snew[UMZ]) * rhoInv;
            SrE = new_ke - old_ke;
        } else {
            SrE = (A4(uold,i,j,k,UMX) * Sr[0] + A4(uold,i,j,k,UMY) * Sr[1] + A4(uold,i,j,k,UMZ) * Sr[2]) * rhoInv;
        }
        src[UEDEN] = SrE;
        snew[UEDEN] += dt * SrE;

        for (int n = 0; n < NSRC; ++n) A4(source,i,j,k,n) += src[n];
    }
}

/* Castro::construct_new_gravity_source, Castro_gravity.cpp:384-596: source += corrector at t^{n+1}.
 * mflux[d] = mass_fluxes[d] (dt * area * rho flux) on the faces of [lo,hi]. */
void ora_new_gravity_source(const int lo[3], const int hi[3], ora_a4 uold, ora_a4 unew, ora_a4 source,
                            const ora_a4 mflux[3], const double grav[3], int grav_source_type, double dt,
                            const double dx[3])
{
    const double vol = dx[0] * dx[1] * dx[2];
    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        double src[NSRC];
        for (int n = 0; n < NSRC; ++n) src[n] = 0.0;
        double hdtInv = 0.5 / dt;

        double rhoo = A4(uold,i,j,k,URHO);
        double rhooinv = 1.0 / A4(uold,i,j,k,URHO);
        double rhon = A4(unew,i,j,k,URHO);
        double rhoninv = 1.0 / A4(unew,i,j,k,URHO);

        double snew[NUM_STATE];
        for (int n = 0; n < NUM_STATE; ++n) snew[n] = A4(unew,i,j,k,n);

        double old_ke = 0.5 * (snew[UMX] * snew[UMX] + snew[UMY] * snew[UMY] + snew[UMZ] * snew[UMZ]) * rhoninv;

        double vold[3], Sr_old[3], vnew[3], Sr_new[3];
        for (int n = 0; n < 3; ++n) vold[n] = A4(uold,i,j,k,UMX+n) * rhooinv;
        for (int n = 0; n < 3; ++n) Sr_old[n] = rhoo * grav[n];
        double SrE_old = vold[0] * Sr_old[0] + vold[1] * Sr_old[1] + vold[2] * Sr_old[2];

        for (int n = 0; n < 3; ++n) vnew[n] = snew[UMX+n] * rhoninv;
        for (int n = 0; n < 3; ++n) Sr_new[n] = rhon * grav[n];
        double SrE_new = vnew[0] * Sr_new[0] + vnew[1] * Sr_new[1] + vnew[2] * Sr_new[2];

        double Srcorr[3];
        for (int n = 0; n < 3; ++n) Srcorr[n] = 0.5 * (Sr_new[n] - Sr_old[n]);
        for (int n = 0; n < 3; ++n) {
            src[UMX+n] = Srcorr[n];
            snew[UMX+n] += dt * src[UMX+n];
        }

        double SrEcorr;
        if (grav_source_type == 1) {
            SrEcorr = 0.5 * (SrE_new - SrE_old);
        } else if (grav_source_type == 2) {
            for (int n = 0; n < 3; ++n) vnew[n] = snew[UMX+n] * rhoninv;
            SrE_new = vnew[0] * Sr_new[0] + vnew[1] * Sr_new[1] + vnew[2] * Sr_new[2];
            SrEcorr = 0.5 * (SrE_new - SrE_old);
        } else if (grav_source_type == 3) {
            double new_ke = 0.5 * (snew[UMX] * snew[UMX] + snew[UMY] * snew[UMY] + snew[UMZ] * snew[UMZ]) * rhoninv;
            SrEcorr = new_ke - old_ke;
        } else {
            SrEcorr = - SrE_old;
            /* time-averaged edge-centred gravity; gold == gnew == grav in every zone */
            double g[3];
            for (int n = 0; n < 3; ++n) g[n] = 0.5 * (grav[n] + grav[n]);
            double gxl = 0.5 * (g[0] + 0.5 * (grav[0] + grav[0]));
            double gxr = 0.5 * (g[0] + 0.5 * (grav[0] + grav[0]));
            double gyl = 0.5 * (g[1] + 0.5 * (grav[1] + grav[1]));
            double gyr = 0.5 * (g[1] + 0.5 * (grav[1] + grav[1]));
            double gzl = 0.5 * (g[2] + 0.5 * (grav[2] + grav[2]));
            double gzr = 0.5 * (g[2] + 0.5 * (grav[2] + grav[2]));

            SrEcorr += hdtInv * (A4(mflux[0],i  ,j,k,0) * gxl * dx[0] +
                                 A4(mflux[0],i+1,j,k,0) * gxr * dx[0] +
                                 A4(mflux[1],i,j  ,k,0) * gyl * dx[1] +
                                 A4(mflux[1],i,j+1,k,0) * gyr * dx[1] +
                                 A4(mflux[2],i,j,k  ,0) * gzl * dx[2] +
                                 A4(mflux[2],i,j,k+1,0) * gzr * dx[2]) / vol;
        }
        src[UEDEN] = SrEcorr;
        snew[UEDEN] += dt * SrEcorr;

        for (int n = 0; n < NSRC; ++n) A4(source,i,j,k,n) += src[n];
    }
}

/* MultiFab::Saxpy(dst, a, src, 0, 0, ncomp, 0): dst += a * src */
void ora_saxpy(const int lo[3], const int hi[3], ora_a4 dst, double a, ora_a4 src, int ncomp)
{
    for (int n = 0; n < ncomp; ++n)
    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) A4(dst,i,j,k,n) += a * A4(src,i,j,k,n);
}
