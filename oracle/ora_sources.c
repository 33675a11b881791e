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

/* ------------------------------------------------------------------------------------------------
 * Rotation source terms, state_in_rotating_frame = 1 (Source/rotation/Rotation.H:10-95,
 * Source/rotation/rotation_sources.cpp:9-500, Source/driver/math.H:9-17, Castro_util.H:87-140)
 * ---------------------------------------------------------------------------------------------- */
static void cross_product(const double a[3], const double b[3], double c[3])
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

/* position(): cell centre, with the periodic wrap of Castro_util.H:117-134 */
static void position(int i, int j, int k, const ora_geom *G, double loc[3])
{
    const int idx[3] = { i, j, k };
    for (int d = 0; d < 3; ++d) {
        double offset = G->problo[d] + 0.5 * G->dx[d];
        if (G->lo_bc[d] == BC_INTERIOR && G->hi_bc[d] == BC_INTERIOR) {
            if (idx[d] < G->domlo[d]) offset += G->probhi[d] - G->problo[d];
            if (idx[d] > G->domhi[d]) offset += G->problo[d] - G->probhi[d];
        }
        loc[d] = offset + (double)idx[d] * G->dx[d];
    }
}

/* rotational_acceleration, Rotation.H:24-75 */
static void rotational_acceleration(const ora_rotation *R, const double r[3], const double v[3], int coriolis, double Sr[3])
{
    Sr[0] = 0.0; Sr[1] = 0.0; Sr[2] = 0.0;
    const int c1 = R->include_centrifugal == 1;
    const int c2 = R->include_coriolis == 1 && coriolis;
    double omega_cross_v[3];
    cross_product(R->omega, v, omega_cross_v);
    if (c1) {
        double omega_cross_r[3], omega_cross_omega_cross_r[3];
        cross_product(R->omega, r, omega_cross_r);
        cross_product(R->omega, omega_cross_r, omega_cross_omega_cross_r);
        for (int d = 0; d < 3; ++d) Sr[d] -= omega_cross_omega_cross_r[d];
    }
    if (c2) {
        for (int d = 0; d < 3; ++d) Sr[d] -= 2.0 * omega_cross_v[d];
    }
}

/* rotational_potential, Rotation.H:77-95 */
static double rotational_potential(const ora_rotation *R, const double r[3])
{
    double phi = 0.0;
    if (R->include_centrifugal == 1) {
        double omega_cross_r[3];
        cross_product(R->omega, r, omega_cross_r);
        for (int d = 0; d < 3; ++d) phi -= 0.5 * omega_cross_r[d] * omega_cross_r[d];
    }
    return phi;
}

/* Castro::fill_rotational_potential, Rotation.cpp:6-39: the PhiRot data corrrsrc reads.  Its zone centre is
 * problo + dx * (i + 1/2) - center, NOT position()'s (problo + dx / 2) + i * dx - center (found by the stub probe) */
static double phi_at(const ora_rotation *R, const ora_geom *G, int i, int j, int k)
{
    const int idx[3] = { i, j, k };
    double r[3];
    for (int d = 0; d < 3; ++d) r[d] = G->problo[d] + G->dx[d] * ((double)idx[d] + 0.5) - R->center[d];
    return rotational_potential(R, r);
}

/* Castro::rsrc, rotation_sources.cpp:9-137 */
void ora_old_rotation_source(const int lo[3], const int hi[3], ora_a4 uold, ora_a4 source, const ora_rotation *R,
                             const ora_geom *G, double dt)
{
    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        double Sr[3], src[NSRC], snew[NUM_STATE], loc[3], v[3];
        for (int n = 0; n < NSRC; ++n) src[n] = 0.0;
        position(i, j, k, G, loc);
        for (int d = 0; d < 3; ++d) loc[d] -= R->center[d];
        double rho = A4(uold,i,j,k,URHO);
        double rhoInv = 1.0 / rho;
        for (int n = 0; n < NUM_STATE; ++n) snew[n] = A4(uold,i,j,k,n);
        double old_ke = 0.5 * (snew[UMX] * snew[UMX] + snew[UMY] * snew[UMY] + snew[UMZ] * snew[UMZ]) * rhoInv;
        v[0] = A4(uold,i,j,k,UMX) * rhoInv;
        v[1] = A4(uold,i,j,k,UMY) * rhoInv;
        v[2] = A4(uold,i,j,k,UMZ) * rhoInv;
        rotational_acceleration(R, loc, v, 1, Sr);
        for (int n = 0; n < 3; ++n) Sr[n] = rho * Sr[n];
        src[UMX] = Sr[0]; src[UMY] = Sr[1]; src[UMZ] = Sr[2];
        snew[UMX] += dt * src[UMX];
        snew[UMY] += dt * src[UMY];
        snew[UMZ] += dt * src[UMZ];
        double SrE;
        if (R->rot_source_type == 3) {
            double new_ke = 0.5 * (snew[UMX] * snew[UMX] + snew[UMY] * snew[UMY] + snew[UMZ] * snew[UMZ]) * rhoInv;
            SrE = new_ke - old_ke;
        } else {
            SrE = A4(uold,i,j,k,UMX) * rhoInv * Sr[0] + A4(uold,i,j,k,UMY) * rhoInv * Sr[1] + A4(uold,i,j,k,UMZ) * rhoInv * Sr[2];
        }
        src[UEDEN] += SrE;
        for (int n = 0; n < NSRC; ++n) A4(source,i,j,k,n) += src[n];
    }
}

/* Castro::corrrsrc, rotation_sources.cpp:140-500 */
void ora_new_rotation_source(const int lo[3], const int hi[3], ora_a4 uold, ora_a4 unew, ora_a4 source,
                             const ora_a4 mflux[3], const ora_rotation *R, const ora_geom *G, double dt)
{
    const double vol = G->dx[0] * G->dx[1] * G->dx[2];
    double dt_omega[3], M[3][3];
    for (int l = 0; l < 3; ++l) for (int m = 0; m < 3; ++m) M[l][m] = 0.0;
    if (R->implicit_rotation_update == 1) {
        for (int d = 0; d < 3; ++d) dt_omega[d] = (R->include_coriolis == 1) ? dt * R->omega[d] : 0.0;
        M[0][0] = 1.0 + dt_omega[0] * dt_omega[0];
        M[0][1] = dt_omega[0] * dt_omega[1] + dt_omega[2];
        M[0][2] = dt_omega[0] * dt_omega[2] - dt_omega[1];
        M[1][0] = dt_omega[1] * dt_omega[0] - dt_omega[2];
        M[1][1] = 1.0 + dt_omega[1] * dt_omega[1];
        M[1][2] = dt_omega[1] * dt_omega[2] + dt_omega[0];
        M[2][0] = dt_omega[2] * dt_omega[0] + dt_omega[1];
        M[2][1] = dt_omega[2] * dt_omega[1] - dt_omega[0];
        M[2][2] = 1.0 + dt_omega[2] * dt_omega[2];
        for (int l = 0; l < 3; ++l)
            for (int m = 0; m < 3; ++m)
                M[l][m] /= (1.0 + dt_omega[0] * dt_omega[0] + dt_omega[1] * dt_omega[1] + dt_omega[2] * dt_omega[2]);
    }
    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        double Sr_old[3], Sr_new[3], Srcorr[3], src[NSRC], snew[NUM_STATE], loc[3];
        for (int n = 0; n < NSRC; ++n) src[n] = 0.0;
        position(i, j, k, G, loc);
        for (int d = 0; d < 3; ++d) loc[d] -= R->center[d];
        double rhoo = A4(uold,i,j,k,URHO);
        double rhooinv = 1.0 / A4(uold,i,j,k,URHO);
        double rhon = A4(unew,i,j,k,URHO);
        double rhoninv = 1.0 / A4(unew,i,j,k,URHO);
        for (int n = 0; n < NUM_STATE; ++n) snew[n] = A4(unew,i,j,k,n);
        double old_ke = 0.5 * (snew[UMX] * snew[UMX] + snew[UMY] * snew[UMY] + snew[UMZ] * snew[UMZ]) * rhoninv;
        double vold[3], vnew[3];
        vold[0] = A4(uold,i,j,k,UMX) * rhooinv;
        vold[1] = A4(uold,i,j,k,UMY) * rhooinv;
        vold[2] = A4(uold,i,j,k,UMZ) * rhooinv;
        rotational_acceleration(R, loc, vold, 1, Sr_old);
        for (int n = 0; n < 3; ++n) Sr_old[n] = rhoo * Sr_old[n];
        double SrE_old = vold[0] * Sr_old[0] + vold[1] * Sr_old[1] + vold[2] * Sr_old[2];
        vnew[0] = A4(unew,i,j,k,UMX) * rhoninv;
        vnew[1] = A4(unew,i,j,k,UMY) * rhoninv;
        vnew[2] = A4(unew,i,j,k,UMZ) * rhoninv;
        rotational_acceleration(R, loc, vnew, 1, Sr_new);
        for (int n = 0; n < 3; ++n) Sr_new[n] = rhon * Sr_new[n];
        double SrE_new = vnew[0] * Sr_new[0] + vnew[1] * Sr_new[1] + vnew[2] * Sr_new[2];
        for (int n = 0; n < 3; ++n) Srcorr[n] = 0.5 * (Sr_new[n] - Sr_old[n]);
        if (R->implicit_rotation_update == 1) {
            double acc[3], new_mom_tmp[3], new_mom[3] = { 0.0, 0.0, 0.0 };
            rotational_acceleration(R, loc, vnew, 0, acc);
            for (int n = 0; n < 3; ++n)
                new_mom_tmp[n] = A4(unew,i,j,k,UMX+n) - 0.5 * Sr_old[n] * dt + 0.5 * rhon * acc[n] * dt;
            for (int l = 0; l < 3; ++l)
                for (int m = 0; m < 3; ++m) new_mom[l] += M[l][m] * new_mom_tmp[m];
            for (int n = 0; n < 3; ++n) Srcorr[n] = (new_mom[n] - A4(unew,i,j,k,UMX+n)) / dt;
        }
        src[UMX] = Srcorr[0]; src[UMY] = Srcorr[1]; src[UMZ] = Srcorr[2];
        snew[UMX] += dt * src[UMX];
        snew[UMY] += dt * src[UMY];
        snew[UMZ] += dt * src[UMZ];
        double SrEcorr;
        if (R->rot_source_type == 1) {
            SrEcorr = 0.5 * (SrE_new - SrE_old);
        } else if (R->rot_source_type == 2) {
            double vn[3], acc[3];
            vn[0] = snew[UMX] * rhoninv; vn[1] = snew[UMY] * rhoninv; vn[2] = snew[UMZ] * rhoninv;
            rotational_acceleration(R, loc, vn, 1, acc);
            Sr_new[0] = rhon * acc[0]; Sr_new[1] = rhon * acc[1]; Sr_new[2] = rhon * acc[2];
            double SrE_new2 = vn[0] * Sr_new[0] + vn[1] * Sr_new[1] + vn[2] * Sr_new[2];
            SrEcorr = 0.5 * (SrE_new2 - SrE_old);
        } else if (R->rot_source_type == 3) {
            double new_ke = 0.5 * (snew[UMX] * snew[UMX] + snew[UMY] * snew[UMY] + snew[UMZ] * snew[UMZ]) * rhoninv;
            SrEcorr = new_ke - old_ke;
        } else {
            SrEcorr = - SrE_old;
            /* phi_old == phi_new: the potential of a steady rotation */
            double p0 = phi_at(R, G, i, j, k);
            double phi = 0.5 * (p0 + p0);
            double pxl = phi_at(R, G, i-1, j, k), pxr = phi_at(R, G, i+1, j, k);
            double pyl = phi_at(R, G, i, j-1, k), pyr = phi_at(R, G, i, j+1, k);
            double pzl = phi_at(R, G, i, j, k-1), pzr = phi_at(R, G, i, j, k+1);
            double phixl = 0.5 * (pxl + pxl), phixr = 0.5 * (pxr + pxr);
            double phiyl = 0.5 * (pyl + pyl), phiyr = 0.5 * (pyr + pyr);
            double phizl = 0.5 * (pzl + pzl), phizr = 0.5 * (pzr + pzr);
            SrEcorr = SrEcorr - (0.5 / dt) * ( A4(mflux[0],i  ,j,k,0) * (phi - phixl) -
                                               A4(mflux[0],i+1,j,k,0) * (phi - phixr) +
                                               A4(mflux[1],i,j  ,k,0) * (phi - phiyl) -
                                               A4(mflux[1],i,j+1,k,0) * (phi - phiyr) +
                                               A4(mflux[2],i,j,k  ,0) * (phi - phizl) -
                                               A4(mflux[2],i,j,k+1,0) * (phi - phizr) ) / vol;
        }
        src[UEDEN] = SrEcorr;
        for (int n = 0; n < NSRC; ++n) A4(source,i,j,k,n) += src[n];
    }
}
