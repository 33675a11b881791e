/*
 * ora_state.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see castro_oracle.h).
 * Restates the per-zone state maintenance around the hydro advance:
 *   clean_state            Source/driver/Castro.cpp:4238-4278
 *     do_enforce_minimum_density   Source/hydro/advection_util.cpp:1080-1172
 *     normalize_species            Source/driver/Castro.cpp:2902-2948
 *     reset_internal_energy        Source/driver/Castro.cpp:3353-3414
 *     computeTemp (EOS(re) -> T)   Source/driver/Castro.cpp:3682-3707
 *   estdt_cfl              Source/driver/timestep.cpp:31-140
 *   physical-BC ghost fill AMReX FillPatch/GpuBndryFuncFab semantics [3P],
 *                          SURVEY.md D.2; BC tables Castro_setup.cpp:40-53
 *   Sedov / Sod initial data  Exec/hydro_tests/{Sedov,Sod}/problem_initialize*.H
 */
#include "ora_internal.h"

/* The zone-local sweeps of this file run under OpenMP like the reference's MFIter loops (`#pragma omp parallel`,
 * Castro.cpp:2902-3092, 3575-3775, timestep.cpp:31-140).  Default: one thread (unit tests); the level driver
 * sets its own thread count. */
static int ora_state_threads = 1;
void ora_set_state_threads(int n) { ora_state_threads = n > 0 ? n : 1; }
#define ORA_PFOR _Pragma("omp parallel for num_threads(ora_state_threads) schedule(static)")

/* ------------------------------------------------------------------ */
void ora_clean_state(const int lo[3], const int hi[3], ora_a4 u, const ora_params *P)
{
    const double small_dens = P->small_dens;

    /* enforce_min_density */
    ORA_PFOR
    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        if (A4(u,i,j,k,URHO) < small_dens) {
            for (int ip = 0; ip < NPASSIVE; ip++) {
                int n = upassmap(ip);
                A4(u,i,j,k,n) *= (small_dens / A4(u,i,j,k,URHO));
            }
            ora_eos_t es;
            es.rho = small_dens;
            es.T = P->small_temp;
            es.xn = A4(u,i,j,k,UFS) / small_dens;      /* advection_util.cpp:1127 */
            ora_eos_rt(P, &es);

            A4(u,i,j,k,URHO) = es.rho;
            A4(u,i,j,k,UTEMP) = es.T;

            A4(u,i,j,k,UMX) = 0.0;
            A4(u,i,j,k,UMY) = 0.0;
            A4(u,i,j,k,UMZ) = 0.0;

            A4(u,i,j,k,UEINT) = es.rho * es.e;
            A4(u,i,j,k,UEDEN) = A4(u,i,j,k,UEINT);
        }
    }

    /* enforce_speed_limit (Castro.cpp:3049-3092) */
    if (P->speed_limit > 0.0) {
        ORA_PFOR
        for (int k = lo[2]; k <= hi[2]; ++k)
        for (int j = lo[1]; j <= hi[1]; ++j)
        for (int i = lo[0]; i <= hi[0]; ++i) {
            double rho = A4(u,i,j,k,URHO);
            double rhoInv = 1.0 / rho;
            double vx = A4(u,i,j,k,UMX) * rhoInv;
            double vy = A4(u,i,j,k,UMY) * rhoInv;
            double vz = A4(u,i,j,k,UMZ) * rhoInv;
            double v = sqrt(vx * vx + vy * vy + vz * vz);
            if (v > P->speed_limit) {
                double reduce_factor = P->speed_limit / v;
                A4(u,i,j,k,UMX) *= reduce_factor;
                A4(u,i,j,k,UMY) *= reduce_factor;
                A4(u,i,j,k,UMZ) *= reduce_factor;
                A4(u,i,j,k,UEDEN) -= 0.5 * rhoInv * (rho * vx * rho * vx - A4(u,i,j,k,UMX) * A4(u,i,j,k,UMX) +
                                                     rho * vy * rho * vy - A4(u,i,j,k,UMY) * A4(u,i,j,k,UMY) +
                                                     rho * vz * rho * vz - A4(u,i,j,k,UMZ) * A4(u,i,j,k,UMZ));
            }
        }
    }

    /* normalize_species */
    ORA_PFOR
    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        double rhoX_sum = 0.0;
        for (int n = 0; n < NUMSPEC; ++n) {
            A4(u,i,j,k,UFS+n) = amax(P->small_x * A4(u,i,j,k,URHO), amin(A4(u,i,j,k,URHO), A4(u,i,j,k,UFS+n)));
            rhoX_sum += A4(u,i,j,k,UFS+n);
        }
        double fac = A4(u,i,j,k,URHO) / rhoX_sum;
        for (int n = 0; n < NUMSPEC; ++n) A4(u,i,j,k,UFS+n) *= fac;
    }

    /* computeTemp: reset_internal_energy ... */
    ORA_PFOR
    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        double rhoInv = 1.0 / A4(u,i,j,k,URHO);
        double Up = A4(u,i,j,k,UMX) * rhoInv;
        double Vp = A4(u,i,j,k,UMY) * rhoInv;
        double Wp = A4(u,i,j,k,UMZ) * rhoInv;
        double ke = 0.5 * (Up * Up + Vp * Vp + Wp * Wp);

        ora_eos_t es;
        es.rho = A4(u,i,j,k,URHO);
        es.T = P->small_temp;
        es.xn = A4(u,i,j,k,UFS) * rhoInv;              /* Castro.cpp:3376 */
        ora_eos_rt(P, &es);

        double small_e = es.e;

        A4(u,i,j,k,UEINT) = amax(A4(u,i,j,k,UEINT), A4(u,i,j,k,URHO) * small_e);
        A4(u,i,j,k,UEDEN) = amax(A4(u,i,j,k,UEDEN), A4(u,i,j,k,URHO) * (small_e + ke) + 0.0);

        double rho_eint = A4(u,i,j,k,UEDEN) - A4(u,i,j,k,URHO) * ke - 0.0;

        if (rho_eint > P->dual_energy_eta2 * A4(u,i,j,k,UEDEN)) {
            A4(u,i,j,k,UEINT) = rho_eint;
        }
    }

    /* ... then T from EOS(re) */
    ORA_PFOR
    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        double rhoInv = 1.0 / A4(u,i,j,k,URHO);
        ora_eos_t es;
        es.rho = A4(u,i,j,k,URHO);
        es.T = A4(u,i,j,k,UTEMP);
        es.e = A4(u,i,j,k,UEINT) * rhoInv;
        es.xn = A4(u,i,j,k,UFS) * rhoInv;              /* Castro.cpp:3694 */
        ora_eos_re(P, &es);
        A4(u,i,j,k,UTEMP) = es.T;
    }
}

/* ------------------------------------------------------------------ */
/* guard != 0: the form the checks INSIDE an advance use (a NaN zone enters as -1e300, so that the step is rejected --
 * the deliberate deviation described in ora_internal.h).  guard == 0: Castro::estdt_cfl as the reference has it
 * (timestep.cpp:131-137: a std::min fold, which drops a NaN) -- what estTimeStep returns outside an advance, where
 * nothing could retry. */
static double estdt_cfl_impl(const int lo[3], const int hi[3], ora_a4 u, const ora_geom *G, const ora_params *P, int guard)
{
    double estdt = 1.e200;
    _Pragma("omp parallel for num_threads(ora_state_threads) schedule(static) reduction(min:estdt)")
    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        double rhoInv = 1.0 / A4(u,i,j,k,URHO);

        ora_eos_t es;
        es.rho = A4(u,i,j,k,URHO);
        es.T = A4(u,i,j,k,UTEMP);
        es.e = A4(u,i,j,k,UEINT) * rhoInv;
        es.xn = A4(u,i,j,k,UFS) * rhoInv;              /* timestep.cpp:67 */
        ora_eos_re(P, &es);

        double ux = A4(u,i,j,k,UMX) * rhoInv;
        double uy = A4(u,i,j,k,UMY) * rhoInv;
        double uz = A4(u,i,j,k,UMZ) * rhoInv;

        double c = es.cs;

        double dt1 = G->dx[0] / (c + fabs(ux));
        double dt2 = G->dx[1] / (c + fabs(uy));
        double dt3 = G->dx[2] / (c + fabs(uz));

        double d = amin3(dt1, dt2, dt3);
        if (guard) d = ora_nan_guard(d);
        estdt = amin(estdt, d);
    }
    return estdt;
}

double ora_estdt_cfl(const int lo[3], const int hi[3], ora_a4 u, const ora_geom *G, const ora_params *P)
{
    return estdt_cfl_impl(lo, hi, u, G, P, 0);
}

double ora_estdt_cfl_guarded(const int lo[3], const int hi[3], ora_a4 u, const ora_geom *G, const ora_params *P)
{
    return estdt_cfl_impl(lo, hi, u, G, P, 1);
}

double ora_min_density(const int lo[3], const int hi[3], ora_a4 u)
{
    double m = 1.e300;
    _Pragma("omp parallel for num_threads(ora_state_threads) schedule(static) reduction(min:m)")
    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) m = amin(m, ora_nan_guard(A4(u,i,j,k,URHO)));
    return m;
}

/* ------------------------------------------------------------------ */
/* BC component tables, Castro_setup.cpp:40-53 (+ set_*_vel_bc :78-130):
 * 0 = copy nearest interior (FOEXTRAP; EXT_DIR is converted to FOEXTRAP by
 *     ca_statefill, Castro_bc_fill_nd.cpp:26-39),
 * +1 = REFLECT_EVEN, -1 = REFLECT_ODD, 2 = leave alone (INT_DIR). */
static int bc_kind(int phys, int comp, int dir)
{
    if (phys == BC_INTERIOR) return 2;
    if (phys == BC_INFLOW || phys == BC_OUTFLOW) return 0;
    /* Symmetry / SlipWall / NoSlipWall: norm_vel_bc = REFLECT_ODD, tang_vel_bc and
     * scalar_bc = REFLECT_EVEN for all three wall types in this version's tables */
    if (comp == UMX + dir) return -1;
    return +1;
}

/* Fill every ghost cell of u that lies outside the problem domain.  x first,
 * then y, then z, each sweep over the full extent already filled, so edges and
 * corners inherit from filled neighbours. */
void ora_bc_fill(ora_a4 u, const ora_geom *G)
{
    for (int dir = 0; dir < 3; ++dir) {
        const int dlo = G->domlo[dir], dhi = G->domhi[dir];
        for (int n = 0; n < u.nc; ++n) {
            const int klo = bc_kind(G->lo_bc[dir], n, dir);
            const int khi = bc_kind(G->hi_bc[dir], n, dir);
            ORA_PFOR
            for (int k = u.lo[2]; k <= u.hi[2]; ++k)
            for (int j = u.lo[1]; j <= u.hi[1]; ++j)
            for (int i = u.lo[0]; i <= u.hi[0]; ++i) {
                int idx[3] = {i, j, k};
                int c = idx[dir];
                if (c < dlo && klo != 2) {
                    int s[3] = {i, j, k};
                    if (klo == 0) { s[dir] = dlo; A4(u,i,j,k,n) = A4(u,s[0],s[1],s[2],n); }
                    else { s[dir] = 2 * dlo - c - 1; A4(u,i,j,k,n) = (double)klo * A4(u,s[0],s[1],s[2],n); }
                } else if (c > dhi && khi != 2) {
                    int s[3] = {i, j, k};
                    if (khi == 0) { s[dir] = dhi; A4(u,i,j,k,n) = A4(u,s[0],s[1],s[2],n); }
                    else { s[dir] = 2 * dhi - c + 1; A4(u,i,j,k,n) = (double)khi * A4(u,s[0],s[1],s[2],n); }
                }
            }
        }
    }
}

void ora_fill_interior_copy(ora_a4 dst, ora_a4 src, const int lo[3], const int hi[3])
{
    for (int n = 0; n < src.nc; ++n)
    ORA_PFOR
    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) A4(dst,i,j,k,n) = A4(src,i,j,k,n);
}

/* ------------------------------------------------------------------ */
/* Exec/hydro_tests/Sedov/problem_initialize.H:8-113 and
 * problem_initialize_state_data.H:8-148 (coord_type 0, 3-D) */
void ora_sedov_init(const int lo[3], const int hi[3], ora_a4 state, const ora_geom *G, const ora_params *P,
                    double r_init, double p_ambient, double exp_energy, double dens_ambient, int nsub)
{
    double center[3];
    for (int n = 0; n < 3; ++n) center[n] = 0.5 * (G->problo[n] + G->probhi[n]);

    ora_eos_t es;
    es.rho = dens_ambient;
    es.p = p_ambient;
    es.T = 1.e9;
    es.xn = 1.0;                                       /* problem_initialize.H:25-29 */
    ora_eos_rp(P, &es);
    const double e_ambient = es.e;
    const double temp_ambient = es.T;

    const double vctr = (4.0 / 3.0) * M_PI * r_init * r_init * r_init;
    const double e_exp = exp_energy / vctr / dens_ambient;

    const double *dx = G->dx;
    double ds[3] = { dx[0] / nsub, dx[1] / nsub, dx[2] / nsub };

    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        double xmin = G->problo[0] + dx[0] * (double)i;
        double ymin = G->problo[1] + dx[1] * (double)j;
        double zmin = G->problo[2] + dx[2] * (double)k;

        double vol_pert = 0.0, vol_ambient = 0.0;

        for (int kk = 0; kk <= nsub - 1; ++kk) {
            double zz = zmin + ds[2] * ((double)kk + 0.5);
            for (int jj = 0; jj <= nsub - 1; ++jj) {
                double yy = ymin + ds[1] * ((double)jj + 0.5);
                for (int ii = 0; ii <= nsub - 1; ++ii) {
                    double xx = xmin + ds[0] * ((double)ii + 0.5);

                    double dist = (center[0] - xx) * (center[0] - xx) +
                                  (center[1] - yy) * (center[1] - yy) +
                                  (center[2] - zz) * (center[2] - zz);

                    if (dist <= r_init * r_init) vol_pert = vol_pert + 1.0;
                    else vol_ambient = vol_ambient + 1.0;
                }
            }
        }

        double e_zone = (vol_pert * e_exp + vol_ambient * e_ambient) / (vol_pert + vol_ambient);
        double eint = dens_ambient * e_zone;

        A4(state,i,j,k,URHO) = dens_ambient;
        A4(state,i,j,k,UMX) = 0.e0;
        A4(state,i,j,k,UMY) = 0.e0;
        A4(state,i,j,k,UMZ) = 0.e0;

        A4(state,i,j,k,UEDEN) = eint +
            0.5e0 * (A4(state,i,j,k,UMX) * A4(state,i,j,k,UMX) / A4(state,i,j,k,URHO) +
                     A4(state,i,j,k,UMY) * A4(state,i,j,k,UMY) / A4(state,i,j,k,URHO) +
                     A4(state,i,j,k,UMZ) * A4(state,i,j,k,UMZ) / A4(state,i,j,k,URHO));

        A4(state,i,j,k,UEINT) = eint;
        A4(state,i,j,k,UTEMP) = temp_ambient;
        A4(state,i,j,k,UFS) = A4(state,i,j,k,URHO);
    }
}
