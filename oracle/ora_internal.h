/* ora_internal.h -- shared helpers for the CPU oracle (test infrastructure only). */
#ifndef ORA_INTERNAL_H
#define ORA_INTERNAL_H

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "castro_oracle.h"

/* Array4::operator() */
#define A4(a, i, j, k, n) \
    ((a).p[((long)(i) - (a).lo[0]) + (a).sy * ((long)(j) - (a).lo[1]) + (a).sz * ((long)(k) - (a).lo[2]) + (a).sn * (long)(n)])

/* amrex::min / amrex::max are std::min / std::max: ties and signed zeros
 * resolve to the FIRST argument.  Keep that exactly. */
static inline double amin(double a, double b) { return (b < a) ? b : a; }
static inline double amax(double a, double b) { return (a < b) ? b : a; }
/* Deliberate deviation from the reference, on both sides of every parity test: the reference's reductions
 * (S_new.min(URHO), the CFL minimum) are std::min folds, which DROP a NaN, so a NaN state passes its density and
 * time-step checks silently.  Here a NaN enters the minimum as -1e300 and the step is rejected (negative density /
 * time-step validity), which hands it to the retry logic. */
static inline double ora_nan_guard(double x) { return (x != x) ? -1.e300 : x; }
static inline double amin3(double a, double b, double c) { return amin(amin(a, b), c); }

/* Castro_util.H:24-50 with NumAdv=0, NumSpec=1, NumAux=0 */
static inline int upassmap(int ip) { return UFS + ip; }
static inline int qpassmap(int ip) { return QFS + ip; }

/* reconstruction.H:4-8 */
enum { im2 = 0, im1 = 1, i0 = 2, ip1 = 3, ip2 = 4 };

/* riemann.H:6-10 */
#define RC_SMLP1 1.e-10
#define RC_SMALL 1.e-8
#define RC_SMALLU 1.e-12

/* Castro.H:23-24 */
#define HISTORY_SIZE 40
#define PSTAR_BISECT_FACTOR 5

typedef struct { double rho, p, rhoe, gamc, un, ut, utt; } RiemannState; /* riemann.H:13-31 */
typedef struct { double csmall, cavg, bnd_fac; } RiemannAux;             /* riemann.H:34-39 */

/* scratch FArrayBox */
typedef struct { ora_a4 a; size_t cap; } ora_fab;
static inline void fab_resize(ora_fab *f, const int lo[3], const int hi[3], int nc)
{
    long nx = hi[0] - lo[0] + 1, ny = hi[1] - lo[1] + 1, nz = hi[2] - lo[2] + 1;
    size_t need = (size_t)nx * ny * nz * nc;
    if (need > f->cap) {
        free(f->a.p);
        f->a.p = (double *)malloc(need * sizeof(double));
        f->cap = need;
    }
    for (int d = 0; d < 3; ++d) { f->a.lo[d] = lo[d]; f->a.hi[d] = hi[d]; }
    f->a.nc = nc; f->a.sy = nx; f->a.sz = nx * ny; f->a.sn = nx * ny * nz;
}
static inline void fab_free(ora_fab *f) { free(f->a.p); f->a.p = NULL; f->cap = 0; }

/* riemann internals shared between files */
void ora_riemannus(const RiemannState *ql, const RiemannState *qr, const RiemannAux *raux,
                   RiemannState *qint, const ora_params *P);
void ora_riemanncg(const RiemannState *ql, const RiemannState *qr, const RiemannAux *raux,
                   RiemannState *qint, const ora_params *P);

#endif
