// Microbenchmark: is a stencil-style kernel bound by vector-memory instruction issue (TA/TCP) rather
// than by bytes?  Same bytes, same zones; 8-byte loads (one zone per thread) vs 16-byte loads (two
// zones per thread) vs 32-byte (four zones per thread, two x4 loads).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int NX = 264, NY = 264, NZ = 264;
constexpr long NC = (long)NX * NY * NZ;

template <int NP, int NS, int W, int XS = 0>
__global__ void __launch_bounds__(256) k_stencil(const double* __restrict__ in, double* __restrict__ out, int nout, int shift)
{
    // W zones per thread along x; NS = number of y-offset stencil points per plane
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long z0 = t * W + shift;
    if (z0 + W > NC - 8L * NX || z0 < 512) return;
    double acc[W];
#pragma unroll
    for (int w = 0; w < W; ++w) acc[w] = 0.0;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const double* a = in + (long)p * NC + z0 + (XS ? (long)(s - NS / 2) * XS : (long)s * NX);
            if (W == 1) { acc[0] += a[0]; }
            else if (W == 2) { double2 v = *reinterpret_cast<const double2*>(a); acc[0] += v.x; acc[1] += v.y; }
            else { double2 v = *reinterpret_cast<const double2*>(a); double2 u = *reinterpret_cast<const double2*>(a + 2);
                   acc[0] += v.x; acc[1] += v.y; acc[2] += u.x; acc[3] += u.y; }
        }
    }
    for (int p = 0; p < nout; ++p) {
        double* o = out + (long)p * NC + z0;
        if (W == 1) o[0] = acc[0] + p;
        else if (W == 2) *reinterpret_cast<double2*>(o) = make_double2(acc[0] + p, acc[1] + p);
        else { *reinterpret_cast<double2*>(o) = make_double2(acc[0] + p, acc[1] + p);
               *reinterpret_cast<double2*>(o + 2) = make_double2(acc[2] + p, acc[3] + p); }
    }
}

template <int NP, int NS, int W, int XS = 0>
float run(const double* in, double* out, int nout, int shift = 0)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const long nt = NC / W;
    const unsigned nb = (unsigned)((nt + 255) / 256);
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k_stencil<NP, NS, W, XS>), dim3(nb), dim3(256), 0, 0, in, out, nout, shift);
    hipEventRecord(a, 0);
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((k_stencil<NP, NS, W, XS>), dim3(nb), dim3(256), 0, 0, in, out, nout, shift);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}

int main()
{
    double *in, *out;
    constexpr int NPMAX = 24, NOUT = 8;
    CK(hipMalloc(&in, sizeof(double) * NC * NPMAX));
    CK(hipMalloc(&out, sizeof(double) * NC * NOUT));
    CK(hipMemset(in, 0, sizeof(double) * NC * NPMAX));
    printf("zones %ld; planes are %.1f MB\n", NC, NC * 8 / 1e6);
#define ROW(NP, NS) { float t1 = run<NP, NS, 1>(in, out, NOUT), t2 = run<NP, NS, 2>(in, out, NOUT), t4 = run<NP, NS, 4>(in, out, NOUT); \
        double gb = (NP + NOUT) * NC * 8 / 1e9; \
        printf("planes %2d x stencil %d (%3d loads + %d stores / zone, %.2f GB alg): x2 %.3f ms (%.2f TB/s)  x4 %.3f ms (%.2f TB/s)  2*x4 %.3f ms (%.2f TB/s)\n", \
               NP, NS, NP * NS, NOUT, gb, t1, gb / t1, t2, gb / t2, t4, gb / t4); }
    { float a = run<24, 4, 2>(in, out, NOUT, 0), b = run<24, 4, 2>(in, out, NOUT, 1), c = run<24, 4, 4>(in, out, NOUT, 1), d = run<8, 1, 2>(in, out, NOUT, 1);
      printf("24x4 x4 loads: aligned %.3f ms, shifted by one double %.3f ms; 2*x4 shifted %.3f ms; 8x1 x4 shifted %.3f ms\n", a, b, c, d); }
    { float a = run<8, 5, 1, 0>(in, out, NOUT), b = run<8, 5, 1, 1>(in, out, NOUT), c = run<8, 5, 1, 16>(in, out, NOUT), d = run<8, 5, 2, 2>(in, out, NOUT), e = run<8, 5, 1, 64>(in, out, NOUT);
      printf("8 planes x 5-point stencil, 8-B loads: along y %.3f ms; along x (shift 1 zone) %.3f ms; along x (shift 16 zones = 1 line) %.3f ms; shift 64 zones %.3f; x4 loads shift 2 zones %.3f ms\n", a, b, c, e, d); }
    ROW(8, 1) ROW(24, 1) ROW(8, 3) ROW(8, 5) ROW(24, 3) ROW(24, 4) ROW(16, 5)
    hipFree(in); hipFree(out);
    return 0;
}
