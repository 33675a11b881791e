// Microbenchmark: many concurrent streams (component planes) per wavefront -- plane-major (SoA, stride
// = whole plane) against row-interleaved (component stride = one x-row) layouts of the same data.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int NX = 264, NY = 264, NZ = 264;
constexpr long NC = (long)NX * NY * NZ;
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));

// NP planes read, NOUT planes written, two zones per thread; offsets: element index = i + rs*(row) + ps*plane
template <int NP, int NOUT, int NS>
__global__ void __launch_bounds__(256) k_streams(const double* __restrict__ in, double* __restrict__ out, long ps, long rs, long zs)
{
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const int ip = (int)(t % (NX / 2));
    const long row = t / (NX / 2);
    if (row >= (long)NY * (NZ - 4)) return;
    const long base = 2L * ip + rs * row;
    d2u acc = {0.0, 0.0};
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int s = 0; s < NS; ++s) acc += *reinterpret_cast<const d2u*>(in + base + ps * p + zs * s);
    }
#pragma unroll
    for (int p = 0; p < NOUT; ++p) *reinterpret_cast<d2u*>(out + base + ps * p) = acc + (double)p;
}

template <int NP, int NOUT, int NS>
void run(const double* in, double* out, const char* what, int KP)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const long nt = (long)(NX / 2) * NY * (NZ - 4);
    const unsigned nb = (unsigned)((nt + 255) / 256);
    for (int lay = 0; lay < 2; ++lay) {
        const long ps = lay ? NX : NC, rs = lay ? (long)KP * NX : NX, zs = rs * NY;
        for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k_streams<NP, NOUT, NS>), dim3(nb), dim3(256), 0, 0, in, out, ps, rs, zs);
        hipEventRecord(a, 0);
        for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((k_streams<NP, NOUT, NS>), dim3(nb), dim3(256), 0, 0, in, out, ps, rs, zs);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b); ms /= 5;
        const double gb = (double)(NP + NOUT) * nt * 16 / 1e9;
        printf("%-28s %-16s %7.3f ms  %5.2f TB/s (unique bytes)\n", what, lay ? "row-interleaved" : "plane-major", ms, gb / ms);
    }
}

int main()
{
    constexpr int KP = 64;
    double *in, *out;
    CK(hipMalloc(&in, sizeof(double) * NC * KP));
    CK(hipMalloc(&out, sizeof(double) * NC * KP));
    CK(hipMemset(in, 0, sizeof(double) * NC * KP));
    run<8, 8, 1>(in, out, "8 in, 8 out", KP);
    run<16, 16, 1>(in, out, "16 in, 16 out", KP);
    run<32, 16, 1>(in, out, "32 in, 16 out", KP);
    run<48, 16, 1>(in, out, "48 in, 16 out", KP);
    run<64, 16, 1>(in, out, "64 in, 16 out", KP);
    run<32, 16, 2>(in, out, "32 in x2 (z,z+1), 16 out", KP);
    run<16, 16, 4>(in, out, "16 in x4 z-planes, 16 out", KP);
    run<8, 50, 1>(in, out, "8 in, 50 out (trace-like)", KP);
    run<1, 32, 1>(in, out, "1 in, 32 out", KP);
    run<32, 1, 1>(in, out, "32 in, 1 out", KP);
    return 0;
}
