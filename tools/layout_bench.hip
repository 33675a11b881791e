// layout_bench.hip -- does the HBM system care how the component planes of a streaming kernel are laid out?
//   tools/layout_bench  (built by: hipcc -O3 --offload-arch=gfx950 tools/layout_bench.hip -o tools/layout_bench)
// A stand-in for the final-stage kernels: every thread (two x-adjacent zones, 16-byte accesses, rows of 264 zones like the
// 256^3 scratch space) reads NIN values per zone from NIN component arrays and writes NOUT, in two layouts:
//   0  plane-major   a[comp][k][j][i]     (the scratch layout of rounds 1-4: one 147 MB plane per component)
//   1  row-major     a[k][j][comp][i]     (the components of a row adjacent: a wave touches 1-2 DRAM pages instead of NIN)
// Prints the sustained GB/s of both for several (NIN, NOUT).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2), aligned(8)));

template <int NIN, int NOUT, int LAYOUT>
__global__ void __launch_bounds__(256) k_stream(const double* __restrict__ in, double* __restrict__ out, int nx, long nrows)
{
    const long pairs_per_row = nx / 2;
    const long tid = (long)blockIdx.x * 256 + threadIdx.x;
    if (tid >= pairs_per_row * nrows) return;
    const long row = tid / pairs_per_row, p = tid - row * pairs_per_row;
    const long plane = (long)nx * nrows;
    double sa = 0.0, sb = 0.0;
#pragma unroll
    for (int n = 0; n < NIN; ++n) {
        const long idx = LAYOUT == 0 ? n * plane + row * nx + 2 * p : (row * NIN + n) * nx + 2 * p;
        const d2 v = *reinterpret_cast<const d2*>(in + idx);
        sa += v.x * (n + 1); sb += v.y * (n + 2);
    }
#pragma unroll
    for (int n = 0; n < NOUT; ++n) {
        const long idx = LAYOUT == 0 ? n * plane + row * nx + 2 * p : (row * NOUT + n) * nx + 2 * p;
        d2 v; v.x = sa + n; v.y = sb - n;
        __builtin_nontemporal_store(v, reinterpret_cast<d2*>(out + idx));
    }
}

template <int NIN, int NOUT, int LAYOUT>
static double run(const double* in, double* out, int nx, long nrows)
{
    const long threads = (long)(nx / 2) * nrows;
    const unsigned nb = (unsigned)((threads + 255) / 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_stream<NIN, NOUT, LAYOUT>), dim3(nb), dim3(256), 0, 0, in, out, nx, nrows);
    hipEventRecord(e0, 0);
    const int it = 5;
    for (int w = 0; w < it; ++w) hipLaunchKernelGGL((k_stream<NIN, NOUT, LAYOUT>), dim3(nb), dim3(256), 0, 0, in, out, nx, nrows);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)(NIN + NOUT) * nx * nrows * 8.0 / (ms / it * 1e-3) / 1e9;
}

int main()
{
    const int nx = 264;
    const long nrows = 264L * 264L;
    const size_t maxp = 48;
    double *in = nullptr, *out = nullptr;
    if (hipMalloc(&in, maxp * nx * nrows * 8) != hipSuccess || hipMalloc(&out, maxp * nx * nrows * 8) != hipSuccess) { std::puts("alloc failed"); return 1; }
    hipMemset(in, 0, maxp * nx * nrows * 8);
    std::printf("%-28s %12s %12s\n", "planes in + out", "plane-major", "row-major");
#define CASE(NI, NO) std::printf("%2d + %2d %20s %9.0f GB/s %9.0f GB/s\n", NI, NO, "", run<NI, NO, 0>(in, out, nx, nrows), run<NI, NO, 1>(in, out, nx, nrows));
    CASE(1, 1) CASE(8, 8) CASE(16, 8) CASE(37, 18) CASE(48, 9) CASE(14, 42) CASE(48, 42)
    // the compulsory plane mixes of the `contract` kernels of round 6: what a kernel that ONLY moves those planes takes
    std::printf("%-36s %10s %10s\n", "kernel (planes in + out)", "GB/s", "ms");
#define MIX(name, NI, NO) { const double r = run<NI, NO, 0>(in, out, nx, nrows); \
        std::printf("%-36s %10.0f %10.3f\n", name " (" #NI " + " #NO ")", r, (double)(NI + NO) * nx * nrows * 8.0 / r / 1e6); }
    MIX("k_ctoprim", 6, 6) MIX("k_trace_pair", 7, 36) MIX("k_trans1_tile", 36, 36) MIX("k_final<y,z>", 30, 17) MIX("k_finalx_consup", 46, 17)
    return 0;
}
