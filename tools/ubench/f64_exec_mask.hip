// Micro-benchmark: does a wave64 fp64 VALU instruction cost less when part of EXEC is off?  (CDNA executes a wave64
// instruction as four passes of 16 lanes; if passes with no active lane were skipped, a 1e5-column spectrum could be cut
// into 2 048 waves of 48-49 columns -- one round of the chip -- instead of 1 563 full waves in 1.5 rounds.)
// Lanes [0, L) of every wave run a chain of independent v_fma_f64; L = 64, 48, 32, 16.  Two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/f64_exec_mask.hip -o /tmp/f64_exec_mask && /tmp/f64_exec_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
#define ITERS 400
__global__ __launch_bounds__(256) void k(double *out, int L, double seed)
{
    double a0 = seed + threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double c = 1.0000001, d = 1e-9;
    if ((int)(threadIdx.x & 63) < L) {
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int r = 0; r < REP / 8; ++r)
                asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                             "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main()
{
    const int blocks = 256 * 2, threads = 256;
    double *out;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int Ls[] = {64, 49, 48, 33, 32, 17, 16, 1};
    for (int rep = 0; rep < 2; ++rep)
        for (int L : Ls) {
            for (int w = 0; w < 20; ++w) k<<<blocks, threads>>>(out, L, 1.5);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int w = 0; w < 10; ++w) k<<<blocks, threads>>>(out, L, 1.5);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("active lanes %2d: %.3f ns per wave-instruction per SIMD (two waves per SIMD)\n", L,
                   ms / 10 * 1e6 / ((double)ITERS * REP) / 2);
        }
    hipFree(out);
    return 0;
}
