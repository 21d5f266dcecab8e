// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the fp64 VALU
// instructions the Toon kernels use.  One wave per SIMD x 4 SIMDs x 256 CUs, 8 independent
// chains per lane, timed with s_memtime inside the kernel.  Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/f64_rates.hip -o /tmp/f64_rates && /tmp/f64_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 64
#define ITERS 200
template <int OP>
__global__ __launch_bounds__(256) void k(double* out, long long* cyc, double seed)
{
    double a0 = seed + threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double c = 1.0000001, d = 1e-9;
    int e0 = 1, e1 = 2;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (OP == 0) { asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d)); }
            if (OP == 1) { asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); }
            if (OP == 2) { asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(d)); }
            if (OP == 3) { asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
            if (OP == 4) { asm volatile("v_rndne_f64 %0, %0\n v_rndne_f64 %1, %1\n v_rndne_f64 %2, %2\n v_rndne_f64 %3, %3\n v_rndne_f64 %4, %4\n v_rndne_f64 %5, %5\n v_rndne_f64 %6, %6\n v_rndne_f64 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
            if (OP == 5) { asm volatile("v_ldexp_f64 %0, %0, %8\n v_ldexp_f64 %1, %1, %9\n v_ldexp_f64 %2, %2, %8\n v_ldexp_f64 %3, %3, %9\n v_ldexp_f64 %4, %4, %8\n v_ldexp_f64 %5, %5, %9\n v_ldexp_f64 %6, %6, %8\n v_ldexp_f64 %7, %7, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(e0), "v"(-e0)); }
            if (OP == 6) { int i0, i1, i2, i3; asm volatile("v_cvt_i32_f64 %0, %4\n v_cvt_i32_f64 %1, %5\n v_cvt_i32_f64 %2, %6\n v_cvt_i32_f64 %3, %7\n v_cvt_i32_f64 %0, %8\n v_cvt_i32_f64 %1, %9\n v_cvt_i32_f64 %2, %10\n v_cvt_i32_f64 %3, %11" : "=&v"(i0), "=&v"(i1), "=&v"(i2), "=&v"(i3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7)); e1 += i0 + i1 + i2 + i3; }
            if (OP == 7) { asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n v_rsq_f64 %4, %4\n v_rsq_f64 %5, %5\n v_rsq_f64 %6, %6\n v_rsq_f64 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
            if (OP == 8) { asm volatile("v_max_f64 %0, %0, %8\n v_max_f64 %1, %1, %8\n v_max_f64 %2, %2, %8\n v_max_f64 %3, %3, %8\n v_max_f64 %4, %4, %8\n v_max_f64 %5, %5, %8\n v_max_f64 %6, %6, %8\n v_max_f64 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(d)); }
            if (OP == 9) { asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %4\n v_mov_b64 %4, %5\n v_mov_b64 %5, %6\n v_mov_b64 %6, %7\n v_mov_b64 %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
            if (OP == 10) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc" : "+v"(e0), "+v"(e1) : : "vcc"); }
            if (OP == 11) { asm volatile("v_lshl_add_u32 %0, %0, 1, %1\n v_lshl_add_u32 %1, %1, 1, %0\n v_lshl_add_u32 %0, %0, 1, %1\n v_lshl_add_u32 %1, %1, 1, %0\n v_lshl_add_u32 %0, %0, 1, %1\n v_lshl_add_u32 %1, %1, 1, %0\n v_lshl_add_u32 %0, %0, 1, %1\n v_lshl_add_u32 %1, %1, 1, %0" : "+v"(e0), "+v"(e1)); }
            if (OP == 15) { int f0=e0,f1=e1,f2=e0+1,f3=e1+1; asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %0, %0, %5, vcc\n v_cndmask_b32 %1, %1, %5, vcc\n v_cndmask_b32 %2, %2, %5, vcc\n v_cndmask_b32 %3, %3, %5, vcc" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(e0), "v"(e1) : "vcc"); e0 += f0+f1+f2+f3; }
            if (OP == 16) { int f0=e0,f1=e1,f2=e0+1,f3=e1+1; asm volatile("v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]\n v_cndmask_b32_e64 %2, %2, %4, s[20:21]\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]\n v_cndmask_b32_e64 %0, %0, %5, s[20:21]\n v_cndmask_b32_e64 %1, %1, %5, s[20:21]\n v_cndmask_b32_e64 %2, %2, %5, s[20:21]\n v_cndmask_b32_e64 %3, %3, %5, s[20:21]" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(e0), "v"(e1) : "s20", "s21"); e0 += f0+f1+f2+f3; }
            if (OP == 17) { asm volatile("v_fma_f64 %0, %0, %8, %9\n s_nop 0\n v_fma_f64 %0, %0, %8, %9\n s_nop 0\n v_fma_f64 %0, %0, %8, %9\n s_nop 0\n v_fma_f64 %0, %0, %8, %9\n s_nop 0\n v_fma_f64 %0, %0, %8, %9\n s_nop 0\n v_fma_f64 %0, %0, %8, %9\n s_nop 0\n v_fma_f64 %0, %0, %8, %9\n s_nop 0\n v_fma_f64 %0, %0, %8, %9\n s_nop 0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d)); }
            if (OP == 18) { asm volatile("v_cmp_gt_f64 vcc, %0, %8\n v_cmp_gt_f64 vcc, %1, %8\n v_cmp_gt_f64 vcc, %2, %8\n v_cmp_gt_f64 vcc, %3, %8\n v_cmp_gt_f64 vcc, %4, %8\n v_cmp_gt_f64 vcc, %5, %8\n v_cmp_gt_f64 vcc, %6, %8\n v_cmp_gt_f64 vcc, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(d) : "vcc"); }
            if (OP == 19) { asm volatile("v_readlane_b32 s20, %0, 0\n v_readlane_b32 s21, %0, 1\n v_readlane_b32 s20, %1, 0\n v_readlane_b32 s21, %1, 1\n v_readlane_b32 s20, %0, 2\n v_readlane_b32 s21, %0, 3\n v_readlane_b32 s20, %1, 2\n v_readlane_b32 s21, %1, 3" : "+v"(e0), "+v"(e1) : : "s20", "s21"); }
            if (OP == 12) { asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %0, %0, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d)); }
            if (OP == 13) { asm volatile("v_sqrt_f64 %0, %0\n v_sqrt_f64 %1, %1\n v_sqrt_f64 %2, %2\n v_sqrt_f64 %3, %3\n v_sqrt_f64 %4, %4\n v_sqrt_f64 %5, %5\n v_sqrt_f64 %6, %6\n v_sqrt_f64 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
            if (OP == 14) { asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(c), "v"(d)); }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + e0 + e1;
    if (threadIdx.x % 64 == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
}
template <int OP>
void run(const char* name, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd, threads = 256;
    double* out; long long* cyc;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    hipMalloc(&cyc, sizeof(long long) * blocks * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, threads>>>(out, cyc, 1.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, threads>>>(out, cyc, 1.5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks * 4, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += v; mean /= h.size();
    const double ninst = (double)ITERS * REP;
    // s_memtime ticks at 100 MHz on gfx9 (constant clock): convert with wall time too
    printf("%-18s waves/SIMD %d  memtime ticks/inst %.3f   wall ns/inst(per wave) %.3f  -> cycles@2.4GHz/inst per SIMD %.2f\n", name, waves_per_simd,
           mean / ninst, ms * 1e6 / ninst, ms * 1e6 / ninst * 2.4 / waves_per_simd);
    hipFree(out); hipFree(cyc);
}
int main()
{
    for (int w = 1; w <= 2; ++w) {
        run<0>("v_fma_f64", w); run<14>("v_fma_f64(sgpr)", w); run<12>("v_fma_f64 dep", w); run<1>("v_mul_f64", w); run<2>("v_add_f64", w); run<8>("v_max_f64", w);
        run<3>("v_rcp_f64", w); run<7>("v_rsq_f64", w); run<13>("v_sqrt_f64", w); run<4>("v_rndne_f64", w); run<5>("v_ldexp_f64", w); run<6>("v_cvt_i32_f64", w);
        run<9>("v_mov_b64", w); run<10>("v_cndmask_b32 dep", w); run<15>("v_cndmask vcc ind", w); run<16>("v_cndmask sgpr ind", w); run<11>("v_lshl_add_u32", w);
        run<17>("v_fma dep + s_nop", w); run<18>("v_cmp_gt_f64", w); run<19>("v_readlane_b32", w);
    }
}
