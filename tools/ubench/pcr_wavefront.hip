// Micro-benchmark for the mapping BASELINE.json's north_star sketches ("one wavelength bin per wavefront with
// the nlayer tridiagonal ... held in LDS"): how long does the TRIDIAGONAL SOLVE ALONE take in that mapping,
// for the headline problem (1e5 wavelengths x 90 layers x 5 angles)?
//
// One wavefront per wavelength.  Toon's two-stream system has 2*nlayer = 180 unknowns (fluxes.py:212-288,
// setup_tri_diag) and one right-hand side per incident angle (the direct-beam terms), i.e. 5 here.  With the
// layers across the lanes the sequential Thomas elimination of the reference (tri_diag_solve, fluxes.py:291-352)
// is replaced by parallel cyclic reduction: 8 steps (2^8 >= 180), each combining row r with rows r -+ 2^k
// through LDS; 180 rows on 64 lanes = 3 rows per lane.  Everything else the real kernel does per wavelength
// (reading eleven planes, the layer coefficients with their exponentials and square root, the per-angle
// source-function integration, the disk sum) is left OUT, and the coefficients are made up in registers, so
// the time printed is a lower bound for a wavefront-per-wavelength kernel.  It is compared with the whole
// lane-per-wavelength sweep kernel of the library (k_reflected_toa, 0.235-0.242 ms for the same problem).
//
// The solution is checked against a Thomas elimination of the same systems on the host.
// Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/pcr_wavefront.hip -o /tmp/pcr_wavefront && /tmp/pcr_wavefront
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int N = 180, NRHS = 5, ROWS = 192, RPL = ROWS / 64, WAVES = 4;   // rows per lane, waves (systems) per block

__device__ __forceinline__ double frcp(double b)      // as the library's device_math.hpp: v_rcp_f64 + two Newton steps
{
    double y = __builtin_amdgcn_rcp(b);
    double e = fma(-b, y, 1.0);
    y = fma(y, e, y);
    e = fma(-b, y, 1.0);
    return fma(y, e, y);
}

// LDS hand-over between the lanes of one wave: no s_barrier needed (one wave owns its rows), but the compiler
// must not move LDS accesses across the point (a lane never reads its own row, so nothing else orders them)
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// the same made-up, diagonally dominant systems on device and host
__host__ __device__ inline void coeffs(long sys, int r, double &a, double &b, double &c, double (&d)[NRHS])
{
    // a few cheap operations per value (the generator is inside the timed kernel): a triangle wave (continuous,
    // so host and device agree whatever their rounding of the argument)
    const double t = 0.37 * (double)(sys % 1009) + 0.11 * r;
    auto fr = [](double x) { return fabs(2.0 * (x - floor(x)) - 1.0); };
    a = (r == 0) ? 0.0 : -(0.3 + 0.2 * fr(t));
    c = (r == N - 1) ? 0.0 : -(0.25 + 0.2 * fr(1.3 * t));
    b = 1.2 + 0.1 * fr(0.7 * t);
    for (int k = 0; k < NRHS; ++k) d[k] = 0.5 + 0.4 * fr(t * (1.0 + 0.1 * k) + 0.3 * k);
}

__global__ __launch_bounds__(64 * WAVES) void k_pcr(long nsys, double *out)
{
    __shared__ double sa[WAVES][ROWS], sb[WAVES][ROWS], sc[WAVES][ROWS], sd[WAVES][NRHS][ROWS];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long sys = (long)blockIdx.x * WAVES + wv;
    if (sys >= nsys) return;                       // whole wave
    double a[RPL], b[RPL], c[RPL], d[RPL][NRHS];
#pragma unroll
    for (int j = 0; j < RPL; ++j) {
        const int r = lane + 64 * j;
        if (r < N) coeffs(sys, r, a[j], b[j], c[j], d[j]);
        else {                                     // padding rows: identity
            a[j] = c[j] = 0.0; b[j] = 1.0;
            for (int k = 0; k < NRHS; ++k) d[j][k] = 0.0;
        }
    }
    for (int s = 1; s < 256; s <<= 1) {
        // publish this step's rows, then combine with the rows at distance s
#pragma unroll
        for (int j = 0; j < RPL; ++j) {
            const int r = lane + 64 * j;
            sa[wv][r] = a[j]; sb[wv][r] = b[j]; sc[wv][r] = c[j];
#pragma unroll
            for (int k = 0; k < NRHS; ++k) sd[wv][k][r] = d[j][k];
        }
        wave_sync();
#pragma unroll
        for (int j = 0; j < RPL; ++j) {
            const int r = lane + 64 * j, rm = r - s, rp = r + s;
            const bool hm = rm >= 0, hp = rp < ROWS;
            const double bm = hm ? sb[wv][hm ? rm : 0] : 1.0, bp = hp ? sb[wv][hp ? rp : 0] : 1.0;
            const double k1 = hm ? a[j] * frcp(bm) : 0.0, k2 = hp ? c[j] * frcp(bp) : 0.0;
            const double am = hm ? sa[wv][hm ? rm : 0] : 0.0, cm = hm ? sc[wv][hm ? rm : 0] : 0.0;
            const double ap = hp ? sa[wv][hp ? rp : 0] : 0.0, cp = hp ? sc[wv][hp ? rp : 0] : 0.0;
            b[j] = fma(-cm, k1, fma(-ap, k2, b[j]));
            a[j] = -am * k1;
            c[j] = -cp * k2;
#pragma unroll
            for (int k = 0; k < NRHS; ++k) {
                const double dm = hm ? sd[wv][k][hm ? rm : 0] : 0.0, dp = hp ? sd[wv][k][hp ? rp : 0] : 0.0;
                d[j][k] = fma(-dm, k1, fma(-dp, k2, d[j][k]));
            }
        }
        wave_sync();                               // all reads done before the next step's writes
    }
    // x[r] = d[r]/b[r]; keep the result alive as the real kernel would use it (top row per angle + a checksum)
    double sum[NRHS];
#pragma unroll
    for (int k = 0; k < NRHS; ++k) sum[k] = 0.0;
#pragma unroll
    for (int j = 0; j < RPL; ++j) {
        const double ib = frcp(b[j]);
        const int r = lane + 64 * j;
#pragma unroll
        for (int k = 0; k < NRHS; ++k) {
            const double x = d[j][k] * ib;
            if (r < N) sum[k] += x * (1.0 + 0.001 * r);
        }
    }
#pragma unroll
    for (int k = 0; k < NRHS; ++k) {
        double v = sum[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) out[sys * NRHS + k] = v;
    }
}

static void thomas(long sys, double (&res)[NRHS])
{
    std::vector<double> a(N), b(N), c(N), cp(N);
    std::vector<double> d((size_t)N * NRHS), x((size_t)N * NRHS);
    for (int r = 0; r < N; ++r) {
        double dd[NRHS];
        coeffs(sys, r, a[r], b[r], c[r], dd);
        for (int k = 0; k < NRHS; ++k) d[(size_t)r * NRHS + k] = dd[k];
    }
    std::vector<double> bb(b);
    for (int r = 1; r < N; ++r) {
        const double m = a[r] / bb[r - 1];
        bb[r] -= m * c[r - 1];
        for (int k = 0; k < NRHS; ++k) d[(size_t)r * NRHS + k] -= m * d[(size_t)(r - 1) * NRHS + k];
    }
    for (int k = 0; k < NRHS; ++k) {
        x[(size_t)(N - 1) * NRHS + k] = d[(size_t)(N - 1) * NRHS + k] / bb[N - 1];
        for (int r = N - 2; r >= 0; --r)
            x[(size_t)r * NRHS + k] = (d[(size_t)r * NRHS + k] - c[r] * x[(size_t)(r + 1) * NRHS + k]) / bb[r];
        res[k] = 0.0;
        for (int r = 0; r < N; ++r) res[k] += x[(size_t)r * NRHS + k] * (1.0 + 0.001 * r);
    }
}

int main(int argc, char **argv)
{
    const long nsys = argc > 1 ? atol(argv[1]) : 100000;
    double *out;
    hipMalloc(&out, sizeof(double) * nsys * NRHS);
    const dim3 grid((unsigned)((nsys + WAVES - 1) / WAVES)), block(64 * WAVES);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 300; ++i) hipLaunchKernelGGL(k_pcr, grid, block, 0, 0, nsys, out);   // clock ramp
    hipDeviceSynchronize();
    double best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_pcr, grid, block, 0, 0, nsys, out);
        hipEventRecord(e0, 0);
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_pcr, grid, block, 0, 0, nsys, out);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms / 50 < best) best = ms / 50;
    }
    std::vector<double> h((size_t)nsys * NRHS);
    hipMemcpy(h.data(), out, sizeof(double) * h.size(), hipMemcpyDeviceToHost);
    double worst = 0.0;
    for (long sys : {0L, 1L, 777L, nsys / 2, nsys - 1}) {
        double ref[NRHS];
        thomas(sys, ref);
        for (int k = 0; k < NRHS; ++k) {
            worst = fmax(worst, fabs(h[sys * NRHS + k] - ref[k]) / fabs(ref[k]));
            if (getenv("PCR_DEBUG")) printf("sys %ld rhs %d: gpu %.15g host %.15g\n", sys, k, h[sys * NRHS + k], ref[k]);
        }
    }
    printf("{\"systems\": %ld, \"rows\": %d, \"rhs\": %d, \"pcr_only_ms\": %.4f, \"max_rel_diff_vs_thomas\": %.2e, "
           "\"note\": \"tridiagonal solve alone, one wavefront per wavelength, coefficients made up in registers\"}\n",
           nsys, N, NRHS, best, worst);
    return 0;
}
