// Developer microbenchmark: do FP64 MFMA and FP64 vector instructions of DIFFERENT waves of a SIMD execute beside each other
// on gfx950, or do they share one pipe?   hipcc --offload-arch=gfx950 -O3 tools/micro/fp64_coexec.hip -o /tmp/fp64_coexec && /tmp/fp64_coexec
// mode 0: every wave runs a loop of independent v_fma_f64; mode 1: every wave a loop of independent v_mfma_f64_16x16x4_f64;
// mode 2: even waves the first, odd waves the second (half the work of each kind per SIMD).  Four waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, int iters)
{
    const int wave = threadIdx.x >> 6;
    const bool mf = MODE == 1 || (MODE == 2 && ((wave + blockIdx.x) & 1));
    double r = 0.0;
    if (!mf) {
        double a0 = threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
        const double z = 0.999999 + 1e-9 * threadIdx.x;
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {       // 64 FMAs per trip
                a0 = fma(a0, z, 1e-9); a1 = fma(a1, z, 1e-9); a2 = fma(a2, z, 1e-9); a3 = fma(a3, z, 1e-9);
                a4 = fma(a4, z, 1e-9); a5 = fma(a5, z, 1e-9); a6 = fma(a6, z, 1e-9); a7 = fma(a7, z, 1e-9);
            }
        }
        r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    } else {
        d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        const double a = 1.0 + 1e-9 * threadIdx.x, b = 1e-6;
        for (int i = 0; i < iters; i++) {       // 4 MFMAs per trip = 4 x 64 cycles = the 64 FMAs x 4 cycles of the other loop
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
        r = c0[0] + c1[1] + c2[2] + c3[3];
    }
    if (r == 12345.678) out[0] = r;
}

template <int MODE>
float run(double *d, int blocks, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    double *d;
    hipMalloc(&d, 8);
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = 256 * wps;         // blocks of 4 waves: wps blocks per CU = wps waves per SIMD
        const float t0 = run<0>(d, blocks, iters), t1 = run<1>(d, blocks, iters), t2 = run<2>(d, blocks, iters);
        const double flop0 = 2.0 * 64 * 64 * iters * 4.0 * blocks, flop1 = 2.0 * 4 * 16 * 16 * 4 * iters * 4.0 * blocks;
        printf("%d wave(s)/SIMD: vector only %.2f ms (%.1f TFLOP/s)   matrix only %.2f ms (%.1f TFLOP/s)   half and half %.2f ms "
               "(independent pipes would give %.2f, one shared pipe %.2f)\n", wps, t0, flop0 / t0 / 1e9, t1, flop1 / t1 / 1e9, t2,
               (t0 > t1 ? t0 : t1) / (wps == 1 ? 1.0 : 1.0), (t0 + t1) / 2);
    }
    return 0;
}
