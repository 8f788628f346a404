// Sustained dense bf16 MFMA rate of THIS box under its power management: every wave runs independent v_mfma_f32_16x16x32_bf16 chains
// out of registers (no memory, no LDS), 2 waves per SIMD, for long enough that the clocks settle.  The number to hold the step's
// MFMA classes against (MI355X_MICROARCH quotes 2.5 PFLOP/s at 2.4 GHz; the clocks under a chip-wide matrix load are lower).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_peak.hip -o tools/probes/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, long long* clk) {
    bf16x8 a, b, av[8];
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 1e-3f + e); b[e] = (__bf16)(blockIdx.x * 1e-3f - e); }
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 8; ++e) av[i][e] = (__bf16)(threadIdx.x * 1e-3f + e + 0.25f * i);      // distinct chains (identical ones get merged / skewed by hipcc)
    const long long t0 = __builtin_readcyclecounter();
    float r = 0.f;
    if constexpr (SHAPE == 16) {
        f32x4 c[8];
        for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c[i]) : "v"(av[i]), "v"(b));   // in place (the builtin form got a rotating AGPR allocation with copies in the loop)
        }
        for (int i = 0; i < 8; ++i) r += c[i][0] + c[i][3];
    } else {
        f32x16 c[4];
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) c[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c[i]) : "v"(av[i]), "v"(b));
        }
        for (int i = 0; i < 4; ++i) r += c[i][0] + c[i][15];
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
    if (r == 123.456f) out[0] = r;
}
int main() {
    float* out; long long* clk; hipMalloc(&out, 4); hipMalloc(&clk, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int wgs = 512;                                            // 2 workgroups of 4 waves per CU: 2 waves per SIMD
    for (int shape : {16, 32})
    for (int iters : {20000, 20000, 200000, 200000, 1000000}) {
        hipEventRecord(e0, 0);
        if (shape == 16) hipLaunchKernelGGL(mfma_loop<16>, dim3(wgs), dim3(256), 0, 0, out, iters, clk);
        else hipLaunchKernelGGL(mfma_loop<32>, dim3(wgs), dim3(256), 0, 0, out, iters, clk);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
        const double per = shape == 16 ? 8 * 2.0 * 16 * 16 * 32 : 4 * 2.0 * 32 * 32 * 16;          // flops per wave and iteration
        const double flops = per * iters * (double)wgs * 4;
        printf("mfma %dx%d bf16: %8d iters  %8.3f ms  %7.1f TFLOP/s  (%.0f MHz s_memtime-equivalent: %lld ticks)\n", shape, shape, iters, ms, flops / (ms * 1e-3) / 1e12,
               c / (ms * 1e-3) / 1e6, c);
    }
    return 0;
}
