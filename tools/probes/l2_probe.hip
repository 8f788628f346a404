// Micro-probe: L2 -> LDS fill rate (global_load_lds_dwordx4) with EVERY CU active and an L2-resident working set that misses the
// 32-KiB vector L1: what one XCD's L2 delivers to its 32 CUs.  private: each CU streams its own window; shared: the CUs of an XCD
// (blockIdx % 8) stream the same window, as the tiles of a GEMM panel do.
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int NW, int G>
__global__ __launch_bounds__(NW * 64) void probe(const char* src, long long* cyc, float* sink, int iters, int steps, int shared, size_t window) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lrow = lane >> 3, ps = lane & 7;
    const int ld = steps * 128;                                          // NW*8 rows of `steps` 128-B pieces
    const char* base = src + (size_t)(shared ? (blockIdx.x & 7) : blockIdx.x) * window + (size_t)(wave * 8 + lrow) * ld + ps * 16;
    int phase = shared ? (blockIdx.x >> 3) * 3 : 0;                      // sharers run a few pieces apart, like tiles of one panel
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < G; ++g) glds16(base + ((it * G + g + phase) % steps) * 128, smem + (wave * G + g) * 1024);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * NW * 64 + threadIdx.x] = ((float*)smem)[threadIdx.x];
}
template <int NW, int G>
void run(const char* name, int blocks, const char* src, int steps, int shared) {
    long long* cyc; float* sink;
    hipMalloc(&cyc, blocks * 8); hipMalloc(&sink, (size_t)blocks * NW * 64 * 4);
    const int iters = 2000;
    const size_t window = (size_t)NW * 8 * steps * 128;
    hipFuncSetAttribute((const void*)probe<NW, G>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<NW, G>), dim3(blocks), dim3(NW * 64), NW * G * 1024, 0, src, cyc, sink, iters, steps, shared, window);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<NW, G>), dim3(blocks), dim3(NW * 64), NW * G * 1024, 0, src, cyc, sink, iters, steps, shared, window);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[256]; hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < blocks; ++i) avg += (double)h[i]; avg /= blocks;
    const double bytes = (double)blocks * iters * NW * G * 1024.0;
    printf("%-34s window %5zu KB x %3d  blocks %3d: %6.1f B/clk/CU (in-kernel cycles)  %6.2f TB/s aggregate (events)\n", name, window >> 10,
           shared ? 8 : blocks, blocks, NW * G * 1024.0 * iters / avg, bytes / (ms * 1e-3) / 1e12);
    hipFree(cyc); hipFree(sink);
}
int main() {
    char* src; hipMalloc(&src, 256 << 20); hipMemset(src, 1, 256 << 20);
    run<8, 8>("private, vL1D-resident", 256, src, 2, 0);          // 16 KB per CU
    run<8, 8>("private, L2-resident", 256, src, 8, 0);            // 64 KB per CU, 2 MB per XCD
    run<8, 8>("private, L2-resident", 128, src, 8, 0);
    run<8, 8>("private, L2-resident", 64, src, 8, 0);
    run<8, 8>("private, L2-resident (96 KB)", 256, src, 12, 0);   // 3 MB per XCD
    run<8, 8>("private, beyond L2", 256, src, 64, 0);             // 512 KB per CU, 16 MB per XCD: MALL
    run<8, 8>("shared by the XCD's 32 CUs", 256, src, 64, 1);     // 512 KB per XCD
    run<8, 8>("shared by the XCD's 32 CUs", 256, src, 256, 1);    // 2 MB per XCD
    run<4, 8>("private, L2-resident, 4 waves", 256, src, 16, 0);
    run<16, 4>("private, L2-resident, 16 waves", 256, src, 4, 0);
    return 0;
}
