// Micro-probe: LDS-DMA (global_load_lds_dwordx4) issue cost and per-CU throughput vs wave count / address pattern.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
// PATTERN 0: lane-linear 8 rows x 128 B (row stride ld); 1: XOR-swizzled 16-B chunks inside each 128-B row;
//         2: fully contiguous 1 KiB; 3: plain global_load_dwordx4 to VGPR (no LDS), 8 rows x 128 B
template <int NW, int G, int PATTERN>
__global__ __launch_bounds__(NW * 64) void probe(const char* src, long long* cyc, float* sink, int iters, int ld) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lrow = lane >> 3, ps = lane & 7;
    const int key = (lrow >> 1) | ((wave & 1) << 2);
    const int chunk = PATTERN == 1 ? (ps ^ key) : ps;
    const char* base = src + (size_t)blockIdx.x * 65536 + (size_t)(wave * 8 + lrow) * (PATTERN == 2 ? 128 : ld) + chunk * 16;
    float4 accv = make_float4(0, 0, 0, 0);
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const char* p = base + ((it * G + g) & 31) * 128;        // walk along K inside an L2-resident window
            if (PATTERN == 3) { const float4 v = *(const float4*)p; accv.x += v.x; accv.y += v.y; accv.z += v.z; accv.w += v.w; }
            else glds16(p, smem + (wave * G + g) * 1024);
        }
        if (PATTERN != 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * NW * 64 + threadIdx.x] = accv.x + accv.y + accv.z + accv.w + ((float*)smem)[threadIdx.x];
}
template <int NW, int G, int PATTERN>
void run(const char* name, int blocks, const char* src, int ld) {
    long long* cyc; float* sink;
    hipMalloc(&cyc, blocks * 8); hipMalloc(&sink, (size_t)blocks * NW * 64 * 4);
    const int iters = 1000;
    hipFuncSetAttribute((const void*)probe<NW, G, PATTERN>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((probe<NW, G, PATTERN>), dim3(blocks), dim3(NW * 64), NW * G * 1024, 0, src, cyc, sink, iters, ld);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double per_iter = (double)h / iters;
    printf("%-44s NW=%2d G=%d blocks=%3d: %7.1f cyc/iter, %6.1f cyc per glds per wave, %5.1f B/clk/CU\n", name, NW, G, blocks, per_iter,
           per_iter / G, NW * G * 1024.0 / per_iter);
    hipFree(cyc); hipFree(sink);
}
int main() {
    char* src; hipMalloc(&src, 64 << 20); hipMemset(src, 1, 64 << 20);
    run<8, 8, 1>("swizzled rows ld=2560, all CUs", 256, src, 2560);
    run<8, 8, 2>("contiguous 1 KiB pieces, all CUs", 256, src, 2560);
    run<8, 8, 1>("swizzled rows ld=2560, 128 CUs", 128, src, 2560);
    run<8, 8, 2>("contiguous 1 KiB pieces, 128 CUs", 128, src, 2560);
    run<8, 8, 1>("swizzled rows ld=2560, 32 CUs", 32, src, 2560);
    run<8, 8, 1>("swizzled rows ld=2560, 1 CU", 1, src, 2560);
    run<4, 8, 1>("swizzled rows, 4 waves, all CUs", 256, src, 2560);
    run<16, 4, 1>("swizzled rows, 16 waves, all CUs", 256, src, 2560);
    run<8, 8, 3>("plain dwordx4 loads to VGPR, all CUs", 256, src, 2560);
    return 0;
}
