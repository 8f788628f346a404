// Micro-probe: cost of the GEMM k-loop skeleton (barrier + swizzled ds_read_b128 fragments) per iteration.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NW, int NREAD, bool BARRIER, bool MFMA, int GROUP>
__global__ __launch_bounds__(NW * 64) void probe(float* out, long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 40960; i += NW * 64) ((float*)smem)[i] = (float)i;
    __syncthreads();
    const int l31 = lane & 31, hi = lane >> 5;
    const int row = (wave * 32 + l31) % 256;
    const int key = (row >> 1) & 7;
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (BARRIER) __builtin_amdgcn_s_barrier();
        const char* base = smem + (it % 3) * 53248;
#pragma unroll
        for (int g = 0; g < NREAD / GROUP; ++g) {
            bf16x8 f[GROUP];
#pragma unroll
            for (int r = 0; r < GROUP; ++r) {
                const int idx = g * GROUP + r;
                f[r] = *(const bf16x8*)(base + ((row + 32 * (idx % 5)) % 400) * 128 + ((((idx & 3) * 2 + hi) ^ key) << 4));
            }
            if (MFMA) {
#pragma unroll
                for (int r = 0; r + 1 < GROUP; ++r) acc[r & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[r], f[r + 1], acc[r & 3], 0, 0, 0);
            } else {
#pragma unroll
                for (int r = 0; r < GROUP; ++r) asm volatile("" ::"v"(f[r]));
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    float s = 0; for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * NW * 64 + threadIdx.x] = s;
}

template <int NW, int NREAD, bool BARRIER, bool MFMA, int GROUP>
void run(const char* name, int blocks) {
    float* out; long long* cyc;
    hipMalloc(&out, blocks * NW * 64 * 4); hipMalloc(&cyc, blocks * 8);
    const int iters = 2000;
    hipFuncSetAttribute((const void*)probe<NW, NREAD, BARRIER, MFMA, GROUP>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<NW, NREAD, BARRIER, MFMA, GROUP>), dim3(blocks), dim3(NW * 64), 163840, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[4]; hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-46s blocks=%3d  cycles/iter = %7.1f  (%.2f cycles per ds_read per wave)\n", name, blocks, (double)h[0] / iters, (double)h[0] / iters / NREAD);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<8, 24, false, false, 6>("8 waves, 24 reads, groups of 6, no barrier", 1);
    run<8, 24, true, false, 6>("8 waves, 24 reads, groups of 6, barrier", 1);
    run<8, 24, true, false, 24>("8 waves, 24 reads, one group of 24, barrier", 1);
    run<8, 24, true, false, 12>("8 waves, 24 reads, groups of 12, barrier", 1);
    run<4, 16, true, false, 4>("4 waves, 16 reads, groups of 4, barrier", 1);
    run<4, 16, true, false, 16>("4 waves, 16 reads, one group, barrier", 1);
    run<1, 16, false, false, 16>("1 wave, 16 reads, one group", 1);
    run<1, 16, false, false, 1>("1 wave, 16 reads, dependent singly", 1);
    run<8, 24, true, true, 6>("8 waves, 24 reads + 20 MFMA, barrier", 1);
    run<8, 24, true, true, 6>("8 waves, 24 reads + 20 MFMA, barrier, 256 blk", 256);
    run<8, 24, true, false, 6>("8 waves, 24 reads, barrier, 256 blk", 256);
    return 0;
}
