// GPU-side floor of back-to-back dependent kernel launches vs block size / LDS size / grid.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_empty(float* p) { if (threadIdx.x == 0 && blockIdx.x == 9999) p[0] = 1; }
__global__ void k_touch(float* p) { extern __shared__ float s[]; s[threadIdx.x] = 1; __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 9999) p[0] = s[3]; }
template <typename F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) f();
    hipEventRecord(a, 0); for (int i = 0; i < 200; ++i) f(); hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 200 * 1e3;
}
int main() {
    float* p; hipMalloc(&p, 1024);
    hipFuncSetAttribute((const void*)k_touch, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    for (int grid : {1, 224, 2048})
        for (int thr : {256, 512, 768}) {
            float e = timeit([&] { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(thr), 0, 0, p); });
            float l0 = timeit([&] { hipLaunchKernelGGL(k_touch, dim3(grid), dim3(thr), 4096, 0, p); });
            float l1 = timeit([&] { hipLaunchKernelGGL(k_touch, dim3(grid), dim3(thr), 159744, 0, p); });
            printf("grid %4d threads %3d: empty %.2f us   4KB-LDS %.2f us   156KB-LDS %.2f us\n", grid, thr, e, l0, l1);
        }
    return 0;
}
