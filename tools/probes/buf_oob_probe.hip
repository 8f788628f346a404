// What does an out-of-range lane of `buffer_load_dwordx4 ... offen lds` (LDS-DMA through a buffer descriptor) leave in LDS:
// zeros, nothing (stale bytes), or garbage?  The 3x3-convolution loader of gemm16.hip wants zero fill for the padding taps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned* src, unsigned nbytes, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) lds[i] = 0xABABABABu;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    // even lanes in range, odd lanes far out of range
    const int voff = (lane & 1) ? 0x7fff0000 : lane * 16;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 256; i += 64) out[i] = lds[i];
}
int main() {
    unsigned *src, *out; hipMalloc(&src, 4096); hipMalloc(&out, 1024);
    std::vector<unsigned> h(1024); for (int i = 0; i < 1024; ++i) h[i] = 0x1000 + i;
    hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, 4096u, out);
    std::vector<unsigned> o(256); hipMemcpy(o.data(), out, 1024, hipMemcpyDeviceToHost);
    int zeros = 0, stale = 0, data_ok = 0, other = 0;
    for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 4; ++j) {
        const unsigned v = o[lane * 4 + j];
        if (lane & 1) { if (v == 0) ++zeros; else if (v == 0xABABABABu) ++stale; else ++other; }
        else { if (v == 0x1000u + lane * 4 + j) ++data_ok; else ++other; }
    }
    printf("buffer_load lds OOB lanes: zeros %d stale %d other %d (of 128 dwords); in-range lanes correct %d of 128\n", zeros, stale, other, data_ok);
    return 0;
}
