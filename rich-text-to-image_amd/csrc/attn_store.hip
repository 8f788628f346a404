// Head-averaged attention probabilities for the token-map producer.
//
// The reference returns softmax(QK^T) averaged over heads from every attention module
// (models/attention_processor.py:166-171,543) and its token-map hooks accumulate the conditional half
// over the sampling steps on the CPU (models/region_diffusion.py:403-426, models/region_diffusion_sdxl.py:965-992).
// The fused attention kernel never materialises P, so for the few layers whose maps are consumed
// (32x32 self-attention maps and the listed cross-attention maps, utils/attention_utils.py:12-67,243-248) this
// kernel recomputes the scores of ONE stream with MFMA in two passes (row max / row sum, then normalised
// probabilities), averages over heads in an LDS tile and adds the tile to an fp32 accumulator in HBM.
// One wavefront owns 32 query rows: no atomics, deterministic.
#include "common.h"
#include <math.h>

template <int DP>
__global__ __launch_bounds__(64) void attn_store_kernel(AttnStoreArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* acc = (float*)smem;                       // [32][ldt]
    const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
    const int ldt = p.NKpad + 1;
    const int q0 = blockIdx.x * 32;
    for (int i = lane; i < 32 * ldt; i += 64) acc[i] = 0.f;
    __syncthreads();
    const int q = q0 + l31 < p.N ? q0 + l31 : p.N - 1;
    const float invH = 1.f / (float)p.H;
    for (int h = 0; h < p.H; ++h) {
        const bf16_t* qptr = p.Q + ((size_t)p.q_row0 + q) * p.ldq + h * DP + hi * 8;
        bf16x8 qf[DP / 16];
#pragma unroll
        for (int ks = 0; ks < DP / 16; ++ks) qf[ks] = *(const bf16x8*)(qptr + ks * 16);
        const bf16_t* kbase = p.K + (size_t)p.k_row0 * p.ldk + h * DP + hi * 8;
        float m = -1e30f, l = 0.f;
        for (int pass = 0; pass < 2; ++pass) {
            float inv = 0.f;
            if (pass == 1) { l += __shfl_xor(l, 32); inv = invH / l; }
            for (int kt = 0; kt < p.NKpad / 32; ++kt) {
                f32x16 s;
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = 0.f;
                const int krow = kt * 32 + l31;
                const bf16_t* kp = kbase + (size_t)(krow < p.NKrows ? krow : p.NKrows - 1) * p.ldk;
#pragma unroll
                for (int ks = 0; ks < DP / 16; ++ks) {
                    const bf16x8 kf = *(const bf16x8*)(kp + ks * 16);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
                }
                // lane holds keys kt*32 + (r&3) + 8*(r>>2) + 4*hi of query l31
                if (pass == 0) {
                    float mx = m;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        if (key >= p.NK) s[r] = -INFINITY;
                        mx = fmaxf(mx, s[r]);
                    }
                    mx = fmaxf(mx, __shfl_xor(mx, 32));
                    float rs = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) rs += exp2f(s[r] - mx);
                    l = l * exp2f(m - mx) + rs;
                    m = mx;
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        if (key < p.NK) acc[l31 * ldt + key] += exp2f(s[r] - m) * inv;
                    }
                }
            }
        }
    }
    __syncthreads();
    // tile -> accumulator (row-contiguous)
    for (int i = lane; i < 32 * p.NK; i += 64) {
        const int r = i / p.NK, c = i - r * p.NK;
        if (q0 + r < p.N) {
            float* dst = p.out + (size_t)(q0 + r) * p.NK + c;
            *dst = p.overwrite ? acc[r * ldt + c] : *dst + acc[r * ldt + c];
        }
    }
}

void launch_attn_store(const AttnStoreArgs& a, hipStream_t st) {
    RT_REQUIRE(a.NKpad % 32 == 0 && a.NKpad <= 1024 && a.NK <= a.NKpad, "attn_store: at most 1024 keys (32x32 maps)");
    const size_t lds = (size_t)32 * (a.NKpad + 1) * 4;
    dim3 grid(cdiv(a.N, 32)), block(64);
#define LAUNCH(D)                                                                                                   \
    {                                                                                                               \
        static bool attr = false;                                                                                   \
        if (!attr) { HIP_CHECK(hipFuncSetAttribute((const void*)attn_store_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 1025 * 4)); attr = true; } \
        hipLaunchKernelGGL(attn_store_kernel<D>, grid, block, lds, st, a);                                         \
    }
    switch (a.DP) {
        case 32: LAUNCH(32) break;
        case 64: LAUNCH(64) break;
        case 96: LAUNCH(96) break;
        case 160: LAUNCH(160) break;
        default: throw rt_error(RT_E_UNSUPPORTED, "attn_store: unsupported padded head dim");
    }
#undef LAUNCH
    HIP_CHECK(hipGetLastError());
}
