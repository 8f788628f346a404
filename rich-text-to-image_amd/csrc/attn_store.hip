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

// Keys are processed in chunks of at most 1024 (the LDS tile is [32][chunk+1] fp32): pass 0 computes the softmax statistics
// (running max / sum over ALL keys) of every head and parks them in LDS, then every chunk recomputes its scores per head,
// accumulates the head average in the tile and flushes it.  Total MFMA work = 2 passes over the scores, whatever the key count,
// so 64x64 self-attention maps (4096 keys: what the SDXL token-map hook sees on every attn1 layer, xl.py:980-992) cost the same
// per score as 32x32 ones.
#define RT_STORE_CHUNK 1024
template <int DP>
__global__ __launch_bounds__(64) void attn_store_kernel(AttnStoreArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
    const int chunk = p.NKpad < RT_STORE_CHUNK ? p.NKpad : RT_STORE_CHUNK;
    const int ldt = chunk + 1;
    float* acc = (float*)smem;                       // [32][ldt]
    float* stat = acc + 32 * ldt;                    // [H][64][2]: running max, 1 / (H * sum) of (head, lane's query)
    const int q0 = blockIdx.x * 32;
    const int q = q0 + l31 < p.N ? q0 + l31 : p.N - 1;
    const float invH = 1.f / (float)p.H;
    auto load_q = [&](int h, bf16x8 (&qf)[DP / 16]) {
        const bf16_t* qptr = p.Q + ((size_t)p.q_row0 + q) * p.ldq + h * DP + hi * 8;
#pragma unroll
        for (int ks = 0; ks < DP / 16; ++ks) qf[ks] = *(const bf16x8*)(qptr + ks * 16);
    };
    auto scores = [&](int h, int kt, const bf16x8 (&qf)[DP / 16]) {
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const int krow = kt * 32 + l31;
        const bf16_t* kp = p.K + ((size_t)p.k_row0 + (krow < p.NKrows ? krow : p.NKrows - 1)) * p.ldk + h * DP + hi * 8;
#pragma unroll
        for (int ks = 0; ks < DP / 16; ++ks) {
            const bf16x8 kf = *(const bf16x8*)(kp + ks * 16);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
        }
        return s;                                    // lane holds keys kt*32 + (r&3) + 8*(r>>2) + 4*hi of query l31
    };
    // ---- pass 0: softmax statistics per head
    for (int h = 0; h < p.H; ++h) {
        bf16x8 qf[DP / 16];
        load_q(h, qf);
        float m = -1e30f, l = 0.f;
        for (int kt = 0; kt < p.NKpad / 32; ++kt) {
            f32x16 s = scores(h, kt, qf);
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
        }
        l += __shfl_xor(l, 32);
        stat[(h * 64 + lane) * 2] = m;
        stat[(h * 64 + lane) * 2 + 1] = invH / l;
    }
    // ---- pass 1: normalised probabilities, head-averaged per key chunk
    for (int c0 = 0; c0 < p.NKpad; c0 += chunk) {
        for (int i = lane; i < 32 * ldt; i += 64) acc[i] = 0.f;
        __syncthreads();
        const int cend = c0 + chunk < p.NKpad ? c0 + chunk : p.NKpad;
        for (int h = 0; h < p.H; ++h) {
            bf16x8 qf[DP / 16];
            load_q(h, qf);
            const float m = stat[(h * 64 + lane) * 2], inv = stat[(h * 64 + lane) * 2 + 1];
            for (int kt = c0 / 32; kt < cend / 32; ++kt) {
                const f32x16 s = scores(h, kt, qf);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key < p.NK) acc[l31 * ldt + key - c0] += exp2f(s[r] - m) * inv;
                }
            }
        }
        __syncthreads();
        // tile -> accumulator (row-contiguous)
        const int ncol = (cend < p.NK ? cend : p.NK) - c0;
        for (int i = lane; i < 32 * ncol; i += 64) {
            const int r = i / ncol, c = i - r * ncol;
            if (q0 + r < p.N) {
                float* dst = p.out + (size_t)(q0 + r) * p.NK + c0 + c;
                *dst = p.overwrite ? acc[r * ldt + c] : *dst + acc[r * ldt + c];
            }
        }
        __syncthreads();
    }
}

void launch_attn_store(const AttnStoreArgs& a, hipStream_t st) {
    RT_REQUIRE(a.NKpad % 32 == 0 && a.NK >= 1 && a.NK <= a.NKpad && a.NKrows >= 1 && a.H >= 1 && a.H <= 32, "attn_store: keys are padded to a multiple of 32; at most 32 heads");
    const int chunk = a.NKpad < RT_STORE_CHUNK ? a.NKpad : RT_STORE_CHUNK;
    const size_t lds = (size_t)32 * (chunk + 1) * 4 + (size_t)a.H * 64 * 2 * 4;
    dim3 grid(cdiv(a.N, 32)), block(64);
#define LAUNCH(D)                                                                                                   \
    {                                                                                                               \
        static bool attr = false;                                                                                   \
        if (!attr) { HIP_CHECK(hipFuncSetAttribute((const void*)attn_store_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * (RT_STORE_CHUNK + 1) * 4 + 32 * 64 * 2 * 4)); attr = true; } \
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
