// CLIP text-encoder pieces that the UNet kernels do not already cover: token + position embedding, causal
// self-attention over <= 128 tokens, and the MLP activations (quick_gelu for CLIP ViT-L, gelu for OpenCLIP bigG).  The text
// encoders run once per prompt set (77 tokens x a handful of prompts): these are latency-sized kernels, every contraction with
// weights still goes through the MFMA GEMM.  Replaces transformers' CLIPTextModel / CLIPTextModelWithProjection as called at
// models/region_diffusion.py:53-66 and models/region_diffusion_sdxl.py:330-356.
#include "common.h"
#include <math.h>

// out[r][c] = tok[ids[r]][c] + pos[r % N][c]      (fp32 trunk)
__global__ __launch_bounds__(256) void embed_kernel(const int* __restrict__ ids, const float* __restrict__ tok, const float* __restrict__ pos,
                                                    float* __restrict__ out, int rows, int N, int C, int vocab) {
    const int r = blockIdx.x;
    int id = ids[r]; if (id < 0) id = 0; if (id >= vocab) id = vocab - 1;
    const float* t = tok + (size_t)id * C;
    const float* p = pos + (size_t)(r % N) * C;
    for (int c = threadIdx.x; c < C; c += 256) out[(size_t)r * C + c] = t[c] + p[c];
}
void launch_embed(const int* ids, const float* tok, const float* pos, float* out, int rows, int N, int C, int vocab, hipStream_t st) {
    RT_REQUIRE(rows > 0 && N > 0 && C > 0 && vocab > 0, "embed: empty problem");
    hipLaunchKernelGGL(embed_kernel, dim3(rows), dim3(256), 0, st, ids, tok, pos, out, rows, N, C, vocab);
    HIP_CHECK(hipGetLastError());
}

// kind 0: quick_gelu x * sigmoid(1.702 x); kind 1: gelu (erf)
__global__ __launch_bounds__(256) void activation_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, size_t n, int kind) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i >= n) return;
    float v[2] = {bf16_to_f32(x[i]), i + 1 < n ? bf16_to_f32(x[i + 1]) : 0.f};
#pragma unroll
    for (int e = 0; e < 2; ++e)
        v[e] = kind == 0 ? v[e] / (1.f + __expf(-1.702f * v[e])) : 0.5f * v[e] * (1.f + erff(v[e] * 0.70710678118654752440f));
    out[i] = f32_to_bf16(v[0]);
    if (i + 1 < n) out[i + 1] = f32_to_bf16(v[1]);
}
void launch_activation(const bf16_t* x, bf16_t* out, size_t n, int kind, hipStream_t st) {
    RT_REQUIRE(n > 0 && (kind == 0 || kind == 1), "activation: bad arguments");
    hipLaunchKernelGGL(activation_kernel, dim3((unsigned)((n / 2 + 255) / 256 + 1)), dim3(256), 0, st, x, out, n, kind);
    HIP_CHECK(hipGetLastError());
}

// Causal attention, one workgroup per (head, batch entry), one thread per query (N <= 128): K and V of the head are staged in LDS
// as fp32, the thread keeps q, the running max / sum and its output row in registers (online softmax in the natural-exp domain).
template <int D>
__global__ __launch_bounds__(128) void causal_attn_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                           int ld, bf16_t* __restrict__ out, int ldo, int N, int d, float scale) {
    extern __shared__ float sm[];
    float* ks = sm;                       // [N][D]
    float* vs = sm + (size_t)N * D;
    const int h = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    for (int i = t; i < N * D; i += 128) {
        const int r = i / D, c = i - r * D;
        const size_t off = ((size_t)b * N + r) * ld + (size_t)h * d + c;
        ks[i] = c < d ? bf16_to_f32(k[off]) : 0.f;
        vs[i] = c < d ? bf16_to_f32(v[off]) : 0.f;
    }
    __syncthreads();
    if (t >= N) return;
    float qr[D], o[D];
#pragma unroll
    for (int c = 0; c < D; ++c) {
        qr[c] = c < d ? bf16_to_f32(q[((size_t)b * N + t) * ld + (size_t)h * d + c]) * scale : 0.f;
        o[c] = 0.f;
    }
    float m = -1e30f, l = 0.f;
    for (int j = 0; j <= t; ++j) {         // causal: keys 0..t
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < D; ++c) s += qr[c] * ks[j * D + c];
        const float mn = fmaxf(m, s);
        const float a = __expf(m - mn), pj = __expf(s - mn);
        l = l * a + pj;
#pragma unroll
        for (int c = 0; c < D; ++c) o[c] = o[c] * a + pj * vs[j * D + c];
        m = mn;
    }
    const float inv = 1.f / l;
#pragma unroll
    for (int c = 0; c < D; ++c)
        if (c < d) out[((size_t)b * N + t) * ldo + (size_t)h * d + c] = f32_to_bf16(o[c] * inv);
}
template <int D>
static void launch_causal_t(const bf16_t* q, const bf16_t* k, const bf16_t* v, int ld, bf16_t* out, int ldo, int B, int H, int N, int d,
                            float scale, hipStream_t st) {
    const size_t lds = (size_t)2 * N * D * sizeof(float);
    static bool attr = false;
    if (!attr) {
        HIP_CHECK(hipFuncSetAttribute((const void*)causal_attn_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 128 * D * (int)sizeof(float)));
        attr = true;
    }
    hipLaunchKernelGGL((causal_attn_kernel<D>), dim3(H, B), dim3(128), lds, st, q, k, v, ld, out, ldo, N, d, scale);
    HIP_CHECK(hipGetLastError());
}
void launch_causal_attention(const bf16_t* q, const bf16_t* k, const bf16_t* v, int ld, bf16_t* out, int ldo, int B, int H, int N, int d,
                             float scale, hipStream_t st) {
    RT_REQUIRE(B > 0 && H > 0 && N > 0 && N <= 128 && d > 0 && d <= 128, "causal attention: N <= 128 tokens, head dim <= 128");
    if (d <= 32) launch_causal_t<32>(q, k, v, ld, out, ldo, B, H, N, d, scale, st);
    else if (d <= 64) launch_causal_t<64>(q, k, v, ld, out, ldo, B, H, N, d, scale, st);
    else launch_causal_t<128>(q, k, v, ld, out, ldo, B, H, N, d, scale, st);
}
