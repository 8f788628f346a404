// Kernels that exist only for the colour-guidance gradient through the VAE decoder:
// GroupNorm(+SiLU) backward, materialised single-head attention softmax forward/backward, 2x2 sum-pool (adjoint of
// the nearest-2x upsample), the masked-mean colour loss and its gradient, and the tiny 1x1 post_quant_conv.
// Every contraction (conv / linear forward and backward-data) reuses the MFMA GEMM of gemm.hip.
#include "common.h"
#include "vae.h"
#include <math.h>

#define GN_MAXC 2560
__device__ __forceinline__ void ld4(const void* x, int bf16in, size_t off, float v[4]) {
    if (bf16in) {
        const uint2 u = *(const uint2*)((const bf16_t*)x + off);
        v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
        v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    } else {
        const float4 f = *(const float4*)((const float*)x + off);
        v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
    }
}
// forward statistics as launch_groupnorm left them: partial[b][0][g] = (mean, rstd) (gn_finalize_kernel, norm.hip)
__device__ __forceinline__ void fwd_stats(const GroupNormBwdArgs& p, int b, float* mean, float* rstd) {
    if (threadIdx.x < p.G) {
        mean[threadIdx.x] = p.fwd_partial[(size_t)b * p.nchunk * 2 * p.G + 2 * threadIdx.x];
        rstd[threadIdx.x] = p.fwd_partial[(size_t)b * p.nchunk * 2 * p.G + 2 * threadIdx.x + 1];
    }
}
// Raw operands of 4 channels of one row: requested for GN_UB rows TOGETHER before any of them is used (round 6).  The SD VAE's maps (8 - 134 MB:
// L2 / Infinity-Cache resident) gain 1.6 % of a guidance call; the SDXL VAE's 1024^2 x 128 fp32 maps stream at the memory system's rate either way.
struct GnRaw { float x[4], da[4]; };
__device__ __forceinline__ void gn_raw_load(const GroupNormBwdArgs& p, size_t off, GnRaw& w) {
    ld4(p.x, p.x_bf16, off, w.x);
    ld4(p.dA, 1, off, w.da);
    if (p.dA_lo) { float dl[4]; ld4(p.dA_lo, 1, off, dl); w.da[0] += dl[0]; w.da[1] += dl[1]; w.da[2] += dl[2]; w.da[3] += dl[3]; }
}
// dxh for 4 channels of one row
__device__ __forceinline__ void dxhat4(const GroupNormBwdArgs& p, const GnRaw& w, const float* ga, const float* be, const float* mu,
                                       const float* rs, float xh[4], float dxh[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        xh[e] = (w.x[e] - mu[e]) * rs[e];
        float dy = w.da[e];
        if (p.silu) {
            const float y = xh[e] * ga[e] + be[e];
            const float sg = 1.f / (1.f + __expf(-y));
            dy *= sg * (1.f + y * (1.f - sg));
        }
        dxh[e] = dy * ga[e];
    }
}
#define GN_UB 4

__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(GroupNormBwdArgs p) {
    __shared__ float sh_s[GN_MAXC], sh_q[GN_MAXC], mean[32], rstd[32];
    const int C = p.C, cpg = C / p.G, nv = C >> 2;
    const int b = blockIdx.y, chunk = blockIdx.x;
    fwd_stats(p, b, mean, rstd);
    __syncthreads();
    const int r0 = chunk * p.rows_per_chunk, r1 = min(r0 + p.rows_per_chunk, p.HW);
    const int nrl = nv >= 256 ? 1 : 256 / nv;
    const int rl = nv >= 256 ? 0 : threadIdx.x / nv;
    const int v0 = nv >= 256 ? threadIdx.x : threadIdx.x % nv;
    if (rl < nrl) {
        for (int vec = v0; vec < nv; vec += 256) {
            const int c = vec * 4;
            float ga[4], be[4], mu[4], rs[4], s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 4; ++e) { const int g = (c + e) / cpg; ga[e] = p.gamma[c + e]; be[e] = p.beta[c + e]; mu[e] = mean[g]; rs[e] = rstd[g]; }
            for (int r = r0 + rl; r < r1; r += GN_UB * nrl) {
                GnRaw w[GN_UB];
#pragma unroll
                for (int u = 0; u < GN_UB; ++u) {
                    const int rr = r + u * nrl < r1 ? r + u * nrl : r;          // clamped address, masked below
                    gn_raw_load(p, ((size_t)b * p.HW + rr) * C + c, w[u]);
                }
#pragma unroll
                for (int u = 0; u < GN_UB; ++u) {                               // rows in ascending order: the sums are those of the one-row loop
                    if (r + u * nrl >= r1) continue;
                    float xh[4], dxh[4];
                    dxhat4(p, w[u], ga, be, mu, rs, xh, dxh);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { s1[e] += dxh[e]; s2[e] += dxh[e] * xh[e]; }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { sh_s[rl * C + c + e] = s1[e]; sh_q[rl * C + c + e] = s2[e]; }
        }
    }
    __syncthreads();
    if (threadIdx.x < p.G) {
        float s = 0.f, q = 0.f;
        for (int k = 0; k < nrl; ++k)
            for (int c = threadIdx.x * cpg; c < (threadIdx.x + 1) * cpg; ++c) { s += sh_s[k * C + c]; q += sh_q[k * C + c]; }
        float* dst = p.bwd_partial + ((size_t)b * p.nchunk + chunk) * 2 * p.G + 2 * threadIdx.x;
        dst[0] = s; dst[1] = q;
    }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(GroupNormBwdArgs p) {
    __shared__ float mean[32], rstd[32], m1[32], m2[32];
    const int C = p.C, cpg = C / p.G, nv = C >> 2;
    const int b = blockIdx.y, chunk = blockIdx.x;
    fwd_stats(p, b, mean, rstd);
    if (threadIdx.x < p.G) {           // finalized by gn_finalize_kernel (mode 1)
        m1[threadIdx.x] = p.bwd_partial[(size_t)b * p.nchunk * 2 * p.G + 2 * threadIdx.x];
        m2[threadIdx.x] = p.bwd_partial[(size_t)b * p.nchunk * 2 * p.G + 2 * threadIdx.x + 1];
    }
    __syncthreads();
    const int r0 = chunk * p.rows_per_chunk, r1 = min(r0 + p.rows_per_chunk, p.HW);
    const int nrl = nv >= 256 ? 1 : 256 / nv;
    const int rl = nv >= 256 ? 0 : threadIdx.x / nv;
    const int v0 = nv >= 256 ? threadIdx.x : threadIdx.x % nv;
    if (rl >= nrl) return;
    for (int vec = v0; vec < nv; vec += 256) {
        const int c = vec * 4;
        float ga[4], be[4], mu[4], rs[4], a1[4], a2[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int g = (c + e) / cpg; ga[e] = p.gamma[c + e]; be[e] = p.beta[c + e]; mu[e] = mean[g]; rs[e] = rstd[g]; a1[e] = m1[g]; a2[e] = m2[g]; }
        for (int r = r0 + rl; r < r1; r += GN_UB * nrl) {
            GnRaw w[GN_UB];
            float4 ad[GN_UB];
#pragma unroll
            for (int u = 0; u < GN_UB; ++u) {
                const int rr = r + u * nrl < r1 ? r + u * nrl : r;
                const size_t off = ((size_t)b * p.HW + rr) * C + c;
                gn_raw_load(p, off, w[u]);
                ad[u] = p.add ? *(const float4*)(p.add + off) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < GN_UB; ++u) {
                if (r + u * nrl >= r1) continue;
                const size_t off = ((size_t)b * p.HW + r + u * nrl) * C + c;
                float xh[4], dxh[4], o[4];
                dxhat4(p, w[u], ga, be, mu, rs, xh, dxh);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rs[e] * (dxh[e] - a1[e] - xh[e] * a2[e]);
                if (p.add) { o[0] += ad[u].x; o[1] += ad[u].y; o[2] += ad[u].z; o[3] += ad[u].w; }
                if (p.out) *(float4*)(p.out + off) = make_float4(o[0], o[1], o[2], o[3]);
                if (p.out_bf16) { uint2 wv; wv.x = pack_bf16x2(o[0], o[1]); wv.y = pack_bf16x2(o[2], o[3]); *(uint2*)(p.out_bf16 + off) = wv; }
                if (p.out_bf16_lo) { uint2 wv; wv.x = pack_bf16x2_lo(o[0], o[1]); wv.y = pack_bf16x2_lo(o[2], o[3]); *(uint2*)(p.out_bf16_lo + off) = wv; }
            }
        }
    }
}
void launch_groupnorm_bwd(const GroupNormBwdArgs& a, hipStream_t st) {
    RT_REQUIRE(a.G >= 1 && a.G <= 32 && a.C % a.G == 0 && a.C % 4 == 0 && a.C <= GN_MAXC, "groupnorm_bwd: bad channel/group count");
    dim3 grid(a.nchunk, a.B), block(256);
    hipLaunchKernelGGL(gn_bwd_stats_kernel, grid, block, 0, st, a);
    launch_gn_finalize(a.bwd_partial, a.B, a.nchunk, a.G, (double)(a.C / a.G) * a.HW, 0.f, 1, st);
    hipLaunchKernelGGL(gn_bwd_apply_kernel, grid, block, 0, st, a);
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------- bf16 transpose through LDS
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int R, int C) {
    __shared__ bf16_t tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        tile[r][c] = (r0 + r < R && c0 + c < C) ? in[(size_t)(r0 + r) * C + c0 + c] : (bf16_t)0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i >> 6, r = i & 63;
        if (c0 + c < C && r0 + r < R) out[(size_t)(c0 + c) * R + r0 + r] = tile[r][c];
    }
}
void launch_transpose_bf16(const bf16_t* in, bf16_t* out, int R, int C, hipStream_t st) {
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3(cdiv(C, 64), cdiv(R, 64)), dim3(256), 0, st, in, out, R, C);
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------- row softmax (one block per row)
__device__ __forceinline__ float block_reduce(float v, bool is_max, float* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const float t = __shfl_xor(v, o); v = is_max ? fmaxf(v, t) : v + t; }
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float r = sh[0];
    for (int i = 1; i < 4; ++i) r = is_max ? fmaxf(r, sh[i]) : r + sh[i];
    return r;
}
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, bf16_t* __restrict__ p, bf16_t* __restrict__ p_lo, int cols, float scale) {
    __shared__ float sh[4];
    const float* row = s + (size_t)blockIdx.x * cols;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < cols; c += 256) mx = fmaxf(mx, row[c] * scale);
    mx = block_reduce(mx, true, sh);
    float sum = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) sum += __expf(row[c] * scale - mx);
    sum = block_reduce(sum, false, sh);
    const float inv = 1.f / sum;
    for (int c = threadIdx.x; c < cols; c += 256) {
        const float v = __expf(row[c] * scale - mx) * inv;
        p[(size_t)blockIdx.x * cols + c] = f32_to_bf16(v);
        if (p_lo) p_lo[(size_t)blockIdx.x * cols + c] = f32_to_bf16(bf16_residual(v));
    }
}
void launch_softmax_rows(const float* s, bf16_t* p, int rows, int cols, float scale, hipStream_t st, bf16_t* p_lo) {
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, st, s, p, p_lo, cols, scale);
    HIP_CHECK(hipGetLastError());
}
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const bf16_t* __restrict__ p, const bf16_t* __restrict__ p_lo, const float* __restrict__ dp,
                                                          bf16_t* __restrict__ ds, bf16_t* __restrict__ ds_lo, int cols, float scale) {
    __shared__ float sh[4];
    const size_t base = (size_t)blockIdx.x * cols;
    auto prob = [&](int c) { return bf16_to_f32(p[base + c]) + (p_lo ? bf16_to_f32(p_lo[base + c]) : 0.f); };
    float dot = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) dot += prob(c) * dp[base + c];
    dot = block_reduce(dot, false, sh);
    for (int c = threadIdx.x; c < cols; c += 256) {
        const float v = scale * prob(c) * (dp[base + c] - dot);
        ds[base + c] = f32_to_bf16(v);
        if (ds_lo) ds_lo[base + c] = f32_to_bf16(bf16_residual(v));
    }
}
void launch_softmax_bwd(const bf16_t* p, const float* dp, bf16_t* ds, int rows, int cols, float scale, hipStream_t st, const bf16_t* p_lo, bf16_t* ds_lo) {
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3(rows), dim3(256), 0, st, p, p_lo, dp, ds, ds_lo, cols, scale);
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------- adjoint of nearest-2x upsample
__global__ void sumpool2x2_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, int C4) {
    const size_t n = (size_t)B * H * W * C4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4); size_t t = i / C4;
        const int x = (int)(t % W); t /= W;
        const int y = (int)(t % H); const int b = (int)(t / H);
        const float4* src = (const float4*)in + (((size_t)b * 2 * H + 2 * y) * 2 * W + 2 * x) * C4 + c;
        const float4 a = src[0], bb = src[C4], cc = src[(size_t)2 * W * C4], d = src[(size_t)2 * W * C4 + C4];
        ((float4*)out)[i] = make_float4(a.x + bb.x + cc.x + d.x, a.y + bb.y + cc.y + d.y, a.z + bb.z + cc.z + d.z, a.w + bb.w + cc.w + d.w);
    }
}
void launch_sumpool2x2(const float* in, float* out, int B, int H, int W, int C, hipStream_t st) {
    RT_REQUIRE(C % 4 == 0, "sumpool: C % 4");
    const size_t n = (size_t)B * H * W * (C / 4);
    int grid = (int)((n + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(sumpool2x2_kernel, dim3(grid), dim3(256), 0, st, in, out, B, H, W, C / 4);
    HIP_CHECK(hipGetLastError());
}
__global__ void add_f32_kernel(const float* a, const float* b, float* out, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 x = ((const float4*)a)[i], y = ((const float4*)b)[i];
        ((float4*)out)[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
}
void launch_add_f32(const float* a, const float* b, float* out, size_t n, hipStream_t st) {
    RT_REQUIRE(n % 4 == 0, "add: n % 4");
    int grid = (int)((n / 4 + 255) / 256); if (grid > 4096) grid = 4096; if (grid < 1) grid = 1;
    hipLaunchKernelGGL(add_f32_kernel, dim3(grid), dim3(256), 0, st, a, b, out, n / 4);
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------- post_quant_conv (1x1, 4 -> 4) fused with predict_x0 / scaling
// z0 = (c_lat * lat + c_eps * eps)  [= predict_x0(lat, eps) / scaling_factor];  out = W z0 + b
__global__ void pq_conv_fwd_kernel(const float* lat, const float* eps, float c_lat, float c_eps, const float* W, const float* b,
                                   bf16_t* out, bf16_t* out_lo, int HW) {
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= HW) return;
    float z[4], o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) z[c] = c_lat * lat[c * HW + pix] + (eps ? c_eps * eps[c * HW + pix] : 0.f);
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = b[k] + W[k * 4 + 0] * z[0] + W[k * 4 + 1] * z[1] + W[k * 4 + 2] * z[2] + W[k * 4 + 3] * z[3];
    uint4 q; q.x = pack_bf16x2(o[0], o[1]); q.y = pack_bf16x2(o[2], o[3]); q.z = 0; q.w = 0;
    *(uint4*)(out + (size_t)pix * 8) = q;
    if (out_lo) { uint4 l; l.x = pack_bf16x2_lo(o[0], o[1]); l.y = pack_bf16x2_lo(o[2], o[3]); l.z = 0; l.w = 0; *(uint4*)(out_lo + (size_t)pix * 8) = l; }
}
void launch_pq_conv_fwd(const float* lat, const float* eps, float c_lat, float c_eps, const float* W, const float* b, bf16_t* out, int HW,
                        hipStream_t st, bf16_t* out_lo) {
    hipLaunchKernelGGL(pq_conv_fwd_kernel, dim3(cdiv(HW, 256)), dim3(256), 0, st, lat, eps, c_lat, c_eps, W, b, out, out_lo, HW);
    HIP_CHECK(hipGetLastError());
}
__global__ void pq_conv_bwd_update_kernel(const float* dz, int ldz, const float* W, float gscale, float weight, const float* mask_all,
                                          float* lat, float* grad_out, int HW) {
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= HW) return;
    float d[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) d[o] = dz[(size_t)pix * ldz + o];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float g = (W[0 * 4 + c] * d[0] + W[1 * 4 + c] * d[1] + W[2 * 4 + c] * d[2] + W[3 * 4 + c] * d[3]) * gscale;
        if (grad_out) grad_out[c * HW + pix] = g;
        lat[c * HW + pix] -= g * weight * mask_all[c * HW + pix];
    }
}
void launch_pq_conv_bwd_update(const float* dz, int ldz, const float* W, float gscale, float weight, const float* mask_all, float* lat,
                               float* grad_out, int HW, hipStream_t st) {
    hipLaunchKernelGGL(pq_conv_bwd_update_kernel, dim3(cdiv(HW, 256)), dim3(256), 0, st, dz, ldz, W, gscale, weight, mask_all, lat, grad_out, HW);
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------- colour loss (rd.py:158-166 / xl.py:857-864) and its gradient
// imgs = clamp(img/2 + 0.5, 0, 1); avg[k][c] = sum(imgs_c * m_k) / sum(m_k); L = sum_k 100 * mean_c (avg - target)^2
__global__ __launch_bounds__(256) void color_sums_kernel(ColorLossArgs p) {
    __shared__ float sh[4][4];
    for (int k = 0; k < p.n; ++k) {
        float s[4] = {0, 0, 0, 0};
        for (int pix = blockIdx.x * 256 + threadIdx.x; pix < p.HWi; pix += p.nblk * 256) {
            const float m = p.masks[(size_t)k * p.HWi + pix];
#pragma unroll
            for (int c = 0; c < 3; ++c) s[c] += fminf(fmaxf(p.img[(size_t)pix * p.ldi + c] * 0.5f + 0.5f, 0.f), 1.f) * m;
            s[3] += m;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = s[c];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            __syncthreads();
            if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][c] = v;
            __syncthreads();
            if (threadIdx.x == 0) p.partial[((size_t)blockIdx.x * p.n + k) * 4 + c] = sh[0][c] + sh[1][c] + sh[2][c] + sh[3][c];
        }
    }
}
__global__ __launch_bounds__(256) void color_grad_kernel(ColorLossArgs p) {
    __shared__ float coef[RT_MAXB][4];
    if (threadIdx.x < p.n) {
        const int k = threadIdx.x;
        double s[4] = {0, 0, 0, 0};
        for (int b = 0; b < p.nblk; ++b)
            for (int c = 0; c < 4; ++c) s[c] += (double)p.partial[((size_t)b * p.n + k) * 4 + c];
        float loss = 0.f;
        for (int c = 0; c < 3; ++c) {
            const float avg = (float)(s[c] / s[3]);
            const float d = avg - p.target[k * 3 + c];
            loss += d * d;
            coef[k][c] = (200.f / 3.f) * d / (float)s[3];
        }
        coef[k][3] = loss * (100.f / 3.f);
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0 && p.loss_out) { float l = 0.f; for (int k = 0; k < p.n; ++k) l += coef[k][3]; *p.loss_out = l; }
    for (int pix = blockIdx.x * 256 + threadIdx.x; pix < p.HWi; pix += gridDim.x * 256) {
        float g[3] = {0, 0, 0};
        for (int k = 0; k < p.n; ++k) {
            const float m = p.masks[(size_t)k * p.HWi + pix];
#pragma unroll
            for (int c = 0; c < 3; ++c) g[c] += coef[k][c] * m;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = p.img[(size_t)pix * p.ldi + c] * 0.5f + 0.5f;
            g[c] = (v >= 0.f && v <= 1.f) ? 0.5f * g[c] : 0.f;        // torch.clamp passes the gradient on [min, max]
        }
        uint4 q; q.x = pack_bf16x2(g[0], g[1]); q.y = pack_bf16x2(g[2], 0.f); q.z = 0; q.w = 0;
        *(uint4*)(p.dimg + (size_t)pix * 8) = q;
        if (p.dimg_lo) { uint4 l; l.x = pack_bf16x2_lo(g[0], g[1]); l.y = pack_bf16x2_lo(g[2], 0.f); l.z = 0; l.w = 0; *(uint4*)(p.dimg_lo + (size_t)pix * 8) = l; }
    }
}
void launch_color_loss_grad(const ColorLossArgs& a, hipStream_t st) {
    RT_REQUIRE(a.n >= 1 && a.n <= RT_MAXB, "color loss: 1..16 regions");
    hipLaunchKernelGGL(color_sums_kernel, dim3(a.nblk), dim3(256), 0, st, a);
    hipLaunchKernelGGL(color_grad_kernel, dim3(1024), dim3(256), 0, st, a);
    HIP_CHECK(hipGetLastError());
}

// decoder output [HW, 4] (channel 3 = padding) -> [3, HW]
__global__ void nhwc4_to_nchw3_kernel(const float* in, float* out, int HW) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
        const float4 v = ((const float4*)in)[i];
        out[i] = v.x; out[(size_t)HW + i] = v.y; out[(size_t)2 * HW + i] = v.z;
    }
}
void launch_nhwc4_to_nchw3(const float* in, float* out, int HW, hipStream_t st) {
    hipLaunchKernelGGL(nhwc4_to_nchw3_kernel, dim3(min(4096, (HW + 255) / 256)), dim3(256), 0, st, in, out, HW);
    HIP_CHECK(hipGetLastError());
}
