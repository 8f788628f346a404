// cross77_kernel: the 77-key cross-attention of attn2 (head dim 64, cached K / V^T) - the engine's kernel behind the to_q projection.
// Built with -fno-honor-nans (Makefile): its softmax maxima run over finite-or--inf scores; the all-masked row is handled explicitly.
#include "xb_common.h"
#include <type_traits>

// ---------------------------------------------------------------------------------------------- cross77_kernel: stand-alone 77-key cross-attention
// O = softmax_fs(Q K[prompt]^T) V[prompt] for the cached cross-attention keys (models/attention_processor.py:476-545, font-size softmax
// :386-401) - the attention unit of xblock_kernel above with Q read from and O written to HBM.  The generic attn_kernel (attention.hip)
// runs this shape as a one-tile flash loop: 19.6 us for 7 x 1024 tokens x 20 heads, 28 us at 4096 tokens x 10 heads - 2.6 TB/s of its
// Q + O bytes.  Here a workgroup owns 64 (or 128) queries x 1 head (2 behind debug bit 21); K / V^T of the head (80 staged keys, 22 KB)
// arrive in LDS by coalesced LDS-DMA once, up to six workgroups share a CU (24 waves hide each other's fragment and softmax latency),
// every wave runs its 16 (32) queries against the head: S^T = K Q^T (Q rows straight from HBM as the B operand), masked / font-size-biased scores, P^T from the
// accumulators, O^T = V^T P^T with the V^T rows staged in the pi order, so a lane ends with 8 CONSECUTIVE d of its query per pair of
// accumulator tiles = one 16-B store.
#define C77_LDS(HPW) ((HPW) * XB_KVH + 768)
template <int T, int HPW>                                             // 16-query tiles per wave, heads per workgroup: a workgroup owns 64 T queries x HPW heads
__global__ __launch_bounds__(256, HPW == 2 ? 3 : 4) void cross77_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q4 = lane >> 4;
    // (head pair, stream, query block) order, contiguous per XCD: the query blocks of one (stream, head pair) share K / V^T in one L2
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, qq = nwg >> 3, rr = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    }
    const int nqb = p.N / (64 * T);
    const int qb = bid % nqb; bid /= nqb;
    const int b = bid % p.B, hg = bid / p.B;                         // hg: head group (HPW heads)
    const int prompt = p.k_src[b], wset = p.wset[b];
    const bool fs = wset >= 0;
    float tw = 1.f, tsg = 1.f;
    if (fs && tid < 96) { tw = p.wabs[wset * p.NK + tid]; tsg = p.wsgn[wset * p.NK + tid]; }
    // ---- K / V^T of heads 2 hg, 2 hg + 1: 44 pieces of 1 KB, 11 per wave (layout of xblock_kernel's K / V^T tiles)
    const int lrow = lane >> 3, pslot = lane & 7;
    const int voff_v = xb_pi(lane) * p.ldvt * 2;
#pragma unroll
    for (int i = 0; i < (22 * HPW + 3) / 4; ++i) {
        const int pidx = i * 4 + wave;
        if (pidx >= 22 * HPW) break;                                 // (HPW = 1: 22 pieces, waves 2 and 3 issue five)
        const int h2 = pidx >= 22 ? 1 : 0, pp = pidx - 22 * h2;
        const int head = hg * HPW + h2;
        char* dst = smem + h2 * XB_KVH + pp * 1024;
        if (pp < 10) {
            const int rho = pp * 8 + lrow, j = rho >> 4, i16 = rho & 15;
            const int key = j < 4 ? 32 * (j >> 1) + 8 * (i16 >> 2) + 4 * (j & 1) + (i16 & 3) : 64 + i16;
            glds16_buf(p.K, (key * p.ldk + ((pslot ^ ((rho >> 1) & 7)) << 3)) * 2, (prompt * p.NK * p.ldk + head * 64) * 2, dst);
        } else {
            const int cc = pp - 10, keyoff = cc < 8 ? 8 * cc : 64 + 4 * (cc - 8);
            glds16_buf(p.VT, voff_v, (head * 64 * p.ldvt + prompt * p.NK + keyoff) * 2, dst);
        }
    }
    // ---- Q fragments of this wave's 16 T queries, both heads: d = 32 ks + 8 q4 .. + 7
    const int q0 = qb * 64 * T + wave * 16 * T + l15;
    const bf16_t* qp = p.Q + ((size_t)p.q_src[b] * p.N + q0) * p.ldq + hg * (64 * HPW) + 8 * q4;
    bf16x8 qf[T][HPW][2];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int h2 = 0; h2 < HPW; ++h2) {
            qf[t][h2][0] = *(const bf16x8*)(qp + (size_t)t * 16 * p.ldq + h2 * 64);
            qf[t][h2][1] = *(const bf16x8*)(qp + (size_t)t * 16 * p.ldq + h2 * 64 + 32);
        }
    float* tabw = (float*)(smem + HPW * XB_KVH);
    if (tid < 96) {
        tabw[tid] = tid < p.nk_valid ? __builtin_amdgcn_logf(tw) : -INFINITY;          // v_log_f32 = log2; log2(0) = -inf
        tabw[96 + tid] = tsg;
    }
    // hipcc's waitcnt pass orders LDS *stores* behind a pending LDS-DMA, not LDS reads, and __syncthreads() does not wait for VMEM loads:
    // every wave must see its own pieces landed BEFORE the barrier (without this wait the kernel raced)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const float* tab = tabw;
    const int key = (l15 >> 1) & 7;
    const int c0 = ((q4 ^ key) << 4), c1 = (((4 + q4) ^ key) << 4);
    bf16_t* orow = p.O + ((size_t)b * p.N + q0) * p.ldo + hg * (64 * HPW) + 8 * q4;
#pragma unroll
    for (int h2 = 0; h2 < HPW; ++h2) {
        const char* kp = smem + h2 * XB_KVH + l15 * 128;
        const char* vp = smem + h2 * XB_KVH + 10240 + (q4 * 64 + l15) * 16;
        f32x4 s[T][5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const f32x4 bias = *(const f32x4*)(tab + (j < 4 ? 32 * (j >> 1) + 8 * q4 + 4 * (j & 1) : 64 + 4 * q4));
            const bf16x8 k0 = *(const bf16x8*)(kp + j * 2048 + c0), k1 = *(const bf16x8*)(kp + j * 2048 + c1);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                s[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf[t][h2][0], bias, 0, 0, 0);
                s[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf[t][h2][1], s[t][j], 0, 0, 0);
            }
        }
        float inv[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            float mx = xb_max3(s[t][0][0], s[t][0][1], s[t][0][2]);
            mx = xb_max3(mx, s[t][0][3], s[t][1][0]);
            mx = xb_max3(mx, s[t][1][1], s[t][1][2]);
#pragma unroll
            for (int j = 2; j < 5; ++j) { mx = xb_max3(mx, s[t][j - 1][3], s[t][j][0]); mx = xb_max3(mx, s[t][j][1], s[t][j][2]); }
            mx = xb_rowmax(xb_max(mx, s[t][4][3]));
            // every valid key carries font-size weight 0 (log2 0 = -inf on every score): the reference's softmax yields NaN for the row
            // (0 / 0, attention_processor.py:392-396).  This file is built with -fno-honor-nans, under which (-inf) - (-inf) would be
            // unspecified: take 0 as the reference instead - the exponentials are exactly 0, their sum is 0, inv = +inf, and the
            // row leaves as 0 * inf = NaN by the hardware's own arithmetic, as in the generic attn_kernel
            mx = mx == -INFINITY ? 0.f : mx;
            f32x2 sum2 = {0.f, 0.f};
            const f32x2 nmx = {-mx, -mx};
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const f32x2 d0 = f32x2{s[t][j][0], s[t][j][1]} + nmx, d1 = f32x2{s[t][j][2], s[t][j][3]} + nmx;      // Q carries d^-1/2 log2 e
                s[t][j][0] = __builtin_amdgcn_exp2f(d0.x); s[t][j][1] = __builtin_amdgcn_exp2f(d0.y);
                s[t][j][2] = __builtin_amdgcn_exp2f(d1.x); s[t][j][3] = __builtin_amdgcn_exp2f(d1.y);
                sum2 += f32x2{s[t][j][0], s[t][j][1]} + f32x2{s[t][j][2], s[t][j][3]};
            }
            inv[t] = 1.f / xb_rowsum(sum2.x + sum2.y);
            // token-map capture (plain pass): P(q, k) = exp2(s_k - mx) / sum for the valid keys of this head - what attn_store_apply2_kernel needs
            if (p.stats != nullptr && b == p.stats_b && q4 == 0)
                ((float2*)p.stats)[(size_t)(hg * HPW + h2) * p.N + (q0 - l15) + t * 16 + l15] = make_float2(mx, inv[t] / (float)p.H);
            if (fs) {                                                // sign of a negative font size on the normalised probability
#pragma unroll
                for (int j = 0; j < 5; ++j) s[t][j] = s[t][j] * *(const f32x4*)(tab + 96 + (j < 4 ? 32 * (j >> 1) + 8 * q4 + 4 * (j & 1) : 64 + 4 * q4));
            }
        }
        f32x4 o[T][4];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < 3; ++st) {
            bf16x8 pf[T];
#pragma unroll
            for (int t = 0; t < T; ++t) pf[t] = xb_pack8(s[t][2 * st], st < 2 ? s[t][2 * st + 1] : zero4);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 vf = *(const bf16x8*)(vp + st * 4096 + dt * 256);
#pragma unroll
                for (int t = 0; t < T; ++t) o[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[t], o[t][dt], 0, 0, 0);
            }
        }
        // accumulator tiles (2 s2, 2 s2 + 1) of lane (l15, q4): d = 32 s2 + 8 q4 + [0, 8) of query l15
#pragma unroll
        for (int t = 0; t < T; ++t) {
            *(bf16x8*)(orow + (size_t)t * 16 * p.ldo + h2 * 64) = xb_pack8(o[t][0] * inv[t], o[t][1] * inv[t]);
            *(bf16x8*)(orow + (size_t)t * 16 * p.ldo + h2 * 64 + 32) = xb_pack8(o[t][2] * inv[t], o[t][3] * inv[t]);
        }
    }
}

int g_c77_t1 = 0;       // debug bits 20 / 21 (A/B): one 16-query tile per wave for every shape / two heads per workgroup
bool cross77_supported(int H, int DP, int tokens, int NK, int nk_valid) {
    return DP == 64 && H >= 1 && tokens % 64 == 0 && NK == 96 && nk_valid >= 1 && nk_valid <= 80;
}

void launch_cross77(const AttnArgs& a, hipStream_t st) {
    RT_REQUIRE(a.cross && cross77_supported(a.H, a.DP, a.N, a.NK, a.nk_valid), "cross77: shape");
    RT_REQUIRE(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldvt % 8 == 0 && a.ldo % 8 == 0, "cross77: leading dimensions");
    RT_REQUIRE((long)RT_MAXB * a.NK * a.ldk * 2 < 0x7fffffffL && (long)a.H * 64 * a.ldvt * 2 < 0x7fffffffL, "cross77: K / V^T cache beyond the 2 GiB descriptor range");
    for (int b = 0; b < a.B; ++b) RT_REQUIRE(a.wset[b] < 0 || (a.wabs && a.wsgn), "cross77: multiplier tables");
    for (int b = 0; b < a.B; ++b) RT_REQUIRE(a.k_src[b] == a.v_src[b], "cross77: K and V of one prompt");
    // ONE head per workgroup (22.5 KB of LDS: up to 6 workgroups = 24 waves per CU hide each other's latency chain; 12.3 vs 13.9 us at
    // 7 x 1024 x 20 heads, 19.3 vs 21.4 at 4096 x 10, profiles/r5_cross77_probe.txt) and two 16-query tiles per wave (K / V^T staged once
    // per 128 queries, every fragment read feeds two MFMAs) when that still leaves two workgroups per CU.  A query's arithmetic does
    // not depend on these choices (bit-identical either way), so they may look at the batch.
    // g_c77_mode (debug bits 20 / 21): bit 0 = one tile per wave always, bit 1 = TWO heads per workgroup (the first form of the kernel)
    const bool one_head = (g_c77_t1 & 2) == 0 || (a.H & 1);
    const int hpw = one_head ? 1 : 2;
    const int T = (g_c77_t1 & 1) ? 1 : ((a.N % 128 == 0 && (a.N / 128) * a.B * (a.H / hpw) >= 512) ? 2 : 1);
    static bool attr = false;
    if (!attr) {
        HIP_CHECK(hipFuncSetAttribute((const void*)cross77_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, C77_LDS(2)));
        HIP_CHECK(hipFuncSetAttribute((const void*)cross77_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, C77_LDS(2)));
        HIP_CHECK(hipFuncSetAttribute((const void*)cross77_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, C77_LDS(1)));
        HIP_CHECK(hipFuncSetAttribute((const void*)cross77_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, C77_LDS(1)));
        attr = true;
    }
    const dim3 grid((a.N / (64 * T)) * a.B * (a.H / hpw));
    if (hpw == 2) {
        if (T == 2) hipLaunchKernelGGL((cross77_kernel<2, 2>), grid, dim3(256), C77_LDS(2), st, a);
        else hipLaunchKernelGGL((cross77_kernel<1, 2>), grid, dim3(256), C77_LDS(2), st, a);
    } else {
        if (T == 2) hipLaunchKernelGGL((cross77_kernel<2, 1>), grid, dim3(256), C77_LDS(1), st, a);
        else hipLaunchKernelGGL((cross77_kernel<1, 1>), grid, dim3(256), C77_LDS(1), st, a);
    }
    HIP_CHECK(hipGetLastError());
}

