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
#include <type_traits>
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

// ---------------------------------------------------------------------------------------------- round 4: the one-pass form
// The kernel above gives ONE wave 32 query rows and walks heads x key tiles twice with its operands fetched straight from
// global memory behind every MFMA: 32 single-wave workgroups for a 1024 x 1024 map = 1.4 ms per launch, 74 launches per capturing
// step of the SDXL plain pass = 105 ms on top of a 28 ms step (profiles/r4_plain_pass_before.json).  Here a workgroup of 8 waves
// owns 16 query rows x ALL keys (<= 1024): wave w holds the key tiles w, w + 8, ... (16 keys each, at most 8) - its slice of S^T for
// one head (<= 32 fp32 per lane) AND its slice of the head-average accumulator (<= 32 fp32 per lane) stay in registers, so the scores
// are computed once: local row max -> LDS exchange over the 8 waves -> exp2 -> local row sum -> LDS exchange -> acc += e / (H sum).
// Two barriers per head (statistics buffers alternate with the head's parity), the K fragments of head h + 1 are requested as soon as
// the MFMAs of head h have consumed theirs.  Swapped product as in attention.hip (S^T = K Q^T: a lane owns one query), heads are
// summed in ascending order in registers: deterministic, no atomics.  out (+)= is a read-modify-write of 64-B row segments.
template <int DP>
__global__ __launch_bounds__(512) void attn_store16_kernel(AttnStoreArgs p) {
    constexpr int KSN = DP / 32;                     // 32-deep k steps of the 16x16x32 MFMA
    constexpr int MT = 8;                            // key tiles per wave (8 waves x 8 tiles x 16 keys = 1024 keys)
    typedef __attribute__((ext_vector_type(4))) float f32x4_t;
    __shared__ float smax[2][8][16], ssum[2][8][16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q4 = lane >> 4;
    const int q0 = blockIdx.x * 16;
    const int qrow = q0 + l15 < p.N ? q0 + l15 : p.N - 1;
    const int ntile = p.NKpad >> 4;
    const int nt = wave < ntile ? (ntile - wave + 7) >> 3 : 0;                  // tiles of this wave (wave-uniform), <= MT
    const bf16_t* qbase = p.Q + ((size_t)p.q_row0 + qrow) * p.ldq + q4 * 8;
    const bf16_t* kbase[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int krow = (wave + 8 * i) * 16 + l15; if (krow > p.NKrows - 1) krow = p.NKrows - 1;
        kbase[i] = p.K + ((size_t)p.k_row0 + krow) * p.ldk + q4 * 8;
    }
    f32x4_t acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8 kf[MT][KSN], qf[KSN];
    auto load_head = [&](int h) {
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks) qf[ks] = *(const bf16x8*)(qbase + h * DP + ks * 32);
#pragma unroll
        for (int i = 0; i < MT; ++i)
            if (i < nt) {
#pragma unroll
                for (int ks = 0; ks < KSN; ++ks) kf[i][ks] = *(const bf16x8*)(kbase[i] + h * DP + ks * 32);
            }
    };
    load_head(0);
    const float invH = 1.f / (float)p.H;
    for (int h = 0; h < p.H; ++h) {
        const int par = h & 1;
        f32x4_t s[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            s[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (i < nt) {
#pragma unroll
                for (int ks = 0; ks < KSN; ++ks) s[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[i][ks], qf[ks], s[i], 0, 0, 0);
            }
        }
        if (h + 1 < p.H) load_head(h + 1);           // the fragment registers are free: next head's operands fly behind the softmax
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < MT; ++i)
            if (i < nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = (wave + 8 * i) * 16 + 4 * q4 + r;
                    if (key >= p.NK) s[i][r] = -INFINITY;
                    mx = fmaxf(mx, s[i][r]);
                }
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (q4 == 0) smax[par][wave][l15] = mx;
        // raw barriers: a __syncthreads() carries s_waitcnt vmcnt(0) and would wait for the next head's operands right here
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float gm = smax[par][0][l15];
#pragma unroll
        for (int w = 1; w < 8; ++w) gm = fmaxf(gm, smax[par][w][l15]);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i)
            if (i < nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[i][r] = __builtin_amdgcn_exp2f(s[i][r] - gm); sum += s[i][r]; }
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        if (q4 == 0) ssum[par][wave][l15] = sum;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float gs = ssum[par][0][l15];
#pragma unroll
        for (int w = 1; w < 8; ++w) gs += ssum[par][w][l15];
        const float sc = invH / gs;
#pragma unroll
        for (int i = 0; i < MT; ++i)
            if (i < nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][r] += s[i][r] * sc;
            }
    }
    if (q0 + l15 < p.N) {
        float* orow = p.out + (size_t)(q0 + l15) * p.NK;
        const bool vec = (p.NK & 3) == 0 && (((uintptr_t)p.out) & 15) == 0;      // rows stay 16-B aligned
#pragma unroll
        for (int i = 0; i < MT; ++i)
            if (i < nt) {
                const int key = (wave + 8 * i) * 16 + 4 * q4;
                if (vec && key + 4 <= p.NK) {
                    float4* d = (float4*)(orow + key);
                    float4 v = p.overwrite ? make_float4(0.f, 0.f, 0.f, 0.f) : *d;
                    v.x += acc[i][0]; v.y += acc[i][1]; v.z += acc[i][2]; v.w += acc[i][3];
                    *d = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (key + r < p.NK) orow[key + r] = p.overwrite ? acc[i][r] : orow[key + r] + acc[i][r];
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------- round 4b: statistics + apply (large maps)
// The one-pass kernel streams ALL keys of every head through each 16-query workgroup: 64 workgroups x 2.6 MB through 64 vector L1s
// (64 B/clk each) = 41 k cycles before anything else - 102 us measured for a 1024 x 1024 map (profiles/r4_attn_store_bench.txt), i.e.
// neither HBM nor MFMA but 64 of 256 CUs pulling K through their L1.  For maps of >= 256 keys the work is therefore cut along the
// KEYS as well, which needs the softmax statistics up front:
//   attn_store_stats_kernel: one wave per (head, 16 queries), all keys, online (max, sum) -> stats[h][q] = (m, 1 / (H sum))
//   attn_store_apply_kernel: one workgroup per (16 queries, 128 keys); wave w sums the heads h = w (mod 4) in registers,
//                            the four partial sums meet in LDS in the order w = 0..3 (deterministic), out (+)= the tile.
// Every CU now pulls 1 / 4 of a megabyte instead of 2.6.
template <int DP>
__global__ __launch_bounds__(256) void attn_store_stats_kernel(AttnStoreArgs p) {
    constexpr int KSN = DP / 32;
    typedef __attribute__((ext_vector_type(4))) float f32x4_t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, q4 = lane >> 4;
    const int h = blockIdx.y;
    const int q0 = (blockIdx.x * 4 + wave) * 16;
    if (q0 >= p.N) return;                                           // wave-uniform; no barriers in this kernel
    const int qrow = q0 + l15 < p.N ? q0 + l15 : p.N - 1;
    const bf16_t* qp = p.Q + ((size_t)p.q_row0 + qrow) * p.ldq + h * DP + q4 * 8;
    bf16x8 qf[KSN];
#pragma unroll
    for (int ks = 0; ks < KSN; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 32);
    const int ntile = p.NKpad >> 4;
    float m = -1e30f, l = 0.f;
    bf16x8 kf[4][KSN];
    auto load4 = [&](int t0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int krow = (t0 + i) * 16 + l15; if (krow > p.NKrows - 1) krow = p.NKrows - 1;
            const bf16_t* kp = p.K + ((size_t)p.k_row0 + krow) * p.ldk + h * DP + q4 * 8;
#pragma unroll
            for (int ks = 0; ks < KSN; ++ks) kf[i][ks] = *(const bf16x8*)(kp + ks * 32);
        }
    };
    load4(0);
    for (int t0 = 0; t0 < ntile; t0 += 4) {
        f32x4_t s[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KSN; ++ks) s[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[i][ks], qf[ks], s[i], 0, 0, 0);
        }
        if (t0 + 4 < ntile) load4(t0 + 4);
        float mx = m;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = (t0 + i) * 16 + 4 * q4 + r;
                if (key >= p.NK) s[i][r] = -INFINITY;
                mx = fmaxf(mx, s[i][r]);
            }
        float rs = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) rs += __builtin_amdgcn_exp2f(s[i][r] - mx);
        l = l * __builtin_amdgcn_exp2f(m - mx) + rs;
        m = mx;
    }
    // the four lanes of a query (q4 = 0..3) hold disjoint key sets: merge (fixed order)
#pragma unroll
    for (int d = 16; d <= 32; d <<= 1) {
        const float mo = __shfl_xor(m, d), lo = __shfl_xor(l, d);
        const float mn = fmaxf(m, mo);
        // both partners compute the same two products; add them in the order (lower lane, upper lane) so the result is symmetric
        const float a = l * __builtin_amdgcn_exp2f(m - mn), b = lo * __builtin_amdgcn_exp2f(mo - mn);
        l = (lane & d) ? b + a : a + b;
        m = mn;
    }
    if (q4 == 0 && q0 + l15 < p.N) {
        float2* st = (float2*)p.stats + (size_t)h * p.N + q0 + l15;
        *st = make_float2(m, 1.f / ((float)p.H * l));
    }
}

template <int DP>
__global__ __launch_bounds__(256) void attn_store_apply_kernel(AttnStoreArgs p) {
    constexpr int KSN = DP / 32;
    constexpr int MT = 8;                                            // 128 keys per workgroup
    typedef __attribute__((ext_vector_type(4))) float f32x4_t;
    __shared__ f32x4_t part[4][MT][64];                              // 32 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q4 = lane >> 4;
    const int q0 = blockIdx.x * 16, t0 = blockIdx.y * MT;
    const int qrow = q0 + l15 < p.N ? q0 + l15 : p.N - 1;
    const bf16_t* qbase = p.Q + ((size_t)p.q_row0 + qrow) * p.ldq + q4 * 8;
    const bf16_t* kbase[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int krow = (t0 + i) * 16 + l15; if (krow > p.NKrows - 1) krow = p.NKrows - 1;
        kbase[i] = p.K + ((size_t)p.k_row0 + krow) * p.ldk + q4 * 8;
    }
    const float2* stat = (const float2*)p.stats + qrow;
    f32x4_t acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8 kf[MT][KSN], qf[KSN];
    float2 ms = make_float2(0.f, 0.f);
    auto load_head = [&](int h) {
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks) qf[ks] = *(const bf16x8*)(qbase + h * DP + ks * 32);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int ks = 0; ks < KSN; ++ks) kf[i][ks] = *(const bf16x8*)(kbase[i] + h * DP + ks * 32);
        ms = stat[(size_t)h * p.N];
    };
    if (wave < p.H) load_head(wave);
    for (int h = wave; h < p.H; h += 4) {
        f32x4_t s[MT];
        const float m = ms.x, sc = ms.y;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            s[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KSN; ++ks) s[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[i][ks], qf[ks], s[i], 0, 0, 0);
        }
        if (h + 4 < p.H) load_head(h + 4);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = (t0 + i) * 16 + 4 * q4 + r;
                const float e = key < p.NK ? __builtin_amdgcn_exp2f(s[i][r] - m) * sc : 0.f;
                acc[i][r] += e;
            }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) part[wave][i][lane] = acc[i];
    __syncthreads();
    // wave w finishes tiles 2w, 2w + 1: partial sums of the four head groups in the order 0, 1, 2, 3
    if (q0 + l15 < p.N) {
        float* orow = p.out + (size_t)(q0 + l15) * p.NK;
        const bool vec = (p.NK & 3) == 0 && (((uintptr_t)p.out) & 15) == 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = wave * 2 + j;
            f32x4_t v = part[0][i][lane];
            v += part[1][i][lane]; v += part[2][i][lane]; v += part[3][i][lane];
            const int key = (t0 + i) * 16 + 4 * q4;
            if (vec && key + 4 <= p.NK) {
                float4* d = (float4*)(orow + key);
                float4 o = p.overwrite ? make_float4(0.f, 0.f, 0.f, 0.f) : *d;
                o.x += v[0]; o.y += v[1]; o.z += v[2]; o.w += v[3];
                *d = o;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (key + r < p.NK) orow[key + r] = p.overwrite ? v[r] : orow[key + r] + v[r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- round 5: apply kernel, K through LDS
// attn_store_apply_kernel fetches every MFMA operand from global memory row-per-lane (a 16-query workgroup pulls 128 keys x H heads
// through its vector L1: 25 us per 32x32 map, latency bound).  Here a workgroup owns 64 queries x 32 keys and walks the heads: the
// head's K tile (32 keys x 64 d = 4 KB) arrives in LDS by ONE coalesced LDS-DMA piece per wave, two heads ahead (3 buffers, one
// barrier per head); wave w owns queries 16 w .. + 15 for ALL heads, so the head sum stays in its registers (no cross-wave
// reduction) and the order of the sum is h = 0, 1, 2, ... for every element.  d = 64 (DP = 64) only; other head dims keep the
// round-4 kernel.
#define AS2_MT 2                      // 16-key tiles per workgroup
__global__ __launch_bounds__(256) void attn_store_apply2_kernel(AttnStoreArgs p) {
    constexpr int DP = 64, MT = AS2_MT;
    typedef __attribute__((ext_vector_type(4))) float f32x4_t;
    __shared__ __attribute__((aligned(16))) char kbuf[3][MT * 16 * 128];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q4 = lane >> 4;
    const int q0 = blockIdx.x * 64 + wave * 16, t0 = blockIdx.y * MT;
    const int qrow = q0 + l15 < p.N ? q0 + l15 : p.N - 1;
    const bf16_t* qbase = p.Q + ((size_t)p.q_row0 + qrow) * p.ldq + q4 * 8;
    const float2* stat = (const float2*)p.stats + qrow;
    // this wave's LDS-DMA piece of a head's K tile: rows 8 wave .. + 7 (MT * 16 = 32 rows = 4 pieces), 16-B slots XOR-swizzled
    const int prow = wave * 8 + (lane >> 3), pslot = lane & 7;
    int krow = t0 * 16 + prow; if (krow > p.NKrows - 1) krow = p.NKrows - 1;
    const int kvoff = (int)(((size_t)krow * p.ldk + ((pslot ^ ((prow >> 1) & 7)) << 3)) * 2);
    const bf16_t* kbase = p.K + (size_t)p.k_row0 * p.ldk;
    bf16x8 qf[3][2];
    float2 ms[3];
    auto fetch = [&](int h, int slot) {
        static_assert(MT == 2, "32 rows = one 8-row piece per wave");
        glds16_buf(kbase, kvoff, h * DP * 2, kbuf[slot] + wave * 1024);
        qf[slot][0] = *(const bf16x8*)(qbase + h * DP); qf[slot][1] = *(const bf16x8*)(qbase + h * DP + 32);
        ms[slot] = stat[(size_t)h * p.N];
    };
    f32x4_t acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    fetch(0, 0);
    if (p.H > 1) fetch(1, 1); else fetch(0, 1);
    const int key = (l15 >> 1) & 7;
    const int c0 = l15 * 128 + ((q4 ^ key) << 4), c1 = l15 * 128 + (((4 + q4) ^ key) << 4);
    // three heads per trip: static buffer / register indices
    auto head = [&](int h, auto slot_c) {
        constexpr int SL = decltype(slot_c)::value;
        // loads issued behind head h's: head h + 1's piece, 2 Q fragments and statistics pair
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                               // head h's tile landed; every wave left head h - 1's buffer ...
        fetch(h + 2 < p.H ? h + 2 : p.H - 1, (SL + 2) % 3);        // ... which head h + 2 takes (beyond the last head: a harmless re-fetch keeps the cadence)
        f32x4_t s[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            s[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(kbuf[SL] + i * 2048 + c0), qf[SL][0], f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            s[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(kbuf[SL] + i * 2048 + c1), qf[SL][1], s[i], 0, 0, 0);
        }
        const float m = ms[SL].x, sc = ms[SL].y;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kk = (t0 + i) * 16 + 4 * q4 + r;
                acc[i][r] += kk < p.NK ? __builtin_amdgcn_exp2f(s[i][r] - m) * sc : 0.f;
            }
    };
    int h = 0;
    for (; h + 3 <= p.H; h += 3) {
        head(h, std::integral_constant<int, 0>{}); head(h + 1, std::integral_constant<int, 1>{}); head(h + 2, std::integral_constant<int, 2>{});
    }
    if (h < p.H) { head(h, std::integral_constant<int, 0>{}); ++h; }
    if (h < p.H) { head(h, std::integral_constant<int, 1>{}); ++h; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the tail re-fetches: nothing may land in LDS after the workgroup left
    if (q0 + l15 < p.N) {
        float* orow = p.out + (size_t)(q0 + l15) * p.NK;
        const bool vec = (p.NK & 3) == 0 && (((uintptr_t)p.out) & 15) == 0;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int kk = (t0 + i) * 16 + 4 * q4;
            if (vec && kk + 4 <= p.NK) {
                float4* d = (float4*)(orow + kk);
                float4 o = p.overwrite ? make_float4(0.f, 0.f, 0.f, 0.f) : *d;
                o.x += acc[i][0]; o.y += acc[i][1]; o.z += acc[i][2]; o.w += acc[i][3];
                *d = o;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kk + r < p.NK) orow[kk + r] = p.overwrite ? acc[i][r] : orow[kk + r] + acc[i][r];
            }
        }
    }
}

int g_store_apply_v1 = 0;     // debug bit 18: round 4's apply kernel for every head dim
int g_store_own_stats = 0;    // debug bit 17: the store computes the statistics itself although the attention launch could leave them
// would launch_attn_store run the statistics + apply pair for this map?  (the caller may then let the self-attention launch write the statistics)
bool attn_store_takes_stats(int N, int NK, int DP);
int g_store_legacy = 0;      // A/B timing (rt_op_gemm_debug): bit 0 (debug bit 5) the round-1 two-pass single-wave kernel for every shape; bit 1 (debug bit 6) no statistics + apply pair
void launch_attn_store(const AttnStoreArgs& a, hipStream_t st) {
    RT_REQUIRE(a.NKpad % 32 == 0 && a.NK >= 1 && a.NK <= a.NKpad && a.NKrows >= 1 && a.H >= 1 && a.H <= 32, "attn_store: keys are padded to a multiple of 32; at most 32 heads");
    if (a.stats && a.NKpad >= 256 && a.NKpad <= 1024 && a.NKpad % 16 == 0 && a.DP <= 96 && !(g_store_legacy & 3)) {
        // large maps: statistics pass + key-split accumulation (see above)
        dim3 gs(cdiv(a.N, 64), a.H), ga(cdiv(a.N, 16), cdiv(a.NKpad, 128)), blk(256);
        switch (a.DP) {
            // stats_ready: the self-attention launch of this layer left the statistics (AttnArgs.stats): the statistics pass is skipped
            case 32: if (!a.stats_ready) hipLaunchKernelGGL(attn_store_stats_kernel<32>, gs, blk, 0, st, a); hipLaunchKernelGGL(attn_store_apply_kernel<32>, ga, blk, 0, st, a); break;
            case 64:
                if (!a.stats_ready) hipLaunchKernelGGL(attn_store_stats_kernel<64>, gs, blk, 0, st, a);
                if (!g_store_apply_v1 && (long)a.NKrows * a.ldk * 2 < 0x7fffffffL) hipLaunchKernelGGL(attn_store_apply2_kernel, dim3(cdiv(a.N, 64), cdiv(a.NKpad, 16 * AS2_MT)), blk, 0, st, a);
                else hipLaunchKernelGGL(attn_store_apply_kernel<64>, ga, blk, 0, st, a);
                break;
            case 96: if (!a.stats_ready) hipLaunchKernelGGL(attn_store_stats_kernel<96>, gs, blk, 0, st, a); hipLaunchKernelGGL(attn_store_apply_kernel<96>, ga, blk, 0, st, a); break;
            default: throw rt_error(RT_E_UNSUPPORTED, "attn_store: unsupported padded head dim");
        }
        HIP_CHECK(hipGetLastError());
        return;
    }
    if (a.stats && a.stats_ready && a.DP == 64 && a.NKpad % 32 == 0 && a.NKpad <= 1024 && !g_store_apply_v1 && !(g_store_legacy & 3) &&
        (long)a.NKrows * a.ldk * 2 < 0x7fffffffL) {
        // small maps (the N x 77 cross maps) whose attention launch left the statistics (cross77_kernel, AttnArgs.stats): the apply kernel alone
        hipLaunchKernelGGL(attn_store_apply2_kernel, dim3(cdiv(a.N, 64), cdiv(a.NKpad, 16 * AS2_MT)), dim3(256), 0, st, a);
        HIP_CHECK(hipGetLastError());
        return;
    }
    if (a.NKpad <= 1024 && a.NKpad % 16 == 0 && !(g_store_legacy & 1)) {
        // every map the token-map producer consumes (32x32 self maps, N x 77 cross maps; attention_utils.py:243-248): one pass,
        // 8 waves per 16 query rows
        dim3 grid16(cdiv(a.N, 16)), block16(512);
        switch (a.DP) {
            case 32: hipLaunchKernelGGL(attn_store16_kernel<32>, grid16, block16, 0, st, a); break;
            case 64: hipLaunchKernelGGL(attn_store16_kernel<64>, grid16, block16, 0, st, a); break;
            case 96: hipLaunchKernelGGL(attn_store16_kernel<96>, grid16, block16, 0, st, a); break;
            case 160: hipLaunchKernelGGL(attn_store16_kernel<160>, grid16, block16, 0, st, a); break;
            default: throw rt_error(RT_E_UNSUPPORTED, "attn_store: unsupported padded head dim");
        }
        HIP_CHECK(hipGetLastError());
        return;
    }
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

bool attn_store_takes_stats(int N, int NK, int DP) {
    (void)N;
    return !g_store_own_stats && NK >= 256 && NK <= 1024 && NK % 32 == 0 && (DP == 32 || DP == 64 || DP == 96) && !(g_store_legacy & 3);
}
