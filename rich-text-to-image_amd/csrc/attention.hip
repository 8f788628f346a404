// Fused attention (QK^T -> softmax -> PV) for gfx950, bf16 MFMA 32x32x16, fp32 softmax/accumulate.
//
// Replaces Attention.get_attention_scores + bmm(P, V) of the reference
// (models/attention_processor.py:359-407 and :476-545) for
//   * self-attention (attn1), incl. the rich-text *injection* mode: the reference stores the full
//     per-head probability tensor of the `text_ref` forward (region_diffusion_sdxl.py:1064-1106) and
//     feeds it to the region forwards (:1018-1062, attention_processor.py:522-524).  softmax(QK^T) only
//     depends on Q,K, so injecting P_ref is identical to attending with (Q_ref, K_ref, V_region): the
//     kernel takes Q/K rows from batch entry q_src[b]/k_src[b] and V^T columns from v_src[b] and never
//     materialises P.
//   * cross-attention (attn2) with the font-size re-weighted softmax (attention_processor.py:386-401):
//       e_k = exp(s_k - max) * |fs_k| ;  p_k = sign(fs_k) * e_k / sum_k e_k
//     expressed as two per-key multiplier vectors (wabs, wsgn); padded keys have wabs = 0.
//
// Layout: "swapped" products so that softmax statistics are lane-local:
//   S^T[key, query] = K * Q^T   (A = K tile from LDS, B = Q fragments held in registers)
//   O^T[d,   query] = V^T * P^T (A = V^T tile from LDS, B = P^T straight from the S^T accumulators:
//                                the MFMA k-slot (lane>>5, j) is bound to the same key on both operands,
//                                so no cross-lane shuffle of P is needed)
// One workgroup = 4 waves x 32 queries; K and V^T tiles are staged with global_load_lds into a double
// buffer of 64-byte-row sub-tiles, XOR-swizzled on 16-B slots via the source address (register staging was
// measured and is slower here: 376-423 vs 700 TFLOP/s).  Built with -amdgpu-mfma-vgpr-form (Makefile).
#include "common.h"
#include <math.h>
#include <type_traits>
#include <utility>

template <int... Is, class F>
static __device__ __forceinline__ void att_unroll(std::integer_sequence<int, Is...>, F& f) { (f(std::integral_constant<int, Is>{}), ...); }

#ifdef RT_ATTN_TIMING
__device__ long long g_attn_times[4 * 8];
#define AT_T(i) { const long long now_ = __builtin_readcyclecounter(); tacc[i] += now_ - tlast; tlast = now_; }
#else
#define AT_T(i)
#endif
// RAGGED (self-attention only): NK is not a multiple of the key tile (token grids like 12x12 or 20x20): the last tile re-reads
// clamped rows / 8-key chunks (always valid memory of this batch entry) and masks the keys >= NK.  NK % 8 == 0 is required so
// that the 16-B V^T chunks of every batch entry stay aligned.
// NW = waves per workgroup (4 or 8): every wave owns 32 queries and all of them share the staged K / V^T tiles, so an 8-wave
// workgroup issues half the LDS-DMA pieces per query (2 instead of 4 per wave and 64-key tile) and halves the L2 -> LDS bytes.
// FOLD (self-attention): the running reference m of the online softmax is subtracted BY THE MATRIX PIPE: one more MFMA k step per
// 32-key sub-tile with A = (1, 0, ...) for every key and B = (-m, 0, ...) for the lane's query, so the accumulators come out as
// s - m and the 32 v_sub per tile and wave disappear from the VALU stream that bounds this kernel (LABNOTES 4.3).  m is kept
// bf16-representable (softmax is invariant to the reference; it only has to stay within 2^8 of the true running maximum), so the
// product 1 * (-m) is exact.
// G > 1 (round 6, self-attention of an INJECTED step): the streams of a unit attend with the SAME (Q, K) - text_ref and the region streams
// that take its probabilities (attention_processor.py:522-524: the reference does not recompute softmax(QK^T) for them either) - and differ
// in V only.  The workgroup computes S^T and P^T ONCE per key tile and runs O_g^T += V_g^T P^T for every member g < G: per member the
// same MFMA sequence on the same operands as the one-stream kernel, i.e. bit-identical outputs, for 1 / G of the softmax VALU work (which
// bounds this kernel) and (1 + G) / 2 G of the MFMAs.  The V^T tiles of the members sit behind the K tile in each stage.
template <int DP, int KT, bool CROSS, bool RAGGED, int NW, bool FOLD, int PRIO, int G>
static __device__ __forceinline__ void attn_tile(const AttnArgs& p, const int b, const int h, const int q0) {
    static_assert(G == 1 || (!CROSS && !RAGGED && NW == 4), "shared-probability units: plain self-attention");
    constexpr int NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE = KT * DP * 2;          // bytes of one K (or V^T) tile
    constexpr int STAGE = (1 + G) * TILE;
    constexpr int NCH = KT * DP / 8;           // 16-B chunks per tile
    constexpr int NJ = KT / 32;                // key sub-tiles
    constexpr int ND = DP / 32;                // d sub-tiles
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    if constexpr (PRIO == 3) {                                        // (probe) a fixed priority per wave of the workgroup: 0, 1, 2, 3
        if (wave == 1) __builtin_amdgcn_s_setprio(1);
        else if (wave == 2) __builtin_amdgcn_s_setprio(2);
        else if (wave == 3) __builtin_amdgcn_s_setprio(3);
    }
    const int qb = p.q_src[b], kb = p.k_src[b];
    // members of the unit: V^T source and output batch entry of member g (one-stream launches: the unit IS batch entry b)
    // (the member count is the template parameter, not a run-time bound: behind a run-time `g < ng` every member's two V^T fragment reads,
    //  their wait and their two MFMAs became a basic block of their own - the LDS latency of every pair exposed, the shared kernel no faster
    //  than four one-stream passes; profiles/r6_attn_units_bench_v1.txt)
    int vb[G], ob[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { vb[g] = p.gvs[b][g]; ob[g] = p.gob[b][g]; }

    float* wl = (float*)(smem + 2 * STAGE);    // cross: [2][KT] multipliers
    // wset[b] < 0: plain softmax over the nk_valid keys - no multiplier tables (6 of the 7 streams of a rich-text step); the 48
    // table reads + 2 x 48 multiplies per lane are a quarter of this single-tile kernel's instructions
    const bool plain = CROSS && p.wset[b] < 0;                      // workgroup-uniform
    if (CROSS && !plain) {
        const int ws = p.wset[b];
        for (int i = tid; i < KT; i += NT) {
            wl[i] = p.wabs[ws * p.NK + i];
            wl[KT + i] = p.wsgn[ws * p.NK + i];
        }
    }

    // Q fragments (B operand of S^T): query = lane&31, d = 16*ks + 8*hi + [0,8)
    const int q = q0 + wave * 32 + l31;
    const int qc = q < p.N ? q : p.N - 1;
    const bf16_t* qptr = p.Q + ((size_t)qb * p.N + qc) * p.ldq + h * DP + hi * 8;
    bf16x8 qf[DP / 16];
#pragma unroll
    for (int ks = 0; ks < DP / 16; ++ks) qf[ks] = *(const bf16x8*)(qptr + ks * 16);

    const bf16_t* kbase = p.K + (size_t)kb * p.NK * p.ldk + h * DP;
    const bf16_t* vbase[G];
#pragma unroll
    for (int g = 0; g < G; ++g) vbase[g] = p.VT + (size_t)h * DP * p.ldvt + (size_t)vb[g] * p.NK;

    // Self-attention on whole key tiles: the pieces go through a buffer descriptor (buffer_load ... offen lds) - per piece ONE 32-bit VGPR
    // byte offset computed here, once, and a SCALAR offset that advances with the key tile, instead of a 64-bit address rebuilt per piece
    // and tile (35 of the ~170 VALU / SALU instructions of a key tile; round 6).
    constexpr bool BUFST = !RAGGED && !CROSS;
    constexpr int NIT = (NCH + NT - 1) / NT;
    int kvoff[BUFST ? NIT : 1], vvoff[BUFST ? NIT : 1];
    if constexpr (BUFST) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * NT + tid, rl = idx >> 2, ps = idx & 3;
            { const int sub = rl / KT, row = rl - sub * KT, ls = ps ^ ((row >> 2) & 3); kvoff[it] = (row * p.ldk + sub * 32 + ls * 8) * 2; }
            { const int sub = rl / DP, row = rl - sub * DP, ls = ps ^ ((row >> 2) & 3); vvoff[it] = (row * p.ldvt + sub * 32 + ls * 8) * 2; }
        }
    }
    auto stage = [&](int s, int key0) {
        char* ks_ = smem + s * STAGE;
        char* vs_ = ks_ + TILE;
        if constexpr (BUFST) {
            const int ksoff = key0 * p.ldk * 2, vsoff = key0 * 2;      // scalars
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                if (it * NT + tid < NCH) {   // wave-uniform: NCH % 64 == 0
                    glds16_buf(kbase, kvoff[it], ksoff, ks_ + (it * NT + wave * 64) * 16);
#pragma unroll
                    for (int g = 0; g < G; ++g) glds16_buf(vbase[g], vvoff[it], vsoff, vs_ + g * TILE + (it * NT + wave * 64) * 16);
                }
            }
            return;
        }
#pragma unroll
        for (int c0 = 0; c0 < NCH; c0 += NT) {
            const int idx = c0 + tid;
            if (idx < NCH) {   // wave-uniform: NCH % 64 == 0
                const int rl = idx >> 2, ps = idx & 3;
                {
                    const int sub = rl / KT, row = rl - sub * KT;
                    const int ls = ps ^ ((row >> 2) & 3);
                    int kr = key0 + row; if (RAGGED && kr > p.NK - 1) kr = p.NK - 1;
                    glds16(kbase + (size_t)kr * p.ldk + sub * 32 + ls * 8, ks_ + (c0 + wave * 64) * 16);
                }
                {
                    const int sub = rl / DP, row = rl - sub * DP;
                    const int ls = ps ^ ((row >> 2) & 3);
                    int kc = key0 + sub * 32 + ls * 8; if (RAGGED && kc > p.NK - 8) kc = p.NK - 8;
#pragma unroll
                    for (int g = 0; g < G; ++g) glds16(vbase[g] + (size_t)row * p.ldvt + kc, vs_ + g * TILE + (c0 + wave * 64) * 16);
                }
            }
        }
    };

    f32x16 o[G][ND];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[g][dt][r] = 0.f;
    float m = FOLD ? 0.f : -1e30f, l = 0.f;
    bf16x8 ka, qm;                                                  // FOLD: the constant key-side fragment and the query-side (-m) fragment
#pragma unroll
    for (int e = 0; e < 8; ++e) { ka[e] = (__bf16)0.f; qm[e] = (__bf16)0.f; }
    if (FOLD && hi == 0) ka[0] = (__bf16)1.f;

    const int ntile = (p.NK + KT - 1) / KT;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef RT_ATTN_TIMING
    long long tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
    for (int kt = 0; kt < ntile; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < ntile) stage(cur ^ 1, (kt + 1) * KT);
        AT_T(0)
        const char* ks_ = smem + cur * STAGE;
        const char* vs_ = ks_ + TILE;

        // ---- S^T = K Q^T
        // Wave priority: the four co-resident waves of a SIMD are in different phases; a wave in an MFMA phase issues sparsely (one
        // MFMA per 32 - 64 cycles of matrix-pipe time) while the waves in their softmax keep the issue port busy with quarter-rate
        // v_exp_f32.  Raised priority for the MFMA phases lets those sparse issues go first, so the matrix pipe runs beside the other
        // waves' VALU work instead of queueing behind it.
        if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(2);
        if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(0);
        f32x16 s[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            s[j] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // folds into the first MFMA's inline-constant C operand
            // MFMA row i of this 32-key sub-tile reads key perm(i) = i with bits 2 and 3 swapped: the accumulator register
            // r of lane half `hi` then holds key 16*(r>>3) + 8*hi + (r&7), i.e. each run of 8 registers is 8 CONSECUTIVE
            // keys and the matching V^T fragment is a single ds_read_b128 (no register shuffling in the P.V loop).
            const int row = j * 32 + ((l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1));
            const int key = (row >> 2) & 3;
#pragma unroll
            for (int ks = 0; ks < DP / 16; ++ks) {
                const int ls = ((ks & 1) * 2 + hi) ^ key;
                const bf16x8 kf = *(const bf16x8*)(ks_ + ((ks >> 1) * KT + row) * 64 + ls * 16);
                s[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[j], 0, 0, 0);
            }
            if constexpr (FOLD) s[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qm, s[j], 0, 0, 0);      // s - m
        }
        if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(0);
        if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(2);                   // (probe) the opposite: the softmax phases first
        AT_T(1)
        // ---- online softmax (lane holds 16 of the 32 keys of each sub-tile for its query: 8*hi + [0,8) and 16 + 8*hi + [0,8); partner = lane^32)
        float mx = FOLD ? (kt == 0 ? -INFINITY : 0.f) : m;          // FOLD: s is already relative to m
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (CROSS || RAGGED) {
                    // (uniform test first: only the sub-tiles that reach past the last valid key pay the per-element compare)
                    if (kt * KT + (j + 1) * 32 > (CROSS ? p.nk_valid : p.NK)) {
                        const int key = kt * KT + j * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
                        if (key >= (CROSS ? p.nk_valid : p.NK)) s[j][r] = -INFINITY;
                    }
                }
                mx = fmaxf(mx, s[j][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        // Deferred rescale (guide T13): keep the old running max while it grew by <= 8 (log2 domain): P is then bounded
        // by 2^8 instead of 1, which bf16 P / fp32 accumulation absorb, and the O / l rescale (32 accumulator
        // read-modify-writes per tile) is skipped for almost every tile.  The decision is taken before this tile's P
        // is exponentiated and after the previous tile's P.V completed, so everything at the old scale is scaled once.
        if constexpr (FOLD) {
            if (kt == 0 || !__all(mx <= 8.f)) {
                const float mn = bf16_to_f32(f32_to_bf16(m + mx));   // new reference, exact as an MFMA operand
                const float delta = mn - m;
                m = mn;
                if (hi == 0) qm[0] = (__bf16)(-mn);
                if (kt > 0) {                                        // (first tile: O = l = 0, and delta may be negative: 2^-delta overflows)
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
                    l *= alpha;
#pragma unroll
                    for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[g][dt][r] *= alpha;
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[j][r] -= delta;   // this tile was produced against the old reference
            }
        } else if (!__all(mx - m <= 8.f)) {
            const float alpha = __builtin_amdgcn_exp2f(m - mx);
            m = mx;
            l *= alpha;
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
            for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[g][dt][r] *= alpha;
        }
        float rs = 0.f;
        if (CROSS && !plain) {                                       // font-size stream: e_k = exp(s_k - max) |fs_k|, p_k = sign(fs_k) e_k / sum e
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kl = j * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
                    float pv = __builtin_amdgcn_exp2f(s[j][r] - m) * wl[kl];
                    rs += pv;
                    s[j][r] = pv * wl[KT + kl];
                }
        } else {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(FOLD ? s[j][r] : s[j][r] - m);      // v_exp_f32: argument <= 8, underflow flushes to 0
                    rs += pv;                                      // (v_pk_add_f32 pairs measured slower: 131 extra v_mov)
                    s[j][r] = pv;
                }
        }
        l += rs;
        AT_T(2)

        // ---- O^T += V^T P^T
        if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(2);
        if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(0);
        if constexpr (G == 1) {
#pragma unroll
            for (int kk = 0; kk < KT / 16; ++kk) {
                const int j = kk >> 1, hh = kk & 1;
                bf16x8 pf;
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[e] = (__bf16)s[j][8 * hh + e];
#pragma unroll
                for (int dt = 0; dt < ND; ++dt) {
                    const int row = dt * 32 + l31;
                    const int key = (row >> 2) & 3;
                    const bf16x8 vf = *(const bf16x8*)(vs_ + (j * DP + row) * 64 + (((2 * hh + hi) ^ key) << 4));
                    o[0][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[0][dt], 0, 0, 0);
                }
            }
        } else {
            // (16-key chunk kk, member g) steps in kk-major order; the two V^T fragments of step i + 1 are requested in front of step i's MFMAs.
            // Inline asm for the reads and their counted waits: as compiler-visible loads hipcc sinks every pair of reads to its two MFMAs
            // (one fragment set, s_waitcnt lgkmcnt(0) in front of every pair - the LDS latency of all 16 steps of a key tile exposed) whatever
            // the source order or the scheduling barriers say.  The waits name the fragments as in / out operands, which ties the MFMAs to them.
            static_assert(ND == 2 && DP == 64 && KT == 64, "shared-probability units: d = 64, 64-key tiles");
            constexpr int NP = (KT / 16) * G;
            const int vkey = (l31 >> 2) & 3;
            const unsigned vs_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)vs_;
            const unsigned va0 = vs_lds + l31 * 64 + ((hi ^ vkey) << 4), va1 = vs_lds + l31 * 64 + (((2 + hi) ^ vkey) << 4);
            bf16x8 vfb[2][ND];
#define ATT_LDV(I_, DST_)                                                                                                      \
            {                                                                                                                  \
                constexpr int kk_ = (I_) / G, g_ = (I_) - kk_ * G, j_ = kk_ >> 1, hh_ = kk_ & 1;                               \
                asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"                                  \
                             : "=&v"(DST_[0]), "=&v"(DST_[1]) : "v"(hh_ ? va1 : va0), "n"(g_ * TILE + j_ * DP * 64), "n"(g_ * TILE + j_ * DP * 64 + 2048)); \
            }
            ATT_LDV(0, vfb[0])
            bf16x8 pf;
            auto step = [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int kk = i / G, g = i - kk * G, j = kk >> 1, hh = kk & 1;
                if constexpr (i + 1 < NP) {
                    ATT_LDV(i + 1, vfb[(i + 1) & 1])
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(vfb[i & 1][0]), "+v"(vfb[i & 1][1]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vfb[i & 1][0]), "+v"(vfb[i & 1][1]));
                }
                if constexpr (g == 0) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) pf[e] = (__bf16)s[j][8 * hh + e];
                }
#pragma unroll
                for (int dt = 0; dt < ND; ++dt) o[g][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfb[i & 1][dt], pf, o[g][dt], 0, 0, 0);
            };
            att_unroll(std::make_integer_sequence<int, NP>{}, step);
#undef ATT_LDV
        }
        if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(0);
        AT_T(3)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        AT_T(4)
        __syncthreads();
        AT_T(5)
    }
#ifdef RT_ATTN_TIMING
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0) for (int i = 0; i < 6; ++i) g_attn_times[wave * 8 + i] = tacc[i];
#endif

    l += __shfl_xor(l, 32);
    const float inv = 1.f / l;
    if constexpr (!CROSS) {
        // token-map capture (plain pass): P(q, k) = exp2(s - m) / l for every key of this head - the deferred rescale keeps m and l consistent
        if (p.stats != nullptr && ob[0] == p.stats_b && hi == 0 && q < p.N)
            ((float2*)p.stats)[(size_t)h * p.N + q] = make_float2(m, inv / (float)p.H);
    }
    // Epilogue through LDS: in the accumulator layout a lane owns ONE query row and 4-element runs of d, so direct stores touch
    // 32 rows with 8 B each per instruction (store-issue-bound tail, guide T21).  Every wave transposes its 32 x DP tile in its
    // own slab of the (now idle) K/V buffers - the loop ended with a barrier - and writes whole 16-B chunks, 8 lanes per row.
    {
        constexpr int RS = DP * 2 + 16;                          // slab row stride (16-B pad)
        constexpr int CPR = DP * 2 / 16;                          // 16-B chunks per row
        char* slab = smem + wave * 32 * RS;
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
            for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    uint2 v;
                    v.x = pack_bf16x2(o[g][dt][4 * gg + 0] * inv, o[g][dt][4 * gg + 1] * inv);
                    v.y = pack_bf16x2(o[g][dt][4 * gg + 2] * inv, o[g][dt][4 * gg + 3] * inv);
                    *(uint2*)(slab + l31 * RS + (dt * 32 + 8 * gg + 4 * hi) * 2) = v;
                }
            // LDS operations of one wave execute in order: no barrier between the slab write and read (nor before the next member's write)
#pragma unroll
            for (int idx0 = 0; idx0 < 32 * CPR; idx0 += 64) {
                const int idx = idx0 + lane;
                const int r = idx / CPR, ch = idx - r * CPR;
                const int qq = q0 + wave * 32 + r;
                if (r < 32 && qq < p.N)
                    *(uint4*)(p.O + ((size_t)ob[g] * p.N + qq) * p.ldo + h * DP + ch * 8) = *(const uint4*)(slab + r * RS + ch * 16);
            }
        }
    }
}

// The kernel: G = 1 - the one-stream launches of rounds 1 - 5; G > 1 - a launch that mixes shared units of exactly G members (ng[b] == G)
// with one-stream units (ng[b] == 1), so that the few heavy workgroups of the shared units run beside the light ones instead of leaving
// CUs idle in a launch of their own (one unit of four at 1024 tokens x 20 heads is 160 workgroups for 256 CUs).  Registers and LDS are
// those of the G-member body: 3 (G = 2) or 2 (G >= 3) workgroups per CU; the one-stream body measured the same at 3 and 8 - 10 % slower
// at 2 (profiles/r6_attn_units_bench_v2.txt).
// XCD-aware work mapping.  Workgroups are dealt to the 8 XCDs round-robin by linear id; the query blocks of one (batch entry,
// head) all stream the same K / V^T, so they must share an L2: give every XCD a contiguous run of the (h, b, query block)
// space, query block fastest (bijective for any grid size).  Before this the 8 query blocks of a 1024-token head sat on 8
// different XCDs and every XCD fetched every head's K/V: 4.2x the algorithmic HBM traffic (profiles/r1_pmc_traffic.json).
template <int DP, int KT, bool CROSS, bool RAGGED = false, int NW = 4, bool FOLD = !CROSS, int PRIO = 0, int G = 1>
__global__ __launch_bounds__(NW * 64, G >= 3 ? 2 : (G == 2 || (CROSS && DP <= 64)) ? 3 : 1) void attn_kernel(AttnArgs p) {
    int b, h, q0;
    {
        const int nwg = gridDim.x;
        int bid = blockIdx.x;
        const int qq = nwg >> 3, rr = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
        const int qb = bid % p.nqb; bid /= p.nqb;          // order (head, batch entry, query block): streams that attend with the
        b = bid % p.B; h = bid / p.B;                          // same Q/K source (injection) are neighbours and share K in L2 too
        q0 = qb * (32 * NW);
    }
    if constexpr (G == 1) attn_tile<DP, KT, CROSS, RAGGED, NW, FOLD, PRIO, 1>(p, b, h, q0);
    else if (p.ng[b] == G) attn_tile<DP, KT, CROSS, RAGGED, NW, FOLD, PRIO, G>(p, b, h, q0);      // workgroup-uniform
    else attn_tile<DP, KT, CROSS, RAGGED, NW, FOLD, PRIO, 1>(p, b, h, q0);
}

// s_setprio 2 in the MFMA phases / 0 in the softmax (PRIO = 1): 942 -> 950 TFLOP/s at 4096 tokens, 716 -> 732 at 1024 stand-alone, the
// self-attention class of a step 13.19 -> 12.75 ms per two steps (profiles/r4_attn_probe_setprio.txt, r4_ab_attn_setprio.jsonl).  The
// opposite assignment (softmax first) measured 915 / 699, a fixed priority per wave 800 / 568.  rt_op_gemm_debug bit 14 switches it off.
int g_attn_prio = 1;
extern int g_attn_units;
void attention_set_prio(int on) { g_attn_prio = on ? 1 : 0; }
template <int DP, int KT, bool CROSS, bool RAGGED, int NW, bool FOLD, int PRIO, int G = 1>
static void launch_tp(const AttnArgs& a, hipStream_t st) {
    // K / V^T double buffer (+ the cross-attention multipliers); the epilogue reuses it as NW slabs of 32 x (DP*2 + 16) bytes
    size_t lds = 2 * (1 + G) * (size_t)KT * DP * 2 + (CROSS ? 2 * KT * sizeof(float) : 0);
    const size_t slabs = (size_t)NW * 32 * (DP * 2 + 16);
    if (slabs > lds) lds = slabs;
    if (G == 1 && !CROSS && g_attn_units == 6) lds = 80 * 1024;      // (probe) the one-stream kernel at two workgroups per CU
    if (G == 1 && !CROSS && g_attn_units == 7) lds = 48 * 1024;      // (probe) ... at three
    static size_t attr = 0;
    if (attr < lds) {
        HIP_CHECK(hipFuncSetAttribute((const void*)attn_kernel<DP, KT, CROSS, RAGGED, NW, FOLD, PRIO, G>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = lds;
    }
    AttnArgs aa = a;
    aa.nqb = cdiv(a.N, 32 * NW);
    dim3 grid(aa.nqb * a.H * a.B), block(NW * 64);
    hipLaunchKernelGGL((attn_kernel<DP, KT, CROSS, RAGGED, NW, FOLD, PRIO, G>), grid, block, lds, st, aa);
    HIP_CHECK(hipGetLastError());
}
template <int DP, int KT, bool CROSS, bool RAGGED = false, int NW = 4, bool FOLD = !CROSS>
static void launch_t(const AttnArgs& a, hipStream_t st) {
#ifdef RT_PROBE
    if (g_attn_prio == 2) { launch_tp<DP, KT, CROSS, RAGGED, NW, FOLD, 2>(a, st); return; }
    if (g_attn_prio == 3) { launch_tp<DP, KT, CROSS, RAGGED, NW, FOLD, 3>(a, st); return; }
#endif
    if (g_attn_prio) launch_tp<DP, KT, CROSS, RAGGED, NW, FOLD, 1>(a, st);
    else launch_tp<DP, KT, CROSS, RAGGED, NW, FOLD, 0>(a, st);
}
// shared-probability units (attn_kernel, "G > 1"): d = 64 self-attention, whole key tiles
template <int G>
static void launch_units(const AttnArgs& a, hipStream_t st) {
    if (g_attn_prio) launch_tp<64, 64, false, false, 4, true, 1, G>(a, st);
    else launch_tp<64, 64, false, false, 4, true, 0, G>(a, st);
}
// Self-attention launches in which several batch entries attend with the same (Q, K) source - an injected rich-text step: text_ref and the
// region streams, models/region_diffusion_sdxl.py:1018-1106 - are re-expressed as UNITS: the members of a source are dealt G at a time
// into shared units (G = 4, or the size of the largest source group when that is 2 or 3), what is left over and every other entry is a
// one-stream unit; ONE launch of the G-member kernel runs both kinds, the shared units first within every head.  Bit-identical per stream
// with the one-stream launches (tests/test_kernels_gpu.py); the partition depends on the launch's own source indices only.
//   mode 1: never (the launches of rounds 1 - 5);  2: as above;  3: shared units of two only;  4 / 5: as 2 / 3 with the one-stream units
//   in a launch of their own (measured: profiles/r6_attn_units_bench_v2.txt).  rt_op_gemm_debug bits 24 - 26 (0 = default).
int g_attn_units = 0;
void attention_set_units(int mode) { g_attn_units = mode; }
// Default (measured on the two SDXL levels of an injected config-3 step, profiles/r6_attn_units_bench_v3.txt: 7 x 1024 tokens x 20 heads
// 54 -> 50 us with pairs in one launch, 54 / 57 with units of four; 7 x 4096 x 10 heads 314 -> 279 us with units of four + the one-stream
// units in their own launch, 287 in one launch, 294 / 280 with pairs): short sequences take pairs beside the one-stream units - a unit of
// four there is 160 heavy workgroups -, long ones units of four in their own launch.  A function of the launch's own shape.
// The plan, host-only (also behind rt_op_attention_units_plan: the rule is testable without a GPU).  Fills `u` (the launch of the G-member
// kernel: shared units first, then - unless `split` - the one-stream units) and `s1` (the one-stream launch of a split plan); returns G, or 0
// when the launch has nothing to share / is outside the shared kernel's domain (d = 64 self-attention on whole 64-key tiles, no statistics).
static int attention_units_plan(const AttnArgs& a, int mode_in, AttnArgs& u, AttnArgs& s1) {
    const int mode = mode_in ? mode_in : (a.N >= 2048 ? 4 : 3);
    if (mode <= 1 || mode >= 6 || a.cross || a.DP != 64 || a.NK % 64 != 0 || a.stats != nullptr) return 0;
    const bool split = mode >= 4;
    // source groups in batch order
    int grp[RT_MAXB][RT_MAXB], gn[RT_MAXB], ngrp = 0, largest = 0;
    bool taken[RT_MAXB] = {};
    for (int b = 0; b < a.B; ++b) {
        if (taken[b]) continue;
        int nm = 0;
        for (int c = b; c < a.B; ++c)
            if (!taken[c] && a.q_src[c] == a.q_src[b] && a.k_src[c] == a.k_src[b]) { grp[ngrp][nm++] = c; taken[c] = true; }
        gn[ngrp++] = nm;
        if (nm > largest) largest = nm;
    }
    if (largest < 2) return 0;
    const int G = (mode == 3 || mode == 5) ? 2 : (largest >= 4 ? 4 : largest);
    u = a; s1 = a;
    int nu = 0, n1 = 0;
    auto put = [&](AttnArgs& d, int& k, int src, const int* mem, int n) {
        d.q_src[k] = a.q_src[src]; d.k_src[k] = a.k_src[src]; d.ng[k] = (unsigned char)n;
        for (int g = 0; g < 4; ++g) { const int c = mem[g < n ? g : 0]; d.gvs[k][g] = (unsigned char)a.v_src[c]; d.gob[k][g] = (unsigned char)c; }
        ++k;
    };
    for (int i = 0; i < ngrp; ++i)                                   // the shared units: the heavy workgroups first
        for (int i0 = 0; i0 + G <= gn[i]; i0 += G) put(u, nu, grp[i][0], &grp[i][i0], G);
    for (int i = 0; i < ngrp; ++i)                                   // what is left of every group: one-stream units
        for (int i0 = gn[i] - gn[i] % G; i0 < gn[i]; ++i0) put(split ? s1 : u, split ? n1 : nu, grp[i][0], &grp[i][i0], 1);
    u.B = nu; s1.B = n1;
    return G;
}
static bool launch_attention_units(const AttnArgs& a, hipStream_t st) {
    AttnArgs u, s1;
    const int G = attention_units_plan(a, g_attn_units, u, s1);
    if (G == 0) return false;
    if (G == 4) launch_units<4>(u, st); else if (G == 3) launch_units<3>(u, st); else launch_units<2>(u, st);
    if (s1.B > 0) launch_t<64, 64, false>(s1, st);
    return true;
}
// Host-only view of the plan (include/rtdiff.h, rt_op_attention_units_plan): per batch entry the launch (0 = the G-member kernel, 1 = the
// one-stream launch of a split plan), the unit inside that launch and the unit's member count.
int attention_units_plan_host(const int* q_src, const int* k_src, int B, int N, int DP, int mode, int* launch_of, int* unit_of, int* members_of) {
    AttnArgs a{}, u, s1;
    a.B = B; a.N = N; a.NK = N; a.DP = DP; a.cross = 0; a.stats = nullptr;
    for (int b = 0; b < B; ++b) { a.q_src[b] = q_src[b]; a.k_src[b] = k_src[b]; a.v_src[b] = b; }
    const int G = attention_units_plan(a, mode, u, s1);
    for (int b = 0; b < B; ++b) { launch_of[b] = G ? -1 : 1; unit_of[b] = b; members_of[b] = 1; }
    if (G == 0) return 0;
    for (int l = 0; l < 2; ++l) {
        const AttnArgs& d = l ? s1 : u;
        for (int k = 0; k < d.B; ++k)
            for (int g = 0; g < d.ng[k]; ++g) { const int b = d.gob[k][g]; launch_of[b] = l; unit_of[b] = k; members_of[b] = d.ng[k]; }
    }
    return G;
}
#ifdef RT_PROBE
int g_attn_nw = 0;          // probe override: 4 or 8 waves per workgroup for the d = 64 self-attention kernel
int g_attn_nofold = 0;      // probe override: the round-2 form (v_sub in the softmax) for A/B timing
#endif

void launch_attention(const AttnArgs& a_in, hipStream_t st) {
    AttnArgs a = a_in;
    for (int b = 0; b < RT_MAXB; ++b) {                              // one-stream launches: the unit is the batch entry
        a.ng[b] = 1;
        for (int g = 0; g < 4; ++g) { a.gvs[b][g] = (unsigned char)(b < a.B ? a.v_src[b] : 0); a.gob[b][g] = (unsigned char)b; }
    }
    RT_REQUIRE(a.B >= 1 && a.B <= RT_MAXB, "attention: batch must be in [1,16]");
    RT_REQUIRE(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldvt % 8 == 0 && a.ldo % 8 == 0, "attention: leading dims");
    RT_REQUIRE(((uintptr_t)a.Q & 15) == 0 && ((uintptr_t)a.K & 15) == 0 && ((uintptr_t)a.VT & 15) == 0 &&
               ((uintptr_t)a.O & 15) == 0, "attention: alignment");
    if (a.cross) {
        RT_REQUIRE(a.NK == 96 && a.nk_valid <= 96 && a.wabs && a.wsgn, "cross-attention expects 77 keys padded to 96");
        switch (a.DP) {
            case 32: launch_t<32, 96, true>(a, st); break;
            case 64: launch_t<64, 96, true>(a, st); break;
            case 96: launch_t<96, 96, true>(a, st); break;
            case 160: launch_t<160, 96, true>(a, st); break;
            default: throw rt_error(RT_E_UNSUPPORTED, "attention: unsupported padded head dim");
        }
    } else {
        RT_REQUIRE(a.NK % 8 == 0 && a.NK >= 8, "self-attention: key count must be a multiple of 8");
        if (launch_attention_units(a, st)) return;
        const bool ragged = a.NK % 64 != 0;
        // 256-query workgroups (NW = 8) halve the LDS-DMA pieces per query but measured 751 vs 802 TFLOP/s in the engine and
        // 908 vs 941 / 642 vs 694 stand-alone (N = 4096 / 1024): the kernel is bound by its softmax VALU work, not by the K / V^T
        // fill, and an 8-wave barrier costs more than the saved copies.  Kept for the probe only.
        bool wide = false; (void)wide;
#ifdef RT_PROBE
        if (g_attn_nw) wide = g_attn_nw == 8;
#endif
        switch (a.DP) {
            case 32: ragged ? launch_t<32, 64, false, true>(a, st) : launch_t<32, 64, false>(a, st); break;
            case 64:
                if (ragged) launch_t<64, 64, false, true>(a, st);
#ifdef RT_PROBE
                else if (wide) launch_t<64, 64, false, false, 8>(a, st);
                else if (g_attn_nofold) launch_t<64, 64, false, false, 4, false>(a, st);
#endif
                else launch_t<64, 64, false>(a, st);
                break;
            case 96: ragged ? launch_t<96, 64, false, true>(a, st) : launch_t<96, 64, false>(a, st); break;
            case 160: ragged ? launch_t<160, 64, false, true>(a, st) : launch_t<160, 64, false>(a, st); break;
            default: throw rt_error(RT_E_UNSUPPORTED, "attention: unsupported padded head dim");
        }
    }
}
