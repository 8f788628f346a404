// bf16 MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950.
//
//   C[M,N] = A[M,K] * W[N,K]^T (+bias, +epilogue)       fp32 accumulate
//
// Replaces every nn.Linear / nn.Conv2d of the reference UNet (models/attention.py:209-304 GEGLU FF,
// models/attention_processor.py:137-152 q/k/v/out projections, models/resnet.py:505-560 conv1/conv2/
// shortcut, models/resnet.py:103-222 up/down-sample convs, models/transformer_2d.py:139-177 proj_in/out).
//
// Structure (CDNA4): BMxBNx64 block tile, WMxWN wavefronts, each wave a grid of v_mfma_f32_32x32x16_bf16
// tiles; both operands are staged HBM -> LDS with global_load_lds_dwordx4 (16 B/lane, no VGPR round trip)
// into an S-deep ring; the LDS image is XOR-swizzled on 16-B slots (slot ^= (row>>1)&7) by permuting the
// per-lane *source* address, so the ds_read_b128 fragment reads are bank-conflict free (guide 5.4 rule 21 /
// T2).  The ring is driven with COUNTED s_waitcnt vmcnt(N) + a raw s_barrier per K tile, so S-2 tiles stay in
// flight across the barrier (guide T3/T4): the bytes in flight per CU, not the MFMA rate, bound this GEMM
// (Little's law against ~2k cycles of L2/MALL latency), hence the large-tile configurations:
//     cfg 0: 128x128, 4 waves (2x2), 2 stages, 2 blocks/CU    small problems, short K
//     cfg 1: 256x128, 8 waves (4x2), 3 stages, 1 block/CU
//     cfg 2: 256x160, 8 waves (8x1), 3 stages, 1 block/CU     N = 1280 / 640 / 320 families (no GEGLU)
//     cfg 3: 256x256, 16 waves (4x4), 2 stages, 1 block/CU    wide N
//     cfg 4: 256x160, 8 compute waves (8x1) + 4 LOADER waves, 3-slot ring (gemm_ws_kernel)
//     cfg 5: 256x128, 8 compute waves (4x2) + 4 loader waves
//     cfg 6: 256x160, 8 waves in two groups one barrier apart (gemm_pp_kernel): long-K problems
//     cfg 7: 256x256, 8 waves (2x4), 8-phase half-tile pipeline (gemm8_kernel): dense, long K
//     cfg 8: 256x320, 8 waves (4x2, 64x160 each), 2 stages: N = 640 (28672x640x640 33.9 vs 38.0 us, x2560 99.9 vs 105-112 us)
// The per-shape choice is made in situ on the real launches (Tuner below); under-filled problems go through split-K slices of the
// 128x128 kernel with a 4-slot ring, eligible 3x3 convolutions through conv3p_kernel.
// (A ping-pong variant with two wave groups half an iteration apart was measured and dropped: 3.2k cycles per
//  K tile against 2.2k here, because the ~1.0k cycles of LDS-DMA issue sit in one of the two phases.)
// All configurations accumulate every output element in the same k order with the same MFMA shape, so the
// result is bit-identical whichever configuration the launcher picks.
// For convolutions the A operand is gathered on the fly from the NHWC bf16 activation (im2col never
// materialised): K index = tap*Cin + c.
#include "common.h"
#include <algorithm>
#include <cmath>
#include <map>
#include <tuple>
#include <type_traits>



#define BK 64
#ifndef RT_ABLATE
#define RT_ABLATE 0
#endif

// ---- epilogue (shared by both main-loop variants), specialised at compile time on the epilogue kind.
// With swapped operands the 32x32 accumulator tile is D[n][m]: m = lane&31 (row of C), n = (r&3) + 8*(r>>2) +
// 4*(lane>>5) (column of C), i.e. a lane owns ONE output row and 4 consecutive columns per register quad.
// Storing straight from that layout makes every store instruction touch 32 different rows with 16-32 B each
// (measured: 7-9 us of the 12 us fixed cost of a launch).  Instead every wave transposes its 32-row slab through
// its private slice of the (now idle) LDS ring and writes / reads HBM in row-contiguous 16-B chunks, so stores
// (and the fp32 residual reads) move whole cache lines.  N % 4 == 0 (fp32) / N % 8 == 0 (bf16) is required.
// PATCH (3x3-conv patch kernel): the tile's 256 rows are a 16x16 pixel patch, local row r -> output row
// prow_base + (r>>4)*pW + (r&15); wrow0 is then the wave's first LOCAL row.
template <int EPI, int TM, int TN, int NWC, int LDS_BYTES, bool PATCH = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x16 (&acc)[TM][TN], int wrow0, int wcol0, int lane, int wave_c,
                                              char* smem, int prow_base = 0, int pW = 0) {
    auto grow = [&](int r) { return PATCH ? prow_base + (r >> 4) * pW + (r & 15) : r; };
    constexpr bool F16 = EPI == EPI_F16;               // fp32 slab like EPI_F32, fp16 in HBM (output and residual)
    constexpr bool F32 = EPI == EPI_F32 || F16;
    constexpr int ES = F32 ? 4 : 2;
    constexpr int TO = EPI == EPI_GEGLU ? TN / 2 : TN;                 // 32-column output tiles per wave
    static_assert(EPI != EPI_GEGLU || TN % 2 == 0, "GEGLU needs value/gate tile pairs in one wave");
    constexpr int MAXROW = LDS_BYTES / NWC / 32;                        // bytes per slab row this wave may use
    constexpr int TS_RAW = (MAXROW - 16) / (32 * ES);
    constexpr int TS = TS_RAW >= TO ? TO : TS_RAW;                      // output tiles staged per pass
    static_assert(TS >= 1, "LDS ring too small for the epilogue slab");
    constexpr int RS = TS * 32 * ES + 16;                               // slab row stride (16-B pad)
    constexpr int CPR = TS * 32 * ES / 16;                              // 16-B chunks per slab row
    const int l31 = lane & 31, hi = lane >> 5;
    char* slab = smem + (size_t)wave_c * 32 * RS;
    const int ocol0 = EPI == EPI_GEGLU ? (wcol0 >> 1) : wcol0;         // first output column of this wave
    const int NO = EPI == EPI_GEGLU ? (p.N >> 1) : p.N;                 // output columns
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int t0 = 0; t0 < TO; t0 += TS) {
            // ---- registers -> slab (this wave's 32 rows x up to TS*32 columns)
#pragma unroll
            for (int tt = 0; tt < TS; ++tt) {
                const int t = t0 + tt;
                if (t >= TO) break;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cl = 8 * g + 4 * hi;                      // column inside the 32-wide tile
                    float v[4];
                    if constexpr (EPI == EPI_GEGLU) {
                        const int cb = wcol0 + 2 * t * 32;              // packed rows: per 64-block [32 value | 32 gate]
                        float bv[4] = {0, 0, 0, 0}, bg[4] = {0, 0, 0, 0};
                        if (p.bias && cb + 64 <= p.N) {
                            const float4 b0 = *(const float4*)(p.bias + cb + cl), b1 = *(const float4*)(p.bias + cb + 32 + cl);
                            bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bg[0] = b1.x; bg[1] = b1.y; bg[2] = b1.z; bg[3] = b1.w;
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float val = acc[i][2 * t][4 * g + e] + bv[e], gt = acc[i][2 * t + 1][4 * g + e] + bg[e];
                            v[e] = val * gelu_erf(gt);
                        }
                    } else {
                        const int col = wcol0 + t * 32 + cl;
                        float bv[4] = {0, 0, 0, 0};
                        if (p.bias && col < p.N) { const float4 b0 = *(const float4*)(p.bias + col); bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; }
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[i][t][4 * g + e] + bv[e];
                        if constexpr (EPI == EPI_BF16_TEMB) {
                            int row = grow(wrow0 + i * 32 + l31); if (row >= p.M) row = p.M - 1;
                            if (col < p.N) {
                                const float4 tv = *(const float4*)(p.temb + (size_t)(row / p.rows_per_batch) * p.temb_ld + col);
                                v[0] += tv.x; v[1] += tv.y; v[2] += tv.z; v[3] += tv.w;
                            }
                        }
                    }
                    char* dst = slab + l31 * RS + (tt * 32 + cl) * ES;
                    if constexpr (F32) *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
                    else { uint2 w; w.x = pack_bf16x2(v[0], v[1]); w.y = pack_bf16x2(v[2], v[3]); *(uint2*)dst = w; }
                }
            }
            // ---- slab -> HBM, row-contiguous 16-B chunks (LDS operations of one wave execute in order)
            const int ncol_pass = (TO - t0 < TS ? TO - t0 : TS) * 32;   // columns staged in this pass
            const int cpr_pass = ncol_pass * ES / 16;
            if constexpr (F16) {
                // 8 columns per lane: two 16-B slab reads, one 16-B residual read, ONE 16-B store
                const int cpr8 = cpr_pass >> 1;
#pragma unroll
                for (int idx0 = 0; idx0 < 16 * CPR; idx0 += 64) {
                    const int idx = idx0 + lane;
                    const int r = idx / cpr8, c8 = idx - r * cpr8;
                    if (r >= 32) continue;
                    const int row = grow(wrow0 + i * 32 + r);
                    const int col = ocol0 + t0 * 32 + c8 * 8;
                    if (row >= p.M || col >= NO) continue;
                    const float4 a0 = *(const float4*)(slab + r * RS + c8 * 32), a1 = *(const float4*)(slab + r * RS + c8 * 32 + 16);
                    float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                    if (p.res) {
                        const uint4 rq = *(const uint4*)((const f16_t*)p.res + (size_t)row * p.ldres + col);
                        const f16_t* rh = (const f16_t*)&rq;
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)rh[e];
                    }
                    uint4 o; f16_t* oh = (f16_t*)&o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) oh[e] = (f16_t)v[e];
                    *(uint4*)((f16_t*)p.out + (size_t)row * p.ldo + col) = o;
                }
            } else {
#pragma unroll
            for (int idx0 = 0; idx0 < 32 * CPR; idx0 += 64) {
                const int idx = idx0 + lane;
                const int r = idx / cpr_pass, ch = idx - r * cpr_pass;
                if (r >= 32) continue;
                const int row = grow(wrow0 + i * 32 + r);
                const int col = ocol0 + t0 * 32 + ch * (16 / ES);
                if (row >= p.M || col >= NO) continue;
                const uint4 q = *(const uint4*)(slab + r * RS + ch * 16);
                if constexpr (F32) {
                    float4 o = make_float4(__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w));
                    if (p.res) { const float4 rv = *(const float4*)((const float*)p.res + (size_t)row * p.ldres + col); o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w; }
                    if (PATCH && p.pair_lo) {                          // the bf16 (hi, lo) pair of the fp32 value instead of the value (workgroup-uniform)
                        uint2 h, l;
                        h.x = pack_bf16x2(o.x, o.y); h.y = pack_bf16x2(o.z, o.w);
                        l.x = pack_bf16x2_lo(o.x, o.y); l.y = pack_bf16x2_lo(o.z, o.w);
                        *(uint2*)((bf16_t*)p.out + (size_t)row * p.ldo + col) = h;
                        *(uint2*)(p.pair_lo + (size_t)row * p.ldo + col) = l;
                    } else
                    *(float4*)((float*)p.out + (size_t)row * p.ldo + col) = o;
                } else {
                    *(uint4*)((bf16_t*)p.out + (size_t)row * p.ldo + col) = q;
                }
            }
            }
        }
    }
}

template <int MODE, int EPI, int BM, int BN, int WM, int WN, int S>
__global__ __launch_bounds__(WM * WN * 64) void gemm_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = WM * WN;                    // waves
    constexpr int STAGE = (BM + BN) * BK * 2;      // bytes per stage: A tile then B tile, 128-B rows
    constexpr int GA = BM / 8, GB = BN / 8;        // 8-row groups (one wave-wide glds each)
    constexpr int NA = (GA + NW - 1) / NW, NB = (GB + NW - 1) / NW;   // glds per thread per stage
    constexpr int P = NA + NB;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;               // MFMA tiles per wave
    static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "wave tile must be a multiple of 32x32");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- XCD-aware bijective block remap (each XCD gets a contiguous run of tiles)
    const int ntn = (p.N + BN - 1) / BN;
    const int ntm = (p.M + BM - 1) / BM;
    const int nwg = ntm * ntn;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // grouped ordering: GRP tile-rows x all tile-columns per group, column-major inside the group, so the
    // tiles resident on one XCD share a few A panels and W panels in its 4 MiB L2
    constexpr int GRP = BM >= 256 ? 4 : 8;
    const int gsz = GRP * ntn;
    const int first_m = (bid / gsz) * GRP;
    const int gm = (ntm - first_m) < GRP ? (ntm - first_m) : GRP;
    const int tm = first_m + (bid % gsz) % gm, tn = (bid % gsz) / gm;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- loader geometry: a wave-wide glds moves one 8-row group: lane -> (row lrow, physical slot pslot)
    const int lrow = lane >> 3, pslot = lane & 7;
    int a_off[NA];                 // element offset of the row (dense) or of the image (conv)
    int a_g[NA];                   // row-group index inside the tile (duplicates at the tail are benign)
    short cy[NA], cx[NA];
    int b_off[NB], b_g[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        int g = i * NW + wave; if (g > GA - 1) g = GA - 1;
        a_g[i] = g;
        int row = m0 + g * 8 + lrow;
        if (row >= p.M) row = p.M - 1;
        if (MODE == A_DENSE) {
            a_off[i] = row * p.lda; cy[i] = cx[i] = 0;
        } else {
            const int b = row / p.rows_per_batch;
            const int pix = row - b * p.rows_per_batch;
            const int y = pix / p.Wout, x = pix - y * p.Wout;
            a_off[i] = b * p.Hin * p.Win * p.Cin;
            cy[i] = (short)y; cx[i] = (short)x;
        }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        int g = i * NW + wave; if (g > GB - 1) g = GB - 1;
        b_g[i] = g;
        int row = n0 + g * 8 + lrow;
        if (row >= p.N) row = p.N - 1;
        b_off[i] = row * p.ldw;
    }

    // part < 0: issue every copy of the stage; part in [0,4): issue the copies with (index % 4) == part, so the
    // LDS-DMA issue slots (tens of cycles each) are spread between the MFMA groups of the k loop
    auto stage = [&](int s, int k0, int part) {
        char* sa = smem + s * STAGE;
        char* sb = sa + BM * BK * 2;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (part >= 0 && (i & 3) != part) continue;
            const int key = ((a_g[i] << 2) | (lrow >> 1)) & 7;          // (row>>1)&7 of the tile row
            const int k = k0 + ((pslot ^ key) << 3);
            const bf16_t* src = p.zero;
            if (MODE == A_DENSE) {
                src = k < p.K ? p.A + a_off[i] + k : p.zero;
            } else if (k < p.K) {
                {
                    const int tap = k / p.Cin, c = k - tap * p.Cin, ky = tap / 3, kx = tap - ky * 3;
                    int yy, xx; bool ok;
                    if (MODE == A_CONV3) {
                        yy = cy[i] + ky - 1; xx = cx[i] + kx - 1;
                        ok = yy >= 0 && yy < p.Hin && xx >= 0 && xx < p.Win;
                    } else if (MODE == A_CONV3_S2) {
                        yy = cy[i] * 2 + ky - 1; xx = cx[i] * 2 + kx - 1;
                        ok = yy >= 0 && yy < p.Hin && xx >= 0 && xx < p.Win;
                    } else {   // nearest 2x upsample fused into the gather: conv runs on the (2Hin x 2Win) grid
                        yy = cy[i] + ky - 1; xx = cx[i] + kx - 1;
                        ok = yy >= 0 && yy < 2 * p.Hin && xx >= 0 && xx < 2 * p.Win;
                        yy >>= 1; xx >>= 1;
                    }
                    if (ok) src = p.A + a_off[i] + (yy * p.Win + xx) * p.Cin + c;
                }
            }
            glds16(src, sa + a_g[i] * 1024);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (part >= 0 && ((i + NA) & 3) != part) continue;
            const int key = ((b_g[i] << 2) | (lrow >> 1)) & 7;
            const int k = k0 + ((pslot ^ key) << 3);
            const bf16_t* src = k < p.K ? p.W + b_off[i] + k : p.zero;
            glds16(src, sb + b_g[i] * 1024);
        }
    };

    // ---- compute geometry
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int fa_off[TM], fa_key[TM], fb_off[TN], fb_key[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ra = wm * (BM / WM) + i * 32 + l31;
        fa_off[i] = ra * 128; fa_key[i] = (ra >> 1) & 7;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int rb = wn * (BN / WN) + j * 32 + l31;
        fb_off[j] = BM * BK * 2 + rb * 128; fb_key[j] = (rb >> 1) & 7;
    }

    // split-K (launch_gemm_splitk): blockIdx.y owns the K tiles [kt_first, kt_first + nk) and writes raw fp32 partial sums
    const int nk_all = (p.K + BK - 1) / BK;
    const int kt_first = gridDim.y > 1 ? (int)((long)blockIdx.y * nk_all / gridDim.y) : 0;
    const int nk = gridDim.y > 1 ? (int)((long)(blockIdx.y + 1) * nk_all / gridDim.y) - kt_first : nk_all;
#define KMAP(j) ((kt_first + (j)) * BK)
#pragma unroll
    for (int s = 0; s < S - 1; ++s)
        if (s < nk) stage(s, KMAP(s), -1);

    // one K tile: wait for it, barrier, then (reads ks+1 | DMA issue of tile kt+S-1 | MFMA ks) per k-step.
    // MORE / LAST are compile-time so the loop body is straight-line code (no control flow between MFMAs).
    auto ktile = [&](int kt, auto more_c, auto behind_c) {
        constexpr bool MORE = decltype(more_c)::value;     // another tile will be staged during this one
        constexpr int BEHIND = decltype(behind_c)::value;  // tiles that stay in flight behind tile kt (0 .. S-2)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BEHIND * P) : "memory");
        __builtin_amdgcn_s_barrier();            // every wave's part of tile kt is in LDS; ring slot (kt-1)%S is free
        __builtin_amdgcn_sched_barrier(0);
        const int ns = (kt + S - 1) % S, nk0 = KMAP(kt + S - 1);
        const char* sbase = smem + (kt % S) * STAGE;
        // register double-buffered fragments: the ds_reads of k-step ks+1 are in flight while the MFMAs of ks run
        bf16x8 fa[2][TM], fb[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[0][i] = *(const bf16x8*)(sbase + fa_off[i] + (((0 * 2 + hi) ^ fa_key[i]) << 4));
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[0][j] = *(const bf16x8*)(sbase + fb_off[j] + (((0 * 2 + hi) ^ fb_key[j]) << 4));
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < BK / 16) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[nxt][i] = *(const bf16x8*)(sbase + fa_off[i] + ((((ks + 1) * 2 + hi) ^ fa_key[i]) << 4));
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[nxt][j] = *(const bf16x8*)(sbase + fb_off[j] + ((((ks + 1) * 2 + hi) ^ fb_key[j]) << 4));
            }
#if RT_ABLATE != 1          // (probe builds only: 1 = no staging in the loop, 2 = no MFMA; tools/probes/gemm_bench.hip)
            if (MORE) stage(ns, nk0, ks);
#endif
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    // operands swapped (D^T = W A^T): the lane then owns ONE output row m and 4 consecutive
                    // output columns per register quad, so the epilogue moves 16-B / 8-B vectors
#if RT_ABLATE == 2
                    asm volatile("" ::"v"(fb[cur][j]), "v"(fa[cur][i]));
#else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cur][j], fa[cur][i], acc[i][j], 0, 0, 0);
#endif
            __builtin_amdgcn_sched_barrier(0);   // keep the (reads ks+1 | DMA issue | MFMA ks) grouping
        }
    };
    static_assert(S >= 2 && S <= 4 && (S - 2) * P <= 63, "ring depth");
    int kt = 0;
    for (; kt + S - 1 < nk; ++kt) ktile(kt, std::true_type{}, std::integral_constant<int, S - 2>{});
    // drain: tile kt has min(S - 2, nk - 1 - kt) younger tiles in flight
    if (S >= 4) for (; kt + 2 < nk; ++kt) ktile(kt, std::false_type{}, std::integral_constant<int, (S >= 4 ? 2 : 0)>{});
    if (S >= 3) for (; kt + 1 < nk; ++kt) ktile(kt, std::false_type{}, std::integral_constant<int, (S >= 3 ? 1 : 0)>{});
    for (; kt < nk; ++kt) ktile(kt, std::false_type{}, std::integral_constant<int, 0>{});
    __syncthreads();      // every wave is done with the LDS ring: it becomes the epilogue's transpose slabs
    if (gridDim.y > 1) {
        GemmArgs q = p;       // partial [split][M][ldo] fp32 (the launcher passes EPI_F32 without bias / residual)
        q.out = (float*)p.out + (size_t)blockIdx.y * p.M * p.ldo;
        gemm_epilogue<EPI, TM, TN, NW, S * STAGE>(q, acc, m0 + wm * (BM / WM), n0 + wn * (BN / WN), lane, wave, smem);
        return;
    }
    gemm_epilogue<EPI, TM, TN, NW, S * STAGE>(p, acc, m0 + wm * (BM / WM), n0 + wn * (BN / WN), lane, wave, smem);
}

// ================================================================================================
// Ping-pong variant: the 8 waves form two groups of 4 (one wave of each group per SIMD) that run ONE BARRIER apart,
// so while a group executes a pure-MFMA section the other group issues its ds_reads and LDS-DMA copies
// (guide T3/T4/T5: wave role split, counted vmcnt, setprio).  Motivation (tools/probes/gemm_bench ablation,
// MI355X): with all waves in lock-step the kernel above spends ~2.8k cycles per K tile at 256x160 where the MFMA
// work is 1.3k, the loads alone 1.7k and the MFMA+ds_read loop alone 2.1k - every wave queues on the CU's single
// texture-address path at the same time and nobody feeds the matrix pipe meanwhile.
//   phase = 2 k-steps (BK/32 phases per K tile), each phase  L: LDS-DMA part (+ counted vmcnt)                    ; s_barrier
//                                                            M: setprio 1 ; 2*TM*TN MFMA | ds_read next phase  ; s_barrier
//   group 1 executes one extra s_barrier first, so its L overlaps group 0's M and vice versa.
// Ring (3 slots, K tile kt in slot kt % 3), per wave, tile kt:
//   L(kt,0): stage the A part of tile kt+2 (slot of tile kt-1) ; s_waitcnt vmcnt(NA) => this wave's share of tile kt+1 landed
//   M(kt,0): MFMA k-steps 0,1 | read k-steps 2,3 of tile kt          ; lgkmcnt(0)
//   L(kt,1): stage the B part of tile kt+2
//   M(kt,1): MFMA k-steps 2,3 | read k-steps 0,1 of tile kt+1        ; lgkmcnt(0)
// RAW: every wave has passed its L(kt,0) wait at the barrier closing group 1's L(kt,0); the first reads of tile kt+1 are
//      in M(kt,1), which starts after that barrier for both groups.
// WAR: a slot's last reads (M(kt-1,0)) are retired by the lgkmcnt(0) before the barrier closing that section; the DMA
//      into the slot is issued in L(kt,0) / L(kt,1), which start after that barrier for both groups.
// Same k order and MFMA shape as gemm_kernel => bit-identical results.
#ifdef RT_PP_TIMING
__device__ long long g_pp_times[8 * 8];
#define PP_T(i) { const long long now_ = __builtin_readcyclecounter(); tacc[i] += now_ - tlast; tlast = now_; }
#else
#define PP_T(i)
#endif
template <int MODE, int EPI, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64) void gemm_pp_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int S = 3;
    constexpr int NW = WM * WN;
    constexpr int STAGE = (BM + BN) * BK * 2;
    constexpr int GA = BM / 8, GB = BN / 8;
    constexpr int NA = (GA + NW - 1) / NW, NB = (GB + NW - 1) / NW;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(NW == 8 && BK == 64, "two groups of four waves, two phases per K tile");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool grp1 = wave >= NW / 2;

    const int ntn = (p.N + BN - 1) / BN;
    const int ntm = (p.M + BM - 1) / BM;
    const int nwg = ntm * ntn;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    constexpr int GRP = 4;
    const int gsz = GRP * ntn;
    const int first_m = (bid / gsz) * GRP;
    const int gm = (ntm - first_m) < GRP ? (ntm - first_m) : GRP;
    const int tm = first_m + (bid % gsz) % gm, tn = (bid % gsz) / gm;
    const int m0 = tm * BM, n0 = tn * BN;

    const int lrow = lane >> 3, pslot = lane & 7;
    int a_off[NA], a_g[NA];
    short cy[NA], cx[NA];
    int b_off[NB], b_g[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        int g = i * NW + wave; if (g > GA - 1) g = GA - 1;
        a_g[i] = g;
        int row = m0 + g * 8 + lrow;
        if (row >= p.M) row = p.M - 1;
        if (MODE == A_DENSE) {
            a_off[i] = row * p.lda; cy[i] = cx[i] = 0;
        } else {
            const int b = row / p.rows_per_batch;
            const int pix = row - b * p.rows_per_batch;
            const int y = pix / p.Wout, x = pix - y * p.Wout;
            a_off[i] = b * p.Hin * p.Win * p.Cin;
            cy[i] = (short)y; cx[i] = (short)x;
        }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        int g = i * NW + wave; if (g > GB - 1) g = GB - 1;
        b_g[i] = g;
        int row = n0 + g * 8 + lrow;
        if (row >= p.N) row = p.N - 1;
        b_off[i] = row * p.ldw;
    }
    auto stage_a = [&](int s, int k0) {
        char* sa = smem + s * STAGE;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int key = ((a_g[i] << 2) | (lrow >> 1)) & 7;
            const int k = k0 + ((pslot ^ key) << 3);
            const bf16_t* src = p.zero;
            if (MODE == A_DENSE) {
                src = k < p.K ? p.A + a_off[i] + k : p.zero;
            } else if (k < p.K) {
                const int tap = k / p.Cin, c = k - tap * p.Cin, ky = tap / 3, kx = tap - ky * 3;
                int yy, xx; bool ok;
                if (MODE == A_CONV3) {
                    yy = cy[i] + ky - 1; xx = cx[i] + kx - 1;
                    ok = yy >= 0 && yy < p.Hin && xx >= 0 && xx < p.Win;
                } else if (MODE == A_CONV3_S2) {
                    yy = cy[i] * 2 + ky - 1; xx = cx[i] * 2 + kx - 1;
                    ok = yy >= 0 && yy < p.Hin && xx >= 0 && xx < p.Win;
                } else {
                    yy = cy[i] + ky - 1; xx = cx[i] + kx - 1;
                    ok = yy >= 0 && yy < 2 * p.Hin && xx >= 0 && xx < 2 * p.Win;
                    yy >>= 1; xx >>= 1;
                }
                if (ok) src = p.A + a_off[i] + (yy * p.Win + xx) * p.Cin + c;
            }
            glds16(src, sa + a_g[i] * 1024);
        }
    };
    auto stage_b = [&](int s, int k0) {
        char* sb = smem + s * STAGE + BM * BK * 2;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int key = ((b_g[i] << 2) | (lrow >> 1)) & 7;
            const int k = k0 + ((pslot ^ key) << 3);
            const bf16_t* src = k < p.K ? p.W + b_off[i] + k : p.zero;
            glds16(src, sb + b_g[i] * 1024);
        }
    };

    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // all fragments of a lane share the swizzle key: (row>>1)&7 with row = 32*t + (lane&31)
    const int key = (l31 >> 1) & 7;
    const int a_base = (wm * (BM / WM) + l31) * 128;
    const int b_base = BM * BK * 2 + (wn * (BN / WN) + l31) * 128;
    int kx[BK / 16];
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) kx[ks] = ((ks * 2 + hi) ^ key) << 4;

    const int nk = (p.K + BK - 1) / BK;
    stage_a(0, 0); stage_b(0, 0);
    if (nk > 1) {
        stage_a(1, BK); stage_b(1, BK);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NB) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                  // tile 0 is in LDS for everyone
    // fragments: fr[c] feeds phase c of a tile; M(kt,0) prefetches fr[1] (k-steps 2,3), M(kt,1) prefetches fr[0] of tile kt+1
    bf16x8 fa[2][2][TM], fb[2][2][TN];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[0][h][i] = *(const bf16x8*)(smem + a_base + i * 4096 + kx[h]);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[0][h][j] = *(const bf16x8*)(smem + b_base + j * 4096 + kx[h]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (grp1) __builtin_amdgcn_s_barrier();        // group 1 runs one barrier behind group 0
    __builtin_amdgcn_sched_barrier(0);
#ifdef RT_PP_TIMING
    long long tacc[5] = {0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
    for (int kt = 0; kt < nk; ++kt) {
        const char* sbase = smem + (kt % S) * STAGE;
        const char* snext = smem + ((kt + 1) % S) * STAGE;
        const bool more2 = kt + 2 < nk;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            // ---------------- L section: LDS-DMA issue only (the wave sits in the copy queue's back-pressure)
            if (ph == 0) {
                if (more2) {
                    stage_a((kt + 2) % S, (kt + 2) * BK);
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA) : "memory");      // this wave's share of tile kt+1 landed
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            } else {
                if (more2) stage_b((kt + 2) % S, (kt + 2) * BK);
            }
            PP_T(0)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            PP_T(1)
            // ---------------- M section: MFMAs of this phase, interleaved with the ds_reads of the next phase
            const char* rb = ph == 0 ? sbase : snext;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ph][h][j], fa[ph][h][i], acc[i][j], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[ph ^ 1][h][i] = *(const bf16x8*)(rb + a_base + i * 4096 + kx[(ph ^ 1) * 2 + h]);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[ph ^ 1][h][j] = *(const bf16x8*)(rb + b_base + j * 4096 + kx[(ph ^ 1) * 2 + h]);
            }
            // one read behind every MFMA (the reads of a 32-row fragment never outnumber 2*(TM*TN) + TM + TN here)
#pragma unroll
            for (int r = 0; r < 2 * (TM + TN); ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * TM * TN, 0);
            __builtin_amdgcn_s_setprio(0);
            PP_T(2)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // next phase's fragments are in registers; slot reads retired
            PP_T(3)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            PP_T(4)
        }
    }
#ifdef RT_PP_TIMING
    if (blockIdx.x == 0 && lane == 0) for (int i = 0; i < 5; ++i) g_pp_times[wave * 8 + i] = tacc[i];
#endif
    if (!grp1) __builtin_amdgcn_s_barrier();       // pairs with group 1's last barrier
    __syncthreads();                               // the ring becomes the epilogue's transpose slabs
    gemm_epilogue<EPI, TM, TN, NW, S * STAGE>(p, acc, m0 + wm * (BM / WM), n0 + wn * (BN / WN), lane, wave, smem);
}

// ================================================================================================
// Wave-specialised variant: NL loader waves + WMxWN compute waves, 3-slot LDS ring, one s_barrier per K tile.
// Measured motivation (s_memtime phase probe on MI355X, LABNOTES.md 4.1): inside the GEMM loop one
// global_load_lds_dwordx4 costs the ISSUING wave ~110-140 cycles, i.e. ~1000 cycles per K tile when every
// wave stages its own share, during which that wave issues no MFMA (in-order issue); the matrix pipe then
// idles ~40 % of the time.  Here the DMA issue slots are paid by waves that do nothing else, the compute
// waves only do ds_read + MFMA (register double-buffered fragments), and the loaders run two tiles ahead
// behind a counted vmcnt so HBM/L2 latency stays covered.
//   loader  kt: s_waitcnt vmcnt(PL)  (tile kt landed, tile kt+1 may be in flight) ; s_barrier ; stage(kt+2)
//   compute kt:                                                                     s_barrier ; compute(kt)
// Ring safety: stage(kt+2) overwrites slot (kt-1)%3, whose last readers (compute(kt-1)) all arrived at
// barrier kt after their MFMAs consumed the fragments.
template <int MODE, int EPI, int BM, int BN, int WM, int WN, int NL>
__global__ __launch_bounds__((WM * WN + NL) * 64) void gemm_ws_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NWC = WM * WN;
    constexpr int STAGE = (BM + BN) * BK * 2;
    constexpr int GA = BM / 8, GB = BN / 8, GT = GA + GB;
    constexpr int PL = (GT + NL - 1) / NL;                           // glds per loader wave per stage
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(PL <= 60, "vmcnt immediate range");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM, nwg = ntm * ntn;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    constexpr int GRP = 4;
    const int gsz = GRP * ntn;
    const int first_m = (bid / gsz) * GRP;
    const int gm = (ntm - first_m) < GRP ? (ntm - first_m) : GRP;
    const int tm = first_m + (bid % gsz) % gm, tn = (bid % gsz) / gm;
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk = (p.K + BK - 1) / BK;

    if (wave >= NWC) {
        // ------------------------------------------------------------------ loader wave
        const int lw = wave - NWC;
        const int lrow = lane >> 3, pslot = lane & 7;
        const bf16_t* base[PL];      // dense: row pointer incl. swizzled chunk; conv: image pointer
        int kofs[PL];                // element offset of this lane's 16-B chunk inside the K tile
        int ldst[PL];                // LDS byte offset of the 8-row group inside a stage
        short cy[PL], cx[PL];
        bool isA[PL];
#pragma unroll
        for (int i = 0; i < PL; ++i) {
            int g = i * NL + lw; if (g > GT - 1) g = GT - 1;          // tail duplicates are benign (same data, same slot)
            const bool a = g < GA;
            const int gl = a ? g : g - GA;
            const int key = ((gl << 2) | (lrow >> 1)) & 7;            // (tile row >> 1) & 7
            kofs[i] = (pslot ^ key) << 3;
            ldst[i] = (a ? 0 : BM * BK * 2) + gl * 1024;
            isA[i] = a; cy[i] = cx[i] = 0;
            if (a) {
                int row = m0 + gl * 8 + lrow; if (row >= p.M) row = p.M - 1;
                if (MODE == A_DENSE) base[i] = p.A + (size_t)row * p.lda;
                else {
                    const int b = row / p.rows_per_batch, pix = row - b * p.rows_per_batch;
                    const int y = pix / p.Wout, x = pix - y * p.Wout;
                    base[i] = p.A + (size_t)b * p.Hin * p.Win * p.Cin; cy[i] = (short)y; cx[i] = (short)x;
                }
            } else {
                int row = n0 + gl * 8 + lrow; if (row >= p.N) row = p.N - 1;
                base[i] = p.W + (size_t)row * p.ldw;
            }
        }
        auto stage = [&](int s, int k0) {
            char* sb = smem + s * STAGE;
#pragma unroll
            for (int i = 0; i < PL; ++i) {
                const int k = k0 + kofs[i];
                const bf16_t* src = p.zero;
                if (MODE == A_DENSE || !isA[i]) {
                    src = k < p.K ? base[i] + k : p.zero;
                } else if (k < p.K) {
                    const int tap = k / p.Cin, c = k - tap * p.Cin, ky = tap / 3, kx = tap - ky * 3;
                    int yy, xx; bool ok;
                    if (MODE == A_CONV3) { yy = cy[i] + ky - 1; xx = cx[i] + kx - 1; ok = yy >= 0 && yy < p.Hin && xx >= 0 && xx < p.Win; }
                    else if (MODE == A_CONV3_S2) { yy = cy[i] * 2 + ky - 1; xx = cx[i] * 2 + kx - 1; ok = yy >= 0 && yy < p.Hin && xx >= 0 && xx < p.Win; }
                    else { yy = cy[i] + ky - 1; xx = cx[i] + kx - 1; ok = yy >= 0 && yy < 2 * p.Hin && xx >= 0 && xx < 2 * p.Win; yy >>= 1; xx >>= 1; }
                    if (ok) src = base[i] + ((size_t)yy * p.Win + xx) * p.Cin + c;
                }
                glds16(src, sb + ldst[i]);
            }
        };
#define KMAPW(j) ((j) * BK)
        stage(0, KMAPW(0));
        if (nk > 1) stage(1, KMAPW(1));
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PL) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + 2 < nk) stage((kt + 2) % 3, KMAPW(kt + 2));
        }
        __builtin_amdgcn_s_barrier();      // pairs with the compute waves' barrier in front of their epilogue
        return;
    }
    // ---------------------------------------------------------------------- compute wave
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // all fragments of a lane share the swizzle key: (row>>1)&7 with row = 32*t + (lane&31)
    const int key = (l31 >> 1) & 7;
    const int a_base = (wm * (BM / WM) + l31) * 128;
    const int b_base = BM * BK * 2 + (wn * (BN / WN) + l31) * 128;
    int kx[BK / 16];
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) kx[ks] = ((ks * 2 + hi) ^ key) << 4;
    for (int kt = 0; kt < nk; ++kt) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const char* sbase = smem + (kt % 3) * STAGE;
        bf16x8 fa[2][TM], fb[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[0][i] = *(const bf16x8*)(sbase + a_base + i * 4096 + kx[0]);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[0][j] = *(const bf16x8*)(sbase + b_base + j * 4096 + kx[0]);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < BK / 16) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[nxt][i] = *(const bf16x8*)(sbase + a_base + i * 4096 + kx[ks + 1]);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[nxt][j] = *(const bf16x8*)(sbase + b_base + j * 4096 + kx[ks + 1]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cur][j], fa[cur][i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_s_barrier();          // all compute waves left the ring (the loaders arrive here too)
    gemm_epilogue<EPI, TM, TN, NWC, 3 * STAGE>(p, acc, m0 + wm * (BM / WM), n0 + wn * (BN / WN), lane, wave, smem);
}

// ================================================================================================
// 8-phase 256x256 kernel (cfg 7) for the wide-N dense GEMMs (GEGLU 7168x10240x1280, the stacked Q|K projection): the guide's
// phase-interleaved schedule (cdna_hip_programming.md "256^2 8-phase template", T2-T5) on this file's 32x32x16 MFMA, swizzle
// and epilogue, so results stay bit-identical with every other configuration.
//   8 waves as 2 (M) x 4 (N), 128 x 64 outputs per wave (4 x 2 MFMA tiles: 24 ds_read_b128 per 32 MFMAs instead of 32 per 32
//   in the 16-wave kernel).  A K tile is four 16-KB HALF tiles: A0/A1 = rows [0,64) / [64,128) of both M waves, B0/B1 = columns
//   [0,32) / [32,64) of all four N waves.  LDS = 2 K tiles x 4 half tiles = 128 KB.
//   A K tile is two sections of two C quadrants x K=64 each (16 MFMAs):  a = (A0,B0) (A0,B1),  b = (A1,B1) (A1,B0);  B0 / B1 stay
//   in registers.  (Measured with one quadrant per section: load 350 | barrier 90 | MFMA 320 | barrier 105 cycles - the barrier
//   pair was a quarter of the loop.)
//   section:  ds_read what it needs | issue TWO half tiles (4 LDS-DMA pieces per wave) | counted vmcnt | lgkmcnt(0) | s_barrier |
//             setprio 1 | 16 MFMA | setprio 0 | s_barrier
//   The two M wave groups run ONE barrier apart: while one group is in its MFMA section the other issues reads and copies.
// Pipeline: half tile h = 4*tile + j (j: A0,B0,B1,A1 = consumption order) is issued six half tiles ahead of its consumption,
// i.e. three to four half tiles (48-64 KB per CU) are in flight behind every barrier; it lands in slot (tile&1, j), whose
// previous occupant's reads were retired (lgkmcnt(0) in front of a barrier every wave has passed) at least one full section
// earlier (WAR), and is first read >= 2 barriers after the counted s_waitcnt vmcnt that retires it (RAW, also across the two
// staggered groups: every wave's wait precedes the barrier the reader has passed).
#ifdef RT_G8_TIMING
__device__ long long g_g8_times[8 * 4];
#endif
template <int EPI>
__global__ __launch_bounds__(512) void gemm8_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 256, BN = 256, HALF = 128 * BK * 2, KTILE = 4 * HALF;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM, nwg = ntm * ntn;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    constexpr int GRP = 4;      // (tile -> XCD mapping variants, GRP 2..16 and no remap, all measured within +-2 % on MI355X)
    const int gsz = GRP * ntn;
    const int first_m = (bid / gsz) * GRP;
    const int gm = (ntm - first_m) < GRP ? (ntm - first_m) : GRP;
    const int tm = first_m + (bid % gsz) % gm, tn = (bid % gsz) / gm;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging: per half tile this wave copies the 8-row pieces q = wave and wave + 8 (local rows 8q .. 8q+7 of the half)
    const int lrow = lane >> 3, pslot = lane & 7;
    int a_off[2][2], b_off[2][2], skey[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int lr = (wave + 8 * u) * 8 + lrow;                  // local row inside the half tile
        skey[u] = (lr >> 1) & 7;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int ra = m0 + (lr >> 6) * 128 + h * 64 + (lr & 63); if (ra >= p.M) ra = p.M - 1;
            int rb = n0 + (lr >> 5) * 64 + h * 32 + (lr & 31); if (rb >= p.N) rb = p.N - 1;
            a_off[h][u] = ra * p.lda; b_off[h][u] = rb * p.ldw;
        }
    }
    // half tile h = 4*tile + j; j: 0 = A0, 1 = B0, 2 = B1, 3 = A1
    auto stage = [&](int h) {
        const int t = h >> 2, j = h & 3;
        char* dst = smem + (t & 1) * KTILE + j * HALF;
        const bool isA = j == 0 || j == 3;
        const int hh = j >> 1;                                      // A0, B0 -> half 0; B1, A1 -> half 1
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = t * BK + ((pslot ^ skey[u]) << 3);
            const bf16_t* src = p.zero;
            if (k < p.K) src = isA ? p.A + (hh ? a_off[1][u] : a_off[0][u]) + k : p.W + (hh ? b_off[1][u] : b_off[0][u]) + k;
            glds16(src, dst + (wave + 8 * u) * 1024);
        }
    };

    // ---- fragments: this lane's rows inside the half tiles
    const int l31 = lane & 31, hi = lane >> 5;
    const int key = (l31 >> 1) & 7;                                 // local rows are 32*x + l31: the key depends on l31 only
    int kx[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kx[ks] = ((ks * 2 + hi) ^ key) << 4;
    const int a_row = (wr * 64 + l31) * 128;                        // + i2 * 4096 for the second 32-row tile of the half
    const int b_row = (wc * 32 + l31) * 128;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
#pragma unroll
    for (int h = 0; h < 6; ++h) stage(h);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");               // A0, B0, B1 of tile 0 landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();                     // group 1 runs one barrier behind group 0
    __builtin_amdgcn_sched_barrier(0);

    bf16x8 fa[2][4], fb0[4], fb1[4];
#ifdef RT_G8_TIMING
    long long tacc[4] = {0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#define G8_T(i) { const long long now_ = __builtin_readcyclecounter(); tacc[i] += now_ - tlast; tlast = now_; }
#else
#define G8_T(i)
#endif
    // The MFMA builtins are pure values to the compiler: nothing orders them against s_barrier / s_setprio (the instruction
    // selector happily sinks a whole quadrant into the next phase's load section).  Two empty asm statements pin the group: the
    // first makes the fragments opaque AFTER the barrier (no MFMA can be hoisted above it), the second consumes the accumulators
    // BEFORE the closing barrier (no MFMA can sink below it) - guide 5.7 item 3.
    // One MFMA section = two C quadrants (16 MFMAs): with one quadrant per section the two barriers cost ~200 of ~870 cycles.
#define RT_G8_MFMA2(IA, J0, F0, J1, F1)                                                                              \
    {                                                                                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* also retires this section's reads before anyone re-stages */ \
        G8_T(0)                                                                                                      \
        __builtin_amdgcn_s_barrier();                                                                                \
        G8_T(1)                                                                                                      \
        asm volatile("" : "+v"(F0[0]), "+v"(F0[1]), "+v"(F0[2]), "+v"(F0[3]), "+v"(F1[0]), "+v"(F1[1]), "+v"(F1[2]), "+v"(F1[3]), \
                          "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]), "+v"(fa[1][0]), "+v"(fa[1][1]),    \
                          "+v"(fa[1][2]), "+v"(fa[1][3]));                                                           \
        __builtin_amdgcn_s_setprio(1);                                                                               \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                           \
            acc[IA][J0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F0[ks], fa[0][ks], acc[IA][J0], 0, 0, 0);          \
            acc[IA + 1][J0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F0[ks], fa[1][ks], acc[IA + 1][J0], 0, 0, 0);  \
            acc[IA][J1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F1[ks], fa[0][ks], acc[IA][J1], 0, 0, 0);          \
            acc[IA + 1][J1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F1[ks], fa[1][ks], acc[IA + 1][J1], 0, 0, 0);  \
        }                                                                                                            \
        asm volatile("" : "+v"(acc[IA][J0]), "+v"(acc[IA + 1][J0]), "+v"(acc[IA][J1]), "+v"(acc[IA + 1][J1]));       \
        __builtin_amdgcn_s_setprio(0);                                                                               \
        G8_T(2)                                                                                                      \
        __builtin_amdgcn_s_barrier();                                                                                \
        G8_T(3)                                                                                                      \
    }
    for (int t = 0; t < nk; ++t) {
        const char* base = smem + (t & 1) * KTILE;
        const int h0 = 4 * t + 6;
        // ---- section a: quadrants (A0, B0), (A0, B1)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fb0[ks] = *(const bf16x8*)(base + 1 * HALF + b_row + kx[ks]);
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fa[i2][ks] = *(const bf16x8*)(base + 0 * HALF + a_row + i2 * 4096 + kx[ks]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fb1[ks] = *(const bf16x8*)(base + 2 * HALF + b_row + kx[ks]);
        stage(h0); stage(h0 + 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");           // A1 of this tile landed
        RT_G8_MFMA2(0, 0, fb0, 1, fb1)
        // ---- section b: quadrants (A1, B1), (A1, B0); B0 / B1 still in registers
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fa[i2][ks] = *(const bf16x8*)(base + 3 * HALF + a_row + i2 * 4096 + kx[ks]);
        stage(h0 + 2); stage(h0 + 3);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");           // A0, B0, B1 of the next tile landed
        RT_G8_MFMA2(2, 1, fb1, 0, fb0)
    }
#undef RT_G8_MFMA2
#ifdef RT_G8_TIMING
    if (blockIdx.x == 0 && lane == 0) for (int i = 0; i < 4; ++i) g_g8_times[wave * 4 + i] = tacc[i];
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the dummy copies behind the last tile
    if (wr == 0) __builtin_amdgcn_s_barrier();                     // pairs with group 1's last barrier
    __syncthreads();                                               // the ring becomes the epilogue's transpose slabs
    gemm_epilogue<EPI, 4, 2, 8, 2 * KTILE>(p, acc, m0 + wr * 128, n0 + wc * 64, lane, wave, smem);
}

// ================================================================================================
// 3x3 stride-1 convolution on 16x16 pixel patches (conv3p_kernel).  The implicit-GEMM kernels above fetch every input
// pixel nine times (once per tap) through the L2 -> LDS copy path, which is what bounds them (LABNOTES.md 4.1).  Here a
// block owns a 16x16 output patch x 160 output channels; per 64-channel chunk the 18x18 input HALO is copied to LDS
// once and the nine taps read their A fragments from it at shifted rows, so the copy traffic per k step drops from
// 53 KB (A 32 KB + W 20 KB) to 25 KB (halo/9 = 4.6 KB + W 20 KB).  k order: (chunk, tap, c) - different from the
// implicit-GEMM kernels' (tap, chunk, c), so results agree with them to rounding, not bit for bit; eligible convolutions
// ALWAYS take this kernel (no tuning), which keeps runs reproducible.
//   LDS: 2 halo slots (328 rows x 128 B) + 3-slot ring of W tiles (160 rows x 128 B) = 145,408 B
//   step s = chunk*9 + tap:  k-steps 0..2 ; lgkmcnt(0) ; s_waitcnt vmcnt(N) ; s_barrier ; issue [halo part of chunk+1 | W tile of
//   step s+3] ; k-step 3 while the first fragments of step s+1 are fetched (the barrier never exposes a ds_read)
// Requirements: mode A_CONV3 (stride 1, pad 1), H % 16 == 0, W % 16 == 0, Cin % 64 == 0.
// UP2: the nearest-2x upsample of Upsample2D (resnet.py:137-172) folded in: the 16x16 OUTPUT patch reads a 10x10 input halo
// (input pixel = output pixel >> 1), so the halo is 100 rows instead of 324.
// (Round 6, measured and dropped: waves as 4 pixel groups x 2 column halves at TN = 4 - 2 A + 2 W fragments per 4 MFMAs instead of 1 + 4, 20 % less
//  LDS traffic, bit-identical: precise VAE guidance call 82.9 vs 83.2 ms, single pass 38.2 vs 37.8 ms.  The loop is not bound by LDS bytes.)
// TN = 32-channel column tiles per workgroup (BN = 32 TN output channels): 5 by default; 3 or 2 when a launch would otherwise put
// fewer workgroups on the chip than it has CUs (SD-v1.5: 48-160 workgroups at BN = 160).  The k order does not depend on TN, so
// every TN gives bit-identical results and the launcher may pick it from the actual batch size.
template <int EPI, bool UP2, int TN = 5>
__global__ __launch_bounds__(512) void conv3p_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BN = 32 * TN, NW = 8;
    constexpr int HW_ = UP2 ? 10 : 18;                                    // halo width (input pixels)
    constexpr int HROWS = UP2 ? 104 : 328, ASLOT = HROWS * 128, BSLOT = BN * 128;
    constexpr int GA = HROWS / 8, GB = BN / 8;
    constexpr int NAH = (GA + NW - 1) / NW, NB = (GB + NW - 1) / NW;      // 6 halo pieces, 3 W pieces per wave
    static_assert(NAH <= 9, "one halo piece per tap step");
    char* const sA = smem;
    char* const sB = smem + 2 * ASLOT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int ptx = p.Wout >> 4, npatch = (p.Hout >> 4) * ptx;
    const int ntm = (p.M / p.rows_per_batch) * npatch;
    const int ntn = (p.N + BN - 1) / BN;
    const int nwg = ntm * ntn;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    constexpr int GRP = 4;
    const int gsz = GRP * ntn;
    const int first_m = (bid / gsz) * GRP;
    const int gm = (ntm - first_m) < GRP ? (ntm - first_m) : GRP;
    const int tm = first_m + (bid % gsz) % gm, tn = (bid % gsz) / gm;
    const int n0 = tn * BN;
    const int img = tm / npatch, pp = tm - img * npatch;
    const int y0 = (pp / ptx) << 4, x0 = (pp % ptx) << 4;

    // ---- loaders
    const int lrow = lane >> 3, pslot = lane & 7;
    int a_off[NAH], a_g[NAH];                       // element offset of the halo pixel's channel vector, -1 = outside the image
#pragma unroll
    for (int i = 0; i < NAH; ++i) {
        int g = i * NW + wave; if (g > GA - 1) g = GA - 1;
        a_g[i] = g;
        const int hr = g * 8 + lrow;
        const int hy = hr / HW_, hx = hr - hy * HW_;
        const int yy = (UP2 ? (y0 >> 1) : y0) - 1 + hy, xx = (UP2 ? (x0 >> 1) : x0) - 1 + hx;
        const bool ok = hr < HW_ * HW_ && yy >= 0 && yy < p.Hin && xx >= 0 && xx < p.Win;
        a_off[i] = ok ? ((img * p.Hin + yy) * p.Win + xx) * p.Cin : -1;
    }
    int b_off[NB], b_g[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        int g = i * NW + wave; if (g > GB - 1) g = GB - 1;
        b_g[i] = g;
        int row = n0 + g * 8 + lrow;
        if (row >= p.N) row = p.N - 1;
        b_off[i] = row * p.ldw;
    }
    // Precise VAE (p.A_lo != null): the three passes A W, A_lo W, A W_lo run as ONE K loop of 3 nc1 chunks - chunk c belongs to pass
    // c / nc1 and reads channel chunk c % nc1 of that pass's operands - so the fp32 output is written once instead of being
    // read-modify-written by two more launches (537 MB each way at 1024^2 x 128 channels: those launches were HBM bound).
    const int nc1 = p.Cin >> 6;
    auto stage_halo = [&](int slot, int chunk, int i) {
        const int key = ((a_g[i] << 2) | (lrow >> 1)) & 7;
        const int pass = chunk >= 2 * nc1 ? 2 : (chunk >= nc1 ? 1 : 0);
        const bf16_t* base = pass == 1 ? p.A_lo : p.A;
        const bf16_t* src = a_off[i] >= 0 ? base + a_off[i] + (chunk - pass * nc1) * 64 + ((pslot ^ key) << 3) : p.zero;
        glds16(src, sA + slot * ASLOT + a_g[i] * 1024);
    };
    auto stage_w = [&](int slot, int c3, int t3) {          // W tile of (chunk c3, tap t3)
        const int pass = c3 >= 2 * nc1 ? 2 : (c3 >= nc1 ? 1 : 0);
        const bf16_t* base = pass == 2 ? p.W_lo : p.W;
        const int k0 = t3 * p.Cin + (c3 - pass * nc1) * 64;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int key = ((b_g[i] << 2) | (lrow >> 1)) & 7;
            glds16(base + b_off[i] + k0 + ((pslot ^ key) << 3), sB + slot * BSLOT + b_g[i] * 1024);
        }
    };

    // ---- compute geometry: wave w owns patch rows 2w, 2w+1 (32 pixels) x 160 channels
    const int l31 = lane & 31, hi = lane >> 5;
    const int py = 2 * wave + (l31 >> 4), px = l31 & 15;                  // this lane's pixel inside the patch
    // halo row read by tap (ky, kx): plain: (py+ky, px+kx); UP2: (((py+ky-1)>>1)+1, ((px+kx-1)>>1)+1)
    auto halo_row = [&](int tap) {
        const int ky = tap / 3, kx = tap - ky * 3;
        return UP2 ? (((py + ky - 1) >> 1) + 1) * HW_ + ((px + kx - 1) >> 1) + 1 : (py + ky) * HW_ + px + kx;
    };
    const int hr0 = halo_row(0);
    const int keyb = (l31 >> 1) & 7;
    const int b_base = l31 * 128;
    int kxb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kxb[ks] = ((ks * 2 + hi) ^ keyb) << 4;
    f32x16 acc[1][TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;

    // split over the input-channel chunks (launch_conv3p_splitk): blockIdx.y owns the chunks [c_first, c_first + nc) and writes raw fp32
    // partial sums; ring slots and halo parity follow the LOCAL chunk index, so one slice (gridDim.y == 1) is the code it always was
    const int nc_all = nc1 * (p.A_lo ? 3 : 1);
    const int c_first = gridDim.y > 1 ? (int)((long)blockIdx.y * nc_all / gridDim.y) : 0;
    const int nc = gridDim.y > 1 ? (int)((long)(blockIdx.y + 1) * nc_all / gridDim.y) - c_first : nc_all;
    const int nsteps = nc * 9;
    auto wait_vm = [&](int n) {                     // n is wave-uniform: 0, 1, NB, NB + 1 or 2 NB
        if (n >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (n == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    static_assert(NB >= 1 && NB <= 3, "wait_vm covers up to 3 W pieces (+1 halo piece) per wave and step");
    // prologue: whole halo of chunk 0, W tiles of steps 0..2 (nsteps >= 9); halo + W(0) must have landed
#pragma unroll
    for (int i = 0; i < NAH; ++i) stage_halo(0, c_first, i);
    stage_w(0, c_first, 0);
    stage_w(1, c_first, 1);
    stage_w(2, c_first, 2);
    wait_vm(2 * NB);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // fragment addressing of one (halo slot, tap): lane's halo row, its swizzle key
    bf16x8 fa[2], fb[2][TN];
    {
        const char* arow = sA + hr0 * 128;
        const int keya = (hr0 >> 1) & 7;
        fa[0] = *(const bf16x8*)(arow + ((hi ^ keya) << 4));
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[0][j] = *(const bf16x8*)(sB + b_base + j * 4096 + kxb[0]);
    }
    // Step s = chunk*9 + tap.  The wait + barrier sits between k-steps 2 and 3: it certifies W(s+1) (and, at tap 8, the next
    // halo) for everyone and retires every read of W(s), so k-step 3 can already fetch the first fragments of step s+1 and
    // the W tile of step s+3 can be copied into W(s)'s slot - no ds_read latency is exposed behind a barrier.
    for (int chunk = 0; chunk < nc; ++chunk) {
        const bool next_chunk = chunk + 1 < nc;
        const char* sa = sA + (chunk & 1) * ASLOT;
        const char* sa_next = sA + ((chunk + 1) & 1) * ASLOT;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int s = chunk * 9 + tap;
            const int hr = halo_row(tap);
            const char* arow = sa + hr * 128;
            const int keya = (hr >> 1) & 7;
            const char* sb = sB + (s % 3) * BSLOT + b_base;
            // first fragments of step s+1
            const int ntap = tap == 8 ? 0 : tap + 1;
            const int nhr = halo_row(ntap);
            const char* narow = (tap == 8 ? sa_next : sa) + nhr * 128;
            const int nkeya = (nhr >> 1) & 7;
            const char* nsb = sB + ((s + 1) % 3) * BSLOT + b_base;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int cur = ks & 1, nxt = cur ^ 1;
                if (ks < 3) {
                    fa[nxt] = *(const bf16x8*)(arow + ((((ks + 1) * 2 + hi) ^ keya) << 4));
#pragma unroll
                    for (int j = 0; j < TN; ++j) fb[nxt][j] = *(const bf16x8*)(sb + j * 4096 + kxb[ks + 1]);
                } else {
                    // issue order inside a step is [halo piece, W tile]; step s-2 issued W(s+1), step s-1 issued [halo piece?, W(s+2)]:
                    // W(s+1) and everything older (incl. the whole next halo at tap 8) has landed once only those may still fly.
                    // A halo piece is issued at step (c, t) iff t < NAH and c + 1 < nc.
                    const bool prev_had_halo = tap >= 1 && tap - 1 < NAH && next_chunk;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // every read of W(s) / this tap's halo rows is retired
                    wait_vm((s + 2 < nsteps ? NB : 0) + (prev_had_halo ? 1 : 0));
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    if (tap < NAH && next_chunk) stage_halo((chunk + 1) & 1, c_first + chunk + 1, tap);
                    if (s + 3 < nsteps) {
                        const int s3 = s + 3, c3 = s3 / 9, t3 = s3 - c3 * 9;
                        stage_w(s3 % 3, c_first + c3, t3);
                    }
                    fa[nxt] = *(const bf16x8*)(narow + ((hi ^ nkeya) << 4));
#pragma unroll
                    for (int j = 0; j < TN; ++j) fb[nxt][j] = *(const bf16x8*)(nsb + j * 4096 + kxb[0]);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cur][j], fa[cur], acc[0][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    __syncthreads();
    if (gridDim.y > 1) {
        GemmArgs q = p;       // partial [slice][M][ldo] fp32 (the launcher passes EPI_F32 without bias / residual)
        q.out = (float*)p.out + (size_t)blockIdx.y * p.M * p.ldo;
        gemm_epilogue<EPI, 1, TN, NW, 2 * ASLOT + 3 * BSLOT, true>(q, acc, wave * 32, n0, lane, wave, smem,
                                                                     (img * p.Hout + y0) * p.Wout + x0, p.Wout);
        return;
    }
    gemm_epilogue<EPI, 1, TN, NW, 2 * ASLOT + 3 * BSLOT, true>(p, acc, wave * 32, n0, lane, wave, smem,
                                                                 (img * p.Hout + y0) * p.Wout + x0, p.Wout);
}

// ---------------------------------------------------------------------------------------------- launch
struct TileCfg { int BM, BN, threads, stages, geglu_ok; };
#define RT_NCFG 9
static const TileCfg kCfg[RT_NCFG] = {{128, 128, 256, 2, 1}, {256, 128, 512, 3, 1}, {256, 160, 512, 3, 0}, {256, 256, 1024, 2, 1},
                                       {256, 160, 768, 3, 0}, {256, 128, 768, 3, 1}, {256, 160, 512, 3, 0}, {256, 256, 512, 2, 1},
                                       {256, 320, 512, 2, 0}};

template <int MODE, int EPI, int BM, int BN, int WM, int WN, int S>
static void launch_cfg(const GemmArgs& a, hipStream_t st) {
    if constexpr (EPI == EPI_GEGLU && (BN / WN / 32) % 2 != 0) {
        throw rt_error(RT_E_INVALID, "gemm: this tile configuration cannot run the GEGLU epilogue");
    } else {
        const size_t lds = (size_t)S * (BM + BN) * BK * 2;
        static bool attr = false;
        if (!attr) {
            HIP_CHECK(hipFuncSetAttribute((const void*)gemm_kernel<MODE, EPI, BM, BN, WM, WN, S>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr = true;
        }
        dim3 grid(cdiv(a.M, BM) * cdiv(a.N, BN)), block(WM * WN * 64);
        hipLaunchKernelGGL((gemm_kernel<MODE, EPI, BM, BN, WM, WN, S>), grid, block, lds, st, a);
    }
}

template <int MODE, int EPI, int BM, int BN, int WM, int WN, int NL>
static void launch_ws(const GemmArgs& a, hipStream_t st) {
    if constexpr (EPI == EPI_GEGLU && (BN / WN / 32) % 2 != 0) {
        throw rt_error(RT_E_INVALID, "gemm: this tile configuration cannot run the GEGLU epilogue");
    } else {
        const size_t lds = (size_t)3 * (BM + BN) * BK * 2;
        static bool attr = false;
        if (!attr) {
            HIP_CHECK(hipFuncSetAttribute((const void*)gemm_ws_kernel<MODE, EPI, BM, BN, WM, WN, NL>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr = true;
        }
        dim3 grid(cdiv(a.M, BM) * cdiv(a.N, BN)), block((WM * WN + NL) * 64);
        hipLaunchKernelGGL((gemm_ws_kernel<MODE, EPI, BM, BN, WM, WN, NL>), grid, block, lds, st, a);
    }
}

template <int MODE, int EPI, int BM, int BN, int WM, int WN>
static void launch_pp(const GemmArgs& a, hipStream_t st) {
    if constexpr (EPI == EPI_GEGLU && (BN / WN / 32) % 2 != 0) {
        throw rt_error(RT_E_INVALID, "gemm: this tile configuration cannot run the GEGLU epilogue");
    } else {
        const size_t lds = (size_t)3 * (BM + BN) * BK * 2;
        static bool attr = false;
        if (!attr) {
            HIP_CHECK(hipFuncSetAttribute((const void*)gemm_pp_kernel<MODE, EPI, BM, BN, WM, WN>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr = true;
        }
        dim3 grid(cdiv(a.M, BM) * cdiv(a.N, BN)), block(WM * WN * 64);
        hipLaunchKernelGGL((gemm_pp_kernel<MODE, EPI, BM, BN, WM, WN>), grid, block, lds, st, a);
    }
}

template <int EPI>
static void launch_g8(const GemmArgs& a, hipStream_t st) {
    constexpr int LDS = 2 * 4 * 128 * BK * 2;
    static bool attr = false;
    if (!attr) {
        HIP_CHECK(hipFuncSetAttribute((const void*)gemm8_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr = true;
    }
    hipLaunchKernelGGL((gemm8_kernel<EPI>), dim3(cdiv(a.M, 256) * cdiv(a.N, 256)), dim3(512), LDS, st, a);
}

template <int MODE, int EPI>
static void launch_me(const GemmArgs& a, int cfg, hipStream_t st) {
    switch (cfg) {
        case 0: launch_cfg<MODE, EPI, 128, 128, 2, 2, 2>(a, st); break;
        case 1: launch_cfg<MODE, EPI, 256, 128, 4, 2, 3>(a, st); break;
        case 2: launch_cfg<MODE, EPI, 256, 160, 8, 1, 3>(a, st); break;
        case 3: launch_cfg<MODE, EPI, 256, 256, 4, 4, 2>(a, st); break;
        case 4: launch_ws<MODE, EPI, 256, 160, 8, 1, 4>(a, st); break;
        case 5: launch_ws<MODE, EPI, 256, 128, 4, 2, 4>(a, st); break;
        case 6: launch_pp<MODE, EPI, 256, 160, 8, 1>(a, st); break;
        case 7:
            if constexpr (MODE == A_DENSE) launch_g8<EPI>(a, st);
            else throw rt_error(RT_E_INVALID, "gemm: the 8-phase configuration is dense-only");
            break;
        case 8: launch_cfg<MODE, EPI, 256, 320, 4, 2, 2>(a, st); break;      // 64x160 per wave: the N = 640 problems (2 column tiles, 224 workgroups at M = 28672)
#ifdef RT_PROBE
        case 9: launch_cfg<MODE, EPI, 256, 256, 2, 4, 2>(a, st); break;      // 8 waves, 128x64 per wave (ties cfg 3)
        case 10: launch_cfg<MODE, EPI, 128, 160, 4, 1, 2>(a, st); break;     // two workgroups per CU: pro/epilogues overlap the partner's loop
        case 11: launch_cfg<MODE, EPI, 128, 160, 4, 1, 3>(a, st); break;
#endif
        default: throw rt_error(RT_E_INVALID, "gemm: bad tile configuration");
    }
}

// (A operand mode, epilogue) combinations that exist on the hot path; anything else is rejected loudly.
static void launch_with_cfg(const GemmArgs& a, int cfg, hipStream_t st) {
    const int key = a.mode * 8 + a.epi;
    switch (key) {
        case A_DENSE * 8 + EPI_BF16: launch_me<A_DENSE, EPI_BF16>(a, cfg, st); break;
        case A_DENSE * 8 + EPI_F32: launch_me<A_DENSE, EPI_F32>(a, cfg, st); break;
        case A_DENSE * 8 + EPI_F16: launch_me<A_DENSE, EPI_F16>(a, cfg, st); break;
        case A_DENSE * 8 + EPI_BF16_TEMB: launch_me<A_DENSE, EPI_BF16_TEMB>(a, cfg, st); break;
        case A_DENSE * 8 + EPI_GEGLU: launch_me<A_DENSE, EPI_GEGLU>(a, cfg, st); break;
        case A_CONV3 * 8 + EPI_BF16: launch_me<A_CONV3, EPI_BF16>(a, cfg, st); break;     // VAE decoder convs (no time embedding) and backward-data
        case A_CONV3 * 8 + EPI_F32: launch_me<A_CONV3, EPI_F32>(a, cfg, st); break;
        case A_CONV3 * 8 + EPI_F16: launch_me<A_CONV3, EPI_F16>(a, cfg, st); break;
        case A_CONV3 * 8 + EPI_BF16_TEMB: launch_me<A_CONV3, EPI_BF16_TEMB>(a, cfg, st); break;
        case A_CONV3_S2 * 8 + EPI_F32: launch_me<A_CONV3_S2, EPI_F32>(a, cfg, st); break;
        case A_CONV3_S2 * 8 + EPI_F16: launch_me<A_CONV3_S2, EPI_F16>(a, cfg, st); break;
        case A_CONV3_UP2 * 8 + EPI_F32: launch_me<A_CONV3_UP2, EPI_F32>(a, cfg, st); break;
        case A_CONV3_UP2 * 8 + EPI_F16: launch_me<A_CONV3_UP2, EPI_F16>(a, cfg, st); break;
        default: throw rt_error(RT_E_UNSUPPORTED, "gemm: (operand mode, epilogue) combination not built");
    }
    HIP_CHECK(hipGetLastError());
}

// Tile configuration choice: measured per problem shape on the real launches (see the in-situ tuner below).  Because all
// configurations are bit-identical in their results, tuning never changes outputs.
static int g_force_cfg = -1;
static int g_conv_patch = 1;
static int g_use16 = 1;
static int g_splitk = 1;
static int g_conv16 = 1;
static int g_xattn = 1;
static int g_no_triple = 0;
static int g_pair = 1;
static int g_cross77 = 1;
static int g_xblock = 0;      // measured slower than the separate launches (LABNOTES R5.2): opt-in
static int g_split_small_rows = 1;      // (A/B, debug bit 30)
static int g_conv3p_tn4 = 1;     // (A/B, debug bit 10)
static int g_conv3p_split = 3;   // round 6: under-filled patch-eligible 3x3 convolutions split over their channel chunks on the patch kernel; debug bit 28: on the split-K implicit GEMM (rounds 2 - 5)
static int g_lnfold = 1;      // round 6: LayerNorm folded into its consumers (gemm16.hip, "LNF"); debug bit 22 restores the LayerNorm launches
#ifdef RT_PROBE
int g_conv3p_tn = 0;          // probe override of the patch kernel's column-tile count
#endif
// bit 0: route eligible convs through the implicit-GEMM kernels; bit 1: keep the 16x16x32 family (gemm16.hip) out (A/B tests)
// bit 2: no split-K; bit 3: stride-1 3x3 convolutions stay on the patch kernel (conv3p_kernel) instead of gemm16.hip's implicit GEMM
// bit 4: cross-attention as to_q GEMM + attention launch instead of the fused kernel (gemm16.hip, EPI_XATTN)
void gemm_set_debug(int flags) {
    g_conv_patch = (flags & 1) ? 0 : 1; g_use16 = (flags & 2) ? 0 : 1; g_splitk = (flags & 4) ? 0 : 1; g_conv16 = (flags & 8) ? 0 : 1;
    g_xattn = (flags & 16) ? 0 : 1;
    g_xblock = (flags & 65536) ? 1 : 0;                               // bit 16: the 640-channel cross-attention block as xblock.hip's ONE launch (opt-in: measured slower, LABNOTES R5.2)
    g_cross77 = (flags & 524288) ? 0 : 1;                             // bit 19: cross-attention on the round-4 kernels (EPI_XATTN / attn_kernel<CROSS>) instead of cross77_kernel
    g_split_small_rows = (flags & 1073741824) ? 0 : 1;
    g_conv3p_tn4 = (flags & 1024) ? 0 : 1;
    g_conv3p_split = (flags & 268435456) ? 0 : ((flags & 536870912) ? 1 : 3); /* bit 29: no two-halves rule for the 32x32 maps */                     // bit 28: under-filled 3x3 convolutions on the split-K implicit GEMM instead of the chunk-split patch kernel
    g_lnfold = (flags & 4194304) ? 0 : 1;                             // bit 22: LayerNorm launches + bf16 projections (rounds 1 - 5) instead of the folded form
    g_pair = (flags & 8192) ? 0 : 1;                                  // bit 13: attn1's Q|K and V^T projections as two launches instead of one grouped launch
    g_no_triple = ((flags & 128) ? 1 : 0) | ((flags & 256) ? 2 : 0) | ((flags & 512) ? 4 : 0) | ((flags & 32768) ? 8 : 0);   // bit 15: dense hi / lo contractions as three launches   // (bits 8 / 9: only the gemm16 / only the patch-kernel route)              // bit 7: the precise VAE's contractions as three launches (round 3) instead of one
}
bool gemm_cross77_enabled() { return g_cross77 != 0; }
bool gemm_lnfold_enabled() { return g_lnfold != 0 && g_use16 != 0 && g_force_cfg < 0; }
bool gemm_xblock_enabled() { return g_xblock != 0 && g_use16 != 0 && g_force_cfg < 0; }
bool gemm_xattn_enabled() { return g_xattn != 0 && g_use16 != 0 && g_force_cfg < 0; }

template <int EPI, bool UP2, int TN>
static void launch_conv3p_tn(const GemmArgs& a, int ntm, int slices, hipStream_t st) {
    constexpr int LDS = 2 * (UP2 ? 104 : 328) * 128 + 3 * 32 * TN * 128;
    static bool attr = false;
    if (!attr) {
        HIP_CHECK(hipFuncSetAttribute((const void*)conv3p_kernel<EPI, UP2, TN>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr = true;
    }
    hipLaunchKernelGGL((conv3p_kernel<EPI, UP2, TN>), dim3(ntm * cdiv(a.N, 32 * TN), slices), dim3(512), LDS, st, a);
}
// slices > 1 (launch_conv3p_splitk): every (patch, column tile) runs `slices` workgroups over disjoint input-channel chunks
template <int EPI, bool UP2>
static void launch_conv3p(const GemmArgs& a, hipStream_t st, int slices = 1) {
    const int ntm = (a.M / a.rows_per_batch) * (a.Hout / 16) * (a.Wout / 16) * slices;
    // Narrower column tiles when 160-channel tiles leave CUs idle or start a mostly empty second round (one workgroup per CU: 145 KB
    // of LDS).  Cost model fitted to tools/probes/conv_bench.hip on MI355X: a 96- / 64-channel workgroup takes 0.72 / 0.55 of a
    // 160-channel one; the big workgroups pay whole rounds of 256, the 64-channel ones 0.4 of the rounding.  Results do not
    // depend on TN (same k order), so the choice may follow the actual batch size.
    auto cost = [&](int tn_, double c) {
        const double r = (double)ntm * cdiv(a.N, 32 * tn_) / 256.0;
        if (r <= 1.0) return c;
        const double up = std::ceil(r);
        return c * (tn_ == 2 ? r + 0.4 * (up - r) : up);
    };
    const double c5 = cost(5, 1.0), c3 = cost(3, 0.72), c2 = cost(2, 0.55);
    int tn = 5;
    double best = c5;
    // 128-channel column tiles where the channel count is a multiple of 128 but not of 160 (the VAE's 128 / 256 / 512-wide layers: a
    // 160-wide tile would multiply 20 - 37 % clamped duplicate columns); round 4, same k order as every other TN
    // (round 6: chunk-split launches may take them at any multiple of 128 - 3 images x 8 slices x 10 tiles of 128 channels are 240 workgroups where 160-channel tiles give 192)
    if (a.N % 128 == 0 && (a.N % 160 != 0 || (slices > 1 && g_conv3p_tn4))) { const double c4 = cost(4, 0.86); if (c4 < best * 0.97) { tn = 4; best = c4; } }
    if (c3 < best * 0.97) { tn = 3; best = c3; }
    if (c2 < best * 0.97) { tn = 2; best = c2; }
#ifdef RT_PROBE
    if (g_conv3p_tn) tn = g_conv3p_tn;
#endif
    if (tn == 2) launch_conv3p_tn<EPI, UP2, 2>(a, ntm / slices, slices, st);
    else if (tn == 3) launch_conv3p_tn<EPI, UP2, 3>(a, ntm / slices, slices, st);
    else if (tn == 4) launch_conv3p_tn<EPI, UP2, 4>(a, ntm / slices, slices, st);
    else launch_conv3p_tn<EPI, UP2, 5>(a, ntm / slices, slices, st);
}
static bool conv_patch_eligible(const GemmArgs& a) {
    if (!g_conv_patch || a.mode == A_DENSE || a.rows_per_batch <= 0 || a.Hout % 16 || a.Wout % 16 || a.Cin % 64 || a.M % a.rows_per_batch) return false;
    if (a.mode == A_CONV3) return a.Hin == a.Hout && a.Win == a.Wout && (a.epi == EPI_BF16 || a.epi == EPI_F32 || a.epi == EPI_F16 || a.epi == EPI_BF16_TEMB);
    if (a.mode == A_CONV3_UP2) return a.Hout == 2 * a.Hin && a.Wout == 2 * a.Win && (a.epi == EPI_F32 || a.epi == EPI_F16);
    return false;
}
void gemm_force_config(int cfg) { g_force_cfg = cfg; }

// ---------------------------------------------------------------------------------------------- tile choice (gemm.hip kernels)
// A pure function of the problem shape - no timing, no state (round 2 ranked the configurations in situ with HIP events on the
// caller's stream: different runs / batch sizes ran different kernels, events were recorded on foreign devices' streams, and a
// stream under capture could not be tuned).  All configurations of this file are bit-identical, so the rule only matters for speed.
// What is left here after gemm16.hip took the dense K % 128 == 0 problems: the implicit-GEMM convolutions that are not patch-eligible
// (conv_in / conv_out / stride-2 downsamplers / tiny maps), 1x1 shortcuts with K = 320 / 960, the text encoders, the VAE's 512-wide layers.
static bool cfg_admissible(const GemmArgs& a, int c) {
    if (a.epi == EPI_GEGLU && !kCfg[c].geglu_ok) return false;
    if (c == 7 && a.mode != A_DENSE) return false;                  // the 8-phase kernel is dense-only
    return true;
}
static int pick_config(const GemmArgs& a) {
    if (g_force_cfg >= 0) return cfg_admissible(a, g_force_cfg) ? g_force_cfg : 0;
    if ((long)a.M * a.N < 256L * 256 * 64) return 0;                  // small problems: 128x128 tiles, two workgroups per CU
    auto wgs = [&](int c) { return (long)cdiv(a.M, kCfg[c].BM) * cdiv(a.N, kCfg[c].BN); };
    int c;
    if (a.epi == EPI_GEGLU) c = 3;
    else if (a.N % 320 == 0 && wgs(8) >= 192) c = 8;
    else if (a.N % 160 == 0) c = 2;
    else if (a.N >= 1024 && a.mode == A_DENSE) c = 3;
    else c = 1;
    if (wgs(c) < 160 && wgs(0) > wgs(c)) c = 0;                      // a big tile that leaves > 1/3 of the CUs idle: smaller tiles
    return c;
}

struct ReduceArgs {
    const float* part; int S; size_t slice;       // S slices of `slice` floats, rows of ldp floats
    int M, N, ldp;
    int epi; const float* bias; const void* res; int ldres; const float* temb; int temb_ld, rows_per_batch;
    void* out; int ldo;
};
__global__ __launch_bounds__(256) void splitk_reduce_kernel(ReduceArgs p) {
    const int NO = p.epi == EPI_GEGLU ? p.N >> 1 : p.N;              // output columns
    const int nv = NO >> 2;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < (size_t)p.M * nv; idx += (size_t)gridDim.x * 256) {
        const int row = (int)(idx / nv), oc = (int)(idx - (size_t)row * nv) * 4;
        // GEGLU: output column oc lives in packed block (oc / 32): value columns 64*blk + (oc % 32), gate columns + 32
        const int c0 = p.epi == EPI_GEGLU ? (oc >> 5) * 64 + (oc & 31) : oc;
        float v[4] = {0, 0, 0, 0}, g[4] = {0, 0, 0, 0};
        for (int s = 0; s < p.S; ++s) {
            const float* src = p.part + (size_t)s * p.slice + (size_t)row * p.ldp + c0;
            const float4 a = *(const float4*)src;
            v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
            if (p.epi == EPI_GEGLU) { const float4 b = *(const float4*)(src + 32); g[0] += b.x; g[1] += b.y; g[2] += b.z; g[3] += b.w; }
        }
        if (p.bias) {
            const float4 b = *(const float4*)(p.bias + c0); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
            if (p.epi == EPI_GEGLU) { const float4 b2 = *(const float4*)(p.bias + c0 + 32); g[0] += b2.x; g[1] += b2.y; g[2] += b2.z; g[3] += b2.w; }
        }
        if (p.epi == EPI_GEGLU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= gelu_erf(g[e]);
        } else if (p.epi == EPI_BF16_TEMB) {
            const float4 t = *(const float4*)(p.temb + (size_t)(row / p.rows_per_batch) * p.temb_ld + oc);
            v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
        }
        if (p.epi == EPI_F32) {
            if (p.res) { const float4 r = *(const float4*)((const float*)p.res + (size_t)row * p.ldres + oc); v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w; }
            *(float4*)((float*)p.out + (size_t)row * p.ldo + oc) = make_float4(v[0], v[1], v[2], v[3]);
        } else if (p.epi == EPI_F16) {
            if (p.res) {
                const uint2 rq = *(const uint2*)((const f16_t*)p.res + (size_t)row * p.ldres + oc);
                const f16_t* rh = (const f16_t*)&rq;
                v[0] += (float)rh[0]; v[1] += (float)rh[1]; v[2] += (float)rh[2]; v[3] += (float)rh[3];
            }
            uint2 w; f16_t* oh = (f16_t*)&w;
            oh[0] = (f16_t)v[0]; oh[1] = (f16_t)v[1]; oh[2] = (f16_t)v[2]; oh[3] = (f16_t)v[3];
            *(uint2*)((f16_t*)p.out + (size_t)row * p.ldo + oc) = w;
        } else {
            uint2 w; w.x = pack_bf16x2(v[0], v[1]); w.y = pack_bf16x2(v[2], v[3]);
            *(uint2*)((bf16_t*)p.out + (size_t)row * p.ldo + oc) = w;
        }
    }
}

// number of K slices for a problem (1 = do not split).  Shape-only rule.
static int splitk_slices(const GemmArgs& a) {
    // callers that batch independent streams pass the tile count of ONE stream; the rule then assumes a nominal batch of four
    // streams whatever the real batch is, so every stream sees the same K slicing alone or in any batch
    long tiles = a.split_tiles > 0 ? 4L * a.split_tiles : (long)cdiv(a.M, 128) * cdiv(a.N, 128);
    // streams of fewer than 128 rows (SD-v1.5's 8x8 maps: 64 tokens) SHARE 128-row tiles: four of them are two row tiles, not four -
    // the nominal count above halved the slices such problems need to fill the chip (3 x 8^2 x 1280 -> 1280: 120 workgroups)
    const int rows1 = a.mode == A_DENSE ? a.rows_per_stream : a.rows_per_batch;
    if (g_split_small_rows && a.split_tiles > 0 && !a.weights_on_rows && rows1 > 0 && rows1 < 128) tiles = (long)cdiv(4 * rows1, 128) * cdiv(a.N, 128);
    const int nk = cdiv(a.K, BK);
    if (tiles > 96 || nk < 8) return 1;
    // Round 6 (tools/small_gemm_bench.py): at 256 tokens per stream a K <= 1280 projection is better off unsplit on the 64-row tiles of the
    // 16x16x32 family than as three K slices + a reduction launch (3 x 256 x 1280 x 1280: 9.3 - 9.8 us against 16.8 - 17.1; the 64-token level
    // keeps the slices: 12.0 against 15.9 us at 3 streams).  A function of ONE stream's shape, like the rest of the rule.
    if (a.mode == A_DENSE && !a.weights_on_rows && a.rows_per_stream >= 256 && a.K <= 1280) return 1;
    int s = (int)(256 / tiles); if (s > 16) s = 16;
    if (s > nk / 4) s = nk / 4;
    return s < 2 ? 1 : s;
}

// Round 6: a patch-eligible 3x3 convolution that the rule above would split (SD-v1.5's 16x16 maps: 3 - 5 images of ONE 16x16 patch each,
// 1280 - 2560 input channels) is split over its INPUT-CHANNEL CHUNKS on the patch kernel instead of over K tiles of the 128x128 implicit
// GEMM: the implicit GEMM copies 32 KB through the L2 -> LDS path per 128x128x64 step (every input pixel nine times), which is what
// bounds its workgroups (0.8 us per K tile whatever the MFMA rate: 3 x 16^2 x 2560 -> 1280 took 91 us for 45 GFLOP), the patch kernel
// 24.6 KB per 256x160x64 step.  Slices = chunks / ceil(chunks / 8): ~8 slices of equal length, a function of Cin alone, so a stream's
// sum order does not depend on the batch.  0: the problem does not take this route.
static int conv3p_split_slices(const GemmArgs& a) {
    if (!g_conv3p_split || !g_splitk || a.A_lo || a.W_lo || a.pair_lo || !(a.mode == A_CONV3 || a.mode == A_CONV3_UP2) || !conv_patch_eligible(a)) return 0;
    const int nc1 = a.Cin >> 6;
    // (measured and dropped: the VAE's single-image layers - SD's 64^2 x 512: 16 patches - sliced to fill 256 CUs: 10.80 vs 10.81 ms per guidance call)
    if (splitk_slices(a) <= 1) {
        // one level up (SD-v1.5's 32x32 maps: 4 patches per image, 640 output channels): even the 64-channel column tiles give a nominal
        // batch of four 160 workgroups (120 at config 1's three streams); two halves over the chunks fill the chip
        const long u2 = 4L * (a.Hout >> 4) * (a.Wout >> 4) * cdiv(a.N, 64);
        return (g_conv3p_split & 2) && u2 <= 192 && nc1 >= 8 ? 2 : 0;
    }
    const int per = cdiv(nc1, 8);
    const int s = cdiv(nc1, per);
    return s >= 2 ? s : 0;
}

// host-only: what launch_gemm does with an under-filled problem.  route 0: one launch, 1: K slices of the 128x128 implicit GEMM / dense GEMM
// + reduction, 2: the patch kernel split over input-channel chunks + reduction
void gemm_split_plan(const GemmArgs& a, int* route, int* slices) {
    const int sp = conv3p_split_slices(a);
    if (sp > 1) { *route = 2; *slices = sp; return; }
    const int s = g_splitk ? splitk_slices(a) : 1;
    *route = s > 1 ? 1 : 0; *slices = s;
}

size_t gemm_splitk_scratch_floats(const GemmArgs& a) {
    const int SP = conv3p_split_slices(a);
    if (SP > 1) return (size_t)SP * a.M * ((a.N + 3) & ~3);
    const int S = splitk_slices(a);
    return S > 1 ? (size_t)S * a.M * ((a.N + 3) & ~3) : 0;
}

// Partial-sum scratch.  An engine hands in its own buffer (GemmArgs.splitk_ws, sized by the dry pass of its plan: nothing is
// allocated inside a forward and the launch is capturable).  Stand-alone callers (rt_op_gemm, the text encoders) fall back to one
// buffer per (device, stream): two streams never share partial sums, and growing it synchronises exactly the stream that uses it.
static float* splitk_fallback_buffer(size_t need, hipStream_t st) {
    struct Buf { float* p = nullptr; size_t floats = 0; };
    static thread_local std::map<std::pair<int, hipStream_t>, Buf> bufs;
    int dev = 0; HIP_CHECK(hipGetDevice(&dev));
    Buf& b = bufs[std::make_pair(dev, st)];
    if (need > b.floats) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(st, &cap);
        if (cap != hipStreamCaptureStatusNone)
            throw rt_error(RT_E_STATE, "gemm: split-K scratch must be provided (GemmArgs.splitk_ws) or warmed up before stream capture");
        HIP_CHECK(hipStreamSynchronize(st));
        if (b.p) (void)hipFree(b.p);
        b.p = nullptr; b.floats = 0;
        HIP_CHECK(hipMalloc((void**)&b.p, need * 4));
        b.floats = need;
    }
    return b.p;
}

static void launch_gemm_splitk(const GemmArgs& a, int S, hipStream_t st) {
    const int ldp = (a.N + 3) & ~3;
    const size_t slice = (size_t)a.M * ldp, need = slice * S;
    float* buf = (a.splitk_ws && a.splitk_ws_floats >= need) ? a.splitk_ws : splitk_fallback_buffer(need, st);
    GemmArgs g = a;
    g.epi = EPI_F32; g.bias = nullptr; g.res = nullptr; g.temb = nullptr; g.out = buf; g.ldo = ldp;
    {
        // 4-deep ring: a slice's workgroup runs alone on its CU and streams cold weights, so the K tiles in flight (3 instead of 1)
        // are what hides the HBM latency (config 1, 2-deep ring: the 8x8 / 16x16 convolutions ran at 0.4 TB/s of weight traffic;
        // 99.9 -> 108.7 steps/s with 4 slots, same with 3 slots or with twice the slices on 2 slots)
        constexpr int BM = 128, BN = 128, SS = 4;
        const size_t lds = (size_t)SS * (BM + BN) * BK * 2;
        dim3 grid(cdiv(a.M, BM) * cdiv(a.N, BN), S), block(256);
#define RT_SPLIT_LAUNCH(MODE_)                                                                                                    \
        {                                                                                                                         \
            static bool attr = false;                                                                                             \
            if (!attr) { HIP_CHECK(hipFuncSetAttribute((const void*)gemm_kernel<MODE_, EPI_F32, BM, BN, 2, 2, SS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; } \
            hipLaunchKernelGGL((gemm_kernel<MODE_, EPI_F32, BM, BN, 2, 2, SS>), grid, block, lds, st, g);                         \
        }
        switch (a.mode) {
            case A_DENSE: RT_SPLIT_LAUNCH(A_DENSE) break;
            case A_CONV3: RT_SPLIT_LAUNCH(A_CONV3) break;
            case A_CONV3_S2: RT_SPLIT_LAUNCH(A_CONV3_S2) break;
            default: RT_SPLIT_LAUNCH(A_CONV3_UP2) break;
        }
#undef RT_SPLIT_LAUNCH
    }
    ReduceArgs r{};
    r.part = buf; r.S = S; r.slice = slice; r.M = a.M; r.N = a.N; r.ldp = ldp; r.epi = a.epi; r.bias = a.bias; r.res = a.res; r.ldres = a.ldres;
    r.temb = a.temb; r.temb_ld = a.temb_ld; r.rows_per_batch = a.rows_per_batch; r.out = a.out; r.ldo = a.ldo;
    const size_t work = (size_t)a.M * ((a.epi == EPI_GEGLU ? a.N / 2 : a.N) / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)std::min<size_t>(cdiv((int)std::min<size_t>(work, 1u << 30), 256), 2048)), dim3(256), 0, st, r);
    HIP_CHECK(hipGetLastError());
}

static void launch_splitk_reduce(const GemmArgs& a, float* buf, int S, size_t slice, int ldp, hipStream_t st) {
    ReduceArgs r{};
    r.part = buf; r.S = S; r.slice = slice; r.M = a.M; r.N = a.N; r.ldp = ldp; r.epi = a.epi; r.bias = a.bias; r.res = a.res; r.ldres = a.ldres;
    r.temb = a.temb; r.temb_ld = a.temb_ld; r.rows_per_batch = a.rows_per_batch; r.out = a.out; r.ldo = a.ldo;
    const size_t work = (size_t)a.M * ((a.epi == EPI_GEGLU ? a.N / 2 : a.N) / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)std::min<size_t>(cdiv((int)std::min<size_t>(work, 1u << 30), 256), 2048)), dim3(256), 0, st, r);
    HIP_CHECK(hipGetLastError());
}
static void launch_conv3p_splitk(const GemmArgs& a, int S, hipStream_t st) {
    const int ldp = (a.N + 3) & ~3;
    const size_t slice = (size_t)a.M * ldp, need = slice * S;
    float* buf = (a.splitk_ws && a.splitk_ws_floats >= need) ? a.splitk_ws : splitk_fallback_buffer(need, st);
    GemmArgs g = a;
    g.epi = EPI_F32; g.bias = nullptr; g.res = nullptr; g.temb = nullptr; g.out = buf; g.ldo = ldp;
    if (a.mode == A_CONV3_UP2) launch_conv3p<EPI_F32, true>(g, st, S); else launch_conv3p<EPI_F32, false>(g, st, S);
    launch_splitk_reduce(a, buf, S, slice, ldp, st);
}

static void check_gemm_args(const GemmArgs& a) {
    RT_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty problem");
    RT_REQUIRE(a.K % 8 == 0 && a.ldw % 8 == 0, "gemm: K and ldw must be multiples of 8");
    RT_REQUIRE(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W & 15) == 0, "gemm: operands must be 16-B aligned");
    if (a.mode == A_DENSE) {
        RT_REQUIRE(a.lda % 8 == 0, "gemm: lda must be a multiple of 8");
        RT_REQUIRE((long)a.M * a.lda < (1L << 31) && (long)a.N * a.ldw < (1L << 31), "gemm: operand too large for 32-bit offsets");
    } else {
        RT_REQUIRE(a.Cin % 8 == 0 && a.K == 9 * a.Cin && a.rows_per_batch == a.Hout * a.Wout, "conv: bad geometry");
        RT_REQUIRE((long)(a.M / a.rows_per_batch + 1) * a.Hin * a.Win * a.Cin < (1L << 31), "conv: input too large for 32-bit offsets");
    }
    if (a.epi == EPI_GEGLU) RT_REQUIRE(a.N % 64 == 0, "geglu: N must be a multiple of 64");
    RT_REQUIRE(a.N % 4 == 0, "gemm: N must be a multiple of 4");
    if (a.epi == EPI_BF16 || a.epi == EPI_BF16_TEMB || a.epi == EPI_F16) RT_REQUIRE(a.N % 8 == 0 && a.ldo % 8 == 0, "gemm: 16-bit output needs N % 8 == 0");
    if (a.epi == EPI_F16 && a.res) RT_REQUIRE(a.ldres % 8 == 0, "gemm: fp16 residual needs ldres % 8 == 0");
    RT_REQUIRE(a.epi >= EPI_BF16 && a.epi <= EPI_F16, "gemm: unknown epilogue");
    if (a.epi == EPI_GEGLU) RT_REQUIRE(a.N % 16 == 0 && a.ldo % 8 == 0, "gemm: GEGLU output needs N % 16 == 0");
    RT_REQUIRE(a.ldo % 4 == 0 && ((uintptr_t)a.out & 15) == 0, "gemm: output must be 16-B aligned with ldo % 4 == 0");
    if (a.res) RT_REQUIRE(a.ldres % 4 == 0 && ((uintptr_t)a.res & 15) == 0, "gemm: residual must be 16-B aligned");
    if (a.bias) RT_REQUIRE(((uintptr_t)a.bias & 15) == 0, "gemm: bias must be 16-B aligned");
    if (a.temb) RT_REQUIRE(a.temb_ld % 4 == 0 && ((uintptr_t)a.temb & 15) == 0, "gemm: temb must be 16-B aligned");
}

// pair output (GemmArgs.pair_lo) exists in the patch convolution's epilogue only: true when launch_gemm takes this problem there
bool gemm_pair_output_ok(const GemmArgs& a) {
    if (a.epi != EPI_F32 || !(a.mode == A_CONV3 || a.mode == A_CONV3_UP2) || g_force_cfg >= 0 || !conv_patch_eligible(a)) return false;
    if ((g_splitk ? splitk_slices(a) : 1) != 1) return false;
    if (a.A_lo || a.W_lo) {                                          // hi / lo contraction: only as ONE launch on the patch kernel
        if (!(a.A_lo && a.W_lo) || (g_no_triple & 1) || (g_no_triple & 4)) return false;
        int ws = 0;
        if (a.mode == A_CONV3 && g_use16 && g_conv16 && g_conv_patch && !a.prefer_patch_conv && gemm16_pick(a, 0, &ws) >= 0) return false;
        return true;
    }
    int ws = 0;
    return !(a.mode == A_CONV3 && g_use16 && g_conv16 && g_conv_patch && !a.prefer_patch_conv && gemm16_pick(a, 0, &ws) >= 0);
}

// LayerNorm fold (gemm16.hip, "LNF"): which route launch_gemm takes is a pure function of the shape, so the engine can ask beforehand
// whether a producer would leave the partials / a consumer has the folded instantiation - and keep the LayerNorm launch otherwise.
int gemm_route16(const GemmArgs& a) {
    if (a.mode != A_DENSE || a.A_lo || a.W_lo || a.pair_lo || g_force_cfg >= 0 || !g_use16) return -1;
    if ((g_splitk ? splitk_slices(a) : 1) != 1) return -1;
    int wstat = 0;
    return gemm16_pick(a, a.weights_on_rows, &wstat);
}
int gemm_ln_emit_bn(const GemmArgs& a_in) {
    GemmArgs a = a_in; a.ln_part = nullptr; a.ln_emit = (float*)(uintptr_t)256;          // (any non-null value: host-side shape test only)
    const int v = gemm_route16(a);
    return a.epi == EPI_F16 && gemm16_ln_variant_ok(a, v) ? gemm16_variant_bn(v) : 0;
}
bool gemm_ln_fold_ok(const GemmArgs& a_in) {
    GemmArgs a = a_in; a.ln_emit = nullptr; a.ln_part = (const float*)(uintptr_t)256;
    return gemm16_ln_variant_ok(a, gemm_route16(a));
}

void launch_gemm(const GemmArgs& a, hipStream_t st) {
    check_gemm_args(a);
    if (a.ln_part || a.ln_emit) RT_REQUIRE(gemm16_ln_variant_ok(a, gemm_route16(a)), "gemm: LayerNorm fold asked of a route without it (gemm_ln_fold_ok / gemm_ln_emit_bn)");
    if (a.pair_lo) RT_REQUIRE(gemm_pair_output_ok(a) && a.ldo % 4 == 0 && ((uintptr_t)a.pair_lo & 7) == 0, "gemm: pair output is only built into the patch convolution (gemm_pair_output_ok)");
    // In-place residual (out == res) is safe: every element is read and written by the same thread.
    // split-K is a function of the shape only (not of a forced tile configuration, not of stream capture): the same problem
    // always takes the same path.  Debug bit 2 (rt_op_gemm_debug(4)) switches it off for A/B tests.
    if (a.A_lo || a.W_lo) {
        // precise VAE contraction A W + A_lo W + A W_lo.  One launch where the kernel runs the three passes as one K loop (3x3
        // convolutions on the gemm16 main loop and on the patch kernel, fp32 output); everywhere else three launches that accumulate
        // in the fp32 output, as round 3 did.
        RT_REQUIRE(a.A_lo && a.W_lo && a.epi == EPI_F32, "gemm: a hi / lo contraction needs both low parts and an fp32 output");
        bool fused = false;
        if ((a.mode == A_CONV3 || a.mode == A_CONV3_UP2) && g_force_cfg < 0 && (g_splitk ? splitk_slices(a) : 1) == 1 && !(g_no_triple & 1)) {
            int ws = 0;
            const bool to16 = a.mode == A_CONV3 && g_use16 && g_conv16 && g_conv_patch && !a.prefer_patch_conv && gemm16_pick(a, 0, &ws) >= 0;
            fused = (to16 && !(g_no_triple & 2)) || (!to16 && conv_patch_eligible(a) && !(g_no_triple & 4));        // exactly the two routes below that take these problems
        }
        if (a.mode == A_DENSE && g_force_cfg < 0 && g_use16 && (g_splitk ? splitk_slices(a) : 1) == 1 && !(g_no_triple & 1) && !(g_no_triple & 8)) {
            int ws = 0;
            fused = gemm16_pick(a, a.weights_on_rows, &ws) >= 0;     // dense on the 16x16x32 family (round 4): the route at the bottom of this function
        }
        if (!fused) {
            RT_REQUIRE(!a.pair_lo, "gemm: pair output on a route without it");      // (unreachable: gemm_pair_output_ok mirrors this routing; kept as a guard)
            GemmArgs b = a; b.A_lo = nullptr; b.W_lo = nullptr;
            launch_gemm(b, st);
            b.bias = nullptr; b.res = a.out; b.ldres = a.ldo;
            b.A = a.A_lo; launch_gemm(b, st);
            b.A = a.A; b.W = a.W_lo; launch_gemm(b, st);
            return;
        }
    }
    const int ksl = g_splitk ? splitk_slices(a) : 1;
    // patch convolutions that cannot fill the chip (< 128 workgroups) go through the split-K implicit GEMM as well
    const bool patch_underfilled = ksl > 1 && a.mode != A_DENSE;
    if (a.mode == A_CONV3 && ksl == 1 && g_force_cfg < 0 && g_use16 && g_conv16 && g_conv_patch && !a.prefer_patch_conv) {
        // stride-1 3x3 convolutions with Cin % 64 == 0: implicit GEMM on the 16x16x32 family's main loop (gemm16.hip, MODE = A_CONV3)
        int wstat = 0;
        const int v = gemm16_pick(a, 0, &wstat);
        if (v >= 0) { RT_REQUIRE(!a.pair_lo, "gemm: pair output on a route without it"); launch_gemm16_variant(a, v, 0, st); return; }
    }
    // only the patch kernel's epilogue writes GemmArgs.pair_lo: every other route below and above would store fp32 into the bf16 hi plane
    const bool to_patch = conv_patch_eligible(a) && !patch_underfilled;
    if (a.pair_lo && !to_patch) throw rt_error(RT_E_INVALID, "gemm: pair output on a route without it");
    if (g_force_cfg < 0 && (patch_underfilled || to_patch)) {
        const int sp = conv3p_split_slices(a);
        if (sp > 1) { launch_conv3p_splitk(a, sp, st); return; }
    }
    if (to_patch) {
        if (a.mode == A_CONV3_UP2) { if (a.epi == EPI_F16) launch_conv3p<EPI_F16, true>(a, st); else launch_conv3p<EPI_F32, true>(a, st); }
        else switch (a.epi) {
            case EPI_BF16: launch_conv3p<EPI_BF16, false>(a, st); break;
            case EPI_F32: launch_conv3p<EPI_F32, false>(a, st); break;
            case EPI_F16: launch_conv3p<EPI_F16, false>(a, st); break;
            default: launch_conv3p<EPI_BF16_TEMB, false>(a, st); break;
        }
        return;
    }
    if (ksl > 1) { launch_gemm_splitk(a, ksl, st); return; }
    if (g_force_cfg < 0 && g_use16) {
        // the 16x16x32 family: tile = pure function of the shape (class from the weight side, row tiling from M), no timing involved
        int wstat = 0;
        const int v = gemm16_pick(a, a.weights_on_rows, &wstat);
        if (v >= 0) { launch_gemm16_variant(a, v, wstat, st); return; }
    }
    launch_with_cfg(a, pick_config(a), st);
}

// Two independent dense problems that read the same activations (attn1: the stacked Q|K projection and V^T = Wv X^T): ONE grouped
// launch of the 16x16x32 family where both take it and a grouped instantiation exists (gemm16.hip, gemm16_dual_kernel: the same tile
// bodies, bit-identical results), otherwise one launch each.  Like every routing decision here a function of the shapes only.
bool gemm_pair_is_grouped(const GemmArgs& a, const GemmArgs& b) {
    return g_pair && g_force_cfg < 0 && g_use16 && a.mode == A_DENSE && b.mode == A_DENSE && (g_splitk ? splitk_slices(a) : 1) == 1 &&
           (g_splitk ? splitk_slices(b) : 1) == 1 && gemm16_pair_variant(a, b) >= 0;
}
void launch_gemm_pair(const GemmArgs& a, const GemmArgs& b, hipStream_t st) {
    if (gemm_pair_is_grouped(a, b)) {
        check_gemm_args(a); check_gemm_args(b);
        if (launch_gemm16_pair(a, b, st)) return;
    }
    launch_gemm(a, st);
    launch_gemm(b, st);
}
