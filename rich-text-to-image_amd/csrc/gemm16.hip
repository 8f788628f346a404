// bf16 GEMM family on v_mfma_f32_16x16x32_bf16 for gfx950 (round 3): dense C[M,N] = A[M,K] * W[N,K]^T with the UNet's epilogues.
//
// Why a second family next to gemm.hip (32x32x16 tiles): every GEMM of a rich-text step has M = F * h*w with F = 7 streams
// (models/region_diffusion_sdxl.py:779-872 runs 7 UNet forwards per step), i.e. M = 7168 or 28672 = 7 * 2^k.  256-row tiles give
// 28 x 8 = 224 workgroups for the 1280-channel projections (32 of 256 CUs idle) and 4.375 rounds for the GEGLU GEMM; 224-row
// tiles (14 x 16) give exactly 256 / 1280 workgroups.  224 rows cannot be split over 4 SIMDs with 32x32 MFMA tiles; with 16x16
// tiles two M-waves take 7 row tiles each.  The second reason is the LDS array: the 8(M) x 1(N) wave layout that 256x160 tiles force
// on 32x32 tiles makes every wave read the whole W tile (192 KB of ds_read per K tile + 53 KB of LDS-DMA writes: the LDS array is
// busy ~90 % of the MFMA time).  Here a 224x160 tile is computed by 2(M) x 2(N) x 2(K) waves: each wave owns a 112x80 output
// quadrant for ONE HALF of every 64-deep K tile (7 + 5 fragment reads per 35 MFMAs: 96 KB of ds_read per K tile) and the two K
// halves are added through LDS in the epilogue.
//
//   class B ("K-split"): WK = 2.  out = (sum over even 32-deep k steps) + (sum over odd 32-deep k steps), each ascending.
//   class A            : WK = 1.  out = sum over 32-deep k steps, ascending.
// Within a class every tile shape accumulates every output element in the same order, so the ROW tiling (224 / 256 rows) may follow M
// (the batch size) while the class is a pure function of (epilogue, N, K): a stream's result does not depend on which other streams
// share the launch (batch invariance), by construction instead of by the all-configurations-bit-identical rule of gemm.hip.
//
// Main loop: S-slot LDS ring of 64-deep K tiles filled by global_load_lds_dwordx4 (same XOR swizzle on the source address as
// gemm.hip: slot ^= (row>>1)&7, conflict-free ds_read_b128 for the 16-row fragments of the 16x16x32 MFMA as well); the fragments of
// k step u+1 are read into registers WHILE the MFMAs of step u run (A fragments roll through one register set row by row, W
// fragments are double buffered), so a tile's LDS slot is free as soon as its last fragments were read: all S slots are in flight
// behind the MFMAs (gemm.hip keeps the tile being multiplied in LDS: S-1 slots).  One counted s_waitcnt vmcnt + raw s_barrier per K tile.
#include "common.h"
#include <algorithm>
#include <type_traits>

#define BK16 64

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// probe builds only (tools/probes/gemm16_bench.hip -DRT_G16_TIMING): s_memtime stamps of wave 0 of every workgroup at kernel entry,
// first tile landed, end of the K loop, end of the K-split exchange, all stores retired.  RT_G16_ABLATE: 1 = no LDS-DMA inside the
// loop, 2 = no MFMA (operands kept alive), to see which of the two bounds the loop.
#ifdef RT_G16_TIMING
__device__ long long g_g16_times[8192 * 8];
#define G16_T(i) { if (tid == 0) g_g16_times[blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); }
#else
#define G16_T(i)
#endif
#ifndef RT_G16_ABLATE
#define RT_G16_ABLATE 0
#endif

#define RT_G16_OOB 0x7fff0000       // a voffset beyond every descriptor range used here: the lane's 16 bytes arrive as zeros

// ---------------------------------------------------------------------------------------------- LNF: LayerNorm folded into the projections (round 6)
// BasicTransformerBlock normalises the fp16 trunk three times per block (models/attention.py:150,168,181: norm1 -> attn1, norm2 -> attn2,
// norm3 -> ff) and every LayerNorm was its own launch (210 per SDXL forward: 18 MB in, 18 MB out, 9.3 us + a kernel boundary each).
//     LN(x) W^T + b = rstd_m (x (W diag(gamma))^T - mu_m s_n) + c_n,     s_n = sum_k gamma_k W_nk,   c_n = b_n + sum_k beta_k W_nk
// so the consumer GEMM can read the UN-normalised trunk against W' = bf16(gamma W) (derived once per checkpoint: ln_fold_derive_kernel,
// norm.hip) and correct in its epilogue.  The PRODUCER of the trunk - the fp16-trunk epilogue of to_out / ff.net.2 / proj_in - leaves,
// next to the fp16 trunk, xb = bf16(trunk value) (rounded from the same fp32 number: the MFMA operand) and per-row partial sums
// (sum xb, sum xb^2 over 80-column blocks), from which every consumer workgroup tabulates (mu, rstd) of its tokens.
//   LNF = 0  plain kernel (the measured binaries of rounds 3 - 5, unchanged)
//   LNF = 1  consumer, tokens on the rows: A = xb; the workgroup turns the partials of its 224 (...) token rows into an LDS table (mu, rstd)
//            while its first K tiles are in flight; the epilogue applies rstd (acc - mu s_col) + c_col (GEGLU: in front of the gelu)
//   LNF = 3  consumer, tokens on the COLUMNS (V^T = Wv X^T): the same with per-column statistics and per-row s, c
//   LNF = 2  producer (EPI_F16 only): besides the fp16 trunk it stores xb and reduces sum / sum of squares of xb per row and 80-column
//            block in a fixed order (v_dot2c_f32_bf16 per pair, LDS scratch + quad DPP across the items: deterministic)
// Arithmetic: LN of the bf16-ROUNDED trunk, exactly (mu, rstd are the statistics of the numbers the MFMA multiplies); against the
// LayerNorm launch (fp32 LN of the fp16 trunk, output rounded to bf16) the operand error moves from LN(x) to x: a row's error grows by
// sqrt(1 + mu^2 / sigma^2) (measured: tests/test_lnfold_gpu.py).  var = E[x^2] - mu^2 in fp32 from 8 / 16 hierarchical partials.
// (First built with the fp16 trunk itself as the operand of v_mfma_f32_16x16x32_f16 - exact operand, no copy: parity better than the
//  LayerNorm launch, but the f16 MFMA runs the K loops 14 % slower on this chip (GEGLU 149 -> 171 us, same instruction stream:
//  profiles/r6_lnfold_fp16_kernel_stats.txt), which cost more than the LayerNorm launches saved.)
template <bool UNUSED>
static __device__ __forceinline__ f32x4_t g16_mma(const bf16x8& b, const bf16x8& a, const f32x4_t& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c, 0, 0, 0);
}
#define RT_LNF_NONE 0
#define RT_LNF_ROWS 1
#define RT_LNF_EMIT 2
#define RT_LNF_COLS 3
#define RT_LN_BLOCK 80               // columns per partial of the producer (the wave tile of every 160- / 320-column variant)
#define RT_LN_TAB 5120               // bytes of the table behind the ring: [320] (mu, rstd) | [320] s | [320] c of the tile

// ---------------------------------------------------------------------------------------------- EPI_XATTN: fused to_q + cross-attention
// The SDXL cross-attention block (models/attention.py:169-189, models/attention_processor.py:476-545, font-size softmax :386-401)
// ran as to_q GEMM -> 96-key attention -> to_out GEMM; the attention launch does 2.8 GFLOP and is bound by reading Q and writing O
// (18 + 18 MB per launch at 1024 tokens x 1280 channels x 7 streams, 18.6 us + a kernel boundary).  Here the to_q tile never leaves
// the CU: a 128 x 320 output tile = 128 queries of ONE stream x 5 whole heads (d = 64), computed by the class-A main loop
// (2(M) x 4(N) waves, 64 x 80 per wave), is written to LDS as bf16, and every wave then runs the 77-key attention of one 16-query
// row tile, head after head, against K / V^T of the stream's prompt staged per head by LDS-DMA (requested before the last K tile of
// the main loop, so the first two heads land behind it).  Swapped products as in attention.hip: S^T = K Q^T (a lane owns ONE query
// and 24 of its 96 keys: the row statistics need two cross-lane steps), O^T = V^T P^T with P^T fed straight from the S^T
// accumulators - the K rows of key tile j are staged in the order pi_j(i) = 32 (j/2) + 8 (i/4) + 4 (j%2) + i%4, which makes the
// 8 accumulator values of a lane for key tiles (2s, 2s+1) the keys 32 s + 8 q4 + [0, 8): exactly the k slots of the 16x16x32 B
// operand.  O replaces Q in LDS (same rows / columns, wave-private) and leaves in 16-B row-contiguous chunks.
//
// LDS map (163,840 B = all of it; ring = 2 x 57,344): Q / O tile [0, 83,968) (128 rows x 656 B: 16-B pad, conflict-free fragment
// reads), K/V buffer C [83,968, 108,544), multiplier tables [108,544, 109,312), K/V buffers A, B [114,688, 163,840) (beyond the ring:
// may be filled while the main loop still runs).  One K/V buffer = K [96 staged rows][128 B, XOR-swizzled slots] + V^T [12 key
// chunks][64 d][16 B].
#define XA_QS 656
#define XA_KVA 114688
#define XA_KVB 139264
#define XA_KVC 83968
#define XA_WTAB 108544
#define XA_LDS 163840
// (buffer-descriptor LDS-DMA: behind the flat global_load_lds form hipcc's waitcnt pass puts s_waitcnt vmcnt(0) in front of every
// later LDS access, which would serialise the per-head prefetch)
static __device__ __forceinline__ void xattn_stage_head(const GemmArgs& p, char* smem, int kvoff, int head, int prompt, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int pc = i * 8 + wave;                                 // wave-uniform piece id: 0..11 K rows, 12..23 V^T key chunks
        if (pc < 12) {
            const int rho = pc * 8 + (lane >> 3), pslot = lane & 7;  // staged row: key tile j = rho / 16, position i = rho % 16
            const int j = rho >> 4, i16 = rho & 15;
            const int key = 32 * (j >> 1) + 8 * (i16 >> 2) + 4 * (j & 1) + (i16 & 3);
            const int swz = (rho >> 1) & 7;
            glds16_buf(p.xa_k, ((prompt * 96 + key) * p.xa_ldk + head * 64 + ((pslot ^ swz) << 3)) * 2, 0, smem + kvoff + pc * 1024);
        } else {
            const int c = pc - 12;
            glds16_buf(p.xa_vt, ((head * 64 + lane) * p.xa_ldvt + prompt * 96 + c * 8) * 2, 0, smem + kvoff + 12288 + c * 1024);
        }
    }
}

// One (16-query row tile, head) unit of a wave.  qrow = first LDS row of the tile, hcol = byte column of the head inside the Q tile.
#ifdef RT_G16_TIMING
__device__ long long g_xa_seg[8192 * 8];
#define XA_T(i) { const long long now_ = __builtin_readcyclecounter(); xa_acc[i] += now_ - xa_last; xa_last = now_; }
#else
#define XA_T(i)
#endif
template <bool FS>
static __device__ __forceinline__ void xattn_unit(char* smem, int kvoff, int qrow, int hcol, int nk_valid, int lane
#ifdef RT_G16_TIMING
                                                  , long long* xa_acc, long long& xa_last
#endif
                                                  ) {
    const int l15 = lane & 15, q4 = lane >> 4;
    XA_T(0)                                                          // barrier wait + refill issue in front of the unit
    char* qp = smem + (qrow + l15) * XA_QS + hcol;
    const bf16x8 qf0 = *(const bf16x8*)(qp + q4 * 16), qf1 = *(const bf16x8*)(qp + 64 + q4 * 16);
    const char* kp = smem + kvoff + l15 * 128;
    const int swz = (l15 >> 1) & 7;
    const int o0 = ((q4 ^ swz) << 4), o1 = (((4 + q4) ^ swz) << 4);
    f32x4_t s[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        s[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(kp + j * 2048 + o0), qf0, s[j], 0, 0, 0);
        s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(kp + j * 2048 + o1), qf1, s[j], 0, 0, 0);
    }
    XA_T(1)                                                          // Q / K fragment reads + 12 MFMAs
    // softmax over the 96 (77 valid) keys of the lane's query: 24 values here, the rest in lanes l15 + 16 {1, 2, 3}
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = 32 * (j >> 1) + 8 * q4 + 4 * (j & 1) + r;
            if (32 * (j >> 1) + 32 > nk_valid) { if (key >= nk_valid) s[j][r] = -INFINITY; }      // uniform test first
            mx = fmaxf(mx, s[j][r]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
    const float* wt = (const float*)(smem + XA_WTAB);
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float e = __builtin_amdgcn_exp2f(s[j][r] - mx);           // Q carries d^-1/2 log2 e
            if constexpr (FS) {                                       // e_k = exp(s_k - max) |fs_k|, p_k = sign(fs_k) e_k / sum e
                const int key = 32 * (j >> 1) + 8 * q4 + 4 * (j & 1) + r;
                e *= wt[key];
                sum += e;
                e *= wt[96 + key];
            } else sum += e;
            s[j][r] = e;
        }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.f / sum;
    XA_T(2)                                                          // softmax
    // O^T = V^T P^T: three 32-key steps, four 16-row d tiles
    f32x4_t o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const char* vp = smem + kvoff + 12288 + (q4 * 64 + l15) * 16;
#pragma unroll
    for (int st = 0; st < 3; ++st) {
        union { uint32_t u[4]; bf16x8 v; } pf;
        pf.u[0] = pack_bf16x2(s[2 * st][0], s[2 * st][1]); pf.u[1] = pack_bf16x2(s[2 * st][2], s[2 * st][3]);
        pf.u[2] = pack_bf16x2(s[2 * st + 1][0], s[2 * st + 1][1]); pf.u[3] = pack_bf16x2(s[2 * st + 1][2], s[2 * st + 1][3]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(vp + st * 4096 + dt * 256), pf.v, o[dt], 0, 0, 0);
    }
    // O over Q (the unit's own rows / columns; its Q fragments are in registers).  Written by inline ds_write_b64: in front of a
    // compiler-visible LDS store hipcc's waitcnt pass puts s_waitcnt vmcnt(0) (LDS-DMA write-after-write), i.e. every phase would end
    // by waiting for the K / V^T pieces of the heads that were just requested (measured: 3.7 k cycles per phase instead of ~1.5 k,
    // profiles/r4_xattn_probe_v1.txt).  LDS operations of one wave execute in order; the closing barrier carries lgkmcnt(0).
    const unsigned qaddr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(qp + 8 * q4);
#define XA_OWRITE(DT_)                                                                                                         \
    {                                                                                                                          \
        const unsigned lo_ = pack_bf16x2(o[DT_][0] * inv, o[DT_][1] * inv), hi_ = pack_bf16x2(o[DT_][2] * inv, o[DT_][3] * inv); \
        const unsigned long long w64_ = ((unsigned long long)hi_ << 32) | lo_;                                                 \
        asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(qaddr), "v"(w64_), "n"((DT_) * 32) : "memory");                      \
    }
    XA_T(3)                                                          // V^T reads + 12 MFMAs
    XA_OWRITE(0) XA_OWRITE(1) XA_OWRITE(2) XA_OWRITE(3)
#undef XA_OWRITE
    XA_T(4)
}

template <int TMW, int TNW, int WN>
static __device__ __forceinline__ void xattn_tail(const GemmArgs& p, char* smem, f32x4_t (&acc)[TMW][TNW], int m0, int n0, int wm, int wn,
                                                  int wave, int lane, int tid, int prompt, int head0, int wset) {
    static_assert(TMW == 4 && TNW == 5 && WN == 4, "128 x 320 tile, 2 x 4 waves");
    const int l15 = lane & 15, q4 = lane >> 4;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                    // every wave left the ring
    // Q tile -> LDS (bf16): lane (l15, q4) holds row l15 and columns 4 q4 .. +3 of every 16 x 16 tile
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
            uint2 w; w.x = pack_bf16x2(acc[i][j][0], acc[i][j][1]); w.y = pack_bf16x2(acc[i][j][2], acc[i][j][3]);
            *(uint2*)(smem + (wm * 64 + i * 16 + l15) * XA_QS + (wn * 80 + j * 16 + 4 * q4) * 2) = w;
        }
    xattn_stage_head(p, smem, XA_KVC, head0 + 2, prompt, wave, lane);
    const bool fs = wset >= 0;                                       // workgroup-uniform
    if (fs && tid < 96) {
        float* wt = (float*)(smem + XA_WTAB);
        wt[tid] = p.xa_wabs[wset * 96 + tid];
        wt[96 + tid] = p.xa_wsgn[wset * 96 + tid];
    }
    const int qrow = wave * 16;
    G16_T(3)
#ifdef RT_G16_TIMING
    long long xa_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, xa_last = __builtin_readcyclecounter();
#define XA_TARGS , xa_acc, xa_last
#else
#define XA_TARGS
#endif
    // phase h: every wave does (its row tile, head h).  K/V pieces retire in issue order: heads 0, 1 (main loop), 2 (above), 3, 4.
#define XA_PHASE(H_, KV_, BEHIND_, REFILL_HEAD_, REFILL_KV_)                                                          \
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(BEHIND_) : "memory");                           \
    __builtin_amdgcn_s_barrier();                                                                                     \
    if (REFILL_HEAD_ >= 0) xattn_stage_head(p, smem, REFILL_KV_, head0 + (REFILL_HEAD_ < 0 ? 0 : REFILL_HEAD_), prompt, wave, lane); \
    if (fs) xattn_unit<true>(smem, KV_, qrow, (H_) * 128, p.xa_nk_valid, lane XA_TARGS);                              \
    else xattn_unit<false>(smem, KV_, qrow, (H_) * 128, p.xa_nk_valid, lane XA_TARGS);
    XA_PHASE(0, XA_KVA, 6, -1, 0)
    XA_PHASE(1, XA_KVB, 3, 3, XA_KVA)
    XA_PHASE(2, XA_KVC, 3, 4, XA_KVB)
    XA_PHASE(3, XA_KVA, 3, -1, 0)
    XA_PHASE(4, XA_KVB, 0, -1, 0)
#undef XA_PHASE
#undef XA_TARGS
#ifdef RT_G16_TIMING
    if (tid == 0) for (int i = 0; i < 8; ++i) g_xa_seg[blockIdx.x * 8 + i] = xa_acc[i];
#endif
    G16_T(5)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    G16_T(6)
    // O tile -> HBM: 128 rows x 40 chunks of 16 B.  All LDS reads first: behind a global store hipcc's waitcnt pass puts
    // s_waitcnt vmcnt(0) in front of the next LDS read (the LDS-DMA bookkeeping), which serialised the ten stores of a thread.
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t ov[10];
#pragma unroll
    for (int it = 0; it < 10; ++it) {
        const int id = it * 512 + tid;
        const int r = id / 40, ch = id - r * 40;
        ov[it] = *(const u32x4_t*)(smem + r * XA_QS + ch * 16);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 10; ++it) asm volatile("" : "+v"(ov[it]));   // pins the reads above the stores (the scheduler sinks them back otherwise)
#pragma unroll
    for (int it = 0; it < 10; ++it) {
        const int id = it * 512 + tid;
        const int r = id / 40, ch = id - r * 40;
        if (m0 + r < p.M) *(u32x4_t*)((bf16_t*)p.out + (size_t)(m0 + r) * p.ldo + n0 + ch * 8) = ov[it];
    }
#ifdef RT_G16_TIMING
    G16_T(7)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G16_T(4)
#endif
}

// MODE = A_CONV3: 3x3 stride-1 pad-1 convolution as an implicit GEMM on the SAME main loop (M = pixels of the NHWC activation,
// K tile t = (64-channel chunk t / 9, tap t % 9) - round 3: (tap t / nch, chunk t % nch) -, weights packed [Cout][tap][Cin] as everywhere else): only the A loader differs -
// the lane's pixel address shifted by the tap, with the padding taps sent out of the descriptor's range so that they read as zeros.
// It re-reads every input pixel nine times (tap-inner K order: from the vL1D / L2), which the patch kernel of gemm.hip avoids, but runs the 224-row / K-split
// main loop whose LDS array is not the bottleneck: 32^2 x 1280 -> 1280 for 7 streams is a 7168 x 1280 x 11520 GEMM.
//
// The body of one output tile lives in gemm16_body.inl (see there why it is included twice).  `bid_in` = the workgroup's linear id inside
// ITS problem's grid: blockIdx.x for a plain launch, blockIdx.x minus the first problem's workgroups inside a grouped launch.
template <int MODE, int EPI, int TMW, int TNW, int WM, int WN, int WK, int S, int LNF = 0>
static __device__ __forceinline__ void gemm16_tile(const GemmArgs& p, const int wstat, const unsigned bid_in) {
#include "gemm16_body.inl"
}

template <int MODE, int EPI, int TMW, int TNW, int WM, int WN, int WK, int S, int LNF = 0>
__global__ __launch_bounds__(WM * WN * WK * 64, (WM * WN * WK == 4) ? 2 : 1) void gemm16_kernel(GemmArgs p, int wstat) {     // (four-wave forms: two workgroups per CU)
    const unsigned bid_in = blockIdx.x;
#include "gemm16_body.inl"
}

// Grouped launch: two INDEPENDENT dense problems as one grid - attn1's stacked Q|K projection (class A, 224 x 320 / 224 x 256 tiles)
// and V^T = Wv X^T (class B transposed, 160 x 224 tiles) read the same LayerNorm output (models/attention_processor.py:495-506).  The
// first nwg_a workgroups run problem A's tiles, the rest problem B's, each with the unchanged tile body (bit-identical with the two
// separate launches); B's workgroups start on the CUs A's tiles leave, so A's epilogue burst and B's prologue overlap and one kernel
// boundary (launch ramp + the write-back of what A left dirty) disappears per attn1.  B's XCD-aware tile order survives any nwg_a:
// workgroups are dealt to the XCDs round-robin by blockIdx.x, so B's local id & 7 names the hardware XCD rotated by nwg_a & 7 - a
// relabelling of the XCDs, which is all the bijective remap needs (same label <=> same L2).
template <int TMW_A, int TNW_A, int S_A, int TMW_B, int TNW_B, bool LN = false>
__global__ __launch_bounds__(512) void gemm16_dual_kernel(GemmArgs pa, GemmArgs pb, int nwg_a) {
    if ((int)blockIdx.x < nwg_a) gemm16_tile<A_DENSE, EPI_BF16, TMW_A, TNW_A, 2, 4, 1, S_A, LN ? RT_LNF_ROWS : 0>(pa, 0, blockIdx.x);
    else gemm16_tile<A_DENSE, EPI_BF16, TMW_B, TNW_B, 2, 2, 2, 3, LN ? RT_LNF_COLS : 0>(pb, 0, blockIdx.x - (unsigned)nwg_a);
}

#ifdef RT_PROBE
// ---------------------------------------------------------------------------------------------- probe only: the chained panel form (LABNOTES R6.5)
// VERDICT r5 item 1, form (i), in its smallest instance: attn1.to_out (fp16 trunk + residual, producer of the LayerNorm partials) and the
// attn2.to_q that consumes it (LayerNorm-folded) as the two phases of ONE launch of 256 workgroups.  Both problems are 7168 x 1280 x 1280 on the
// same 224 x 160 K-split tile; a to_q tile reads the 224-row panel of xb + its partial sums, i.e. the outputs of the 8 column tiles of its row
// panel - which the default tile map places on ONE XCD (4 panels x 8 column tiles = one XCD's contiguous run of 32).  Hand-off per panel:
// every wave drains its stores (vmcnt(0)), workgroup barrier, one relaxed agent-scope arrival on the panel's counter; consumers poll the
// counter from one lane with s_sleep, barrier, `buffer_inv sc1` (the CU's vector L1 may hold lines of xb from an earlier layer), then the
// unchanged tile body.  NOT for the product: visibility of plain stores through the shared L2 relies on the producer and the consumer
// sitting on one XCD (a placement, not a guarantee), and the spin is bounded (a launch that is not fully resident falls through with wrong
// results instead of hanging the box).  mode bit 0: no wait at all (wrong results: prices the phases without the seam); bit 1: no L1 invalidate; bit 2: `buffer_inv sc0`.
__device__ unsigned g_chain_flags[64];
__device__ long long g_chain_times[256 * 4];
template <int MODE_UNUSED>
__global__ __launch_bounds__(512, 1) void gemm16_chain_kernel(GemmArgs pa, GemmArgs pb, unsigned target, int mode) {
    const int tid = threadIdx.x;
    if (tid == 0) g_chain_times[blockIdx.x * 4 + 0] = __builtin_readcyclecounter();
    gemm16_tile<A_DENSE, EPI_F16, 7, 5, 2, 2, 2, 3, RT_LNF_EMIT>(pa, 0, blockIdx.x);
    // the tile map of the body (default order): XCD-aware bijective remap + groups of 4 tile rows x all tile columns
    const int ntn = (pa.N + 159) / 160, ntm = (pa.M + 223) / 224, nwg = ntm * ntn;
    int bid = blockIdx.x;
    { const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3; bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx; }
    const int gsz = 4 * ntn, first_m = (bid / gsz) * 4, gm = (ntm - first_m) < 4 ? (ntm - first_m) : 4;
    const int tm = first_m + (bid % gsz) % gm;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this wave's stores have left for L2
    __syncthreads();
    if (tid == 0) {
        g_chain_times[blockIdx.x * 4 + 1] = __builtin_readcyclecounter();
        __hip_atomic_fetch_add(&g_chain_flags[tm], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((mode & 1) == 0) {
            int spins = 0;
            while (__hip_atomic_load(&g_chain_flags[tm], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1 << 18)) __builtin_amdgcn_s_sleep(2);
        }
        g_chain_times[blockIdx.x * 4 + 2] = __builtin_readcyclecounter();
    }
    __syncthreads();
    if ((mode & 6) == 0) asm volatile("buffer_inv sc1" ::: "memory");
    else if (mode & 4) asm volatile("buffer_inv sc0" ::: "memory");                    // (mode & 2: no invalidate at all - prices the instruction)
    gemm16_tile<A_DENSE, EPI_BF16, 7, 5, 2, 2, 2, 3, RT_LNF_ROWS>(pb, 0, blockIdx.x);
    if (tid == 0) g_chain_times[blockIdx.x * 4 + 3] = __builtin_readcyclecounter();
}
void launch_gemm16_chain(const GemmArgs& a, const GemmArgs& b, unsigned target, int mode, hipStream_t st) {
    constexpr int LDS = 3 * (224 + 160) * 128 + RT_LN_TAB;
    static bool attr = false;
    if (!attr) { HIP_CHECK(hipFuncSetAttribute((const void*)gemm16_chain_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); attr = true; }
    RT_REQUIRE(a.M == b.M && a.N == 1280 && b.N == 1280 && a.M % 224 == 0 && (a.M / 224) * 8 == 256, "chain probe: 7168 x 1280 projections (256 workgroups, all resident)");
    hipLaunchKernelGGL(gemm16_chain_kernel<0>, dim3(256), dim3(512), LDS, st, a, b, target, mode);
    HIP_CHECK(hipGetLastError());
}
void gemm16_chain_read_times(long long* dst) { HIP_CHECK(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_chain_times), sizeof(long long) * 256 * 4)); }
void gemm16_chain_reset() { unsigned z[64] = {}; HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_chain_flags), z, sizeof(z))); }
#endif

int g_g16_deep = 1;            // (A/B, rt_op_gemm_debug bit 12 clears it) 64 x 160 tiles on a 5-slot ring where K gives >= 8 tiles
int g_g16_tall = 0;            // (A/B, rt_op_gemm_debug bit 27) GEGLU launches that cannot take the W-stationary order run groups of 8 tile rows
void gemm16_set_tall(int on) { g_g16_tall = on; }
void gemm16_set_deep(int on) { g_g16_deep = on; }
// ---------------------------------------------------------------------------------------------- launch
struct G16Var { int BM, BN, WK, S, geglu_ok; };
// variant ids (probe / tests).  Within a class (A: WK = 1, B: WK = 2) all variants are bit-identical per output element.
static const G16Var kVar[RT_G16_NVAR] = {
    {224, 160, 2, 3, 0},   // 0  class B: N = 1280 / 640 / 320 families, M = 7 * 2^k
    {128, 160, 2, 3, 0},   // 1  class B, 128 rows (M = 2^k: the Q|K projection of the 4 non-injected streams, SD-v1.5 batches)
    {224, 256, 1, 2, 1},   // 2  class A: wide N (GEGLU)
    {256, 256, 1, 2, 1},   // 3
    {224, 320, 1, 2, 0},   // 4  class A, 320 columns
    {256, 320, 1, 2, 0},   // 5
    {160, 224, 2, 3, 0},   // 6  class B transposed: V^T = Wv X^T (rows = head channels, columns = tokens)
    {160, 128, 2, 3, 0},   // 7  class B transposed, 128 token columns
    {128, 256, 1, 3, 1},   // 8  class A, 128 rows
    {64, 160, 2, 3, 0},    // 9  class B, 64 rows: batches that leave half of the chip idle on 128-row tiles (the 2-stream plain pass, SD-v1.5's 3 - 5 streams)
    {128, 320, 1, 2, 0},   // 10 class A, 320 columns, 128 rows  } the 640-channel level of the same small batches (its class is A whatever
    {64, 320, 1, 3, 0},    // 11 class A, 320 columns, 64 rows   } the batch: 224-row tiles give 74 workgroups for 2 streams)
    {160, 64, 2, 3, 0},    // 12 class B transposed, 64 token columns (V^T of the same small batches)
    {128, 160, 1, 2, 0},   // 13 (round 5 probe) FOUR waves, 73.7 KB of LDS: two workgroups per CU whose phases interleave (class A arithmetic: no K split)
    // (measured and dropped, profiles/r3_gemm16_probe_v2_timing.txt: FOUR-wave forms of 224x160 / 224x320 / 224x256 - one wave per
    //  SIMD with its accumulators in AGPRs, no K split / no exchange - run their K loop at 46 k instead of 25.5 k cycles: a single
    //  compiler-scheduled wave does not keep the matrix pipe fed.  The kernel template still takes WM*WN*WK == 4.)
};

template <int MODE, int EPI, int TMW, int TNW, int WM, int WN, int WK, int S, int LNF = 0>
static void launch_v(const GemmArgs& a, int wstat, hipStream_t st) {
    constexpr int BM = WM * TMW * 16, BN = WN * TNW * 16;
    constexpr int LDS = S * (BM + BN) * 128 + ((LNF == RT_LNF_ROWS || LNF == RT_LNF_COLS) ? RT_LN_TAB : 0);
    static bool attr = false;
    if (!attr) {
        HIP_CHECK(hipFuncSetAttribute((const void*)gemm16_kernel<MODE, EPI, TMW, TNW, WM, WN, WK, S, LNF>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr = true;
    }
    const int ntn = cdiv(a.N, BN), ntm = cdiv(a.M, BM);
    if (wstat == 1 && (ntn % 8 != 0)) wstat = 0;
    hipLaunchKernelGGL((gemm16_kernel<MODE, EPI, TMW, TNW, WM, WN, WK, S, LNF>), dim3(ntm * ntn), dim3(WM * WN * WK * 64), LDS, st, a, wstat);
    HIP_CHECK(hipGetLastError());
}

// LayerNorm fold ("LNF" at the top of this file): the instantiations that exist.  Producer of the partials: the fp16-trunk epilogue of
// every variant with 80-column wave tiles (160- / 320-column tiles).  Consumers: every bf16-output / GEGLU variant of the dense family -
// tokens on the rows (LNF = 1) or, for the V^T form, on the columns (LNF = 3).  Same tile shapes, wave layouts and k order as the plain
// kernels of the variant, so a variant's class keeps its meaning (batch invariance: gemm16_pick decides exactly as before).
static bool ln_variant_ok(int epi, int v, int lnf) {
    if (lnf == RT_LNF_EMIT) return epi == EPI_F16 && (v == 0 || v == 1 || v == 4 || v == 5 || v == 9 || v == 10 || v == 11);
    if (lnf == RT_LNF_COLS) return epi == EPI_BF16 && (v == 6 || v == 7 || v == 12);
    if (lnf == RT_LNF_ROWS) {
        if (epi == EPI_GEGLU) return v == 2 || v == 3 || v == 8;
        return epi == EPI_BF16 && (v == 0 || v == 1 || v == 2 || v == 3 || v == 4 || v == 5 || v == 8 || v == 9 || v == 10 || v == 11);
    }
    return false;
}
static void launch_ln(const GemmArgs& a, int v, int wstat, int lnf, hipStream_t st) {
    RT_REQUIRE(a.mode == A_DENSE && ln_variant_ok(a.epi, v, lnf), "gemm16: no LayerNorm-fold instantiation of this variant / epilogue");
    if (lnf == RT_LNF_EMIT) {
        RT_REQUIRE(a.N % RT_LN_BLOCK == 0 && a.ldo == a.N && a.ln_copy && ((uintptr_t)a.ln_copy & 15) == 0, "gemm16: partials are emitted per 80-column block of a dense output, next to its bf16 copy");
        switch (v) {
            case 0: launch_v<A_DENSE, EPI_F16, 7, 5, 2, 2, 2, 3, RT_LNF_EMIT>(a, wstat, st); return;
            case 1: launch_v<A_DENSE, EPI_F16, 4, 5, 2, 2, 2, 3, RT_LNF_EMIT>(a, wstat, st); return;
            case 4: launch_v<A_DENSE, EPI_F16, 7, 5, 2, 4, 1, 2, RT_LNF_EMIT>(a, wstat, st); return;
            case 5: launch_v<A_DENSE, EPI_F16, 8, 5, 2, 4, 1, 2, RT_LNF_EMIT>(a, wstat, st); return;
            case 9: if (g_g16_deep && a.K >= 8 * BK16) launch_v<A_DENSE, EPI_F16, 2, 5, 2, 2, 2, 5, RT_LNF_EMIT>(a, wstat, st); else launch_v<A_DENSE, EPI_F16, 2, 5, 2, 2, 2, 3, RT_LNF_EMIT>(a, wstat, st); return;
            case 10: launch_v<A_DENSE, EPI_F16, 4, 5, 2, 4, 1, 2, RT_LNF_EMIT>(a, wstat, st); return;
            default: launch_v<A_DENSE, EPI_F16, 2, 5, 2, 4, 1, 3, RT_LNF_EMIT>(a, wstat, st); return;
        }
    }
    RT_REQUIRE(a.ln_part && a.ln_s && (a.ln_npair == 1 || a.ln_npair == 2 || a.ln_npair == 4) && a.ln_inv_c > 0.f && a.ln_ld >= (a.weights_on_rows ? a.N : a.M) &&
               ((uintptr_t)a.ln_part & 15) == 0 && ((uintptr_t)a.ln_s & 7) == 0, "gemm16: LayerNorm-fold consumer arguments (1, 2 or 4 tile pairs per token, pair-major with ln_ld >= tokens)");
    if (lnf == RT_LNF_COLS) {
        switch (v) {
            case 6: launch_v<A_DENSE, EPI_BF16, 5, 7, 2, 2, 2, 3, RT_LNF_COLS>(a, wstat, st); return;
            case 7: launch_v<A_DENSE, EPI_BF16, 5, 4, 2, 2, 2, 3, RT_LNF_COLS>(a, wstat, st); return;
            default: launch_v<A_DENSE, EPI_BF16, 5, 2, 2, 2, 2, 3, RT_LNF_COLS>(a, wstat, st); return;
        }
    }
    if (a.epi == EPI_GEGLU) {
        switch (v) {
            case 2: launch_v<A_DENSE, EPI_GEGLU, 7, 4, 2, 4, 1, 2, RT_LNF_ROWS>(a, wstat, st); return;
            case 3: launch_v<A_DENSE, EPI_GEGLU, 8, 4, 2, 4, 1, 2, RT_LNF_ROWS>(a, wstat, st); return;
            default: launch_v<A_DENSE, EPI_GEGLU, 4, 4, 2, 4, 1, 3, RT_LNF_ROWS>(a, wstat, st); return;
        }
    }
    switch (v) {
        case 0: launch_v<A_DENSE, EPI_BF16, 7, 5, 2, 2, 2, 3, RT_LNF_ROWS>(a, wstat, st); return;
        case 1: launch_v<A_DENSE, EPI_BF16, 4, 5, 2, 2, 2, 3, RT_LNF_ROWS>(a, wstat, st); return;
        case 2: launch_v<A_DENSE, EPI_BF16, 7, 4, 2, 4, 1, 2, RT_LNF_ROWS>(a, wstat, st); return;
        case 3: launch_v<A_DENSE, EPI_BF16, 8, 4, 2, 4, 1, 2, RT_LNF_ROWS>(a, wstat, st); return;
        case 4: launch_v<A_DENSE, EPI_BF16, 7, 5, 2, 4, 1, 2, RT_LNF_ROWS>(a, wstat, st); return;
        case 5: launch_v<A_DENSE, EPI_BF16, 8, 5, 2, 4, 1, 2, RT_LNF_ROWS>(a, wstat, st); return;
        case 8: launch_v<A_DENSE, EPI_BF16, 4, 4, 2, 4, 1, 3, RT_LNF_ROWS>(a, wstat, st); return;
        case 9: if (g_g16_deep && a.K >= 8 * BK16) launch_v<A_DENSE, EPI_BF16, 2, 5, 2, 2, 2, 5, RT_LNF_ROWS>(a, wstat, st); else launch_v<A_DENSE, EPI_BF16, 2, 5, 2, 2, 2, 3, RT_LNF_ROWS>(a, wstat, st); return;
        case 10: launch_v<A_DENSE, EPI_BF16, 4, 5, 2, 4, 1, 2, RT_LNF_ROWS>(a, wstat, st); return;
        default: launch_v<A_DENSE, EPI_BF16, 2, 5, 2, 4, 1, 3, RT_LNF_ROWS>(a, wstat, st); return;
    }
}
int gemm16_variant_bn(int v) { return v >= 0 && v < RT_G16_NVAR ? kVar[v].BN : 0; }
bool gemm16_ln_variant_ok(const GemmArgs& a, int v) {
    if (a.mode != A_DENSE || v < 0) return false;
    if (a.ln_part) return (a.ln_npair == 1 || a.ln_npair == 2 || a.ln_npair == 4) && ln_variant_ok(a.epi, v, a.weights_on_rows ? RT_LNF_COLS : RT_LNF_ROWS);
    if (a.ln_emit) return a.N % (2 * kVar[v].BN) == 0 && a.N / kVar[v].BN <= 8 && a.ldo == a.N && ln_variant_ok(a.epi, v, RT_LNF_EMIT);
    return true;
}

template <int MODE, int EPI>
static void launch_e(const GemmArgs& a, int v, int wstat, hipStream_t st) {
    switch (v) {
        case 2: launch_v<MODE, EPI, 7, 4, 2, 4, 1, 2>(a, wstat, st); return;
        case 3: if constexpr (MODE == A_DENSE) { launch_v<MODE, EPI, 8, 4, 2, 4, 1, 2>(a, wstat, st); return; } break;
        case 8: if constexpr (MODE == A_DENSE) { launch_v<MODE, EPI, 4, 4, 2, 4, 1, 3>(a, wstat, st); return; } break;
        default: break;
    }
    if constexpr (EPI != EPI_GEGLU) {
        switch (v) {
            case 0: launch_v<MODE, EPI, 7, 5, 2, 2, 2, 3>(a, wstat, st); return;
            case 4: launch_v<MODE, EPI, 7, 5, 2, 4, 1, 2>(a, wstat, st); return;
            default: break;
        }
        if constexpr (MODE == A_DENSE) {
            switch (v) {
                case 1: launch_v<MODE, EPI, 4, 5, 2, 2, 2, 3>(a, wstat, st); return;
                case 5: launch_v<MODE, EPI, 8, 5, 2, 4, 1, 2>(a, wstat, st); return;
                case 6: launch_v<MODE, EPI, 5, 7, 2, 2, 2, 3>(a, wstat, st); return;
                case 7: launch_v<MODE, EPI, 5, 4, 2, 2, 2, 3>(a, wstat, st); return;
                case 9: if (g_g16_deep && a.K >= 8 * BK16 && !a.A_lo) launch_v<MODE, EPI, 2, 5, 2, 2, 2, 5>(a, wstat, st); else launch_v<MODE, EPI, 2, 5, 2, 2, 2, 3>(a, wstat, st); return;
                case 10: launch_v<MODE, EPI, 4, 5, 2, 4, 1, 2>(a, wstat, st); return;
                case 11: launch_v<MODE, EPI, 2, 5, 2, 4, 1, 3>(a, wstat, st); return;
                case 12: launch_v<MODE, EPI, 5, 2, 2, 2, 2, 3>(a, wstat, st); return;
#ifdef RT_PROBES
                case 13: launch_v<MODE, EPI, 4, 5, 2, 2, 1, 2>(a, wstat, st); return;      // (probe: two co-resident four-wave workgroups, LABNOTES R5.6b)
#endif
                default: break;
            }
        }
    }
    throw rt_error(RT_E_INVALID, "gemm16: variant cannot run this operand mode / epilogue");
}

#ifdef RT_PROBES
bool xattn_fused_supported(int C, int H, int DP, int tokens) {
    if (!(DP == 64 && (H * 64) % 320 == 0 && tokens > 0 && tokens % 128 == 0 && C % BK16 == 0 && C >= 3 * BK16)) return false;
    // ... and only where the 128 x 320 tiles of a step's streams stay within one round of the chip: one stream may contribute at
    // most 40 tiles (1024 tokens x 1280 channels: 32; the 4096-token / 640-channel level gives 64 per stream = 448 workgroups for 7
    // streams, 1.75 rounds of a K = 640 loop: measured 51.4 us against 27.6 + 15 for the two launches, profiles/r4_xattn_probe_v1.txt).
    // A function of ONE stream's shape, never of the batch.
    return (tokens / 128) * (H * 64 / 320) <= 40;
}
void launch_xattn_fused(const GemmArgs& a, hipStream_t st) {
    RT_REQUIRE(a.mode == A_DENSE && a.epi == EPI_XATTN && a.N % 320 == 0 && a.K % BK16 == 0 && a.K >= 3 * BK16 && a.lda % 8 == 0 && a.ldw % 8 == 0 &&
               a.ldo % 8 == 0, "xattn: shape");
    RT_REQUIRE(a.xa_tokens > 0 && a.xa_tokens % 128 == 0 && a.M % a.xa_tokens == 0 && a.M / a.xa_tokens <= RT_MAXB, "xattn: tokens per stream must be a multiple of 128");
    RT_REQUIRE(a.xa_k && a.xa_vt && a.xa_ldk % 8 == 0 && a.xa_ldvt % 8 == 0 && a.xa_nk_valid > 0 && a.xa_nk_valid <= 96, "xattn: K / V^T cache");
    RT_REQUIRE((long)a.M * a.lda * 2 < 0x7fffffffL && (long)a.N * a.ldw * 2 < 0x7fffffffL, "xattn: operand beyond the 2 GiB descriptor range");
    for (int b = 0; b < a.M / a.xa_tokens; ++b) RT_REQUIRE(a.xa_wset[b] < 0 || (a.xa_wabs && a.xa_wsgn), "xattn: multiplier tables");
    using K = void (*)(GemmArgs, int);
    K kern = gemm16_kernel<A_DENSE, EPI_XATTN, 4, 5, 2, 4, 1, 2>;
    static bool attr = false;
    if (!attr) { HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, XA_LDS)); attr = true; }
    hipLaunchKernelGGL(kern, dim3(cdiv(a.M, 128) * (a.N / 320)), dim3(512), XA_LDS, st, a, 0);
    HIP_CHECK(hipGetLastError());
}
#else      // the shipped library: the fused to_q + attention launch (measured neutral, LABNOTES R4.1) is a probe-build kernel
bool xattn_fused_supported(int, int, int, int) { return false; }
void launch_xattn_fused(const GemmArgs&, hipStream_t) { throw rt_error(RT_E_UNSUPPORTED, "EPI_XATTN is built with `make PROBES=1` only"); }
bool xblock_supported(int, int, int, int) { return false; }
void launch_xblock(const XBlockArgs&, hipStream_t) { throw rt_error(RT_E_UNSUPPORTED, "xblock_kernel is built with `make PROBES=1` only"); }
#endif

static bool conv16_geometry(const GemmArgs& a) {
    return a.mode == A_CONV3 && a.Hin == a.Hout && a.Win == a.Wout && a.Cin % BK16 == 0 && a.K == 9 * a.Cin && a.rows_per_batch == a.Hout * a.Wout &&
           a.M % a.rows_per_batch == 0 && (long)a.M * a.Cin * 2 < (long)RT_G16_OOB;
}
bool gemm16_supported(const GemmArgs& a) {
    if (a.mode == A_CONV3) {
        if (!conv16_geometry(a)) return false;
        if (!(a.epi == EPI_BF16 || a.epi == EPI_F32 || a.epi == EPI_F16 || a.epi == EPI_BF16_TEMB)) return false;
        return a.ldw % 8 == 0;
    }
    if (a.mode != A_DENSE || a.K % (2 * BK16) != 0 || a.K < 4 * BK16) return false;      // K % 128 (the K-split class runs pairs of K tiles), >= 4 tiles
    if ((a.A_lo || a.W_lo) && !(a.A_lo && a.W_lo && a.epi == EPI_F32)) return false;     // hi / lo contraction: the fp32-output kernels run it as one K loop of three passes
    if (!(a.epi == EPI_BF16 || a.epi == EPI_F32 || a.epi == EPI_F16 || a.epi == EPI_GEGLU)) return false;
    if (a.lda % 8 || a.ldw % 8) return false;
    // the loaders build signed 32-bit BYTE offsets against a 2 GiB buffer descriptor: larger operands stay on gemm.hip (whose
    // launcher checks its own element-offset bound) instead of reading zeros beyond the range
    if ((long)a.M * a.lda * 2 >= 0x7fffffffL || (long)a.N * a.ldw * 2 >= 0x7fffffffL) return false;
    return true;
}

void launch_gemm16_variant(const GemmArgs& a, int v, int wstat, hipStream_t st) {
    RT_REQUIRE(gemm16_supported(a), "gemm16: problem outside the family's domain (dense K % 128 == 0, K >= 256; or a 3x3 stride-1 convolution with Cin % 64 == 0)");
    RT_REQUIRE(v >= 0 && v < RT_G16_NVAR, "gemm16: variant");
    RT_REQUIRE(!(a.ln_part && a.ln_emit), "gemm16: a launch consumes OR produces LayerNorm partials");
    if (a.ln_part) { launch_ln(a, v, wstat, a.weights_on_rows ? RT_LNF_COLS : RT_LNF_ROWS, st); return; }
    if (a.ln_emit) { launch_ln(a, v, wstat, RT_LNF_EMIT, st); return; }
    if (a.mode == A_CONV3) {
        RT_REQUIRE(kVar[v].WK == 1 || (a.K / BK16) % 2 == 0, "gemm16: the K-split class needs an even number of K tiles");
        switch (a.epi) {
            case EPI_BF16: launch_e<A_CONV3, EPI_BF16>(a, v, wstat, st); break;
            case EPI_F32: launch_e<A_CONV3, EPI_F32>(a, v, wstat, st); break;
            case EPI_F16: launch_e<A_CONV3, EPI_F16>(a, v, wstat, st); break;
            default: launch_e<A_CONV3, EPI_BF16_TEMB>(a, v, wstat, st); break;
        }
        return;
    }
    switch (a.epi) {
        case EPI_BF16: launch_e<A_DENSE, EPI_BF16>(a, v, wstat, st); break;
        case EPI_F32: launch_e<A_DENSE, EPI_F32>(a, v, wstat, st); break;
        case EPI_F16: launch_e<A_DENSE, EPI_F16>(a, v, wstat, st); break;
        default: launch_e<A_DENSE, EPI_GEGLU>(a, v, wstat, st); break;
    }
}

template <int TMW_A, int TNW_A, int S_A, int TMW_B, int TNW_B, bool LN = false>
static void launch_dual_v(const GemmArgs& a, const GemmArgs& b, hipStream_t st) {
    constexpr int BMA = 2 * TMW_A * 16, BNA = 4 * TNW_A * 16, BMB = 2 * TMW_B * 16, BNB = 2 * TNW_B * 16;
    constexpr int LDS_A = S_A * (BMA + BNA) * 128, LDS_B = 3 * (BMB + BNB) * 128, LDS = (LDS_A > LDS_B ? LDS_A : LDS_B) + (LN ? RT_LN_TAB : 0);
    static bool attr = false;
    if (!attr) {
        HIP_CHECK(hipFuncSetAttribute((const void*)gemm16_dual_kernel<TMW_A, TNW_A, S_A, TMW_B, TNW_B, LN>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr = true;
    }
    const int nwg_a = cdiv(a.M, BMA) * cdiv(a.N, BNA), nwg_b = cdiv(b.M, BMB) * cdiv(b.N, BNB);
    hipLaunchKernelGGL((gemm16_dual_kernel<TMW_A, TNW_A, S_A, TMW_B, TNW_B, LN>), dim3(nwg_a + nwg_b), dim3(512), LDS, st, a, b, nwg_a);
    HIP_CHECK(hipGetLastError());
}
// Grouped launch of (a: tokens on the rows, class A) + (b: weights on the rows, class B transposed).  Returns false - nothing launched -
// when the pair of tiles gemm16_pick gives the two problems has no grouped instantiation; the caller then launches them one by one.
int gemm16_pair_variant(const GemmArgs& a_in, const GemmArgs& b_in) {            // id of the grouped instantiation (0..3), -1: none
    if (a_in.mode != A_DENSE || b_in.mode != A_DENSE || a_in.epi != EPI_BF16 || b_in.epi != EPI_BF16 || a_in.weights_on_rows || !b_in.weights_on_rows) return -1;
    if (a_in.res || b_in.res || a_in.A_lo || b_in.A_lo) return -1;
    if ((a_in.ln_part != nullptr) != (b_in.ln_part != nullptr) || a_in.ln_emit || b_in.ln_emit) return -1;       // both fold the same LayerNorm or neither
    int wa = 0, wb = 0;
    const int va = gemm16_pick(a_in, 0, &wa), vb = gemm16_pick(b_in, 1, &wb);
    if (wa || wb) return -1;
    // instantiated pairs (Q|K tile, V^T tile): 7 / 4 streams of a rich-text step at both SDXL attention levels; the 2-stream plain pass
    if (vb == 6 && va == 4) return 0;                                 // 224 x 320 + 160 x 224
    if (vb == 6 && va == 2) return 1;                                 // 224 x 256 + 160 x 224
    if (vb == 12 && va == 8) return 2;                                // 128 x 256 + 160 x 64  (2 x 1024 tokens x 1280 channels: 160 + 256 workgroups)
    if (vb == 7 && va == 2) return 3;                                 // 224 x 256 + 160 x 128 (2 x 4096 tokens x 640 channels: 185 + 256)
    return -1;
}
bool launch_gemm16_pair(const GemmArgs& a_in, const GemmArgs& b_in, hipStream_t st) {
    const int va = gemm16_pair_variant(a_in, b_in);
    if (va < 0) return false;
    if (a_in.ln_part) {
        RT_REQUIRE(a_in.ln_s && b_in.ln_s && (a_in.ln_npair == 1 || a_in.ln_npair == 2 || a_in.ln_npair == 4) && b_in.ln_npair == a_in.ln_npair && a_in.ln_inv_c > 0.f &&
                   a_in.ln_ld >= a_in.M && b_in.ln_ld >= b_in.N, "gemm16 pair: LayerNorm-fold consumer arguments");
        switch (va) {
            case 0: launch_dual_v<7, 5, 2, 5, 7, true>(a_in, b_in, st); break;
            case 1: launch_dual_v<7, 4, 2, 5, 7, true>(a_in, b_in, st); break;
            case 2: launch_dual_v<4, 4, 3, 5, 2, true>(a_in, b_in, st); break;
            default: launch_dual_v<7, 4, 2, 5, 4, true>(a_in, b_in, st); break;
        }
        return true;
    }
    switch (va) {
        case 0: launch_dual_v<7, 5, 2, 5, 7>(a_in, b_in, st); break;
        case 1: launch_dual_v<7, 4, 2, 5, 7>(a_in, b_in, st); break;
        case 2: launch_dual_v<4, 4, 3, 5, 2>(a_in, b_in, st); break;
        default: launch_dual_v<7, 4, 2, 5, 4>(a_in, b_in, st); break;
    }
    return true;
}

// Tile choice, a pure function of the shape (no timing): the CLASS (A / B, i.e. the accumulation order of an output element) follows
// the weight side only - (epilogue, N, K), or (M, K) for the V^T product whose weights are the A operand - and the number of rows ONE
// stream contributes (a.rows_per_stream: identical for a stream computed alone or inside any batch); the tile inside the class follows
// the actual extent of the batched dimension.  Returns -1 when the family has no tile for the shape (the caller stays on gemm.hip).
int gemm16_pick(const GemmArgs& a, int weights_on_rows, int* wstat) {
    *wstat = 0;
    if (!gemm16_supported(a)) return -1;
    const int batched = weights_on_rows ? a.N : a.M;                 // extent of the token dimension
    const int rps = a.mode == A_CONV3 ? a.rows_per_batch : (a.rows_per_stream > 0 ? a.rows_per_stream : batched);
    if (rps < 256) return -1;                                        // small maps stay on gemm.hip (128x128 tiles / split-K)
    const double nk = a.K / 64.0;
    // launch time model in units of one K tile of a 224x160 tile: whole rounds of 256 workgroups (one per CU) x (K tiles x tile
    // area + a fixed prologue / epilogue share); tiles of 128 rows feed the matrix pipe ~20 % worse per flop
    // (measured, profiles/r3_gemm16_probe_v1.txt: the steady-state loop follows the bytes a K tile copies into LDS, (BM + BN) rows)
    auto cost = [&](int BMv, int BNv, double eff) {
        const long tiles = (long)cdiv(a.M, BMv) * cdiv(a.N, BNv);
        return (double)((tiles + 255) / 256) * (nk * (BMv + BNv) / 384.0 * eff + 12.0);
    };
    if (a.mode == A_CONV3) {
        // same classes as the dense problems of the same width; the K-split class needs an even K-tile count (Cin / 64 even)
        // ... and only where ONE image contributes enough tiles that the batches of a step fill the chip (the rule may not look at
        // the batch: a stream must take the same path alone and inside a batch).  SDXL: 38 - 74 tiles per image at every level;
        // SD-v1.5's 32^2 / 16^2 levels: 20 / 16, three to seven of which leave most CUs idle - they stay on the patch kernel
        // (config 1, same box: 126 steps/s with them here, 134 on the patch kernel).
        const bool wide = a.N % 320 == 0 && (long)cdiv(rps, 224) * (a.N / 320) >= 32;
        if (wide) return 4;
        if (a.N % 160 == 0 && (a.K / BK16) % 2 == 0) return (long)cdiv(rps, 224) * (a.N / 160) >= 30 ? 0 : -1;
        if (a.N % 256 == 0) return (long)cdiv(rps, 224) * (a.N / 256) >= 30 ? 2 : -1;
        return -1;
    }
    // Small batches (the 2-stream plain pass, SD-v1.5's 3 - 5 streams): where the tiles above would leave half of the chip idle, 64-row
    // (64-token) tiles of the SAME class that still fit one round of 256 workgroups.  Measured (profiles/r4_gemm16_probe_small_batch.txt):
    // 2048 x 1280 x 5120 39.6 -> 34.8 us, 3072 x 640 x 640 9.6 -> 7.6, 768 x 1280 x 1280 12.8 -> 9.8 (class B); 8192 x 640 x 640
    // 22.2 -> 11.8, 8192 x 640 x 2560 59.0 -> 31.0, 8192 x 1280 x 640 20.5 -> 16.6 (class A); a second round loses (5120 x 640 x 640: 10.7 ->
    // 13.7).  Same class => same bits, so - unlike the class - this choice may look at the batch.
    if (weights_on_rows) {
        if (a.epi != EPI_BF16 || a.M % 160 != 0) return -1;
        if ((long)(a.M / 160) * cdiv(a.N, 64) <= 256) return 12;
        return cost(160, 224, 1.0) <= cost(160, 128, 1.2) ? 6 : 7;
    }
    if (a.epi == EPI_GEGLU) {
        if (a.N % 256 != 0) return -1;
        *wstat = (a.N / 256) % 8 == 0 ? 1 : (g_g16_tall ? 2 : 0);
        const double c2 = cost(224, 256, 1.0), c3 = cost(256, 256, 1.0), c8 = cost(128, 256, 1.2);
        return c2 <= c3 && c2 <= c8 ? 2 : (c3 <= c8 ? 3 : 8);
    }
    // Class A (bit-identical with the 32x32x16 kernels of gemm.hip as well: same k order, measured bit for bit) on 320-column tiles
    // moves the fewest L2 -> LDS bytes per flop (7.6 B/kFLOP against 10.7 for 224x160) and wins wherever ONE stream contributes
    // enough 320-wide tiles to fill the chip at the step's batch sizes; the 1280-channel projections (5 x 4 tiles per stream) do not:
    // they take the K-split class B on 160-column tiles.  The decision uses rows_per_stream, never the batch.
    const bool wide320 = a.N % 320 == 0 && (long)cdiv(rps, 224) * (a.N / 320) >= 32;
    if (wide320) {
        static const int cand[4][3] = {{224, 320, 4}, {224, 256, 2}, {256, 256, 3}, {128, 256, 8}};   // all class A: free choice
        int best = -1; double bc = 1e300;
        for (auto& c : cand) {
            if (a.N % c[1] != 0) continue;
            if (a.ln_emit && c[1] != 320) continue;                  // a producer of LayerNorm partials stays on the 80-column wave tiles (same class, same bits)
            const double cc = cost(c[0], c[1], c[0] == 128 ? 1.2 : 1.0);
            if (cc < bc) { bc = cc; best = c[2]; }
        }
        if (best >= 0 && (long)cdiv(a.M, kVar[best].BM) * (a.N / kVar[best].BN) <= 128) {     // half of the chip or more would idle
            if ((long)cdiv(a.M, 64) * (a.N / 320) <= 256) return 11;
            if ((long)cdiv(a.M, 128) * (a.N / 320) <= 256) return 10;
        }
        return best;
    }
    if (a.N % 160 == 0) {                                                             // class B
        if ((long)cdiv(a.M, 64) * (a.N / 160) <= 256) return 9;
        return cost(224, 160, 1.0) <= cost(128, 160, 1.2) ? 0 : 1;
    }
    if (a.N % 256 == 0) {                                                             // widths that are multiples of 256 only (VAE 256 / 512, CLIP 768 / 1024 / 3072): class A
        const double c2 = cost(224, 256, 1.0), c3 = cost(256, 256, 1.0), c8 = cost(128, 256, 1.2);
        return c2 <= c3 && c2 <= c8 ? 2 : (c3 <= c8 ? 3 : 8);
    }
    return -1;
}


#ifdef RT_G16_TIMING
void gemm16_read_times(long long* dst, int n) { HIP_CHECK(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_g16_times), (size_t)n * 8)); }
void gemm16_read_xa_segments(long long* dst, int n) { HIP_CHECK(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_xa_seg), (size_t)n * 8)); }
#endif
