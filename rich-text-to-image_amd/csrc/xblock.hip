// PROBE BUILD ONLY (make PROBES=1; measured 1.2x slower than three launches, LABNOTES R5.2).
// xblock_kernel: the SDXL cross-attention block of the 640-channel level (BASELINE.json's north-star block, shape A: 4096 tokens x
// 640 channels, 10 heads x 64) as ONE launch in which Q, the probabilities and O never leave the registers of the wave that owns
// their tokens:
//     trunk_out = trunk_in + to_out( softmax_fs( to_q(x) K[prompt]^T ) V[prompt] ) + b_out
// (models/attention.py:169-189; processor arithmetic models/attention_processor.py:476-545, font-size softmax :386-401).
//
// Every contraction is written D^T = A . B with the TOKEN as the n index of v_mfma_f32_16x16x32_bf16: a lane (l15, q4) owns token l15 of
// a 16-token tile for the whole kernel.  The MFMA returns, per lane, 4 consecutive m indices of token l15; the next contraction is over
// that m index, and its B operand wants 8 consecutive k slots of token l15 - which two accumulator tiles (2s, 2s+1) hold if the ROWS of
// the A operand of the producing MFMA were staged in the order pi(16 a + 4 q + r) = 32 (a >> 1) + 8 q + 4 (a & 1) + r (the permutation
// attention.hip / EPI_XATTN use for the key rows).  LDS-DMA takes a per-lane source address, so staging rows in permuted order is free:
//     Q^T  = Wq[pi rows] . x^T          x^T  : registers (streamed from HBM during the first ten tiles, 16 B per lane and 32-channel step)
//     S^T  = K[pi rows]  . Q^T          Q^T  : bf16 pairs of the to_q accumulators
//     O^T  = V^T[pi rows]. P^T          P^T  : bf16 pairs of the softmax values
//     out^T= Wo[rho rows]. O^T + (trunk + bias)     O^T: bf16 pairs of the PV accumulators; the accumulators START as trunk + bias;
// rho puts 16 consecutive output channels into one lane, so the fp16 trunk is read and written in 32-B runs per lane = full 128-B lines
// per token row.
// A workgroup = 4 waves, ONE per SIMD with the whole 512-register file each; a wave owns T = 2 token tiles, so every A fragment read
// from LDS (1 KB) feeds two MFMAs (LDS reads at ~50 % of the array's 256 B/clk).  The A operands stream through a 3-slot LDS ring in
// ONE fixed sequence per workgroup: 20 Wq tiles (chunk c = 320 output columns = 5 heads, K tile kt = 64 channels: 40 KB), 5 K / V^T
// tiles (two heads each, 80 staged keys: 44 KB), 20 Wo tiles.  One barrier per tile; tile i+2 is copied into the slot tile i-1 left
// while tile i is multiplied.  Weights are shared by all workgroups (1.6 MB: L2 resident), x and the trunk are read once.
//
// Keys: 77 valid of 96 cached.  Key tiles 0..3 carry keys 0..63 in the pi order; tile 4 carries keys 64..79 in natural order and is
// paired with zeros in the third k step of O^T = V^T P^T, whose V^T chunk for lane group q is staged from keys 64 + 4 q .. + 7 (the
// upper four meet the zeros).
// Softmax: the key mask and the |font size| multipliers enter as an additive bias on the scores (log2 |fs_k|, -inf for keys >= 77)
// with which the S^T accumulators are initialised: exp2(s + log2 w - max) = w exp2(s - max) / C with the same normaliser C in
// numerator and denominator; the sign of a negative font size multiplies the normalised probability (attention_processor.py:392-396).
//
// vmcnt bookkeeping: a boundary waits until at most N VMEM LOADS issued after the next tile's pieces are outstanding (loads retire in
// order; stores and the compiler's own waits can only make the wait stricter).  N = allowed(idx) below follows the issue order
// exactly: pieces of tile idx+2, the x fragments of K tile idx+3 (first chunk), the trunk rows of an output chunk.
#include "xb_common.h"
#include <type_traits>

#define XB_SLOT 45056
#define XB_TAB (3 * XB_SLOT)                   // [96] log2 |fs| / -inf, [96] sign, [640] to_out bias
#define XB_LDS (XB_TAB + 768 + 2560)

#ifdef RT_XB_TIMING
__device__ long long g_xb_stamp[1024 * 8];
void xblock_read_times(long long* dst, int n) { hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_xb_stamp), (size_t)n * 8); }
#define XB_T(i) if (lane == 0 && wave == 0) g_xb_stamp[blockIdx.x * 8 + (i)] = __builtin_readcyclecounter();
#else
#define XB_T(i)
#endif

namespace xb {
constexpr int KT = 10, NQ = 20, NKV = 5, P1 = NQ + NKV, NT = P1 + NQ;      // C = 640: 10 K tiles per projection, 2 chunks
constexpr int pieces(int i) { return i < 0 || i >= NT ? 0 : (i >= NQ && i < P1 ? 11 : 10); }
// loads issued behind the pieces of tile i+2 while tile i is multiplied: x fragments of K tile i+3 (4), the trunk rows of output
// chunk 0 (tile 24) / chunk 1 (tile 33) (20); i = -1: the prologue's x fragments of K tiles 1 and 2
constexpr int extra(int i) { return i == -1 ? 8 : (i >= 0 && i <= 6 ? 4 : (i == 24 || i == 33 ? 20 : 0)); }
constexpr int allowed(int i) { return extra(i - 1) + pieces(i + 2) + extra(i); }
}

template <int N> static __device__ __forceinline__ void xb_boundary() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_s_barrier();
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void xblock_kernel(XBlockArgs p) {
    using namespace xb;
    constexpr int C = 640, T = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q4 = lane >> 4;
    const int m0 = blockIdx.x * 128;
    const int stream = m0 / p.tokens;          // a 128-token workgroup never straddles two streams (tokens % 128 == 0)
    const int prompt = p.prompt[stream], wset = p.wset[stream];
    const bool fs = wset >= 0;
    XB_T(0)

    // ---- LDS-DMA pieces: piece i of this wave copies 8 LDS rows (1 KB) at slot offset (4 i + wave) * 1024.  W tiles: one VGPR per
    // operand; the per-piece part is a constant that goes into the scalar offset: pi(rho + 32) = pi(rho) + 32, rho(r + 64) = rho(r) + 64
    // (the to_out rows need two bases: pieces i even / odd), swizzle keys repeat every 16 rows.
    const int lrow = lane >> 3, pslot = lane & 7;
    const int rho0 = wave * 8 + lrow;
    const int swz0 = ((pslot ^ ((rho0 >> 1) & 7)) << 3);
    const int voff_q = (xb_pi(rho0) * C + swz0) * 2;
    const int voff_o0 = (xb_rho(rho0) * C + swz0) * 2, voff_o1 = (xb_rho(rho0 + 32) * C + swz0) * 2;
    const int voff_v = xb_pi(lane) * p.ldvt * 2;                     // LDS d row `lane` <- V^T row pi(lane)
    auto stage_piece = [&](int idx, int slot_off, int i) {           // idx, i compile-time after unrolling; slot_off wave-uniform
        if (idx < NQ) {
            const int c = idx / KT, kt = idx - c * KT;
            glds16_buf(p.wq, voff_q, ((c * 320 + 32 * i) * C + kt * 64) * 2, smem + slot_off + (i * 4 + wave) * 1024);
        } else if (idx < P1) {
            const int pidx = i * 4 + wave;                           // 0..43 (wave-uniform): 22 pieces per head
            const int h2 = pidx >= 22 ? 1 : 0, pp = pidx - 22 * h2;
            const int head = (idx - NQ) * 2 + h2;
            char* dst = smem + slot_off + h2 * XB_KVH + pp * 1024;   // K pieces 0..9, V^T chunks 10..21: contiguous
            if (pp < 10) {
                const int rho = pp * 8 + lrow, j = rho >> 4, i16 = rho & 15;                     // staged key row: tile j, position i16
                const int key = j < 4 ? 32 * (j >> 1) + 8 * (i16 >> 2) + 4 * (j & 1) + (i16 & 3) : 64 + i16;
                glds16_buf(p.kc, (key * p.ldk + ((pslot ^ ((rho >> 1) & 7)) << 3)) * 2, (prompt * 96 * p.ldk + head * 64) * 2, dst);
            } else {
                const int cc = pp - 10, keyoff = cc < 8 ? 8 * cc : 64 + 4 * (cc - 8);
                glds16_buf(p.vt, voff_v, (head * 64 * p.ldvt + prompt * 96 + keyoff) * 2, dst);
            }
        } else if (idx < NT) {
            const int u = idx - P1, c2 = u / KT, kt = u - c2 * KT;
            glds16_buf(p.wo, (i & 1) ? voff_o1 : voff_o0, ((c2 * 320 + 64 * (i >> 1)) * C + kt * 64) * 2, smem + slot_off + (i * 4 + wave) * 1024);
        }
    };

    // ---- x^T fragments of the wave's T token tiles: channels 32 kk + 8 q4 .. + 7 of token l15; K tile kt needs kk = 2 kt, 2 kt + 1
    bf16x8 xb_[T][2 * KT];
    const bf16_t* xrow[T];
#pragma unroll
    for (int t = 0; t < T; ++t) xrow[t] = p.x + (size_t)(m0 + wave * 32 + t * 16 + l15) * C + 8 * q4;
    auto load_x = [&](int kt) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            xb_[t][2 * kt] = *(const bf16x8*)(xrow[t] + 64 * kt);
            xb_[t][2 * kt + 1] = *(const bf16x8*)(xrow[t] + 64 * kt + 32);
        }
    };
    // score bias / sign tables of this workgroup's stream and the to_out bias: requested first, written to LDS behind the prologue's
    // copies (the compiler's wait for them then leaves everything younger in flight)
    float tw = 1.f, tsg = 1.f, tb[3];
    if (fs && tid < 96) { tw = p.wabs[wset * 96 + tid]; tsg = p.wsgn[wset * 96 + tid]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) tb[i] = (p.bo && tid + 256 * i < C) ? p.bo[tid + 256 * i] : 0.f;
    // ring prologue: tiles 0 and 1 (tile 2 goes out while tile 0 is multiplied), x of K tiles 0..2
#pragma unroll
    for (int i = 0; i < 10; ++i) stage_piece(0, 0, i);
    load_x(0);
#pragma unroll
    for (int i = 0; i < 10; ++i) stage_piece(1, XB_SLOT, i);
    load_x(1); load_x(2);
    float* tabw = (float*)(smem + XB_TAB);
    if (tid < 96) {
        tabw[tid] = tid < p.nk_valid ? __builtin_amdgcn_logf(tw) : -INFINITY;         // v_log_f32 = log2; log2(0) = -inf
        tabw[96 + tid] = tsg;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) if (tid + 256 * i < C) tabw[192 + tid + 256 * i] = tb[i];
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(10 + 8) : "memory");    // tile 0 and the x fragments of K tile 0 landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    XB_T(1)

    int cur = 0, nxt = XB_SLOT, fre = 2 * XB_SLOT;                  // slots of tiles i, i+1, and the free one (tile i+2's)
    auto advance = [&]() { const int t_ = cur; cur = nxt; nxt = fre; fre = t_; };
    const int key = (l15 >> 1) & 7;
    int aoff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) aoff[ks] = l15 * 128 + (((ks * 4 + q4) ^ key) << 4);
    const float* tab = (const float*)(smem + XB_TAB);

    uint4 rres[T][5][2];                                             // fp16 trunk rows of one output chunk: 16 channels per (tile, 64-block)
    auto load_res = [&](int c2) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const f16_t* rr = p.res + (size_t)(m0 + wave * 32 + t * 16 + l15) * p.ldres + 320 * c2 + 16 * q4;
#pragma unroll
            for (int bb = 0; bb < 5; ++bb) { rres[t][bb][0] = *(const uint4*)(rr + 64 * bb); rres[t][bb][1] = *(const uint4*)(rr + 64 * bb + 8); }
        }
    };

    f32x4 acc[T][20];
    // One W tile (320 rows x 64 k) against the B fragments b0 / b1 (k steps 0 / 1) of both token tiles.  The refill of the free slot
    // with tile IDX + 2 and the tile's extra loads are spread over the MFMA groups; fragments are fetched one group ahead; the wait +
    // barrier for tile IDX + 1 sits in front of the LAST group, which multiplies from registers.
    auto wtile = [&](auto idx_c, const bf16x8 (&b0)[T], const bf16x8 (&b1)[T]) {
        constexpr int IDX = decltype(idx_c)::value;
        constexpr int GR = 4, NG = 40 / GR;                          // 4 m tiles per group, 10 groups per tile (5 per k step)
        constexpr int NP = pieces(IDX + 2);
        bf16x8 fa[2][GR];
        const char* base = smem + cur;
#pragma unroll
        for (int r = 0; r < GR; ++r) fa[0][r] = *(const bf16x8*)(base + aoff[0] + r * 2048);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int ks = g / 5, mg = g % 5;
            if (g < NG - 1) {
                const int nks = (g + 1) / 5, nmg = (g + 1) % 5;
#pragma unroll
                for (int r = 0; r < GR; ++r) fa[(g + 1) & 1][r] = *(const bf16x8*)(base + aoff[nks] + (nmg * GR + r) * 2048);
            } else {
                xb_boundary<allowed(IDX)>();
            }
#pragma unroll
            for (int r = 0; r < GR; ++r)
#pragma unroll
                for (int t = 0; t < T; ++t)
                    acc[t][mg * GR + r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[g & 1][r], ks ? b1[t] : b0[t], acc[t][mg * GR + r], 0, 0, 0);
            if (g < NG - 1) {
                if (g < NP) stage_piece(IDX + 2, fre, g);
                if (g == NG - 2) {
#pragma unroll
                    for (int i = NG - 1; i < NP; ++i) stage_piece(IDX + 2, fre, i);
                    if (IDX <= 6) load_x(IDX + 3 <= KT - 1 ? IDX + 3 : KT - 1);
                    if (IDX == 33) load_res(1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        advance();
    };
#define XB_IC(v) std::integral_constant<int, (v)>{}

    // ---- to_q: Q^T[320 c .. + 319][tokens] chunk after chunk; the accumulators leave as the bf16 B fragments of S^T = K Q^T
    bf16x8 qb[T][KT][2];
#define XB_WQ(c_, kt_) { const bf16x8 b0[T] = {xb_[0][2 * (kt_)], xb_[1][2 * (kt_)]}, b1[T] = {xb_[0][2 * (kt_) + 1], xb_[1][2 * (kt_) + 1]}; wtile(XB_IC((c_) * KT + (kt_)), b0, b1); }
#define XB_WQ_CHUNK(c_)                                                                                              \
    {                                                                                                                \
        _Pragma("unroll") for (int t = 0; t < T; ++t) _Pragma("unroll") for (int m = 0; m < 20; ++m) acc[t][m] = f32x4{0.f, 0.f, 0.f, 0.f}; \
        XB_WQ(c_, 0) XB_WQ(c_, 1) XB_WQ(c_, 2) XB_WQ(c_, 3) XB_WQ(c_, 4) XB_WQ(c_, 5) XB_WQ(c_, 6) XB_WQ(c_, 7) XB_WQ(c_, 8) XB_WQ(c_, 9) \
        _Pragma("unroll") for (int t = 0; t < T; ++t) _Pragma("unroll") for (int h = 0; h < 5; ++h) _Pragma("unroll") for (int s = 0; s < 2; ++s) \
            qb[t][(c_) * 5 + h][s] = xb_pack8(acc[t][4 * h + 2 * s], acc[t][4 * h + 2 * s + 1]);                     \
    }
    XB_WQ_CHUNK(0)
    XB_WQ_CHUNK(1)
    XB_T(2)

    // ---- attention: 5 K / V^T tiles of two heads each; per tile four independent units (head, token tile); O^T leaves as to_out's B fragments
    bf16x8 ob[T][KT][2];
    // One head on both token tiles, in three scheduling groups (sched_barrier between them: left alone, hipcc hoists every fragment
    // read of every unit of a tile to the front and spills): S^T (K fragments shared by the two tiles) | softmax | O^T (V^T fragments shared)
    auto head_unit = [&](const char* kbase, const bf16x8 (&q0)[T], const bf16x8 (&q1)[T], bf16x8 (&o0)[T], bf16x8 (&o1)[T]) {
        const char* kp = kbase + l15 * 128;
        const char* vp = kbase + 10240 + (q4 * 64 + l15) * 16;
        const int c0 = ((q4 ^ key) << 4), c1 = (((4 + q4) ^ key) << 4);
        f32x4 s[T][5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            // score bias of the lane's keys: tiles 0..3: 32 (j / 2) + 8 q4 + 4 (j % 2) + r; tile 4: 64 + 4 q4 + r
            const f32x4 bias = *(const f32x4*)(tab + (j < 4 ? 32 * (j >> 1) + 8 * q4 + 4 * (j & 1) : 64 + 4 * q4));
            const bf16x8 k0 = *(const bf16x8*)(kp + j * 2048 + c0), k1 = *(const bf16x8*)(kp + j * 2048 + c1);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                s[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, q0[t], bias, 0, 0, 0);
                s[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, q1[t], s[t][j], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        float inv[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            float mx = xb_max3(s[t][0][0], s[t][0][1], s[t][0][2]);
            mx = xb_max3(mx, s[t][0][3], s[t][1][0]);
            mx = xb_max3(mx, s[t][1][1], s[t][1][2]);
#pragma unroll
            for (int j = 2; j < 5; ++j) { mx = xb_max3(mx, s[t][j - 1][3], s[t][j][0]); mx = xb_max3(mx, s[t][j][1], s[t][j][2]); }
            mx = xb_rowmax(xb_max(mx, s[t][4][3]));
            f32x2 sum2 = {0.f, 0.f};
            const f32x2 nmx = {-mx, -mx};
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const f32x2 d0 = f32x2{s[t][j][0], s[t][j][1]} + nmx, d1 = f32x2{s[t][j][2], s[t][j][3]} + nmx;     // Q carries d^-1/2 log2 e
                s[t][j][0] = __builtin_amdgcn_exp2f(d0.x); s[t][j][1] = __builtin_amdgcn_exp2f(d0.y);
                s[t][j][2] = __builtin_amdgcn_exp2f(d1.x); s[t][j][3] = __builtin_amdgcn_exp2f(d1.y);
                sum2 += f32x2{s[t][j][0], s[t][j][1]} + f32x2{s[t][j][2], s[t][j][3]};
            }
            const float sum = xb_rowsum(sum2.x + sum2.y);
            inv[t] = 1.f / sum;
            if (fs) {                                                // sign of a negative font size on the normalised probability
#pragma unroll
                for (int j = 0; j < 5; ++j) s[t][j] = s[t][j] * *(const f32x4*)(tab + 96 + (j < 4 ? 32 * (j >> 1) + 8 * q4 + 4 * (j & 1) : 64 + 4 * q4));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
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
#pragma unroll
        for (int t = 0; t < T; ++t) { o0[t] = xb_pack8(o[t][0] * inv[t], o[t][1] * inv[t]); o1[t] = xb_pack8(o[t][2] * inv[t], o[t][3] * inv[t]); }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto kvtile = [&](auto t2_c) {
        constexpr int T2 = decltype(t2_c)::value, IDX = NQ + T2;
        constexpr int NP = pieces(IDX + 2);
        // the refill first: nothing in the units depends on it, and its issue slots would otherwise sit behind the softmax
#pragma unroll
        for (int i = 0; i < NP; ++i) stage_piece(IDX + 2, fre, i);
        if (T2 == 4) load_res(0);
        __builtin_amdgcn_sched_barrier(0);
        const char* kb = smem + cur;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int h = 2 * T2 + h2;
            const bf16x8 q0[T] = {qb[0][h][0], qb[1][h][0]}, q1[T] = {qb[0][h][1], qb[1][h][1]};
            bf16x8 o0[T], o1[T];
            head_unit(kb + h2 * XB_KVH, q0, q1, o0, o1);
#pragma unroll
            for (int t = 0; t < T; ++t) { ob[t][h][0] = o0[t]; ob[t][h][1] = o1[t]; }
        }
        xb_boundary<allowed(IDX)>();
        advance();
    };
    kvtile(XB_IC(0)); kvtile(XB_IC(1)); kvtile(XB_IC(2)); kvtile(XB_IC(3)); kvtile(XB_IC(4));
    XB_T(3)

    // ---- to_out on top of trunk + bias, 320 output channels at a time.  Lane (l15, q4), m tiles 4 bb + a: output channels
    // 320 c2 + 64 bb + 16 q4 + 4 a + r = 16 consecutive channels per 64-block
#define XB_WO(c_, kt_) { const bf16x8 b0[T] = {ob[0][kt_][0], ob[1][kt_][0]}, b1[T] = {ob[0][kt_][1], ob[1][kt_][1]}; wtile(XB_IC(P1 + (c_) * KT + (kt_)), b0, b1); }
#define XB_WO_CHUNK(c_)                                                                                              \
    {                                                                                                                \
        _Pragma("unroll") for (int t = 0; t < T; ++t) _Pragma("unroll") for (int bb = 0; bb < 5; ++bb) _Pragma("unroll") for (int a = 0; a < 4; ++a) { \
            const f32x4 bv = *(const f32x4*)(tab + 192 + 320 * (c_) + 64 * bb + 16 * q4 + 4 * a);                    \
            const f16_t* rh = (const f16_t*)&rres[t][bb][a >> 1];                                                    \
            acc[t][4 * bb + a] = f32x4{bv[0] + (float)rh[4 * (a & 1)], bv[1] + (float)rh[4 * (a & 1) + 1], bv[2] + (float)rh[4 * (a & 1) + 2], bv[3] + (float)rh[4 * (a & 1) + 3]}; \
        }                                                                                                            \
        XB_WO(c_, 0) XB_WO(c_, 1) XB_WO(c_, 2) XB_WO(c_, 3) XB_WO(c_, 4) XB_WO(c_, 5) XB_WO(c_, 6) XB_WO(c_, 7) XB_WO(c_, 8) XB_WO(c_, 9) \
        XB_T(4 + (c_))                                                                                               \
        _Pragma("unroll") for (int t = 0; t < T; ++t) {                                                              \
            f16_t* orow = p.out + (size_t)(m0 + wave * 32 + t * 16 + l15) * p.ldo + 320 * (c_) + 16 * q4;            \
            _Pragma("unroll") for (int bb = 0; bb < 5; ++bb) {                                                       \
                uint4 w0, w1; f16_t* oh0 = (f16_t*)&w0; f16_t* oh1 = (f16_t*)&w1;                                    \
                _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int r = 0; r < 4; ++r) {        \
                    oh0[4 * a + r] = (f16_t)acc[t][4 * bb + a][r]; oh1[4 * a + r] = (f16_t)acc[t][4 * bb + 2 + a][r]; \
                }                                                                                                    \
                *(uint4*)(orow + 64 * bb) = w0; *(uint4*)(orow + 64 * bb + 8) = w1;                                  \
            }                                                                                                        \
        }                                                                                                            \
    }
    XB_WO_CHUNK(0)
    XB_WO_CHUNK(1)
    XB_T(6)
#ifdef RT_XB_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    XB_T(7)
#endif
}

bool xblock_supported(int C, int H, int DP, int tokens) {
    // shape A of SDXL: 640 channels = 10 heads x 64.  (1280 channels would need 2 x 160 registers of x^T fragments per lane.)
    return C == 640 && H == 10 && DP == 64 && tokens % 128 == 0;
}

void launch_xblock(const XBlockArgs& a, hipStream_t st) {
    RT_REQUIRE(xblock_supported(a.C, a.H, 64, a.tokens), "xblock: shape");
    RT_REQUIRE(a.M % a.tokens == 0 && a.M / a.tokens <= RT_MAXB && a.nk_valid > 0 && a.nk_valid <= 80, "xblock: streams / keys (at most 80 of the 96 cached keys are staged)");
    RT_REQUIRE(a.ldk % 8 == 0 && a.ldvt % 8 == 0 && a.ldres % 8 == 0 && a.ldo % 8 == 0, "xblock: leading dimensions");
    RT_REQUIRE((long)96 * RT_MAXB * a.ldk * 2 < 0x7fffffffL && (long)a.H * 64 * a.ldvt * 2 < 0x7fffffffL, "xblock: K / V^T cache beyond the 2 GiB descriptor range");
    for (int b = 0; b < a.M / a.tokens; ++b) RT_REQUIRE(a.wset[b] < 0 || (a.wabs && a.wsgn), "xblock: multiplier tables");
    static bool attr = false;
    if (!attr) { HIP_CHECK(hipFuncSetAttribute((const void*)xblock_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, XB_LDS)); attr = true; }
    hipLaunchKernelGGL(xblock_kernel, dim3(a.M / 128), dim3(256), XB_LDS, st, a);
    HIP_CHECK(hipGetLastError());
}
