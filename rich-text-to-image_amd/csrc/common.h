// Shared device/host helpers for the region-diffusion engine (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <stdexcept>

typedef uint16_t bf16_t;   // raw bf16 bits in memory
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define RT_OK 0
#define RT_E_INVALID (-1)
#define RT_E_HIP (-2)
#define RT_E_STATE (-3)
#define RT_E_MISSING_WEIGHT (-4)
#define RT_E_UNSUPPORTED (-5)

struct rt_error : std::runtime_error {
    int code;
    rt_error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define HIP_CHECK(expr)                                                                        \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            throw rt_error(RT_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));       \
    } while (0)

#define RT_REQUIRE(cond, msg)                                                                  \
    do {                                                                                       \
        if (!(cond)) throw rt_error(RT_E_INVALID, std::string(msg) + " [" #cond "]");          \
    } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// fp32 -> bf16, round-to-nearest-even, NaN stays NaN: gfx950's v_cvt_pk_bf16_f32 (one instruction per PAIR; rounds 1-2 carried a
// 12-instruction integer sequence with an exec-mask branch for the NaN case per VALUE, which made the bf16 epilogues VALU bound)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 hw_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_bf16x2));
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, f) & 0xffffu); }
// "low part" of a value whose high part is its bf16 rounding: v ~= bf16(v) + bf16(v - bf16(v)) to 2^-17 relative.  The precise mode
// of the VAE engine (vae.hip) multiplies such pairs with three bf16 MFMA passes (hi*hi + lo*hi + hi*lo, fp32 accumulation).
__device__ __forceinline__ float bf16_residual(float v) { return v - bf16_to_f32(f32_to_bf16(v)); }
__device__ __forceinline__ uint32_t pack_bf16x2_lo(float a, float b) { return pack_bf16x2(bf16_residual(a), bf16_residual(b)); }

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// gelu(x) = 0.5 x (1 + erf(x / sqrt 2)), erf(x / sqrt 2) as an odd degree-17 polynomial P of clamp(x, +-4.25) with P(4.25) == 1
// (weighted least-squares minimax fit, tools/fit_gelu.py): |gelu error| <= 6.4e-5 absolute everywhere (below half an ulp of the bf16
// output for |gelu| > 0.03, exact saturation 0 / x outside the clamp), no transcendental instruction.  The GEGLU epilogues evaluate
// it on 56 gate values per lane and tile; round 3 measured them VALU bound on the previous rcp + exp2 form (2 quarter-rate
// instructions + 13 scalar FMAs per value).  The x2 form maps to v_pk_mul_f32 / v_pk_fma_f32, two gates per issue slot; both forms
// run the same IEEE fma sequence and are bit-identical.
#define RT_GELU_CLAMP 4.25f
#define RT_GELU_C8 1.203002836e-10f
#define RT_GELU_COEFS(K)                                                                                                                  \
    K(-1.137335648e-08f) K(4.748426363e-07f) K(-1.167462415e-05f) K(1.911683503e-04f) K(-2.242938848e-03f)           \
    K(1.971663348e-02f) K(-1.328222901e-01f) K(7.978754640e-01f)
__device__ __forceinline__ f32x2 gelu_erf_x2(f32x2 x) {
    f32x2 xc;
    xc.x = __builtin_amdgcn_fmed3f(x.x, -RT_GELU_CLAMP, RT_GELU_CLAMP);
    xc.y = __builtin_amdgcn_fmed3f(x.y, -RT_GELU_CLAMP, RT_GELU_CLAMP);
    const f32x2 s = xc * xc;
    f32x2 q = {RT_GELU_C8, RT_GELU_C8};
#define RT_K(c) q = __builtin_elementwise_fma(q, s, (f32x2){c, c});
    RT_GELU_COEFS(RT_K)
#undef RT_K
    const f32x2 pe = xc * q;                                        // erf(x / sqrt 2)
    const f32x2 h = x * 0.5f;
    return __builtin_elementwise_fma(h, pe, h);
}
__device__ __forceinline__ float gelu_erf(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -RT_GELU_CLAMP, RT_GELU_CLAMP);
    const float s = xc * xc;
    float q = RT_GELU_C8;
#define RT_K(c) q = __builtin_fmaf(q, s, c);
    RT_GELU_COEFS(RT_K)
#undef RT_K
    const float pe = xc * q;
    const float h = x * 0.5f;
    return __builtin_fmaf(h, pe, h);
}

// async global -> LDS copy of 16 B per lane (LDS destination = wave-uniform base + lane*16)
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// LDS-DMA through a buffer descriptor: buffer_load_dwordx4 v(voff), s[rsrc], s(soff) offen lds.  `base` must be wave-uniform; the
// descriptor is rebuilt from it at every call site (4 SALU moves, hoisted by the compiler).  Kept in a non-template __device__
// function: the descriptor type exists in device compilation only and a kernel TEMPLATE that names it loses its host-side stub.
static __device__ __forceinline__ void glds16_buf(const void* base, int voff_bytes, int soff_bytes, void* lds_wave_base, unsigned range_bytes = 0x7fffffffu) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, range_bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff_bytes, soff_bytes, 0, 0);
}
// ---------------------------------------------------------------- GEMM / implicit-GEMM conv
enum { A_DENSE = 0, A_CONV3 = 1, A_CONV3_S2 = 2, A_CONV3_UP2 = 3 };
enum { EPI_BF16 = 0, EPI_F32 = 1, EPI_BF16_TEMB = 2, EPI_GEGLU = 3, EPI_F16 = 4,   // EPI_F16: fp16 output (+ fp16 residual): the UNet trunk
       EPI_XATTN = 5 };   // gemm16.hip only: the to_q projection whose tile never leaves the CU - 77-key cross-attention on it, O written (launch_xattn_fused)
typedef _Float16 f16_t;
#define RT_MAXB 16

struct GemmArgs {
    const bf16_t* A;   // dense: [M, lda]; conv: NHWC input [B, Hin, Win, Cin]
    const bf16_t* W;   // [N, ldw] (K contiguous)
    const float* bias; // [N] or null
    void* out;         // bf16 or f32, [M, ldo]
    const void* res;   // optional residual [M, ldres]: fp32 for EPI_F32, fp16 for EPI_F16
    const float* temb; // EPI_BF16_TEMB: [B, temb_ld]
    const bf16_t* zero; // >= 16 B of zeros
    int mode, epi;
    int M, N, K;
    int lda, ldw, ldo, ldres, temb_ld;
    int rows_per_batch;              // Hout*Wout (conv and temb)
    int Hin, Win, Cin, Hout, Wout;   // conv geometry (Cin padded to a multiple of 8; K = 9*Cin)
    int debug;                       // unused by the product kernels (kept so probe builds can pass flags without changing the ABI of the struct)
    int rows_per_stream;             // tokens ONE stream contributes to the batched dimension (0: unknown): keys the tile-class choice of
                                     // gemm16.hip so that a stream's result does not depend on the other streams of the launch
    float* splitk_ws;                // optional split-K partial-sum scratch owned by the caller (engine workspace); null: per-(device, stream) fallback
    size_t splitk_ws_floats;
    int prefer_patch_conv;           // 1: stride-1 3x3 convolution that should stay on the patch kernel (single-image VAE layers whose 224-row GEMM
                                     // tiles would leave most of the chip idle: the caller decides on its own, batch-free, shape)
    int weights_on_rows;             // 1: the V^T product (A = weight rows, W = token rows): tile-class choices key on M instead of N
    int split_tiles;                 // 128x128 output tiles of ONE batch entry's share of the problem (0: unknown).  The split-K rule
                                     // uses it instead of the actual tile count, so a stream's result does not depend on how many
                                     // other streams share the launch (batch invariance); see splitk_slices in gemm.hip
    // Triple-pass contraction of the VAE's precise mode (vae.hip): out = A W^T + A_lo W^T + A W_lo^T accumulated in ONE fp32 accumulator
    // (operands as bf16 pairs v = hi + lo).  Null: single pass.  The 3x3 convolution kernels run it as one launch with a three times
    // longer K loop (no fp32 read-modify-write of the output between passes); launch_gemm falls back to three launches elsewhere.
    const bf16_t* A_lo; const bf16_t* W_lo;
    // EPI_F32 with pair_lo != null (precise VAE, 3x3 convolutions on the patch kernel only: gemm_pair_output_ok): the fp32 result leaves
    // as the bf16 pair the NEXT contraction consumes - `out` = hi plane, `pair_lo` = lo plane, both bf16 [M, ldo] - instead of fp32
    // that a cast kernel would read back and split (8 B of traffic per element saved; same bits: pack_bf16x2 / pack_bf16x2_lo)
    bf16_t* pair_lo;
    // EPI_XATTN (launch_xattn_fused): A = LayerNorm'd tokens [B * xa_tokens, K], W = packed to_q [H * 64, K] (pre-scaled by
    // d^-1/2 log2 e), out = attention output O [M, ldo] bf16.  K / V^T: the per-prompt cross-attention cache (AttnArgs layout).
    const bf16_t* xa_k; const bf16_t* xa_vt;
    int xa_ldk, xa_ldvt;
    int xa_tokens;                   // query tokens per stream (multiple of 128: a 128-row tile never straddles two streams)
    int xa_nk_valid;                 // keys >= xa_nk_valid of the 96 cached rows are masked (77)
    const float* xa_wabs; const float* xa_wsgn;      // [nsets, 96] font-size multipliers (attention_processor.py:386-401)
    int xa_prompt[RT_MAXB], xa_wset[RT_MAXB];        // per stream: prompt index into the cache, multiplier set (-1: plain softmax)
    // LayerNorm folded into the projections (gemm16.hip, "LNF"; models/attention.py:150,168,181).
    //  consumer (ln_part != null; EPI_BF16 / EPI_GEGLU on the 16x16x32 family only): the token operand (A, or W when weights_on_rows) is
    //    xb = the UN-normalised trunk as bf16, the weight operand is W' = bf16(gamma W) (ln_fold_derive), ln_s = [weight rows] (s, c)
    //    pairs: s = row sums of W', c = b + W beta (`bias` is unused); out = rstd (xb W'^T - mu s) + c with (mu, rstd) of every token
    //    from ln_part: one (sum, sum of squares) of xb per token and column tile of the producer, pair-major [ln_npair][ln_ld] float4
    //    = two tiles each (ln_npair = 1, 2 or 4)
    //  producer (ln_emit != null; EPI_F16 on the 16x16x32 family's 80-column wave tiles): besides the fp16 trunk leaves xb in ln_copy
    //    [M, ldo] and that array for its OUTPUT rows in ln_emit (ln_ld = M)
    const float* ln_part; int ln_npair;
    int ln_ld;
    const float* ln_s;
    float ln_inv_c, ln_eps;
    bf16_t* ln_copy;
    float* ln_emit;
};
// LayerNorm fold plumbing (host-only predicates are pure functions of the shape, like every routing decision)
int gemm_route16(const GemmArgs& a);                    // variant of the 16x16x32 family launch_gemm would take for this problem, -1: another route
int gemm_ln_emit_bn(const GemmArgs& a);                 // column-tile width (160 / 320) of the partials launch_gemm(a) with a.ln_emit set would leave; 0: this route has no such epilogue
bool gemm_ln_fold_ok(const GemmArgs& a);                // launch_gemm(a) with a.ln_part set has a folded instantiation
bool gemm16_ln_variant_ok(const GemmArgs& a, int v);    // gemm16.hip: variant v has the LayerNorm-fold instantiation a.ln_part / a.ln_emit ask for
void launch_ln_partials(const f16_t* x, bf16_t* xb, float* part, int rows, int C, int bn, hipStream_t st);      // stand-alone producer of xb + partials (column tiles of bn = 160 / 320) from an fp16 trunk (norm.hip)
void launch_ln_fold_derive(const bf16_t* W, int ldw, const float* bias, const float* gamma, const float* beta, int N, int K,
                           bf16_t* Wf, float* sc, hipStream_t st);                         // W' = bf16(gamma W), sc[n] = (row sum of W', bias + W beta)
int gemm16_variant_bn(int v);                           // column-tile width of a gemm16 variant
// Fused to_q -> 77-key cross-attention (one launch instead of two; Q never reaches HBM).  Eligible: head dim 64, H * 64 % 320 == 0,
// tokens % 128 == 0, C % 64 == 0, C >= 192.
bool xattn_fused_supported(int C, int H, int DP, int tokens);
void launch_xattn_fused(const GemmArgs& a, hipStream_t st);
// xblock.hip: the whole cross-attention block of the 640-channel level (to_q -> 77-key attention -> to_out + bias + fp16 residual) as ONE
// launch with Q, P and O in registers.  x = LayerNorm'd tokens (bf16), wq = packed to_q [H * 64, C] pre-scaled by d^-1/2 log2 e,
// wo = packed to_out [C, H * 64], K / V^T = the per-prompt cache (AttnArgs layout), res / out = the fp16 trunk.
struct XBlockArgs {
    const bf16_t* x; const bf16_t* wq; const bf16_t* wo; const float* bo;
    const bf16_t* kc; const bf16_t* vt;
    const f16_t* res; f16_t* out;
    const float* wabs; const float* wsgn;            // [nsets, 96] font-size multipliers (attention_processor.py:386-401)
    int ldk, ldvt, ldres, ldo;
    int M, tokens, nk_valid, C, H;                   // M = streams * tokens; keys >= nk_valid of the 96 cached rows are masked
    int prompt[RT_MAXB], wset[RT_MAXB];              // per stream: prompt index into the cache, multiplier set (-1: plain softmax)
};
bool xblock_supported(int C, int H, int DP, int tokens);
void launch_xblock(const XBlockArgs& a, hipStream_t st);
void launch_gemm(const GemmArgs& a, hipStream_t st);
bool gemm_pair_output_ok(const GemmArgs& a);            // host-only: launch_gemm would run this problem on a kernel that can write GemmArgs.pair_lo
void gemm_split_plan(const GemmArgs& a, int* route, int* slices);      // host-only: 0 one launch / 1 K slices + reduction / 2 chunk-split patch convolution + reduction
size_t gemm_splitk_scratch_floats(const GemmArgs& a);   // fp32 partial sums launch_gemm needs for this problem (0: not split)
void gemm_force_config(int cfg);   // -1: shape-based choice; 0..8: force a gemm.hip tile configuration (also keeps gemm16.hip out)
void gemm_set_debug(int d);        // bit 0: eligible 3x3 convolutions through the implicit-GEMM kernels; bit 1: keep gemm16.hip out; bit 4: no fused cross-attention
bool gemm_lnfold_enabled();       // LayerNorm folded into its consumers (debug bit 22 clear, gemm16 on, no forced configuration)
bool gemm_xblock_enabled();       // the one-launch cross-attention block (xblock.hip) is switched on (debug bit 16 SET - opt-in; gemm16 on, no forced configuration)
bool gemm_xattn_enabled();         // the fused to_q + cross-attention kernel is allowed (debug bit 4 clear, gemm16 on, no forced configuration)
// gemm16.hip: 16x16x32-MFMA family (224-row tiles, intra-tile K split); dense problems with K % 64 == 0 only
#define RT_G16_NVAR 14
bool gemm16_supported(const GemmArgs& a);
int gemm16_pick(const GemmArgs& a, int weights_on_rows, int* wstat);      // variant id or -1; pure function of the shape
void launch_gemm16_variant(const GemmArgs& a, int variant, int wstat, hipStream_t st);
int gemm16_pair_variant(const GemmArgs& a, const GemmArgs& b);                   // host-only: id of the grouped instantiation (0..3), -1: none
bool launch_gemm16_pair(const GemmArgs& a, const GemmArgs& b, hipStream_t st);   // grouped launch (gemm16.hip); false: not launched, no grouped form
bool gemm_pair_is_grouped(const GemmArgs& a, const GemmArgs& b);                // host-only: would launch_gemm_pair take the grouped launch?
void launch_gemm_pair(const GemmArgs& a, const GemmArgs& b, hipStream_t st);     // gemm.hip: the grouped launch where it exists (debug bit 13 clear), else two launch_gemm calls

// ---------------------------------------------------------------- attention
struct AttnArgs {
    const bf16_t* Q;  int ldq;   // [*, ldq]  row (q_src[b]*N + n), head h at column h*DP; pre-scaled by d^-1/2 * log2(e)
    const bf16_t* K;  int ldk;   // [*, ldk]  row (k_src[b]*NK + key), head h at column h*DP
    const bf16_t* VT; int ldvt;  // [H*DP, ldvt] row (h*DP + d), column (v_src[b]*NK + key)
    bf16_t* O;        int ldo;   // [B*N, ldo] row (b*N + n), head h at column h*DP
    int q_src[RT_MAXB], k_src[RT_MAXB], v_src[RT_MAXB], wset[RT_MAXB];
    // filled by launch_attention (attn_kernel only): a launch runs B UNITS; unit b attends with (q_src[b], k_src[b]) and has ng[b] members
    // that share its probabilities - member g reads V^T of batch entry gvs[b][g] and writes the output rows of batch entry gob[b][g]
    // (one-stream launches: ng = 1, gvs = v_src[b], gob = b)
    unsigned char ng[RT_MAXB], gvs[RT_MAXB][4], gob[RT_MAXB][4];
    const float* wabs;           // cross: [nsets, NK] |font size| multiplier per key; null => plain softmax
    const float* wsgn;           // cross: [nsets, NK] sign multiplier per key
    int B, H, N;                 // queries per batch entry
    int NK;                      // keys per batch entry in K / V^T (multiple of the key tile)
    int nk_valid;                // cross: keys >= nk_valid are masked out
    int DP;                      // padded head dim (multiple of 32)
    int cross;
    int nqb;                     // set by launch_attention: 128-query blocks per (batch entry, head)
    // self-attention only: when non-null, the kernel also leaves the softmax statistics of batch entry stats_b - (running reference m,
    // 1 / (H * sum)) per (head, query), [H][N][2] fp32 - which is exactly what attn_store_apply_kernel needs to rebuild the
    // head-averaged probabilities of that stream: the token-map capture of the plain pass then skips its own statistics pass
    float* stats; int stats_b;
};
void launch_attention(const AttnArgs& a, hipStream_t st);
// xblock.hip: the 77-key cross-attention as its own kernel (d = 64, cached K / V^T): 64 queries x 2 heads per workgroup, K / V^T in LDS
bool cross77_supported(int H, int DP, int tokens, int NK, int nk_valid);
void launch_cross77(const AttnArgs& a, hipStream_t st);
bool gemm_cross77_enabled();      // debug bit 19 clear
int attention_units_plan_host(const int* q_src, const int* k_src, int B, int N, int DP, int mode, int* launch_of, int* unit_of, int* members_of);   // host-only: the partition launch_attention would take
void gemm16_set_tall(int on);        // (A/B) rt_op_gemm_debug bit 27
void gemm16_set_deep(int on);        // (A/B) rt_op_gemm_debug bit 12 clears the 5-slot ring of the 64 x 160 tiles
void attention_set_units(int mode);  // shared-probability units of the self-attention launches (attention.hip, launch_attention_units; rt_op_gemm_debug bits 24 - 26)
void attention_set_prio(int on);   // s_setprio around the MFMA phases of attn_kernel (default on; rt_op_gemm_debug bit 14 clears it)

// head-averaged probabilities of one stream, accumulated over calls (token-map producer)
struct AttnStoreArgs {
    const bf16_t* Q; int ldq; long q_row0;     // query rows q_row0 + [0, N), head h at column h*DP (pre-scaled like AttnArgs.Q)
    const bf16_t* K; int ldk; long k_row0;     // key rows k_row0 + [0, NKrows)
    float* out;                                // [N, NK] fp32 accumulator
    int H, N, NK, NKpad, NKrows, DP;
    int overwrite;                             // 1: out = avg(P); 0: out += avg(P)
    float* stats;                              // optional scratch [H][N][2] fp32 (softmax max, 1 / (H sum) per (head, query)): enables the
                                               // statistics + key-split apply pair for maps of >= 256 keys; null: one-pass kernel
    int stats_ready;                           // 1: `stats` already holds the statistics of this stream (written by the attention
                                               // launch itself, AttnArgs.stats): only the apply kernel runs
};
void launch_attn_store(const AttnStoreArgs& a, hipStream_t st);
bool attn_store_takes_stats(int N, int NK, int DP);   // the statistics + apply pair would run for this map: the self-attention launch may leave the statistics

// ---------------------------------------------------------------- norms / elementwise
struct GroupNormArgs {
    const void* x1; const void* x2;   // x2 may be null; virtual concat along channels [C1 | C2]
    int in_bf16;                       // input element type: 0 fp32, 1 bf16 (x2 must be null), 2 fp16 (UNet trunk)
    int C1, C2, G, B, HW;
    const float* gamma; const float* beta; float eps;
    int silu;
    bf16_t* out;                       // [B, HW, C1+C2] normalised (+SiLU)
    bf16_t* raw_out;                   // optional: un-normalised bf16 copy of the (concatenated) input
    bf16_t* out_lo;                    // optional low parts of out / raw_out (precise VAE mode: operand = hi + lo)
    bf16_t* raw_lo;
    float* partial;                    // workspace [B, nchunk, G, 2]
    int nchunk, rows_per_chunk;        // from groupnorm_nchunk / groupnorm_rows_per_chunk
    int fuse_finalize;                 // 1 (UNet forward, nchunk <= 128): every block of the apply kernel reduces the per-chunk partials
                                       // itself (one launch fewer; `partial` keeps the RAW sums).  0: gn_finalize_kernel leaves
                                       // (mean, rstd) in chunk 0's slot, which the VAE backward reads.
    double fin_n;                      // set by launch_groupnorm: elements per group
    int apply_rows;                    // set by launch_groupnorm: rows per workgroup of the apply kernel
    float* stats; int stats_ld;        // set by launch_groupnorm: where the apply kernel finds (mean, rstd) per (batch entry, group) in the non-fused form
};
int groupnorm_rows_per_chunk(int HW);
int groupnorm_bwd_rows_per_chunk(int HW);              // chunking of launch_groupnorm_bwd (its own partial-sum layout; one image: the forward's (mean, rstd) sit at offset 0 either way)
void launch_groupnorm(const GroupNormArgs& a, hipStream_t st);
void groupnorm_set_chunk_div(int d);                    // A/B
void groupnorm_set_fused(int on);                       // A/B: 0 = always the two-launch form (rt_op_gemm_debug bit 23)
int groupnorm_nchunk(int HW);

void launch_layernorm(const void* x, int x_f16 /* 0: fp32 rows, 1: fp16 rows (UNet trunk) */, const float* gamma, const float* beta,
                      bf16_t* out, int rows, int C, float eps, hipStream_t st);
void launch_cast_f16_bf16(const f16_t* x, bf16_t* out, size_t n, hipStream_t st);
void launch_cast_f32_bf16(const float* x, bf16_t* out, size_t n, hipStream_t st, bf16_t* out_lo = nullptr);
// out[b][n] (+)= sum_k act(a[b][k]) * W[n][k] + bias[n];  B <= 8
void launch_small_linear(const float* a, int lda, const bf16_t* W, int ldw, const float* bias, float* out, int ldo,
                         int B, int N, int K, int silu_in, int accumulate, hipStream_t st);
// one entry per ResnetBlock2D: out[B*first + b*N + n] = bias[n] + sum_k silu(emb[b][k]) * W[n][k]; `first` = prefix sum of N
struct TembEntry { const bf16_t* W; const float* bias; int N; int first; };       // output block of an entry: [B, N] at B * first (batch-independent table)
void launch_temb_all(const float* emb, int lde, float* silu_scratch, const TembEntry* tab, int ntab, int total, int B, int K, float* out,
                     hipStream_t st);
void launch_timestep_embed(const float* t, int n, int dim, float* out, int ldo, hipStream_t st);
void launch_timestep_embed_scalar(float t, int dim, float* out, hipStream_t st);
// CLIP text encoder pieces (text.hip)
void launch_embed(const int* ids, const float* tok, const float* pos, float* out, int rows, int N, int C, int vocab, hipStream_t st);
void launch_activation(const bf16_t* x, bf16_t* out, size_t n, int kind, hipStream_t st);      // 0 quick_gelu, 1 gelu(erf)
void launch_causal_attention(const bf16_t* q, const bf16_t* k, const bf16_t* v, int ld, bf16_t* out, int ldo, int B, int H, int N, int d,
                             float scale, hipStream_t st);

// weight packing (device side): strided gather fp32/fp16/bf16 -> bf16 (or f32) with scale.
//   dst (r, c) <- scale * src[ srow(r)*s_r + (c / c_inner)*s_co + (c % c_inner)*s_ci ]   (0 where invalid)
//   row_map 0: srow = r
//           1: GEGLU interleave: per 64-row block [32 value rows | 32 gate rows]; gate rows start at rows/2
//           2: head padding: srow = (r / rm_a) * rm_b + r % rm_a, valid iff r % rm_a < rm_b   (rm_a = DP, rm_b = d)
//   column validity: (c % c_inner) < ci_valid
enum { PACK_ROWS_ID = 0, PACK_ROWS_GEGLU = 1, PACK_ROWS_HEADPAD = 2 };
struct PackArgs {
    const void* src; int src_dtype;   // 0 f32, 1 f16, 2 bf16
    void* dst; int dst_f32;
    int rows, cols, ld_dst;
    int row_map, rm_a, rm_b;
    int c_inner, ci_valid;
    long s_r, s_co, s_ci;
    float scale;
    long s_base;                      // constant source offset (flipped conv taps for the backward-data weights)
    int lo_part;                      // 1: store the low part bf16(v - bf16(v)) instead of bf16(v) (precise VAE mode)
};
void launch_pack(const PackArgs& a, hipStream_t st);

// NCHW fp32 latents -> NHWC bf16 [B, HW, 8] (channels 4..7 zero), scaled per batch entry
struct PrepArgs {
    const float* src[RT_MAXB];   // each [4, HW]
    float scale[RT_MAXB];
    bf16_t* dst; int B, HW;
};
void launch_prep_latents(const PrepArgs& a, hipStream_t st);

// ---------------------------------------------------------------- VAE decoder forward / backward-data helpers (a13)
// GroupNorm(+SiLU) backward wrt the input: dx = rstd * (dxh - mean_g(dxh) - xh * mean_g(dxh * xh)),
// dxh = dA * silu'(y) * gamma, y = xh*gamma + beta, xh = (x - mean) * rstd; statistics are recomputed from the
// forward pass' partial sums.  `add` (fp32, may be null) is accumulated into the result (residual gradients).
struct GroupNormBwdArgs {
    const void* x; int x_bf16;         // forward input [B, HW, C]
    const bf16_t* dA;                  // gradient wrt the (activated) output [B, HW, C]
    const bf16_t* dA_lo;               // optional low part of dA (precise mode)
    bf16_t* out_bf16_lo;               // optional low part of out_bf16
    const float* fwd_partial;          // forward statistics partials [B, nchunk, G, 2]
    float* bwd_partial;                // workspace [B, nchunk, G, 2]
    const float* gamma; const float* beta; float eps;
    int silu, C, G, B, HW, nchunk, rows_per_chunk;
    const float* add;                  // optional fp32 [B, HW, C]
    float* out;                        // fp32 [B, HW, C] (may be null)
    bf16_t* out_bf16;                  // optional bf16 copy (operand of the next backward conv)
};
void launch_groupnorm_bwd(const GroupNormBwdArgs& a, hipStream_t st);
void launch_gn_finalize(float* partial, int B, int nchunk, int G, double n, float eps, int mode, hipStream_t st);
void launch_transpose_bf16(const bf16_t* in, bf16_t* out, int R, int C, hipStream_t st);        // [R,C] -> [C,R]
void launch_softmax_rows(const float* s, bf16_t* p, int rows, int cols, float scale, hipStream_t st, bf16_t* p_lo = nullptr);
// dS = scale * P o (dP - rowsum(dP o P))   (bf16 out)
void launch_softmax_bwd(const bf16_t* p, const float* dp, bf16_t* ds, int rows, int cols, float scale, hipStream_t st, const bf16_t* p_lo = nullptr,
                        bf16_t* ds_lo = nullptr);
void launch_sumpool2x2(const float* in, float* out, int B, int H, int W, int C, hipStream_t st);   // [B,2H,2W,C] -> [B,H,W,C]
void launch_add_f32(const float* a, const float* b, float* out, size_t n, hipStream_t st);
// per pixel: out[pix][0..7] = bf16(W[4x4] * (in[c][pix] * scale) + b), channels 4..7 zero  (post_quant_conv 1x1)
void launch_pq_conv_fwd(const float* lat_nchw, const float* eps_nchw, float c_lat, float c_eps, const float* W, const float* b,
                        bf16_t* out, int HW, hipStream_t st, bf16_t* out_lo = nullptr);
// dlat[c][pix] = sum_o W[o][c] * dz[pix][o] * gscale ;  lat -= dlat * weight * mask_all
void launch_pq_conv_bwd_update(const float* dz, int ldz, const float* W, float gscale, float weight, const float* mask_all,
                               float* lat_nchw, float* grad_out /* optional [4,HW] */, int HW, hipStream_t st);
struct ColorLossArgs {
    const float* img; int ldi;         // decoder output [HWi, ldi] fp32 (3 valid channels), range [-1, 1]
    const float* masks;                // [n, HWi]
    const float* target;               // [n, 3] device
    int n, HWi;
    float* partial;                    // [nblk, n, 4] workspace
    int nblk;
    bf16_t* dimg;                      // [HWi, 8] bf16 gradient wrt the decoder output
    bf16_t* dimg_lo;                   // optional low part (precise mode)
    float* loss_out;                   // [1] optional
};
void launch_color_loss_grad(const ColorLossArgs& a, hipStream_t st);
