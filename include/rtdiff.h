/* librtdiff — C ABI of the MI355X (gfx950) region-diffusion sampling engine.
 *
 * Drop-in boundary for the denoising hot path of songweige/rich-text-to-image.  The reference has no
 * native code and no FFI: its "operator interface" for this path is the Python call surface cited
 * beside each entry point below (file:line under /root/reference).  The Python facade in
 * rich-text-to-image_amd/ binds these symbols with ctypes and mirrors the reference classes.
 *
 * Conventions
 *   - every function returns int: 0 = ok, negative = RT_E_*; rt_last_error() gives the message
 *   - all tensor pointers are DEVICE pointers unless the parameter is documented as host
 *   - the caller owns every tensor it passes and keeps it alive until the stream is synchronised;
 *     the engine owns its weight arena, K/V caches, workspace and sampler state
 *   - one engine = one device = one HIP stream; calls on one engine are not re-entrant
 *   - no allocation happens inside forward / step calls
 */
#ifndef RTDIFF_H
#define RTDIFF_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_E_INVALID (-1)
#define RT_E_HIP (-2)
#define RT_E_STATE (-3)
#define RT_E_MISSING_WEIGHT (-4)
#define RT_E_UNSUPPORTED (-5)

#define RT_MAX_LEVELS 4
#define RT_MAX_STREAMS 16

enum { RT_DTYPE_F32 = 0, RT_DTYPE_F16 = 1, RT_DTYPE_BF16 = 2 };
enum { RT_SCHED_EULER = 0, RT_SCHED_PNDM = 1 };

/* Architecture of the UNet: the constructor arguments of UNet2DConditionModel that SD-v1.5 / SDXL use
 * (models/unet_2d_condition.py:160-215). */
typedef struct rt_config {
    int n_levels;
    int block_out_channels[RT_MAX_LEVELS];
    int down_has_attn[RT_MAX_LEVELS];   /* CrossAttnDownBlock2D (1) vs DownBlock2D (0) */
    int up_has_attn[RT_MAX_LEVELS];     /* CrossAttnUpBlock2D (1) vs UpBlock2D (0), in up_blocks order */
    int layers_per_block[RT_MAX_LEVELS];
    int transformer_layers[RT_MAX_LEVELS];
    int heads[RT_MAX_LEVELS];           /* `attention_head_dim` of the config = number of heads */
    int cross_attention_dim;
    int norm_groups;
    float norm_eps;
    int use_linear_projection;
    int addition_text_time;             /* addition_embed_type == "text_time" */
    int addition_time_embed_dim;
    int projection_class_embeddings_input_dim;
    int in_channels, out_channels;      /* 4, 4 */
    int latent_h, latent_w;             /* largest latent the workspace is sized for */
    int max_streams;                    /* UNet forwards batched per launch (<= RT_MAX_STREAMS) */
    int max_prompts;                    /* prompts resident in the cross-attention K/V cache */
} rt_config;

typedef struct rt_engine rt_engine;

/* lifecycle -------------------------------------------------------------------------------------- */
int rt_create(const rt_config* cfg, int device, rt_engine** out);   /* RegionDiffusion.__init__ rd.py:16-47, xl.py:56-134 */
int rt_destroy(rt_engine* e);
const char* rt_last_error(rt_engine* e);                             /* e may be NULL: last error of failed rt_create */
int rt_set_stream(rt_engine* e, void* hip_stream);   /* NULL: back to the stream rt_create made.  A step call is hipGraph-capturable on it (tests/test_engine_gpu.py) */
int rt_synchronize(rt_engine* e);

/* weights: names are the reference UNet's state_dict keys (unet.load_state_dict, rd.py:32, xl.py:115) --- */
int rt_weight_count(rt_engine* e);
int rt_weight_info(rt_engine* e, int idx, char* name, int name_cap, int64_t* shape4, int* ndim);
int rt_bind_weight(rt_engine* e, const char* name, const void* dev_ptr, int dtype, const int64_t* shape, int ndim);
int rt_weights_missing(rt_engine* e, char* buf, int cap);            /* returns count; names ';'-joined into buf */
int rt_arena_info(rt_engine* e, void** dev_ptr, uint64_t* bytes);    /* packed arena, for one RCCL broadcast */
int rt_arena_mark_bound(rt_engine* e);                               /* after the arena was filled by a broadcast */

/* per image -------------------------------------------------------------------------------------- */
/* text conditioning: prompt_embeds [P,77,D] f32, pooled [P,Dp] f32 (SDXL) or NULL, time_ids[6] (host) or NULL.
 * rd.py:49-84 / xl.py:256-442,741-762; index 0 = negative prompt, 1..P-2 = region prompts, P-1 = base prompt */
int rt_set_prompts(rt_engine* e, const float* prompt_embeds, const float* pooled, const float* time_ids_host,
                   int n_prompts, int pooled_dim);
/* region masks (model.masks, rd.py:97,119-128 / xl.py:775,810-821): [R,4,h,w] f32 */
int rt_set_masks(rt_engine* e, const float* masks, int n_regions, int h, int w);
/* font-size control (text_format_dict['word_pos'|'font_size'], richtext_utils.py:188-209): HOST arrays; n = 0 disables */
int rt_set_fontsize(rt_engine* e, const int64_t* word_pos_host, const float* font_size_host, int n);
/* scheduler tables (HOST arrays): Euler: sigmas[n+1], timesteps[n]; PNDM: alphas_cumprod[1000], timesteps[n_iter] */
int rt_set_schedule(rt_engine* e, int kind, const float* timesteps_host, int n_timesteps, const float* table_host,
                    int n_table, int num_inference_steps);
/* sampler state: latents [1,4,h,w] f32 (copied in); the reference stream starts as a clone (rd.py:93, xl.py:774) */
int rt_set_latents(rt_engine* e, const float* latents, int h, int w);
int rt_get_latents(rt_engine* e, float* latents_out, float* latents_ref_out /* may be NULL */);

/* hot path ---------------------------------------------------------------------------------------- */
/* one iteration of the rich-text loop (rd.py:99-173 / xl.py:779-872 without colour guidance):
 * all R+1 / R+3 UNet forwards batched, mask combine, CFG, scheduler step, background blend */
/* inject_selfattn / inject_background are DOUBLES: the reference evaluates `t > (1-inject_selfattn)*1000` and
 * `int(inject_background*len(timesteps))` on Python floats (rd.py:103-104, xl.py:783-784); float32 would move the blend step
 * by one for e.g. 0.7 x 50 steps */
int rt_region_step(rt_engine* e, int step_index, float guidance_scale, double inject_selfattn, double inject_background,
                   int xl_semantics, int flags /* bit 0: elide reference forwards that cannot influence the output;
                                                  bit 1: defer the background blend to rt_background_blend() */);
/* the deferred blend of the last step (colour guidance sits between the scheduler step and the blend: rd.py:151-173) */
int rt_background_blend(rt_engine* e);
/* Intra-image split of one rich-text step over `nparts` GPUs (the F forwards at region_diffusion_sdxl.py:787-821 are
 * independent except that injected region forwards consume the text_ref forward's self-attention Q / K and ResNet feature).  Every
 * rank holds the same engine state and calls, per step:
 *   rt_region_step_part(..., part, nparts, &first, &count)   the UNet forwards of ITS contiguous range [first, first + count) of the
 *                                                            step's stream list [uncond, base, uncond_ref, text_ref, regions...];
 *                                                            text_ref and the region streams always share the last range, so nothing
 *                                                            crosses GPUs inside the forward
 *   (exchange the ranges of the eps buffer, rt_eps_info: stream s at byte offset s * bytes_per_stream - one collective per rank)
 *   rt_region_step_finish(...)                               mask combine + CFG + scheduler step + blend on all streams (every rank)
 * Same arguments / flags as rt_region_step; the result is bit-identical with rt_region_step on one GPU. */
int rt_region_step_part(rt_engine* e, int step_index, float guidance_scale, double inject_selfattn, double inject_background, int xl,
                        int flags, int part, int nparts, int* first_stream, int* n_streams,
                        int* plan_info /* optional [3]: streams of the step, its text_ref stream (-1: none), injection on: the arguments
                                          of rt_op_split_range, from which a rank derives the OTHER ranks' ranges without a collective */);
int rt_region_step_finish(rt_engine* e, int step_index, float guidance_scale, double inject_selfattn, double inject_background, int xl,
                          int flags);
int rt_eps_info(rt_engine* e, void** dev_ptr, unsigned long long* bytes_per_stream, int* max_streams);
/* host-only: the stream range of `part` for a step of n_streams streams whose text_ref stream is text_ref_stream (-1: none) */
int rt_op_split_range(int n_streams, int text_ref_stream, int inject, int part, int nparts, int* first_stream, int* n_streams_out);
/* one iteration of the plain-text loop (rd.py:200-214 / xl.py:880-905): batch-2 forward, CFG, step */
int rt_plain_step(rt_engine* e, int step_index, float guidance_scale);
/* The same step shared by the ranks of a process group (round 6; sample.py --gpus N --split_image): part 0 runs the unconditional forward,
 * part 1 the text forward - and is the rank whose engine records the token maps -, further parts none ([first, first + count) of the
 * stream list [uncond, text]); the ranks exchange their slots of the eps buffer (rt_eps_info) and every rank calls rt_plain_step_finish
 * (CFG + scheduler step).  Bit-identical with rt_plain_step on one GPU. */
int rt_plain_step_part(rt_engine* e, int step_index, int part, int nparts, int* first_stream, int* n_streams);
int rt_plain_step_finish(rt_engine* e, int step_index, float guidance_scale);

/* token-map attention store (hooks rd.py:397-443, xl.py:959-1016): head-averaged softmax(QK^T) of the
 * CONDITIONAL stream of rt_plain_step, recorded for the named attention modules (reference module names such as
 * "down_blocks.1.attentions.0.transformer_blocks.0.attn1").  mode 1: accumulate over calls after the module's 10th
 * call (n_maps[name] > 10); mode 2: overwrite after the 10th call (the SD-v1.5 self-attention quirk, rd.py:423);
 * mode 0: off.  Self-attention maps are limited to 32x32 tokens (the only ones get_token_maps consumes). */
int rt_attn_store_enable(rt_engine* e, const char* module_name, int mode);
int rt_attn_store_reset(rt_engine* e);
int rt_attn_store_read(rt_engine* e, const char* module_name, float* dst_dev /* [rows, cols] f32 or NULL */, int* n_calls,
                       int* rows, int* cols);
int rt_attn_module_count(rt_engine* e);
int rt_attn_module_info(rt_engine* e, int idx, char* name, int name_cap, int* max_tokens, int* heads);

/* per-launch HIP-event timing of the MFMA kernels on the engine's stream (bench.py roofline leg).
 * Algorithmic FLOPs are counted per launch (2*M*N*K for GEMM/conv, 4*B*H*N*NK*d for attention; padded
 * head dims / keys are not counted). */
enum { RT_PROF_GEMM_DENSE = 0, RT_PROF_GEMM_CONV = 1, RT_PROF_ATTN_SELF = 2, RT_PROF_ATTN_CROSS = 3,
       RT_PROF_ATTN_STORE = 4 /* head-averaged map accumulation of the plain pass (attn_store_kernel, HBM-side accumulators) */,
       RT_PROF_XATTN_FUSED = 5 /* to_q + 77-key cross-attention as one launch (gemm16 EPI_XATTN): FLOPs = 2*M*HD*C + 4*B*H*N*77*d */,
       RT_PROF_XBLOCK = 6 /* the whole cross-attention block as one launch (xblock.hip): FLOPs = 4*M*HD*C + 4*B*H*N*77*d */ };
int rt_profile_enable(rt_engine* e, int on);     /* on: start recording (clears old records); off: stop */
int rt_profile_read(rt_engine* e, int kernel_class, int* count, double* total_ms, double* total_flops);
/* as rt_profile_read, plus the ALGORITHMIC HBM bytes of the launches (attention store: the fp32 accumulator read + written once and the
 * Q / K rows of the recorded stream read once; 0 for the classes that are priced in FLOPs) */
int rt_profile_read2(rt_engine* e, int kernel_class, int* count, double* total_ms, double* total_flops, double* total_bytes);

/* operator level (parity tests, AttnProcessor / unet(...) seams; unet_2d_condition.py:703-717) ------------ */
/* x [B,4,h,w] f32 NCHW; per-stream: input scale, prompt index, font-size flag, self-attention Q/K source
 * stream (== own index for normal attention), resnet-feature source stream or -1; out [B,4,h,w] f32 NCHW */
int rt_unet_forward(rt_engine* e, const float* x, int B, int h, int w, float timestep, const float* in_scale_host,
                    const int* prompt_idx_host, const int* fontsize_host, const int* qk_src_host,
                    const int* res_src_host, float* out);

/* epi: 0 bf16 out | 1 fp32 out (+ fp32 residual) | 2 bf16 out + per-batch-entry time embedding | 3 GEGLU (bf16 out, N/2 columns)
 *      | 4 fp16 out (+ fp16 residual): the UNet's residual trunk (fp32 arithmetic in the epilogue, fp16 in HBM like the reference's
 *        fp16 pipelines).  `res` is fp32 for epi 1 and fp16 for epi 4. */
int rt_op_gemm(const void* A, const void* W, const float* bias, void* out, const void* res, const float* temb,
               int mode, int epi, int M, int N, int K, int lda, int ldw, int ldo, int ldres, int temb_ld,
               int rows_per_batch, int Hin, int Win, int Cin, int Hout, int Wout, void* stream);
/* wset_host (cross only): per batch entry the row of wabs / wsgn [nsets, NK] to multiply the exponentials / probabilities with
 * (attention_processor.py:386-401), or -1 for plain softmax over the keys < nk_valid (no table reads; NULL = set 0 everywhere). */
int rt_op_attention(const void* Q, int ldq, const void* K, int ldk, const void* VT, int ldvt, void* O, int ldo,
                    const int* q_src_host, const int* k_src_host, const int* v_src_host, const int* wset_host,
                    const float* wabs, const float* wsgn, int B, int H, int N, int NK, int nk_valid, int DP, int cross,
                    void* stream);
/* Host-only (no GPU): the partition of a self-attention launch into shared-probability units (round 6; csrc/attention.hip).  Streams that
 * attend with the same (q_src, k_src) - text_ref and the region streams that take its probabilities, attention_processor.py:522-524 - are dealt
 * G at a time into units that compute softmax(QK^T) once; mode 0 = the shape rule (tokens >= 2048: units of four in their own launch, else
 * pairs beside the one-stream units), 1 = never, 2 - 5 as rt_op_gemm_debug bits 24 - 26.  Per batch entry: launch_of (0 = the G-member
 * kernel's launch, 1 = the separate one-stream launch; 1 for every entry when nothing is shared), unit_of, members_of.  Returns G (0: no sharing),
 * -1 on bad arguments. */
int rt_op_attention_units_plan(const int* q_src, const int* k_src, int B, int tokens, int DP, int mode, int* launch_of, int* unit_of, int* members_of);
/* Host-only query of the shape rule of csrc/gemm16.hip (no device needed; tests/test_gemm16_pick.py): the tile variant a problem of `streams`
 * streams x rows_per_stream rows takes (ids below; -1: not in the family / stays on gemm.hip or the patch convolution; -2: conv3x3 query
 * with a non-square map).  conv3x3 = 1: stride-1 3x3 convolution with Cin = K_or_Cin input channels on a square map of rows_per_stream
 * pixels.  *w_stationary = 1 when the launch uses the W-stationary tile -> XCD order.  The summation CLASS of the answer ({2,3,4,5,8,10,11} /
 * {0,1,9} / {6,7,12} / -1) never depends on `streams`; inside a class the tile shape follows the batch. */
int rt_op_gemm16_pick(int conv3x3, int epi, int streams, int rows_per_stream, int N, int K_or_Cin, int weights_on_rows, int* w_stationary);
/* Host-only query of the split rule of csrc/gemm.hip for problems that cannot fill the chip (no device needed; tests/test_split_plan.py):
 * *route 0: one launch | 1: K slices of the 128x128 (implicit) GEMM + reduction launch | 2 (round 6): the 16x16-patch convolution kernel
 * split over its input-channel chunks + reduction launch; *slices: how many.  conv3x3: 0 dense (K_or_Cin = K), 1 stride-1 3x3 convolution,
 * 3 the nearest-2x up-sample folded in (rows_per_stream = OUTPUT pixels, a square map).  Route and slice count are functions of ONE
 * stream's shape (a nominal batch of four), never of `streams`: a stream's k order does not depend on the batch. */
int rt_op_split_plan(int conv3x3, int epi, int streams, int rows_per_stream, int N, int K_or_Cin, int* route, int* slices);
/* One tile variant of the 16x16x32-MFMA GEMM family (csrc/gemm16.hip; tests / micro-benchmarks - rt_op_gemm picks by shape):
 * 0: 224x160 K-split  1: 128x160 K-split  2: 224x256  3: 256x256  4: 224x320  5: 256x320  6: 160x224 K-split (V^T)  7: 160x128 K-split
 * 8: 128x256  9: 64x160 K-split  10: 128x320  11: 64x320  12: 160x64 K-split (V^T) - 9..12: the small batches of the plain pass / SD-v1.5;
 * -1: the variant the shape rule picks.  Variants {2,3,4,5,8,10,11} (class A) give the same bits as each other and as every
 * tile configuration of csrc/gemm.hip; {0,1,9} and {6,7,12} (class B: two K halves summed at the end) are bit-identical sets.
 * Dense only, K % 128 == 0, K >= 256; epi as rt_op_gemm (0, 1, 3, 4). */
int rt_op_gemm16_variant(const void* A, const void* W, const float* bias, void* out, const void* res, int epi, int M, int N, int K, int lda,
                         int ldw, int ldo, int ldres, int weights_on_rows, int variant, int wstat, void* stream);
/* attn1's two projections of one LayerNorm output X [M, K] bf16 (models/attention_processor.py:495-506: to_q / to_k / to_v read the same
 * hidden states) exactly as the engine launches them: qk[Mqk, Nqk] = X[:Mqk] Wqk^T + bqk (stacked, head-padded to_q | to_k; Mqk <= M: the
 * injected region streams of a rich-text step need no Q / K of their own) and vt[Nv, M] = Wv X^T.  Where the 16x16x32 family has the pair
 * of tiles (SDXL: both attention levels with 7, 4 and - the plain pass - 2 streams) they go out as ONE grouped launch (csrc/gemm16.hip, gemm16_dual_kernel: the
 * unchanged tile bodies, bit-identical with two launches; rt_op_gemm_debug bit 13 forces two launches); *grouped (may be NULL) says which. */
int rt_op_gemm_pair_pick(int streams_qk, int streams, int rows_per_stream, int Nqk, int Nv, int K);   /* host-only: 0..3 = grouped (which tile pair), -1 = two launches */
int rt_op_gemm_qk_vt(const void* X, int ldx, int K, int rows_per_stream, const void* Wqk, const float* bqk, int Mqk, int Nqk, void* qk, int ldqk,
                     const void* Wv, int Nv, int M, void* vt, int ldvt, int* grouped, void* stream);
/* The SDXL cross-attention block the north star names, as one call (replaces attn2 of BasicTransformerBlock, models/attention.py:169-189,
 * processor arithmetic models/attention_processor.py:476-545, font-size softmax :386-401):
 *   trunk_out[B*N, C] (fp16) = trunk_in + to_out(softmax_fs(to_q(x) K[prompt]^T) V[prompt]) + bo
 * x bf16 [B*N, C] (LayerNorm output); wq bf16 [H*DP, C] head-padded and pre-scaled by d^-1/2 log2 e; wo bf16 [C, H*DP]; bo fp32 [C] or NULL;
 * kcache bf16 [P*96, H*DP], vtcache bf16 [H*DP, ldvt] (77 keys padded to 96 per prompt); prompt_host / wset_host: per batch entry the
 * prompt index and the multiplier set in wabs / wsgn [nsets, 96] (the engine's tables: 0 plain softmax, 1 font-size), or -1 = plain softmax
 * over the 77 valid keys without reading the tables (what the engine passes for every stream without a font-size entry: identical bits,
 * fewer instructions); q_scratch, o_scratch bf16 [B*N, H*DP].
 * Launches: where the tiling allows it (DP = 64, H*64 % 320 == 0, N % 128 == 0, C % 64 == 0: the 1280-channel level of SDXL, 60 of its 70
 * blocks) to_q and the attention are ONE kernel (csrc/gemm16.hip, EPI_XATTN: a 128 x 320 tile = 128 queries x 5 heads stays in LDS,
 * q_scratch is not written) followed by the to_out GEMM; otherwise to_q GEMM -> attention -> to_out GEMM. */
int rt_op_cross_attn_block(const void* x, const void* wq, const void* wo, const float* bo, const void* kcache, const void* vtcache, int ldvt,
                           const int* prompt_host, const int* wset_host, const float* wabs, const float* wsgn, const void* trunk_in_f16,
                           void* trunk_out_f16, void* q_scratch, void* o_scratch, int B, int N, int C, int H, int DP, void* stream);
/* in_type: 0 fp32, 1 bf16 (x2 must be NULL), 2 fp16 - the element type of x1 / x2 (the UNet trunk is fp16) */
int rt_op_groupnorm(const void* x1, const void* x2, int in_type, int C1, int C2, int G, int B, int HW,
                    const float* gamma, const float* beta, float eps, int silu, void* out_bf16, void* raw_out_bf16,
                    void* stream);
int rt_op_layernorm(const float* x, const float* gamma, const float* beta, void* out_bf16, int rows, int C, float eps,
                    void* stream);
int rt_op_layernorm_f16(const void* x_f16, const float* gamma, const float* beta, void* out_bf16, int rows, int C, float eps, void* stream);
/* LayerNorm folded into the projection that consumes it (round 6; csrc/gemm16.hip "LNF").  BasicTransformerBlock applies norm1 / norm2 /
 * norm3 in front of attn1 / attn2 / ff (/root/reference/models/attention.py:150,168,181); the engine computes
 *     LN(x) W^T + b = rstd (xb W'^T - mu s) + c,   W' = bf16(gamma W),  s = row sums of W',  c = b + W beta
 * on xb = the UN-normalised trunk as bf16, which the trunk's producer leaves next to the fp16 trunk together with ONE (sum, sum of squares)
 * of xb per token and column tile of its grid (`tile_cols` = 160 or 320 columns), from which (mu, rstd) of a token follow; eps = 1e-5.
 * `partials`: pair-major [C / tile_cols / 2][tokens] float4 = (sum, sumsq) of two neighbouring tiles.
 * rt_op_ln_gemm: W bf16 packed [N, C] (GEGLU: rows interleaved per 64-block [32 value | 32 gate], as rt_op_gemm epi 3);
 *   epi 0: out bf16 [tokens, N] (weights_on_rows = 1: out = [N, tokens], the V^T form) | epi 3: GEGLU, out bf16 [tokens, N / 2];
 *   (xb_bf16, partials) as a producer left them, or both NULL: made here from the fp16 trunk x_f16 [tokens, C] by the stand-alone kernel.
 *   C = 640 or 1280.  RT_E_UNSUPPORTED (-5) when the shape has no folded form (the engine then keeps the LayerNorm launch).
 *   rows_per_stream as rt_op_gemm16_pick.
 * rt_op_gemm_emit_partials: the fp16-trunk GEMM (rt_op_gemm epi 4, optional fp16 residual) that ALSO leaves xb [M, N] bf16 and the
 *   partials of its output rows (N = 640 or 1280; *tile_cols = the width its tile class uses); RT_E_UNSUPPORTED when the shape's tile
 *   variant has no such epilogue. */
int rt_op_ln_gemm(const void* x_f16, const float* gamma, const float* beta, const void* W_bf16, const float* bias, void* out, int tokens,
                  int N, int C, int epi, int weights_on_rows, int rows_per_stream, const void* xb_bf16, const float* partials, int tile_cols,
                  void* stream);
int rt_op_gemm_emit_partials(const void* A, const void* W, const float* bias, void* out_f16, const void* res_f16, int M, int N, int K,
                             int rows_per_stream, void* xb_bf16, float* partials, int* tile_cols, void* stream);
int rt_op_ln_partials(const void* x_f16, void* xb_bf16, float* partials, int rows, int C, int tile_cols, void* stream);      /* the stand-alone producer of (xb, partials) */
int rt_op_small_linear(const float* a, int lda, const void* W_bf16, int ldw, const float* bias, float* out, int ldo,
                       int B, int N, int K, int silu_in, int accumulate, void* stream);
int rt_op_timestep_embed(const float* t, int n, int dim, float* out, int ldo, void* stream);
int rt_op_cast_bf16(const float* x, void* out_bf16, long long n, void* stream);
/* ---- CLIP text encoder pieces (transformers' CLIPTextModel[WithProjection] as called at rd.py:53-66, xl.py:330-356); the linear layers
 * and LayerNorms are rt_op_gemm / rt_op_layernorm.  ids [rows] int32 (device), tok [vocab, C], pos [N, C] fp32 -> out [rows, C] fp32 */
int rt_op_embed(const int* ids, const float* tok, const float* pos, float* out, int rows, int N, int C, int vocab, void* stream);
/* kind 0: quick_gelu (CLIP ViT-L), 1: gelu/erf (OpenCLIP bigG); bf16 -> bf16, n elements */
int rt_op_activation(const void* x_bf16, void* out_bf16, long long n, int kind, void* stream);
/* causal self-attention over N <= 128 tokens: q, k, v bf16 [B*N, ld] with head h at column h*d; out bf16 [B*N, ldo] */
int rt_op_causal_attention(const void* q, const void* k, const void* v, int ld, void* out, int ldo, int B, int H, int N, int d, float scale,
                           void* stream);
/* Head-averaged attention probabilities of ONE batch entry (the `attention_probs_avg` the reference processor returns,
 * attention_processor.py:541-545, reshape_batch_dim_to_heads_and_average): out[N, NK] (=|+=) mean_h softmax(Q_h K_h^T).
 * Q rows q_row0+[0,N), K rows k_row0+[0,NKrows) in the rt_op_attention layouts; NK valid keys (any count: processed in chunks of
 * 1024), NKpad = key count padded to a multiple of 32. */
int rt_op_attention_probs_avg(const void* Q, int ldq, long long q_row0, const void* K, int ldk, long long k_row0, float* out,
                              int H, int N, int NK, int NKpad, int NKrows, int DP, int accumulate, void* stream);
const char* rt_op_last_error(void);
/* GEMM tile choice: -1 (default) = a pure function of the problem shape (csrc/gemm16.hip: 16x16x32-MFMA family, 224-row tiles;
 * csrc/gemm.hip: 32x32x16 family for everything else) - no timing, no per-process state; 0..8 = force one tile configuration of
 * gemm.hip (tests / micro-benchmarks; all of them give bit-identical results). */
int rt_op_probes_built(void);      /* 1: the library was built with `make PROBES=1` and contains the measured-and-rejected kernels (xblock_kernel,
                                    * EPI_XATTN, gemm16 variant 13) that rt_op_gemm_debug bits 16 / 4 and variant 13 select; 0: the shipped build */
int rt_op_gemm_force_config(int cfg);
/* A/B switches (benchmarks; the engine wrapper reads RTDIFF_DEBUG_FLAGS once at load): bit 0 patch-eligible 3x3 convs through the
 * implicit-GEMM kernels; bit 1 keep gemm16.hip out; bit 2 no split-K; bit 3 stride-1 3x3 convs stay on the patch kernel (not on the
 * gemm16 main loop); bit 4 cross-attention as to_q GEMM + attention launch instead of the fused kernel;
 * bit 5 token-map accumulation on the round-1 two-pass kernel (csrc/attn_store.hip); bit 6 large maps on the one-pass kernel instead of
 * the statistics + key-split apply pair; bit 7 the precise VAE's hi / lo contractions as three launches instead of one;
 * bit 13 attn1's Q|K and V^T projections as two launches instead of one grouped launch; bit 14 no raised wave priority (s_setprio) in the
 * MFMA phases of the attention kernel; bit 15 the precise VAE's DENSE hi / lo contractions as three launches */
int rt_op_gemm_debug(int flags);

/* ---- VAE decoder: colour guidance (rd.py:151-168, xl.py:849-867) and plain decode (rd.py:227-236) ----
 * AutoencoderKL.decoder + post_quant_conv (diffusers 0.18.2, third party: architecture restated in oracle/vae.py).
 * Weight names are the AutoencoderKL state_dict keys ("decoder.conv_in.weight", "post_quant_conv.bias", ...). */
typedef struct rt_vae_config {
    int n_blocks;                         /* len(block_out_channels) */
    int block_out_channels[RT_MAX_LEVELS];
    int layers_per_block;                 /* decoder uses layers_per_block + 1 resnets per up block */
    int norm_groups;
    float scaling_factor;                 /* 0.18215 (SD) / 0.13025 (SDXL) */
    int latent_h, latent_w;               /* largest latent the workspace is sized for */
    int precise;                          /* 1: fp32-class contractions (operands as bf16 hi + lo pairs, three MFMA passes): what the SDXL
                                           * pipeline of the reference asks for (xl.py:856 decodes in fp32); 0: one bf16 pass */
} rt_vae_config;
typedef struct rt_vae rt_vae;
int rt_vae_create(const rt_vae_config* cfg, int device, rt_vae** out);
int rt_vae_destroy(rt_vae* v);
const char* rt_vae_last_error(rt_vae* v);
int rt_vae_weight_count(rt_vae* v);
int rt_vae_weight_info(rt_vae* v, int idx, char* name, int name_cap, int64_t* shape4, int* ndim);
int rt_vae_bind_weight(rt_vae* v, const char* name, const void* dev_ptr, int dtype, const int64_t* shape, int ndim);
int rt_vae_synchronize(rt_vae* v);
/* packed decoder weights (forward + backward-data copies), for the start-up broadcast next to rt_arena_info */
int rt_vae_arena_info(rt_vae* v, void** dev_ptr, uint64_t* bytes);
int rt_vae_arena_mark_bound(rt_vae* v);
/* img_out [3, 8h, 8w] f32 in [-1,1] = decode(latents [4,h,w] / scaling_factor if divide_by_scaling) */
int rt_vae_decode(rt_vae* v, const float* latents, int h, int w, int divide_by_scaling, float* img_out);
/* one guidance update, in place on `latents` [4,h,w]:
 *   x0 = (latents - noise_pred*sqrt(1-alpha_t))/sqrt(alpha_t); img = clamp(decode(x0/scaling)/2+.5, 0, 1)
 *   L = sum_k 100 * mse(sum(img*m_k)/sum(m_k), target_k);  latents -= dL/dlatents * weight * mask_all
 * masks_img [n, 8h*8w] (channel 0 of text_format_dict['color_obj_atten']), target_rgb_host [n*3],
 * mask_all [4,h,w] (text_format_dict['color_obj_atten_all']); grad_out [4,h,w] / loss_out_host optional. */
int rt_vae_color_guidance(rt_vae* v, float* latents, const float* noise_pred, float alpha_t, int h, int w, const float* masks_img,
                          const float* target_rgb_host, int n_regions, float weight, const float* mask_all, float* grad_out,
                          float* loss_out_host);
/* engine sampler state pointers for the guidance step: latents [4,h,w] and the CFG-combined noise prediction of the
 * last rt_region_step (noise_pred in rd.py:131-132 / xl.py:824-825) */
int rt_get_state_ptrs(rt_engine* e, float** latents, float** noise_pred);

#ifdef __cplusplus
}
#endif
#endif
