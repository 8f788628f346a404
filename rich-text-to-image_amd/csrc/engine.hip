// Host-side engine: weight arena, execution plan, batched UNet forward and the rich-text step driver.
//
// One engine = one GPU = one HIP stream.  The reference (pure Python, one process, batch 1) runs the
// R+1 / R+3 UNet forwards of a rich-text step sequentially and rewires nn.Module hooks between them
// (models/region_diffusion.py:99-173, models/region_diffusion_sdxl.py:779-872).  Here all forwards of
// a step are ONE batched forward ("streams"); the hook families become per-stream mode words:
//   font-size hooks (rd.py:465-494)        -> fontsize[b]  : multiplier set used by cross-attention
//   selfattn capture + replacement hooks   -> qk_src[b]    : stream whose Q,K drive self-attention
//   resnet feature capture/injection       -> res_src[b]   : stream whose up_blocks.1.resnets.1 residual
//                                                            branch replaces this stream's (resnet.py:639-643)
// The text_ref stream is in the same batch, so its Q/K and resnet feature are available to the region
// streams at every layer without storing 5.9 GB of probabilities per step.
#include "common.h"
#include "../../include/rtdiff.h"
#include <map>
#include <mutex>
#include <vector>
#include <string>
#include <cmath>
#include <cstring>
#include <functional>
#include <algorithm>

static thread_local std::string g_create_error;
static thread_local std::string g_op_error;

// ------------------------------------------------------------------------------------------------
// memory helpers
struct Arena {           // bump allocator over one device allocation; base == nullptr => measuring pass
    char* base = nullptr;
    size_t off = 0, cap = 0;
    void* alloc(size_t bytes) {
        off = (off + 255) & ~(size_t)255;
        void* p = base ? base + off : (void*)(uintptr_t)(off + 256);   // fake non-null pointer when measuring
        off += bytes;
        if (base && off > cap) throw rt_error(RT_E_STATE, "arena overflow");
        return p;
    }
};

struct Workspace {       // stack allocator with mark/release; dry == true measures the peak only
    char* base = nullptr;
    size_t off = 0, cap = 0, peak = 0;
    bool dry = false;
    void* alloc(size_t bytes) {
        off = (off + 255) & ~(size_t)255;
        void* p = base ? base + off : (void*)(uintptr_t)(off + 256);
        off += bytes;
        if (off > peak) peak = off;
        if (!dry && off > cap) throw rt_error(RT_E_STATE, "workspace overflow");
        return p;
    }
    float* f32(size_t n) { return (float*)alloc(n * 4); }
    bf16_t* b16(size_t n) { return (bf16_t*)alloc(n * 2); }
    f16_t* f16(size_t n) { return (f16_t*)alloc(n * 2); }
    size_t mark() const { return off; }
    void release(size_t m) { off = m; }
};
struct Scope {
    Workspace& w; size_t m;
    explicit Scope(Workspace& ws) : w(ws), m(ws.mark()) {}
    ~Scope() { w.release(m); }
};

// ------------------------------------------------------------------------------------------------
// plan
struct NormW { float* g = nullptr; float* b = nullptr; int C = 0; };
struct MatW { bf16_t* w = nullptr; float* b = nullptr; int N = 0, K = 0; };   // packed [N, K] bf16 (+ f32 bias)

struct ResnetP {
    std::string name;
    int cin = 0, cout = 0;
    int temb_first = 0;          // prefix sum of cout over the resnets before this one (index into the per-forward temb buffer)
    NormW n1, n2;
    MatW c1, c2, temb, sc;
    bool has_sc = false;
};
struct TBlockP {
    NormW ln1, ln2, ln3;
    MatW qk1, v1, out1, q2, k2, v2, out2, ff1, ff2;
    // LayerNorm folded into its consumers (gemm16.hip, "LNF"): W' = bf16(gamma W) in the place of w (no .b), and per weight row the pair
    // (s, c) = (row sum of W', b + W beta).  Derived from the packed arena by ensure_fold(); null when the block's width has no folded form.
    MatW qk1f, v1f, q2f, ff1f;
    float* qk1s = nullptr; float* v1s = nullptr; float* q2s = nullptr; float* ff1s = nullptr;
    bf16_t* kcache = nullptr;    // [maxP*96, H*DP]
    bf16_t* vtcache = nullptr;   // [H*DP, maxP*96]
    // token-map attention store: index 0 = attn1, 1 = attn2
    std::string mod_name[2];
    float* store[2] = {nullptr, nullptr};
    int store_mode[2] = {0, 0};            // 0 off, 1 accumulate after the 10th call, 2 overwrite after the 10th call
    int store_calls[2] = {0, 0};
    int store_rows[2] = {0, 0}, store_cols[2] = {0, 0};
    size_t store_cap[2] = {0, 0};
};
struct TransformerP {
    std::string name;
    int C = 0, heads = 0, d = 0, DP = 0, level = 0;
    NormW gn;
    MatW pin, pout;
    std::vector<TBlockP> blocks;
};
struct DownP { std::vector<ResnetP> res; std::vector<TransformerP> attn; bool has_attn = false, has_down = false; MatW down; int C = 0; };
struct UpP { std::vector<ResnetP> res; std::vector<TransformerP> attn; bool has_attn = false, has_up = false; MatW up; int C = 0; };

struct WeightSlot {
    std::string name;
    std::vector<int64_t> shape;
    PackArgs pack;     // everything except src/src_dtype
    bool bound = false;
};

static int pad_head_dim(int d) {
    if (d <= 32) return 32;
    if (d <= 64) return 64;
    if (d <= 96) return 96;
    if (d <= 160) return 160;
    throw rt_error(RT_E_UNSUPPORTED, "head dim > 160 not supported");
}

struct Tensor { f16_t* p; int C; };   // trunk tensor [B, HW, C]: fp16 in HBM (what the reference's fp16 pipelines carry), fp32 in every epilogue's arithmetic

struct FwdIn {
    int B = 0, h = 0, w = 0;
    const float* x[RT_MAXB];
    float scale[RT_MAXB];
    float t = 0;
    int prompt[RT_MAXB], fontsize[RT_MAXB], qk_src[RT_MAXB], res_src[RT_MAXB];
    float* eps_out = nullptr;    // [B, HW, 4] fp32
    int store_stream = -1;       // stream whose head-averaged attention maps are recorded (plain pass: the conditional one)
};

extern int g_store_own_stats;
static inline bool g_store_own_stats_flag() { return g_store_own_stats != 0; }
// step epilogue kernels (defined in step.hip)
struct StepArgs;
void launch_step_epilogue(const StepArgs& a, hipStream_t st);
void launch_gather_add_rows(const float* base, const float* table, const int* /*host*/ idx, float* out, int B, int C, hipStream_t st);
void launch_inject_add(f16_t* out, const f16_t* sc, const float* hres, const int* /*host*/ src, int B, size_t per_batch, hipStream_t st);
void launch_nhwc4_to_nchw(const float* in, float* out, int B, int HW, hipStream_t st);
void launch_pad_ctx(const float* ctx, bf16_t* out, int P, int D, hipStream_t st);
void launch_background_blend(float* lat, const float* lat_ref, const float* mask_last, int n, hipStream_t st);

#include "step.h"

struct rt_engine {
    rt_config cfg;
    int device = 0;
    hipStream_t stream = nullptr;       // the stream every launch of this engine goes to
    hipStream_t own_stream = nullptr;   // created by rt_create; rt_set_stream(NULL) returns to it
    std::string err;

    Arena arena;                 // packed weights only (this is what a multi-GPU launch broadcasts)
    Arena sarena;                // K/V caches + per-image sampler state
    Arena farena;                // LayerNorm-folded copies of the projections that consume a LayerNorm (derived on every rank from the
                                 // packed arena: not part of what a multi-GPU launch broadcasts)
    char* farena_base = nullptr;
    size_t farena_bytes = 0;
    bool fold_dirty = true;      // a weight was bound / the arena was filled since the last derivation
    char* arena_base = nullptr;
    char* sarena_base = nullptr;
    size_t arena_bytes = 0, sarena_bytes = 0;
    Workspace ws;
    bf16_t* zero = nullptr;

    std::vector<WeightSlot> slots;
    std::map<std::string, int> slot_index;

    // plan
    MatW conv_in, conv_out, t1, t2, a1, a2;
    NormW norm_out;
    std::vector<DownP> down;
    std::vector<UpP> up;
    ResnetP mid_r0, mid_r1;
    TransformerP mid_t;
    int temb_dim = 0;
    // every resnet's time_emb_proj(silu(emb)) is computed by ONE launch at the start of a forward (launch_temb_all)
    std::vector<TembEntry> temb_tab;
    int temb_total = 0;
    TembEntry* temb_tab_dev = nullptr;
    float* temb_all = nullptr;       // [resnet][B][cout] of the running forward

    // per-image state
    int n_prompts = 0;
    float* aug_emb = nullptr;        // [maxP, temb_dim] (zeros when no addition embedding)
    float* wabs = nullptr;           // [2, 96]
    float* wsgn = nullptr;
    float* masks = nullptr;          // [R, 4, HW]
    int n_regions = 0, mask_hw = 0;
    float* lat = nullptr;            // [4, HW]
    float* lat_ref = nullptr;
    float* noise_pred = nullptr;     // [4, HW] CFG-combined prediction of the last step (guidance input)
    float* eps = nullptr;            // [maxB, HW, 4]
    float* ets = nullptr;            // PNDM history [4][2][4*HW]
    float* cur_sample = nullptr;     // PNDM [2][4*HW]
    int lat_h = 0, lat_w = 0;
    // schedule (host)
    int sched_kind = 0, num_inference_steps = 0;
    std::vector<float> timesteps, table;
    int pndm_counter = 0, pndm_nets = 0, pndm_head = 0;
    int steps_done = 0;

    // optional per-launch HIP-event profiling of the MFMA kernels (bench.py roofline leg)
    struct ProfRec { int cls; double flops, bytes; hipEvent_t a, b; };
    bool profiling = false;
    std::vector<ProfRec> prof;
    void prof_begin(int cls, double flops, double bytes = 0.0) {
        if (!profiling) return;
        ProfRec r; r.cls = cls; r.flops = flops; r.bytes = bytes;
        HIP_CHECK(hipEventCreate(&r.a)); HIP_CHECK(hipEventCreate(&r.b));
        HIP_CHECK(hipEventRecord(r.a, stream));
        prof.push_back(r);
    }
    void prof_end() { if (profiling) HIP_CHECK(hipEventRecord(prof.back().b, stream)); }

    // ---------------------------------------------------------------------------- plan building
    void add_slot(const std::string& name, std::vector<int64_t> shape, const PackArgs& pk) {
        WeightSlot s; s.name = name; s.shape = std::move(shape); s.pack = pk;
        slot_index[name] = (int)slots.size();
        slots.push_back(s);
    }
    static PackArgs pk_matrix(void* dst, int rows, int cols, int ld, long s_r) {
        PackArgs p{}; p.dst = dst; p.dst_f32 = 0; p.rows = rows; p.cols = cols; p.ld_dst = ld;
        p.row_map = PACK_ROWS_ID; p.c_inner = cols; p.ci_valid = cols; p.s_r = s_r; p.s_co = 0; p.s_ci = 1; p.scale = 1.f;
        return p;
    }
    static PackArgs pk_vec(void* dst, int n) {
        PackArgs p{}; p.dst = dst; p.dst_f32 = 1; p.rows = n; p.cols = 1; p.ld_dst = 1;
        p.row_map = PACK_ROWS_ID; p.c_inner = 1; p.ci_valid = 1; p.s_r = 1; p.s_co = 0; p.s_ci = 0; p.scale = 1.f;
        return p;
    }
    NormW mk_norm(const std::string& name, int C) {
        NormW n; n.C = C;
        n.g = (float*)arena.alloc((size_t)C * 4); n.b = (float*)arena.alloc((size_t)C * 4);
        add_slot(name + ".weight", {C}, pk_vec(n.g, C));
        add_slot(name + ".bias", {C}, pk_vec(n.b, C));
        return n;
    }
    // nn.Linear [N, K] (optionally a 1x1 conv [N, K, 1, 1])
    MatW mk_linear(const std::string& name, int K, int N, bool bias, bool as_conv1x1 = false) {
        MatW m; m.N = N; m.K = K;
        m.w = (bf16_t*)arena.alloc((size_t)N * K * 2);
        std::vector<int64_t> shp = as_conv1x1 ? std::vector<int64_t>{N, K, 1, 1} : std::vector<int64_t>{N, K};
        add_slot(name + ".weight", shp, pk_matrix(m.w, N, K, K, K));
        if (bias) { m.b = (float*)arena.alloc((size_t)N * 4); add_slot(name + ".bias", {N}, pk_vec(m.b, N)); }
        return m;
    }
    // nn.Conv2d 3x3 [Cout, Cin, 3, 3] -> [Cout, 9*CinP], K index = tap*CinP + c
    MatW mk_conv3(const std::string& name, int Cin, int Cout) {
        const int CinP = (Cin + 7) & ~7;
        MatW m; m.N = Cout; m.K = 9 * CinP;
        m.w = (bf16_t*)arena.alloc((size_t)Cout * m.K * 2);
        PackArgs p{}; p.dst = m.w; p.rows = Cout; p.cols = m.K; p.ld_dst = m.K; p.row_map = PACK_ROWS_ID;
        p.c_inner = CinP; p.ci_valid = Cin; p.s_r = (long)Cin * 9; p.s_co = 1; p.s_ci = 9; p.scale = 1.f;
        add_slot(name + ".weight", {Cout, Cin, 3, 3}, p);
        m.b = (float*)arena.alloc((size_t)Cout * 4);
        add_slot(name + ".bias", {Cout}, pk_vec(m.b, Cout));
        return m;
    }
    ResnetP mk_resnet(const std::string& name, int cin, int cout) {
        RT_REQUIRE(cin % 8 == 0 && cout % 8 == 0, "resnet: channel counts must be multiples of 8");
        ResnetP r; r.name = name; r.cin = cin; r.cout = cout;
        r.n1 = mk_norm(name + ".norm1", cin);
        r.c1 = mk_conv3(name + ".conv1", cin, cout);
        r.temb = mk_linear(name + ".time_emb_proj", temb_dim, cout, true);
        r.n2 = mk_norm(name + ".norm2", cout);
        r.c2 = mk_conv3(name + ".conv2", cout, cout);
        r.has_sc = cin != cout;
        if (r.has_sc) r.sc = mk_linear(name + ".conv_shortcut", cin, cout, true, true);
        r.temb_first = temb_total; temb_total += cout;
        temb_tab.push_back(TembEntry{r.temb.w, r.temb.b, cout, r.temb_first});
        return r;
    }
    // q/k/v projection with head padding d -> DP on the output rows
    MatW mk_headproj(const std::string& wname, int K, int H, int d, int DP, float scale, bf16_t* dst_rows, int ld) {
        MatW m; m.N = H * DP; m.K = K; m.w = dst_rows;
        PackArgs p = pk_matrix(dst_rows, H * DP, K, ld, K);
        p.row_map = PACK_ROWS_HEADPAD; p.rm_a = DP; p.rm_b = d; p.scale = scale;
        add_slot(wname + ".weight", {H * d, K}, p);
        return m;
    }
    MatW mk_outproj(const std::string& name, int C, int H, int d, int DP) {
        MatW m; m.N = C; m.K = H * DP;
        m.w = (bf16_t*)arena.alloc((size_t)C * m.K * 2);
        PackArgs p{}; p.dst = m.w; p.rows = C; p.cols = m.K; p.ld_dst = m.K; p.row_map = PACK_ROWS_ID;
        p.c_inner = DP; p.ci_valid = d; p.s_r = (long)H * d; p.s_co = d; p.s_ci = 1; p.scale = 1.f;
        add_slot(name + ".weight", {C, H * d}, p);
        m.b = (float*)arena.alloc((size_t)C * 4);
        add_slot(name + ".bias", {C}, pk_vec(m.b, C));
        return m;
    }
    // widths whose LayerNorms can be folded: C / 80 = 8 or 16 partial sums per token (SDXL's 640 / 1280; SD-v1.5's 320 / 960-free levels
    // have K = 320, outside the 16x16x32 family)
    static bool ln_fold_width(int C) { return C == 640 || C == 1280; }
    // W' / s / c of every folded projection from the packed arena (idempotent; a few ms, once per checkpoint)
    void ensure_fold() {
        if (!fold_dirty || !farena_base) { fold_dirty = false; return; }
        for (auto& sl : slots) if (!sl.bound) return;                   // not all weights are there yet
        for_each_tblock([&](TransformerP& t, TBlockP& k) {
            if (!k.qk1f.w) return;
            launch_ln_fold_derive(k.qk1.w, k.qk1.K, k.qk1.b, k.ln1.g, k.ln1.b, k.qk1.N, k.qk1.K, k.qk1f.w, k.qk1s, stream);
            launch_ln_fold_derive(k.v1.w, k.v1.K, k.v1.b, k.ln1.g, k.ln1.b, k.v1.N, k.v1.K, k.v1f.w, k.v1s, stream);
            launch_ln_fold_derive(k.q2.w, k.q2.K, k.q2.b, k.ln2.g, k.ln2.b, k.q2.N, k.q2.K, k.q2f.w, k.q2s, stream);
            launch_ln_fold_derive(k.ff1.w, k.ff1.K, k.ff1.b, k.ln3.g, k.ln3.b, k.ff1.N, k.ff1.K, k.ff1f.w, k.ff1s, stream);
        });
        fold_dirty = false;
    }
    TransformerP mk_transformer(const std::string& name, int C, int heads, int nlayers, int level) {
        TransformerP t; t.name = name; t.C = C; t.heads = heads; t.d = C / heads; t.DP = pad_head_dim(t.d); t.level = level;
        RT_REQUIRE(C % heads == 0, "channels not divisible by heads");
        const int HD = heads * t.DP, D = cfg.cross_attention_dim;
        const float qscale = (float)(std::pow((double)t.d, -0.5) * 1.4426950408889634);   // d^-1/2 * log2(e)
        t.gn = mk_norm(name + ".norm", C);
        t.pin = mk_linear(name + ".proj_in", C, C, true, !cfg.use_linear_projection);
        for (int li = 0; li < nlayers; ++li) {
            const std::string b = name + ".transformer_blocks." + std::to_string(li);
            TBlockP k;
            k.mod_name[0] = b + ".attn1"; k.mod_name[1] = b + ".attn2";
            k.ln1 = mk_norm(b + ".norm1", C);
            bf16_t* qk = (bf16_t*)arena.alloc((size_t)2 * HD * C * 2);
            k.qk1 = mk_headproj(b + ".attn1.to_q", C, heads, t.d, t.DP, qscale, qk, C);
            mk_headproj(b + ".attn1.to_k", C, heads, t.d, t.DP, 1.f, qk + (size_t)HD * C, C);
            k.qk1.N = 2 * HD;
            bf16_t* v = (bf16_t*)arena.alloc((size_t)HD * C * 2);
            k.v1 = mk_headproj(b + ".attn1.to_v", C, heads, t.d, t.DP, 1.f, v, C);
            k.out1 = mk_outproj(b + ".attn1.to_out.0", C, heads, t.d, t.DP);
            k.ln2 = mk_norm(b + ".norm2", C);
            bf16_t* q2 = (bf16_t*)arena.alloc((size_t)HD * C * 2);
            k.q2 = mk_headproj(b + ".attn2.to_q", C, heads, t.d, t.DP, qscale, q2, C);
            bf16_t* k2 = (bf16_t*)arena.alloc((size_t)HD * D * 2);
            k.k2 = mk_headproj(b + ".attn2.to_k", D, heads, t.d, t.DP, 1.f, k2, D);
            bf16_t* v2 = (bf16_t*)arena.alloc((size_t)HD * D * 2);
            k.v2 = mk_headproj(b + ".attn2.to_v", D, heads, t.d, t.DP, 1.f, v2, D);
            k.out2 = mk_outproj(b + ".attn2.to_out.0", C, heads, t.d, t.DP);
            k.ln3 = mk_norm(b + ".norm3", C);
            // GEGLU: nn.Linear(C, 8C); rows interleaved per 64-block [32 value | 32 gate]
            k.ff1.N = 8 * C; k.ff1.K = C;
            k.ff1.w = (bf16_t*)arena.alloc((size_t)8 * C * C * 2);
            k.ff1.b = (float*)arena.alloc((size_t)8 * C * 4);
            RT_REQUIRE((4 * C) % 32 == 0, "GEGLU width must be a multiple of 32");
            { PackArgs p = pk_matrix(k.ff1.w, 8 * C, C, C, C); p.row_map = PACK_ROWS_GEGLU; add_slot(b + ".ff.net.0.proj.weight", {8 * C, C}, p); }
            { PackArgs p = pk_vec(k.ff1.b, 8 * C); p.row_map = PACK_ROWS_GEGLU; add_slot(b + ".ff.net.0.proj.bias", {8 * C}, p); }
            k.ff2 = mk_linear(b + ".ff.net.2", 4 * C, C, true);
            k.kcache = (bf16_t*)sarena.alloc((size_t)cfg.max_prompts * 96 * HD * 2);
            k.vtcache = (bf16_t*)sarena.alloc((size_t)cfg.max_prompts * 96 * HD * 2);
            if (ln_fold_width(C)) {
                auto mkf = [&](const MatW& w, MatW& f, float*& sv) {
                    f.N = w.N; f.K = w.K;
                    f.w = (bf16_t*)farena.alloc((size_t)w.N * w.K * 2);
                    sv = (float*)farena.alloc((size_t)w.N * 8);
                };
                mkf(k.qk1, k.qk1f, k.qk1s); mkf(k.v1, k.v1f, k.v1s); mkf(k.q2, k.q2f, k.q2s); mkf(k.ff1, k.ff1f, k.ff1s);
            }
            t.blocks.push_back(k);
        }
        t.pout = mk_linear(name + ".proj_out", C, C, true, !cfg.use_linear_projection);
        return t;
    }

    void build_plan() {
        slots.clear(); slot_index.clear(); down.clear(); up.clear(); temb_tab.clear(); temb_total = 0;
        const int L = cfg.n_levels;
        const int* boc = cfg.block_out_channels;
        temb_dim = boc[0] * 4;
        zero = (bf16_t*)sarena.alloc(256);
        conv_in = mk_conv3("conv_in", cfg.in_channels, boc[0]);
        t1 = mk_linear("time_embedding.linear_1", boc[0], temb_dim, true);
        t2 = mk_linear("time_embedding.linear_2", temb_dim, temb_dim, true);
        if (cfg.addition_text_time) {
            a1 = mk_linear("add_embedding.linear_1", cfg.projection_class_embeddings_input_dim, temb_dim, true);
            a2 = mk_linear("add_embedding.linear_2", temb_dim, temb_dim, true);
        }
        int out_c = boc[0];
        for (int i = 0; i < L; ++i) {
            DownP d; const int in_c = out_c; out_c = boc[i]; d.C = out_c;
            d.has_attn = cfg.down_has_attn[i]; d.has_down = i != L - 1;
            const std::string pre = "down_blocks." + std::to_string(i);
            for (int j = 0; j < cfg.layers_per_block[i]; ++j) {
                d.res.push_back(mk_resnet(pre + ".resnets." + std::to_string(j), j == 0 ? in_c : out_c, out_c));
                if (d.has_attn) d.attn.push_back(mk_transformer(pre + ".attentions." + std::to_string(j), out_c, cfg.heads[i], cfg.transformer_layers[i], i));
            }
            if (d.has_down) d.down = mk_conv3(pre + ".downsamplers.0.conv", out_c, out_c);
            down.push_back(d);
        }
        mid_r0 = mk_resnet("mid_block.resnets.0", boc[L - 1], boc[L - 1]);
        mid_t = mk_transformer("mid_block.attentions.0", boc[L - 1], cfg.heads[L - 1], cfg.transformer_layers[L - 1], L - 1);
        mid_r1 = mk_resnet("mid_block.resnets.1", boc[L - 1], boc[L - 1]);
        out_c = boc[L - 1];
        for (int i = 0; i < L; ++i) {
            UpP u; const int prev = out_c; out_c = boc[L - 1 - i];
            const int in_c = boc[L - 1 - std::min(i + 1, L - 1)];
            u.C = out_c; u.has_attn = cfg.up_has_attn[i]; u.has_up = i != L - 1;
            const int nl = cfg.layers_per_block[L - 1 - i] + 1;
            const std::string pre = "up_blocks." + std::to_string(i);
            for (int j = 0; j < nl; ++j) {
                const int skip_c = j == nl - 1 ? in_c : out_c;
                const int res_in = j == 0 ? prev : out_c;
                u.res.push_back(mk_resnet(pre + ".resnets." + std::to_string(j), res_in + skip_c, out_c));
                if (u.has_attn) u.attn.push_back(mk_transformer(pre + ".attentions." + std::to_string(j), out_c, cfg.heads[L - 1 - i], cfg.transformer_layers[L - 1 - i], L - 1 - i));
            }
            if (u.has_up) u.up = mk_conv3(pre + ".upsamplers.0.conv", out_c, out_c);
            up.push_back(u);
        }
        norm_out = mk_norm("conv_norm_out", boc[0]);
        conv_out = mk_conv3("conv_out", boc[0], cfg.out_channels);
        // per-image sampler state
        const size_t HW = (size_t)cfg.latent_h * cfg.latent_w;
        aug_emb = (float*)sarena.alloc((size_t)cfg.max_prompts * temb_dim * 4);
        wabs = (float*)sarena.alloc(2 * 96 * 4); wsgn = (float*)sarena.alloc(2 * 96 * 4);
        masks = (float*)sarena.alloc((size_t)RT_MAXB * 4 * HW * 4);
        lat = (float*)sarena.alloc(4 * HW * 4); lat_ref = (float*)sarena.alloc(4 * HW * 4); noise_pred = (float*)sarena.alloc(4 * HW * 4);
        eps = (float*)sarena.alloc((size_t)cfg.max_streams * HW * 4 * 4);
        ets = (float*)sarena.alloc((size_t)4 * 2 * 4 * HW * 4);
        cur_sample = (float*)sarena.alloc((size_t)2 * 4 * HW * 4);
    }

    // ---------------------------------------------------------------------------- launch helpers
    bool dry() const { return ws.dry; }
    int cur_hw = 0;              // tokens per stream of the block being executed (0 outside the UNet forward): the split-K rule of
                                 // launch_gemm is keyed on ONE stream's share of a GEMM so that results are batch invariant
    // split-K partial sums live in one engine-owned buffer sized by the dry pass of the plan (the former thread_local
    // buffer inside gemm.hip was shared by every stream / device of the thread and grew with a hipMalloc in the middle of a forward)
    float* splitk_buf = nullptr;
    size_t splitk_floats = 0, splitk_need = 0;
    bool run_gemm(GemmArgs& g) {        // false in the dry pass (only records the scratch the launch would need)
        if (dry()) { splitk_need = std::max(splitk_need, gemm_splitk_scratch_floats(g)); return false; }
        g.splitk_ws = splitk_buf; g.splitk_ws_floats = splitk_floats;
        return true;
    }
    // LayerNorm fold of a launch (gemm16.hip, "LNF"): `part` set = consumer (A / X is xb, the un-normalised trunk as bf16; W the folded
    // copy, `s` its (row sum, c) pairs); `emit` set = producer: leaves xb in `copy` and the partials of its output rows in `emit`
    struct LnFold { const float* part = nullptr; int npair = 0; int ld = 0; const float* s = nullptr; float* emit = nullptr; bf16_t* copy = nullptr; };
    void ln_apply(GemmArgs& g, const LnFold* ln, int C) const {
        if (!ln) return;
        g.ln_part = ln->part; g.ln_npair = ln->npair; g.ln_ld = ln->ld; g.ln_s = ln->s; g.ln_emit = ln->emit; g.ln_copy = ln->copy;
        g.ln_inv_c = 1.f / (float)C; g.ln_eps = 1e-5f;
    }
    GemmArgs dense_args(const bf16_t* A, int lda, const MatW& W, int M, void* out, int ldo, int epi, const void* res = nullptr,
                        int ldres = 0, const float* temb = nullptr, int rows_per_batch = 0) const {
        GemmArgs g{}; g.A = A; g.W = W.w; g.bias = W.b; g.out = out; g.res = res; g.temb = temb; g.zero = zero;
        g.mode = A_DENSE; g.epi = epi; g.M = M; g.N = W.N; g.K = W.K; g.lda = lda; g.ldw = W.K; g.ldo = ldo;
        g.ldres = ldres; g.temb_ld = W.N; g.rows_per_batch = rows_per_batch;
        if (cur_hw > 0) { g.split_tiles = cdiv(cur_hw, 128) * cdiv(W.N, 128); g.rows_per_stream = cur_hw; }
        return g;
    }
    GemmArgs vt_args(const MatW& Wv, const bf16_t* X, int ldx, int M, bf16_t* out, int ldo) const {
        GemmArgs g{}; g.A = Wv.w; g.W = X; g.bias = Wv.b; g.out = out; g.zero = zero;
        g.mode = A_DENSE; g.epi = EPI_BF16; g.M = Wv.N; g.N = M; g.K = Wv.K; g.lda = Wv.K; g.ldw = ldx; g.ldo = ldo;
        g.weights_on_rows = 1;
        if (cur_hw > 0) { g.split_tiles = cdiv(Wv.N, 128) * cdiv(cur_hw, 128); g.rows_per_stream = cur_hw; }
        return g;
    }
    void gemm(const bf16_t* A, int lda, const MatW& W, int M, void* out, int ldo, int epi, const void* res = nullptr,
              int ldres = 0, const float* temb = nullptr, int rows_per_batch = 0, const LnFold* ln = nullptr) {
        GemmArgs g = dense_args(A, lda, W, M, out, ldo, epi, res, ldres, temb, rows_per_batch);
        ln_apply(g, ln, ln && ln->part ? W.K : W.N);
        if (!run_gemm(g)) return;
        prof_begin(RT_PROF_GEMM_DENSE, 2.0 * M * W.N * W.K);
        launch_gemm(g, stream);
        prof_end();
    }
    // V^T = Wv [HD, K] x X[M, K]^T -> [HD, M]
    void gemm_vt(const MatW& Wv, const bf16_t* X, int ldx, int M, bf16_t* out, int ldo) {
        GemmArgs g = vt_args(Wv, X, ldx, M, out, ldo);
        if (!run_gemm(g)) return;
        prof_begin(RT_PROF_GEMM_DENSE, 2.0 * M * Wv.N * Wv.K);
        launch_gemm(g, stream);
        prof_end();
    }
    // attn1: the stacked Q|K projection of the first Mqk rows and V^T of all M rows read the same LayerNorm output: ONE grouped launch
    // where gemm16.hip has the pair of tiles (launch_gemm_pair), otherwise one launch each - the same tile bodies either way
    void gemm_qk_vt(const bf16_t* X, int ldx, const MatW& Wqk, int Mqk, bf16_t* qk, int ldqk, const MatW& Wv, int M, bf16_t* vt, int ldvt,
                    const LnFold* lnqk = nullptr, const LnFold* lnv = nullptr) {
        GemmArgs a = dense_args(X, ldx, Wqk, Mqk, qk, ldqk, EPI_BF16);
        GemmArgs b = vt_args(Wv, X, ldx, M, vt, ldvt);
        ln_apply(a, lnqk, Wqk.K); ln_apply(b, lnv, Wv.K);
        const bool ra = run_gemm(a), rb = run_gemm(b);
        if (!ra || !rb) return;
        prof_begin(RT_PROF_GEMM_DENSE, 2.0 * Mqk * Wqk.N * Wqk.K + 2.0 * M * Wv.N * Wv.K);
        launch_gemm_pair(a, b, stream);
        prof_end();
    }
    void conv3(const bf16_t* in, int mode, const MatW& W, int B, int Hin, int Win, int CinP, void* out, int epi,
               const void* res = nullptr, const float* temb = nullptr) {
        int Hout = Hin, Wout = Win;
        if (mode == A_CONV3_S2) { Hout = (Hin + 1) / 2; Wout = (Win + 1) / 2; }   // k3 s2 p1
        if (mode == A_CONV3_UP2) { Hout = Hin * 2; Wout = Win * 2; }
        GemmArgs g{}; g.A = in; g.W = W.w; g.bias = W.b; g.out = out; g.res = res; g.temb = temb; g.zero = zero;
        g.mode = mode; g.epi = epi; g.M = B * Hout * Wout; g.N = W.N; g.K = W.K; g.lda = 0; g.ldw = W.K; g.ldo = W.N;
        g.ldres = W.N; g.temb_ld = W.N; g.rows_per_batch = Hout * Wout;
        g.Hin = Hin; g.Win = Win; g.Cin = CinP; g.Hout = Hout; g.Wout = Wout;
        g.split_tiles = cdiv(Hout * Wout, 128) * cdiv(W.N, 128);
        RT_REQUIRE(W.K == 9 * CinP, "conv: weight/input channel mismatch");
        if (!run_gemm(g)) return;
        prof_begin(RT_PROF_GEMM_CONV, 2.0 * g.M * W.N * W.K);
        launch_gemm(g, stream);
        prof_end();
    }
    void groupnorm(const void* x1, const void* x2, int in_type /* 0 fp32, 1 bf16, 2 fp16 */, int C1, int C2, int B, int HW, const NormW& n, float eps_,
                   bool silu, bf16_t* out, bf16_t* raw) {
        Scope sc(ws);
        const int nchunk = groupnorm_nchunk(HW);
        float* partial = ws.f32((size_t)B * nchunk * cfg.norm_groups * 2);
        if (dry()) return;
        GroupNormArgs a{}; a.x1 = x1; a.x2 = x2; a.in_bf16 = in_type; a.C1 = C1; a.C2 = C2; a.G = cfg.norm_groups; a.B = B;
        a.HW = HW; a.gamma = n.g; a.beta = n.b; a.eps = eps_; a.silu = silu; a.out = out; a.raw_out = raw;
        a.partial = partial; a.nchunk = nchunk; a.rows_per_chunk = groupnorm_rows_per_chunk(HW);
        a.fuse_finalize = 1;
        launch_groupnorm(a, stream);
    }
    void layernorm(const f16_t* x, const NormW& n, bf16_t* out, int rows) {
        if (dry()) return;
        launch_layernorm(x, 1, n.g, n.b, out, rows, n.C, 1e-5f, stream);
    }

    // ---------------------------------------------------------------------------- blocks
    // ResnetBlock2D.forward (models/resnet.py:591-645); x2 = skip tensor of the up path (virtual concat)
    Tensor resnet(const ResnetP& r, const FwdIn& in, int HW, int Hh, int Ww, Tensor x1, const Tensor* x2, const float* emb,
                  bool inject_here) {
        const int B = in.B, M = B * HW;
        const int c1 = x1.C, c2 = x2 ? x2->C : 0;
        cur_hw = HW;
        RT_REQUIRE(c1 + c2 == r.cin, "resnet: input channel mismatch");
        f16_t* out = ws.f16((size_t)M * r.cout);
        {
            Scope sc(ws);
            bf16_t* h1 = ws.b16((size_t)M * r.cin);
            bf16_t* raw = r.has_sc ? ws.b16((size_t)M * r.cin) : nullptr;
            groupnorm(x1.p, x2 ? x2->p : nullptr, 2, c1, c2, B, HW, r.n1, cfg.norm_eps, true, h1, raw);
            const float* tp = temb_all + (size_t)B * r.temb_first;        // computed for all resnets at the start of the forward
            // rich-text feature injection (resnet.py:639-643): out[b] = shortcut(x[b]) + hidden[res_src[b]].  The residual branch
            // (conv1 / norm2 / conv2) of an injected stream is never used, so the trailing run of injected streams is not computed
            // at all (the region streams of a rich-text step are the last ones): Bk streams keep their branch.
            int Bk = B;
            bool any_inject = false;
            if (inject_here) {
                for (int b = 0; b < B; ++b) any_inject |= in.res_src[b] >= 0;
                while (Bk > 1 && in.res_src[Bk - 1] >= 0) --Bk;
                for (int b = 0; b < B; ++b)
                    if (in.res_src[b] >= 0) RT_REQUIRE(in.res_src[b] < Bk && in.res_src[in.res_src[b]] < 0, "resnet: a feature source stream must compute its own residual branch");
            }
            if (dry()) Bk = B;                       // the workspace is sized for the case without injection
            const int Mk = Bk * HW;
            bf16_t* h2 = ws.b16((size_t)Mk * r.cout);
            conv3(h1, A_CONV3, r.c1, Bk, Hh, Ww, r.cin, h2, EPI_BF16_TEMB, nullptr, tp);
            bf16_t* h3 = ws.b16((size_t)Mk * r.cout);
            groupnorm(h2, nullptr, 1, r.cout, 0, Bk, HW, r.n2, cfg.norm_eps, true, h3, nullptr);
            const f16_t* resid = x1.p;
            if (r.has_sc) { gemm(raw, r.cin, r.sc, M, out, r.cout, EPI_F16); resid = out; }
            else RT_REQUIRE(!x2, "resnet without shortcut cannot take a concat input");
            if (!any_inject) {
                conv3(h3, A_CONV3, r.c2, B, Hh, Ww, r.cout, out, EPI_F16, resid);
            } else {
                float* hres = ws.f32((size_t)Mk * r.cout);        // fp32: rounded once, together with the shortcut (launch_inject_add)
                conv3(h3, A_CONV3, r.c2, Bk, Hh, Ww, r.cout, hres, EPI_F32, nullptr);
                int src[RT_MAXB];
                for (int b = 0; b < B; ++b) src[b] = in.res_src[b] >= 0 ? in.res_src[b] : b;
                if (!dry()) launch_inject_add(out, resid, hres, src, B, (size_t)HW * r.cout, stream);
            }
        }
        return Tensor{out, r.cout};
    }

    // Transformer2DModel.forward (models/transformer_2d.py:270-310) + BasicTransformerBlock (attention.py:131-206)
    Tensor transformer(TransformerP& t, const FwdIn& in, int HW, Tensor x) {
        const int B = in.B, M = B * HW, C = t.C, HD = t.heads * t.DP;
        cur_hw = HW;
        RT_REQUIRE(x.C == C, "transformer: channel mismatch");
        RT_REQUIRE(HW % 8 == 0, "transformer: token count (h*w of the attention level) must be a multiple of 8");
        f16_t* out = ws.f16((size_t)M * C);
        {
            Scope sc(ws);
            f16_t* hcur = ws.f16((size_t)M * C);
            // LayerNorm folded into its consumers (gemm16.hip, "LNF"; attention.py:150,168,181) wherever every consumer of the LayerNorm has
            // the folded instantiation - a pure function of the layer's shape (width, tokens per stream), never of the batch.  The trunk's
            // producers (proj_in, to_out, ff.net.2: fp16-trunk epilogues) then leave the per-row partial sums next to it; a producer
            // without that form is followed by the stand-alone partials kernel.  Debug bit 22 restores the LayerNorm launches.
            float* part = ws.f32((size_t)M * 16);                    // up to 4 tile pairs x float4 per token
            bf16_t* xb = ws.b16((size_t)M * C);                      // the trunk as bf16, written next to it by its producer
            bool fold1 = false, fold2 = false, fold3 = false, emit_pin = false, emit_out = false, emit_ff2 = false;
            int ln_bn = 160;                                         // column-tile width of the partials (a function of the class of the producers: of (C, tokens per stream))
            if (gemm_lnfold_enabled() && ln_fold_width(C) && !t.blocks.empty() && t.blocks[0].qk1f.w) {
                const TBlockP& k0 = t.blocks[0];
                const int bn_pin = gemm_ln_emit_bn(dense_args(nullptr, C, t.pin, M, nullptr, C, EPI_F16));
                const int bn_out = gemm_ln_emit_bn(dense_args(nullptr, HD, k0.out1, M, nullptr, C, EPI_F16));
                const int bn_ff2 = gemm_ln_emit_bn(dense_args(nullptr, 4 * C, k0.ff2, M, nullptr, C, EPI_F16));
                ln_bn = bn_out ? bn_out : (bn_ff2 ? bn_ff2 : (bn_pin ? bn_pin : 160));
                emit_pin = bn_pin == ln_bn; emit_out = bn_out == ln_bn; emit_ff2 = bn_ff2 == ln_bn;     // (one class for all three: same N, same tokens per stream)
                const int np = C / ln_bn / 2;
                GemmArgs cq = dense_args(nullptr, C, k0.qk1f, M, nullptr, 2 * HD, EPI_BF16); cq.ln_npair = np;
                GemmArgs cv = vt_args(k0.v1f, nullptr, C, M, nullptr, M); cv.ln_npair = np;
                GemmArgs c2 = dense_args(nullptr, C, k0.q2f, M, nullptr, HD, EPI_BF16); c2.ln_npair = np;
                GemmArgs c3 = dense_args(nullptr, C, k0.ff1f, M, nullptr, 4 * C, EPI_GEGLU); c3.ln_npair = np;
                fold1 = gemm_ln_fold_ok(cq) && gemm_ln_fold_ok(cv);
                fold2 = gemm_ln_fold_ok(c2);
                fold3 = gemm_ln_fold_ok(c3);
            }
            const int npair = C / ln_bn / 2;
            // partials of the trunk as it stands, for the LayerNorm that follows: from the producer's epilogue (`em` was passed to it) or here
            auto partials_after = [&](bool emitted) { if (!emitted && !dry()) launch_ln_partials(hcur, xb, part, M, C, ln_bn, stream); };
            LnFold em; em.emit = part; em.copy = xb;
            {
                Scope s2(ws);
                bf16_t* g = ws.b16((size_t)M * C);
                groupnorm(x.p, nullptr, 2, C, 0, B, HW, t.gn, 1e-6f, false, g, nullptr);
                gemm(g, C, t.pin, M, hcur, C, EPI_F16, nullptr, 0, nullptr, 0, fold1 && emit_pin ? &em : nullptr);
                if (fold1) partials_after(emit_pin);
            }
            for (size_t bi = 0; bi < t.blocks.size(); ++bi) {
                TBlockP& k = t.blocks[bi];
                const bool last = bi + 1 == t.blocks.size();
                Scope s2(ws);
                bf16_t* n = ws.b16((size_t)M * C);
                // --- attn1 (self; attention_processor.py:476-545)
                if (!fold1) layernorm(hcur, k.ln1, n, M);
                bf16_t* qk = ws.b16((size_t)M * 2 * HD);
                bf16_t* vt = ws.b16((size_t)HD * M);
                bf16_t* o = ws.b16((size_t)M * HD);
                float* store_stats = ws.f32((size_t)t.heads * HW * 2);      // softmax statistics of the token-map accumulation (attn_store.hip)
                // Q,K are only needed for streams that some stream attends with (injected region streams use the
                // text_ref stream's Q,K: attention_processor.py:522-524 discards their own scores)
                int nqk = 0;
                for (int b = 0; b < B; ++b) nqk = std::max(nqk, in.qk_src[b] + 1);
                if (fold1) {
                    LnFold lq; lq.part = part; lq.npair = npair; lq.ld = M; lq.s = k.qk1s;
                    LnFold lv; lv.part = part; lv.npair = npair; lv.ld = M; lv.s = k.v1s;
                    gemm_qk_vt(xb, C, k.qk1f, nqk * HW, qk, 2 * HD, k.v1f, M, vt, M, &lq, &lv);
                } else gemm_qk_vt(n, C, k.qk1, nqk * HW, qk, 2 * HD, k.v1, M, vt, M);
                if (!dry()) {
                    AttnArgs a{}; a.Q = qk; a.ldq = 2 * HD; a.K = qk + HD; a.ldk = 2 * HD; a.VT = vt; a.ldvt = M; a.O = o; a.ldo = HD;
                    for (int b = 0; b < B; ++b) { a.q_src[b] = in.qk_src[b]; a.k_src[b] = in.qk_src[b]; a.v_src[b] = b; a.wset[b] = 0; }
                    a.B = B; a.H = t.heads; a.N = HW; a.NK = HW; a.nk_valid = HW; a.DP = t.DP; a.cross = 0;
                    // a layer whose map is recorded in this call: the attention launch leaves the softmax statistics of the recorded stream,
                    // so the store runs its apply kernel only (debug bit 17: the store computes them itself, round 4's two launches)
                    const bool will_store = in.store_stream >= 0 && k.store_mode[0] && k.store_calls[0] + 1 > 10;      // n_maps[name] > 10 (rd.py:422, xl.py:988)
                    const bool stats_from_attn = will_store && attn_store_takes_stats(HW, HW, t.DP) && in.qk_src[in.store_stream] == in.store_stream;
                    if (stats_from_attn) { a.stats = store_stats; a.stats_b = in.store_stream; }
                    prof_begin(RT_PROF_ATTN_SELF, 4.0 * B * t.heads * (double)HW * HW * t.d);
                    launch_attention(a, stream);
                    prof_end();
                    if (in.store_stream >= 0 && k.store_mode[0] && ++k.store_calls[0] > 10) {
                        RT_REQUIRE((size_t)HW * HW <= k.store_cap[0], "attention store: map larger than the enabled buffer");
                        AttnStoreArgs sa{}; sa.Q = qk; sa.ldq = 2 * HD; sa.q_row0 = (long)in.store_stream * HW;
                        sa.K = qk + HD; sa.ldk = 2 * HD; sa.k_row0 = (long)in.store_stream * HW;
                        sa.out = k.store[0]; sa.H = t.heads; sa.N = HW; sa.NK = HW; sa.NKpad = HW; sa.NKrows = HW; sa.DP = t.DP;
                        sa.overwrite = k.store_mode[0] == 2; sa.stats = store_stats; sa.stats_ready = stats_from_attn ? 1 : 0;
                        prof_begin(RT_PROF_ATTN_STORE, 2.0 * 2.0 * t.heads * (double)HW * HW * t.d, 8.0 * HW * HW + 2.0 * 2.0 * HW * HD);
                        launch_attn_store(sa, stream);
                        prof_end();
                        k.store_rows[0] = HW; k.store_cols[0] = HW;
                    }
                }
                gemm(o, HD, k.out1, M, hcur, C, EPI_F16, hcur, C, nullptr, 0, fold2 && emit_out ? &em : nullptr);
                if (fold2) partials_after(emit_out);
                // --- attn2 (cross, K/V from the per-prompt cache; font-size softmax on flagged streams)
                if (!fold2) layernorm(hcur, k.ln2, n, M);
                // to_q and the 77-key attention as ONE launch (gemm16.hip, EPI_XATTN: the Q tile stays in LDS) wherever the tiling
                // allows it - a pure function of the layer's shape; the token-map capture of the plain pass reads Q from HBM and
                // keeps the two-launch form for the layers it records
                const bool capture2 = in.store_stream >= 0 && k.store_mode[1];
                // round 5: the 77-key attention on its own kernel (cross77_kernel, xblock.hip: 64 queries x 2 heads per workgroup, K / V^T in
                // LDS) behind the plain to_q GEMM is faster than the fused launch of round 4 and serves the capturing layers too (Q is in HBM)
                const bool c77 = gemm_cross77_enabled() && cross77_supported(t.heads, t.DP, HW, 96, 77) && t.d == 64;
                const bool fused2 = !c77 && !fold2 && gemm_xattn_enabled() && xattn_fused_supported(C, t.heads, t.DP, HW) && !capture2;
                // the 640-channel level: to_q, attention AND to_out + residual as one launch with Q / P / O in registers (xblock.hip)
                const bool block2 = !fold2 && gemm_xblock_enabled() && xblock_supported(C, t.heads, t.DP, HW) && t.d == 64 && !capture2;
                if (block2) {
                    if (!dry()) {
                        XBlockArgs xa{}; xa.x = n; xa.wq = k.q2.w; xa.wo = k.out2.w; xa.bo = k.out2.b; xa.kc = k.kcache; xa.vt = k.vtcache;
                        xa.res = hcur; xa.out = hcur; xa.wabs = wabs; xa.wsgn = wsgn; xa.ldk = HD; xa.ldvt = cfg.max_prompts * 96; xa.ldres = C; xa.ldo = C;
                        xa.M = M; xa.tokens = HW; xa.nk_valid = 77; xa.C = C; xa.H = t.heads;
                        for (int b = 0; b < B; ++b) { xa.prompt[b] = in.prompt[b]; xa.wset[b] = in.fontsize[b] ? 1 : -1; }
                        RT_REQUIRE(k.q2.K == C && k.out2.K == HD && HD == C, "xblock: packed projection widths");
                        prof_begin(RT_PROF_XBLOCK, 4.0 * M * HD * C + 4.0 * B * t.heads * (double)HW * 77 * t.d);
                        launch_xblock(xa, stream);
                        prof_end();
                    }
                    if (fold3) partials_after(false);
                } else {
                if (fused2) {
                    if (!dry()) {
                        GemmArgs g{}; g.A = n; g.W = k.q2.w; g.out = o; g.zero = zero; g.mode = A_DENSE; g.epi = EPI_XATTN;
                        g.M = M; g.N = HD; g.K = C; g.lda = C; g.ldw = k.q2.K; g.ldo = HD; g.rows_per_stream = HW;
                        g.xa_k = k.kcache; g.xa_vt = k.vtcache; g.xa_ldk = HD; g.xa_ldvt = cfg.max_prompts * 96; g.xa_tokens = HW; g.xa_nk_valid = 77;
                        g.xa_wabs = wabs; g.xa_wsgn = wsgn;
                        for (int b = 0; b < B; ++b) { g.xa_prompt[b] = in.prompt[b]; g.xa_wset[b] = in.fontsize[b] ? 1 : -1; }
                        prof_begin(RT_PROF_XATTN_FUSED, 2.0 * M * HD * C + 4.0 * B * t.heads * (double)HW * 77 * t.d);
                        launch_xattn_fused(g, stream);
                        prof_end();
                    }
                } else {
                if (fold2) {
                    LnFold l2; l2.part = part; l2.npair = npair; l2.ld = M; l2.s = k.q2s;
                    gemm(xb, C, k.q2f, M, qk, HD, EPI_BF16, nullptr, 0, nullptr, 0, &l2);
                } else gemm(n, C, k.q2, M, qk, HD, EPI_BF16);
                if (!dry()) {
                    AttnArgs a{}; a.Q = qk; a.ldq = HD; a.K = k.kcache; a.ldk = HD; a.VT = k.vtcache; a.ldvt = cfg.max_prompts * 96;
                    a.O = o; a.ldo = HD;
                    for (int b = 0; b < B; ++b) { a.q_src[b] = b; a.k_src[b] = in.prompt[b]; a.v_src[b] = in.prompt[b]; a.wset[b] = in.fontsize[b] ? 1 : -1; }      // -1: plain softmax, no multiplier tables
                    a.wabs = wabs; a.wsgn = wsgn;
                    a.B = B; a.H = t.heads; a.N = HW; a.NK = 96; a.nk_valid = 77; a.DP = t.DP; a.cross = 1;
                    // a recorded layer on cross77_kernel: the launch leaves the softmax statistics of the recorded stream (as the self-attention
                    // launch does for attn1), the store runs its apply kernel only
                    const bool stats2 = c77 && capture2 && k.store_calls[1] + 1 > 10 && !g_store_own_stats_flag();
                    if (stats2) { a.stats = store_stats; a.stats_b = in.store_stream; }
                    prof_begin(RT_PROF_ATTN_CROSS, 4.0 * B * t.heads * (double)HW * 77 * t.d);
                    if (c77) launch_cross77(a, stream); else launch_attention(a, stream);
                    prof_end();
                    if (in.store_stream >= 0 && k.store_mode[1] && ++k.store_calls[1] > 10) {
                        RT_REQUIRE((size_t)HW * 77 <= k.store_cap[1], "attention store: map larger than the enabled buffer");
                        AttnStoreArgs sa{}; sa.Q = qk; sa.ldq = HD; sa.q_row0 = (long)in.store_stream * HW;
                        sa.K = k.kcache; sa.ldk = HD; sa.k_row0 = (long)in.prompt[in.store_stream] * 96;
                        sa.out = k.store[1]; sa.H = t.heads; sa.N = HW; sa.NK = 77; sa.NKpad = 96; sa.NKrows = 96; sa.DP = t.DP;
                        sa.overwrite = k.store_mode[1] == 2;
                        if (stats2) { sa.stats = store_stats; sa.stats_ready = 1; }
                        prof_begin(RT_PROF_ATTN_STORE, 2.0 * 2.0 * t.heads * (double)HW * 77 * t.d, 8.0 * HW * 77 + 2.0 * (HW + 96.0) * HD);
                        launch_attn_store(sa, stream);
                        prof_end();
                        k.store_rows[1] = HW; k.store_cols[1] = 77;
                    }
                }
                }
                gemm(o, HD, k.out2, M, hcur, C, EPI_F16, hcur, C, nullptr, 0, fold3 && emit_out ? &em : nullptr);
                if (fold3) partials_after(emit_out);
                }
                // --- GEGLU feed-forward (attention.py:209-304)
                bf16_t* gg = ws.b16((size_t)M * 4 * C);
                if (fold3) {
                    LnFold l3; l3.part = part; l3.npair = npair; l3.ld = M; l3.s = k.ff1s;
                    gemm(xb, C, k.ff1f, M, gg, 4 * C, EPI_GEGLU, nullptr, 0, nullptr, 0, &l3);
                } else {
                    layernorm(hcur, k.ln3, n, M);
                    gemm(n, C, k.ff1, M, gg, 4 * C, EPI_GEGLU);
                }
                // the next block's norm1 reads what ff.net.2 leaves; behind the LAST block proj_out reads the trunk as bf16 - which is
                // exactly the xb an emitting ff.net.2 writes (the cast launch below is then not needed)
                const bool need1 = fold1 && !last, copy_last = last && fold1 && emit_ff2;
                gemm(gg, 4 * C, k.ff2, M, hcur, C, EPI_F16, hcur, C, nullptr, 0, (need1 || copy_last) && emit_ff2 ? &em : nullptr);
                if (need1) partials_after(emit_ff2);
            }
            const bool have_xb = fold1 && emit_ff2 && !t.blocks.empty();
            bf16_t* hb = ws.b16((size_t)M * C);
            if (!dry() && !have_xb) launch_cast_f16_bf16(hcur, hb, (size_t)M * C, stream);
            gemm(have_xb ? xb : hb, C, t.pout, M, out, C, EPI_F16, x.p, C);
        }
        return Tensor{out, C};
    }

    // UNet2DConditionModel.forward (models/unet_2d_condition.py:703-983), batched over streams
    void unet_forward(const FwdIn& in) {
        const int B = in.B, Hh = in.h, Ww = in.w, HW0 = Hh * Ww;
        RT_REQUIRE(B >= 1 && B <= cfg.max_streams && B <= RT_MAXB, "forward: too many streams");
        RT_REQUIRE(Hh <= cfg.latent_h && Ww <= cfg.latent_w, "forward: latent larger than configured");
        RT_REQUIRE((Hh % (1 << (cfg.n_levels - 1))) == 0 && (Ww % (1 << (cfg.n_levels - 1))) == 0, "forward: latent size not divisible");
        if (!dry()) ensure_fold();       // (normally a no-op: rt_set_prompts derived the folded projections already)
        const size_t m0 = ws.mark();
        // time / addition embeddings (unet_2d_condition.py:784-877)
        float* emb = ws.f32((size_t)B * temb_dim);
        {
            Scope sc(ws);
            float* tsin = ws.f32(cfg.block_out_channels[0]);
            float* e1 = ws.f32(temb_dim); float* e2 = ws.f32(temb_dim);
            if (!dry()) {
                launch_timestep_embed_scalar(in.t, cfg.block_out_channels[0], tsin, stream);
                launch_small_linear(tsin, cfg.block_out_channels[0], t1.w, t1.K, t1.b, e1, temb_dim, 1, temb_dim, t1.K, 0, 0, stream);
                launch_small_linear(e1, temb_dim, t2.w, t2.K, t2.b, e2, temb_dim, 1, temb_dim, temb_dim, 1, 0, stream);
                launch_gather_add_rows(e2, aug_emb, in.prompt, emb, B, temb_dim, stream);
            }
        }
        // time_emb_proj(silu(emb)) of every resnet (resnet.py:611-613): one launch, buffer lives for the whole forward
        temb_all = ws.f32((size_t)B * temb_total);
        {
            Scope sc(ws);
            float* semb = ws.f32((size_t)B * temb_dim);
            if (!dry()) {
                launch_temb_all(emb, temb_dim, semb, temb_tab_dev, (int)temb_tab.size(), temb_total, B, temb_dim, temb_all, stream);
            }
        }
        std::vector<Tensor> skips;
        Tensor x;
        {
            f16_t* x0 = ws.f16((size_t)B * HW0 * cfg.block_out_channels[0]);
            Scope sc(ws);
            bf16_t* x8 = ws.b16((size_t)B * HW0 * 8);
            if (!dry()) {
                PrepArgs p{}; p.B = B; p.HW = HW0; p.dst = x8;
                for (int b = 0; b < B; ++b) { p.src[b] = in.x[b]; p.scale[b] = in.scale[b]; }
                launch_prep_latents(p, stream);
            }
            conv3(x8, A_CONV3, conv_in, B, Hh, Ww, 8, x0, EPI_F16);
            x = Tensor{x0, cfg.block_out_channels[0]};
        }
        skips.push_back(x);
        int ch = Hh, cw = Ww;
        for (size_t i = 0; i < down.size(); ++i) {
            DownP& d = down[i];
            for (size_t j = 0; j < d.res.size(); ++j) {
                x = resnet(d.res[j], in, ch * cw, ch, cw, x, nullptr, emb, false);
                if (d.has_attn) x = transformer(d.attn[j], in, ch * cw, x);
                skips.push_back(x);
            }
            if (d.has_down) {
                const int nh = (ch + 1) / 2, nw = (cw + 1) / 2;
                f16_t* y = ws.f16((size_t)B * nh * nw * d.C);
                {
                    Scope sc(ws);
                    bf16_t* xb = ws.b16((size_t)B * ch * cw * d.C);
                    if (!dry()) launch_cast_f16_bf16(x.p, xb, (size_t)B * ch * cw * d.C, stream);
                    conv3(xb, A_CONV3_S2, d.down, B, ch, cw, d.C, y, EPI_F16);
                }
                ch = nh; cw = nw;
                x = Tensor{y, d.C};
                skips.push_back(x);
            }
        }
        x = resnet(mid_r0, in, ch * cw, ch, cw, x, nullptr, emb, false);
        x = transformer(mid_t, in, ch * cw, x);
        x = resnet(mid_r1, in, ch * cw, ch, cw, x, nullptr, emb, false);
        for (size_t i = 0; i < up.size(); ++i) {
            UpP& u = up[i];
            for (size_t j = 0; j < u.res.size(); ++j) {
                Tensor sk = skips.back(); skips.pop_back();
                const bool inject_here = (i == 1 && j == 1);        // 'up_blocks.1.resnets.1' (rd.py:350, xl.py:1101)
                x = resnet(u.res[j], in, ch * cw, ch, cw, x, &sk, emb, inject_here);
                if (u.has_attn) x = transformer(u.attn[j], in, ch * cw, x);
            }
            if (u.has_up) {
                f16_t* y = ws.f16((size_t)B * ch * cw * 4 * u.C);
                {
                    Scope sc(ws);
                    bf16_t* xb = ws.b16((size_t)B * ch * cw * u.C);
                    if (!dry()) launch_cast_f16_bf16(x.p, xb, (size_t)B * ch * cw * u.C, stream);
                    conv3(xb, A_CONV3_UP2, u.up, B, ch, cw, u.C, y, EPI_F16);
                }
                ch *= 2; cw *= 2;
                x = Tensor{y, u.C};
            }
        }
        {
            Scope sc(ws);
            bf16_t* hn = ws.b16((size_t)B * HW0 * x.C);
            groupnorm(x.p, nullptr, 2, x.C, 0, B, HW0, norm_out, cfg.norm_eps, true, hn, nullptr);
            conv3(hn, A_CONV3, conv_out, B, Hh, Ww, x.C, in.eps_out, EPI_F32);
        }
        ws.release(m0);
        cur_hw = 0;
    }

    // ---------------------------------------------------------------------------- per-image setup
    void set_prompts(const float* pe, const float* pooled, const float* time_ids, int P, int pooled_dim) {
        RT_REQUIRE(P >= 1 && P <= cfg.max_prompts, "set_prompts: too many prompts");
        require_bound();
        const int D = cfg.cross_attention_dim;
        Scope sc(ws);
        bf16_t* ctx = ws.b16((size_t)cfg.max_prompts * 96 * D);
        launch_pad_ctx(pe, ctx, P, D, stream);
        auto build = [&](TransformerP& t) {
            const int HD = t.heads * t.DP;
            for (TBlockP& k : t.blocks) {
                gemm(ctx, D, k.k2, P * 96, k.kcache, HD, EPI_BF16);
                gemm_vt(k.v2, ctx, D, P * 96, k.vtcache, cfg.max_prompts * 96);
            }
        };
        // one prompt = one "stream" of 96 rows: the tile class / split-K rule of these GEMMs is keyed on that, so a prompt's cached
        // K / V bits do not depend on how many other prompts share the set
        cur_hw = 96;
        for (auto& d : down) for (auto& t : d.attn) build(t);
        build(mid_t);
        for (auto& u : up) for (auto& t : u.attn) build(t);
        cur_hw = 0;
        HIP_CHECK(hipMemsetAsync(aug_emb, 0, (size_t)cfg.max_prompts * temb_dim * 4, stream));
        if (cfg.addition_text_time) {
            RT_REQUIRE(pooled && time_ids, "set_prompts: SDXL needs pooled embeds and time_ids");
            const int td = cfg.addition_time_embed_dim, Kin = cfg.projection_class_embeddings_input_dim;
            RT_REQUIRE(pooled_dim + 6 * td == Kin, "set_prompts: pooled_dim + 6*time_embed_dim != projection input dim");
            float* tid = ws.f32(8); float* tsin = ws.f32((size_t)6 * td);
            float* add = ws.f32((size_t)P * Kin); float* hmid = ws.f32((size_t)P * temb_dim);
            HIP_CHECK(hipMemcpyAsync(tid, time_ids, 24, hipMemcpyHostToDevice, stream));
            launch_timestep_embed(tid, 6, td, tsin, td, stream);
            for (int p_ = 0; p_ < P; ++p_) {
                HIP_CHECK(hipMemcpyAsync(add + (size_t)p_ * Kin, pooled + (size_t)p_ * pooled_dim, (size_t)pooled_dim * 4, hipMemcpyDeviceToDevice, stream));
                HIP_CHECK(hipMemcpyAsync(add + (size_t)p_ * Kin + pooled_dim, tsin, (size_t)6 * td * 4, hipMemcpyDeviceToDevice, stream));
            }
            launch_small_linear(add, Kin, a1.w, a1.K, a1.b, hmid, temb_dim, P, temb_dim, Kin, 0, 0, stream);
            launch_small_linear(hmid, temb_dim, a2.w, a2.K, a2.b, aug_emb, temb_dim, P, temb_dim, temb_dim, 1, 0, stream);
        }
        n_prompts = P;
        HIP_CHECK(hipStreamSynchronize(stream));   // host buffers (time_ids) may go away
    }

    void set_fontsize(const int64_t* word_pos, const float* font_size, int n) {
        // set 0: plain softmax; set 1: font-size softmax (attention_processor.py:386-396).  wabs = 0 on padded keys.
        float ha[2 * 96], hs[2 * 96];
        for (int s = 0; s < 2; ++s) for (int k = 0; k < 96; ++k) { ha[s * 96 + k] = k < 77 ? 1.f : 0.f; hs[s * 96 + k] = 1.f; }
        for (int i = 0; i < n; ++i) {
            RT_REQUIRE(word_pos[i] >= 0 && word_pos[i] < 77, "set_fontsize: word_pos out of range");
        }
        // torch index_put semantics with repeated indices: the last write wins for '=' and every '*=' on the
        // gathered copy also resolves to a single write per index (attention_processor.py:393,396)
        for (int i = 0; i < n; ++i) {
            const int k = (int)word_pos[i];
            ha[96 + k] = std::fabs(font_size[i]);
            hs[96 + k] = font_size[i] > 0 ? 1.f : (font_size[i] < 0 ? -1.f : 0.f);
        }
        HIP_CHECK(hipMemcpyAsync(wabs, ha, sizeof(ha), hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(wsgn, hs, sizeof(hs), hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
    }

    template <typename F> void for_each_tblock(F f) {
        for (auto& d : down) for (auto& t : d.attn) for (auto& k : t.blocks) f(t, k);
        for (auto& k : mid_t.blocks) f(mid_t, k);
        for (auto& u : up) for (auto& t : u.attn) for (auto& k : t.blocks) f(t, k);
    }
    bool any_store() { bool any = false; for_each_tblock([&](TransformerP&, TBlockP& k) { any |= k.store_mode[0] || k.store_mode[1]; }); return any; }
    void attn_store_enable(const std::string& name, int mode) {
        bool found = false;
        for_each_tblock([&](TransformerP& t, TBlockP& k) {
            for (int w = 0; w < 2; ++w) {
                if (k.mod_name[w] != name) continue;
                found = true;
                const size_t N = (size_t)(cfg.latent_h >> t.level) * (cfg.latent_w >> t.level);
                const size_t cols = w == 0 ? N : 77;
                if (mode != 0 && w == 0 && N > 1024)
                    throw rt_error(RT_E_UNSUPPORTED, "attention store: self-attention maps above 32x32 are not recorded (never consumed: attention_utils.py:243-248)");
                if (mode != 0 && k.store_cap[w] < N * cols) {
                    if (k.store[w]) HIP_CHECK(hipFree(k.store[w]));
                    HIP_CHECK(hipMalloc((void**)&k.store[w], N * cols * 4));
                    k.store_cap[w] = N * cols;
                }
                k.store_mode[w] = mode; k.store_calls[w] = 0; k.store_rows[w] = k.store_cols[w] = 0;
                if (mode != 0) HIP_CHECK(hipMemsetAsync(k.store[w], 0, k.store_cap[w] * 4, stream));
            }
        });
        if (!found) throw rt_error(RT_E_INVALID, "attention store: unknown attention module " + name);
    }
    void attn_store_reset() {
        for_each_tblock([&](TransformerP&, TBlockP& k) {
            for (int w = 0; w < 2; ++w) {
                k.store_calls[w] = 0; k.store_rows[w] = k.store_cols[w] = 0;
                if (k.store[w]) HIP_CHECK(hipMemsetAsync(k.store[w], 0, k.store_cap[w] * 4, stream));
            }
        });
    }

    void require_bound() {
        for (auto& s : slots) if (!s.bound) throw rt_error(RT_E_MISSING_WEIGHT, "weight not bound: " + s.name);
    }

    // ---------------------------------------------------------------------------- step drivers
    void region_step(int i, float g, double inject_selfattn, double inject_background, bool xl, bool elide, bool defer_blend);
    void region_plan(int i, float g, double isa, double ibg, bool xl, bool elide, bool defer_blend, FwdIn& in, StepArgs& a, bool& blend_out, bool& inject_out);
    void region_finish(int i, StepArgs& a, bool blend_deferred);
    // intra-image split (step_driver.inl): the forwards of this rank's contiguous stream range, then - after the ranks exchanged their
    // slices of `eps` - the epilogue on all of them
    void region_step_part(int i, float g, double isa, double ibg, bool xl, bool elide, bool defer_blend, int part, int nparts, int* first, int* count, int* plan_info);
    void region_step_finish(int i, float g, double isa, double ibg, bool xl, bool elide, bool defer_blend);
    bool pending_blend = false;
    void plain_step(int i, float g);
    void plain_forward(int i, int first, int count);
    void plain_finish(int i, float g);
    void plain_step_part(int i, int part, int nparts, int* first, int* count);
};

#include "step_driver.inl"

// ================================================================================================
// C ABI
#define RT_TRY(e, ...)                                                             \
    try { __VA_ARGS__; return RT_OK; }                                                    \
    catch (const rt_error& ex) { (e)->err = ex.what(); return ex.code; }           \
    catch (const std::exception& ex) { (e)->err = ex.what(); return RT_E_INVALID; }

extern "C" {

int rt_create(const rt_config* cfg, int device, rt_engine** out) {
    rt_engine* e = nullptr;
    try {
        RT_REQUIRE(cfg && out, "rt_create: null argument");
        RT_REQUIRE(cfg->n_levels >= 2 && cfg->n_levels <= RT_MAX_LEVELS, "rt_create: n_levels");
        RT_REQUIRE(cfg->max_streams >= 1 && cfg->max_streams <= RT_MAXB, "rt_create: max_streams must be in [1,16]");
        RT_REQUIRE(cfg->in_channels == 4 && cfg->out_channels == 4, "rt_create: latent channels must be 4");
        RT_REQUIRE(cfg->cross_attention_dim % 8 == 0, "rt_create: cross_attention_dim % 8");
        e = new rt_engine();
        e->cfg = *cfg; e->device = device;
        // pass 1: measure the arena (no device needed: lets CPU-only hosts enumerate the weight table)
        e->arena = Arena(); e->sarena = Arena(); e->farena = Arena(); e->ws.dry = true;
        e->build_plan();
        e->arena_bytes = e->arena.off + 256; e->sarena_bytes = e->sarena.off + 256; e->farena_bytes = e->farena.off + 256;
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0 || device < 0) {
            // weight-table-only engine (CPU box): every device call will fail with RT_E_STATE
            e->arena_base = nullptr;
            *out = e;
            return RT_OK;
        }
        HIP_CHECK(hipSetDevice(device));
        HIP_CHECK(hipStreamCreate(&e->stream));
        e->own_stream = e->stream;
        HIP_CHECK(hipMalloc((void**)&e->arena_base, e->arena_bytes));
        HIP_CHECK(hipMemset(e->arena_base, 0, e->arena_bytes));
        HIP_CHECK(hipMalloc((void**)&e->sarena_base, e->sarena_bytes));
        HIP_CHECK(hipMemset(e->sarena_base, 0, e->sarena_bytes));
        HIP_CHECK(hipMalloc((void**)&e->farena_base, e->farena_bytes));
        HIP_CHECK(hipMemset(e->farena_base, 0, e->farena_bytes));
        e->arena = Arena(); e->arena.base = e->arena_base; e->arena.cap = e->arena_bytes;
        e->sarena = Arena(); e->sarena.base = e->sarena_base; e->sarena.cap = e->sarena_bytes;
        e->farena = Arena(); e->farena.base = e->farena_base; e->farena.cap = e->farena_bytes;
        e->build_plan();
        // pass 2: measure the workspace with a dry forward at the largest shape
        {
            FwdIn in{}; in.B = cfg->max_streams; in.h = cfg->latent_h; in.w = cfg->latent_w;
            for (int b = 0; b < in.B; ++b) { in.qk_src[b] = b; in.res_src[b] = b ? 0 : -1; in.prompt[b] = 0; }
            e->ws = Workspace(); e->ws.dry = true;
            e->splitk_need = 0;
            e->unet_forward(in);
            {   // the K / V^T cache GEMMs of rt_set_prompts (one prompt = one 96-row stream) at the largest prompt count
                const int D = cfg->cross_attention_dim, P96 = cfg->max_prompts * 96;
                e->cur_hw = 96;
                e->for_each_tblock([&](TransformerP& t, TBlockP& k) {
                    e->gemm(nullptr, D, k.k2, P96, nullptr, t.heads * t.DP, EPI_BF16);
                    e->gemm_vt(k.v2, nullptr, D, P96, nullptr, P96);
                });
                e->cur_hw = 0;
            }
            size_t peak = e->ws.peak;
            {   // split-K slices GROW as the maps shrink (the rule keys on ONE stream's tile count): an engine built for 128 x 128
                // latents and stepped at 64 x 64 needs more partial-sum scratch than the largest shape measures.  Dry passes at every
                // halved latent size the architecture admits record it, so no forward ever allocates.
                const int align = 1 << (cfg->n_levels - 1);
                for (int s = 2; s <= 16; s *= 2) {
                    FwdIn in2 = in; in2.h = cfg->latent_h / s; in2.w = cfg->latent_w / s;
                    if (in2.h * s != cfg->latent_h || in2.w * s != cfg->latent_w || in2.h % align || in2.w % align || in2.h < 8 || in2.w < 8) break;
                    const int deep = (in2.h / align) * (in2.w / align);
                    if (deep % 8) break;                               // attention levels need h*w % 8 == 0
                    e->unet_forward(in2);
                }
            }
            // set_prompts scratch
            size_t sp = (size_t)cfg->max_prompts * 96 * cfg->cross_attention_dim * 2 + (size_t)cfg->max_prompts * (cfg->projection_class_embeddings_input_dim + e->temb_dim) * 4 + (1 << 16);
            if (sp > peak) peak = sp;
            peak += 1 << 20;
            e->ws = Workspace();
            HIP_CHECK(hipMalloc((void**)&e->ws.base, peak));
            e->ws.cap = peak;
            if (e->splitk_need) {
                HIP_CHECK(hipMalloc((void**)&e->splitk_buf, e->splitk_need * 4));
                e->splitk_floats = e->splitk_need;
            }
            if (!e->temb_tab.empty()) {      // the time_emb_proj table of temb_all_kernel: batch-independent, uploaded once
                HIP_CHECK(hipMalloc((void**)&e->temb_tab_dev, e->temb_tab.size() * sizeof(TembEntry)));
                HIP_CHECK(hipMemcpy(e->temb_tab_dev, e->temb_tab.data(), e->temb_tab.size() * sizeof(TembEntry), hipMemcpyHostToDevice));
            }
        }
        e->set_fontsize(nullptr, nullptr, 0);      // multiplier set 0/1 = plain softmax until rt_set_fontsize is called
        *out = e;
        return RT_OK;
    } catch (const std::exception& ex) {
        g_create_error = ex.what();
        delete e;
        return RT_E_INVALID;
    }
}

int rt_destroy(rt_engine* e) {
    if (!e) return RT_OK;
    if (e->arena_base) {
        (void)hipSetDevice(e->device); (void)hipStreamSynchronize(e->stream);
        for (auto& r : e->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
        (void)hipFree(e->arena_base); (void)hipFree(e->sarena_base); (void)hipFree(e->farena_base); (void)hipFree(e->ws.base); (void)hipFree(e->temb_tab_dev); (void)hipFree(e->splitk_buf); (void)hipStreamDestroy(e->own_stream);
    }
    delete e;
    return RT_OK;
}

const char* rt_last_error(rt_engine* e) { return e ? e->err.c_str() : g_create_error.c_str(); }

static void need_device(rt_engine* e) { if (!e->arena_base) throw rt_error(RT_E_STATE, "engine has no device (weight-table-only)"); }

int rt_set_stream(rt_engine* e, void* s) { RT_TRY(e, { need_device(e); e->stream = s ? (hipStream_t)s : e->own_stream; }) }
int rt_synchronize(rt_engine* e) { RT_TRY(e, { need_device(e); HIP_CHECK(hipStreamSynchronize(e->stream)); }) }

int rt_weight_count(rt_engine* e) { return (int)e->slots.size(); }
int rt_weight_info(rt_engine* e, int idx, char* name, int cap, int64_t* shape4, int* ndim) {
    RT_TRY(e, {
        RT_REQUIRE(idx >= 0 && idx < (int)e->slots.size(), "rt_weight_info: index");
        const WeightSlot& s = e->slots[idx];
        RT_REQUIRE((int)s.name.size() < cap, "rt_weight_info: name buffer too small");
        std::strcpy(name, s.name.c_str());
        *ndim = (int)s.shape.size();
        for (size_t i = 0; i < s.shape.size(); ++i) shape4[i] = s.shape[i];
    })
}
int rt_bind_weight(rt_engine* e, const char* name, const void* ptr, int dtype, const int64_t* shape, int ndim) {
    RT_TRY(e, {
        need_device(e);
        auto it = e->slot_index.find(name);
        if (it == e->slot_index.end()) throw rt_error(RT_E_INVALID, std::string("unknown weight: ") + name);
        WeightSlot& s = e->slots[it->second];
        RT_REQUIRE(ndim == (int)s.shape.size(), "rt_bind_weight: rank mismatch");
        for (int i = 0; i < ndim; ++i)
            if (shape[i] != s.shape[i]) throw rt_error(RT_E_INVALID, std::string("shape mismatch for ") + name);
        RT_REQUIRE(dtype >= 0 && dtype <= 2, "rt_bind_weight: dtype");
        PackArgs p = s.pack; p.src = ptr; p.src_dtype = dtype;
        launch_pack(p, e->stream);
        s.bound = true;
        e->fold_dirty = true;
    })
}
int rt_weights_missing(rt_engine* e, char* buf, int cap) {
    int n = 0; std::string acc;
    for (auto& s : e->slots) if (!s.bound) { ++n; if (acc.size() + s.name.size() + 2 < (size_t)cap) { acc += s.name; acc += ';'; } }
    if (buf && cap > 0) { std::strncpy(buf, acc.c_str(), cap - 1); buf[cap - 1] = 0; }
    return n;
}
int rt_arena_info(rt_engine* e, void** p, uint64_t* bytes) { RT_TRY(e, { need_device(e); *p = e->arena_base; *bytes = e->arena_bytes; }) }
int rt_arena_mark_bound(rt_engine* e) { for (auto& s : e->slots) s.bound = true; e->fold_dirty = true; return RT_OK; }

int rt_set_prompts(rt_engine* e, const float* pe, const float* pooled, const float* tids, int P, int pooled_dim) {
    RT_TRY(e, { need_device(e); e->ensure_fold(); e->set_prompts(pe, pooled, tids, P, pooled_dim); })
}
int rt_set_masks(rt_engine* e, const float* m, int R, int h, int w) {
    RT_TRY(e, {
        need_device(e);
        RT_REQUIRE(R >= 1 && R <= RT_MAXB && h <= e->cfg.latent_h && w <= e->cfg.latent_w, "rt_set_masks: bad shape");
        HIP_CHECK(hipMemcpyAsync(e->masks, m, (size_t)R * 4 * h * w * 4, hipMemcpyDeviceToDevice, e->stream));
        e->n_regions = R; e->mask_hw = h * w;
    })
}
int rt_set_fontsize(rt_engine* e, const int64_t* wp, const float* fs, int n) { RT_TRY(e, { need_device(e); e->set_fontsize(wp, fs, n); }) }
int rt_set_schedule(rt_engine* e, int kind, const float* ts, int nts, const float* table, int ntab, int nsteps) {
    RT_TRY(e, {
        RT_REQUIRE(kind == RT_SCHED_EULER || kind == RT_SCHED_PNDM, "rt_set_schedule: kind");
        e->sched_kind = kind; e->num_inference_steps = nsteps;
        e->timesteps.assign(ts, ts + nts); e->table.assign(table, table + ntab);
        if (kind == RT_SCHED_EULER) RT_REQUIRE(ntab == nts + 1, "euler: need n+1 sigmas");
        e->pndm_counter = 0; e->pndm_nets = 0; e->pndm_head = 0; e->steps_done = 0;
    })
}
int rt_set_latents(rt_engine* e, const float* l, int h, int w) {
    RT_TRY(e, {
        need_device(e);
        RT_REQUIRE(h <= e->cfg.latent_h && w <= e->cfg.latent_w, "rt_set_latents: too large");
        const size_t n = (size_t)4 * h * w * 4;
        HIP_CHECK(hipMemcpyAsync(e->lat, l, n, hipMemcpyDeviceToDevice, e->stream));
        HIP_CHECK(hipMemcpyAsync(e->lat_ref, l, n, hipMemcpyDeviceToDevice, e->stream));
        e->lat_h = h; e->lat_w = w;
        e->pndm_counter = 0; e->pndm_nets = 0; e->pndm_head = 0; e->steps_done = 0;
    })
}
int rt_get_latents(rt_engine* e, float* out, float* out_ref) {
    RT_TRY(e, {
        need_device(e);
        const size_t n = (size_t)4 * e->lat_h * e->lat_w * 4;
        HIP_CHECK(hipMemcpyAsync(out, e->lat, n, hipMemcpyDeviceToDevice, e->stream));
        if (out_ref) HIP_CHECK(hipMemcpyAsync(out_ref, e->lat_ref, n, hipMemcpyDeviceToDevice, e->stream));
    })
}
int rt_get_state_ptrs(rt_engine* e, float** latents, float** noise_pred) {
    RT_TRY(e, { need_device(e); HIP_CHECK(hipStreamSynchronize(e->stream)); *latents = e->lat; *noise_pred = e->noise_pred; })
}
int rt_region_step(rt_engine* e, int i, float g, double isa, double ibg, int xl, int elide) {
    RT_TRY(e, { need_device(e); e->region_step(i, g, isa, ibg, xl != 0, (elide & 1) != 0, (elide & 2) != 0); })
}
int rt_region_step_part(rt_engine* e, int i, float g, double isa, double ibg, int xl, int flags, int part, int nparts, int* first, int* count,
                        int* plan_info) {
    RT_TRY(e, { need_device(e); RT_REQUIRE(first && count, "rt_region_step_part: null outputs");
                e->region_step_part(i, g, isa, ibg, xl != 0, (flags & 1) != 0, (flags & 2) != 0, part, nparts, first, count, plan_info); })
}
int rt_region_step_finish(rt_engine* e, int i, float g, double isa, double ibg, int xl, int flags) {
    RT_TRY(e, { need_device(e); e->region_step_finish(i, g, isa, ibg, xl != 0, (flags & 1) != 0, (flags & 2) != 0); })
}
int rt_eps_info(rt_engine* e, void** dev_ptr, unsigned long long* bytes_per_stream, int* max_streams) {
    RT_TRY(e, { need_device(e); RT_REQUIRE(dev_ptr && bytes_per_stream && max_streams, "rt_eps_info: null outputs");
                RT_REQUIRE(e->lat_h > 0, "rt_eps_info: call rt_set_latents first");
                *dev_ptr = e->eps; *bytes_per_stream = (unsigned long long)e->lat_h * e->lat_w * 4 * 4; *max_streams = e->cfg.max_streams; })
}
int rt_background_blend(rt_engine* e) {
    RT_TRY(e, {
        need_device(e);
        if (e->pending_blend) { launch_background_blend(e->lat, e->lat_ref, e->masks + (size_t)(e->n_regions - 1) * 4 * e->mask_hw, 4 * e->mask_hw, e->stream); e->pending_blend = false; }
    })
}
int rt_plain_step(rt_engine* e, int i, float g) { RT_TRY(e, { need_device(e); e->plain_step(i, g); }) }
int rt_plain_step_part(rt_engine* e, int i, int part, int nparts, int* first, int* count) {
    RT_TRY(e, { need_device(e); RT_REQUIRE(first && count, "rt_plain_step_part: null outputs"); e->plain_step_part(i, part, nparts, first, count); })
}
int rt_plain_step_finish(rt_engine* e, int i, float g) { RT_TRY(e, { need_device(e); e->plain_finish(i, g); }) }

int rt_unet_forward(rt_engine* e, const float* x, int B, int h, int w, float t, const float* in_scale, const int* prompt,
                    const int* fontsize, const int* qk_src, const int* res_src, float* out) {
    RT_TRY(e, {
        need_device(e); e->require_bound();
        RT_REQUIRE(B >= 1 && B <= e->cfg.max_streams, "rt_unet_forward: batch");
        RT_REQUIRE(e->n_prompts > 0, "rt_unet_forward: call rt_set_prompts first");
        FwdIn in{}; in.B = B; in.h = h; in.w = w; in.t = t; in.eps_out = e->eps;
        for (int b = 0; b < B; ++b) {
            in.x[b] = x + (size_t)b * 4 * h * w; in.scale[b] = in_scale ? in_scale[b] : 1.f;
            in.prompt[b] = prompt ? prompt[b] : 0; in.fontsize[b] = fontsize ? fontsize[b] : 0;
            in.qk_src[b] = qk_src ? qk_src[b] : b; in.res_src[b] = res_src ? res_src[b] : -1;
            RT_REQUIRE(in.prompt[b] >= 0 && in.prompt[b] < e->n_prompts, "rt_unet_forward: prompt index");
            RT_REQUIRE(in.qk_src[b] >= 0 && in.qk_src[b] < B && in.res_src[b] < B, "rt_unet_forward: source stream index");
        }
        e->unet_forward(in);
        launch_nhwc4_to_nchw(e->eps, out, B, h * w, e->stream);
    })
}

int rt_attn_store_enable(rt_engine* e, const char* name, int mode) {
    RT_TRY(e, { need_device(e); RT_REQUIRE(mode >= 0 && mode <= 2, "rt_attn_store_enable: mode"); e->attn_store_enable(name, mode); })
}
int rt_attn_store_reset(rt_engine* e) { RT_TRY(e, { need_device(e); e->attn_store_reset(); }) }
int rt_attn_store_read(rt_engine* e, const char* name, float* dst, int* n_calls, int* rows, int* cols) {
    RT_TRY(e, {
        need_device(e);
        bool found = false;
        e->for_each_tblock([&](TransformerP&, TBlockP& k) {
            for (int w = 0; w < 2; ++w) {
                if (k.mod_name[w] != name) continue;
                found = true;
                *n_calls = k.store_calls[w]; *rows = k.store_rows[w]; *cols = k.store_cols[w];
                if (dst && k.store[w] && k.store_rows[w] > 0)
                    HIP_CHECK(hipMemcpyAsync(dst, k.store[w], (size_t)k.store_rows[w] * k.store_cols[w] * 4, hipMemcpyDeviceToDevice, e->stream));
            }
        });
        if (!found) throw rt_error(RT_E_INVALID, std::string("attention store: unknown attention module ") + name);
        HIP_CHECK(hipStreamSynchronize(e->stream));
    })
}
int rt_attn_module_count(rt_engine* e) { int n = 0; e->for_each_tblock([&](TransformerP&, TBlockP&) { n += 2; }); return n; }
int rt_attn_module_info(rt_engine* e, int idx, char* name, int cap, int* max_tokens, int* heads) {
    RT_TRY(e, {
        int i = 0; bool found = false;
        e->for_each_tblock([&](TransformerP& t, TBlockP& k) {
            for (int w = 0; w < 2; ++w, ++i) {
                if (i != idx) continue;
                found = true;
                RT_REQUIRE((int)k.mod_name[w].size() < cap, "rt_attn_module_info: name buffer too small");
                std::strcpy(name, k.mod_name[w].c_str());
                *max_tokens = (e->cfg.latent_h >> t.level) * (e->cfg.latent_w >> t.level);
                *heads = t.heads;
            }
        });
        RT_REQUIRE(found, "rt_attn_module_info: index");
    })
}

int rt_profile_enable(rt_engine* e, int on) {
    RT_TRY(e, {
        need_device(e);
        for (auto& r : e->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
        e->prof.clear();
        e->profiling = on != 0;
    })
}
int rt_profile_read(rt_engine* e, int cls, int* count, double* total_ms, double* total_flops) {
    RT_TRY(e, {
        need_device(e);
        HIP_CHECK(hipStreamSynchronize(e->stream));
        int n = 0; double ms = 0, fl = 0;
        for (auto& r : e->prof) if (r.cls == cls) {
            float t = 0; HIP_CHECK(hipEventElapsedTime(&t, r.a, r.b));
            ++n; ms += t; fl += r.flops;
        }
        *count = n; *total_ms = ms; *total_flops = fl;
    })
}
int rt_profile_read2(rt_engine* e, int cls, int* count, double* total_ms, double* total_flops, double* total_bytes) {
    const int rc = rt_profile_read(e, cls, count, total_ms, total_flops);
    if (rc != RT_OK) return rc;
    double by = 0;
    for (auto& r : e->prof) if (r.cls == cls) by += r.bytes;
    *total_bytes = by;
    return RT_OK;
}

// ---- operator-level entry points (stateless; share one lazily allocated zero page per device) ----
static bf16_t* op_zero_page() {       // one page per DEVICE (a thread that drives two GPUs must not hand device-0 memory to device-1 kernels)
    static std::mutex mu;
    static std::map<int, bf16_t*> pages;
    int dev = 0; HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    bf16_t*& z = pages[dev];
    if (!z) { HIP_CHECK(hipMalloc((void**)&z, 256)); HIP_CHECK(hipMemset(z, 0, 256)); }
    return z;
}
// statistics scratch of rt_op_attention_probs_avg: one buffer per (device, stream), grown on demand (never under capture: null then,
// which selects the one-pass kernel)
static float* op_store_stats(size_t floats, hipStream_t st) {
    struct Buf { float* p = nullptr; size_t n = 0; };
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, Buf> bufs;
    int dev = 0; HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    Buf& b = bufs[std::make_pair(dev, st)];
    if (floats > b.n) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(st, &cap);
        if (cap != hipStreamCaptureStatusNone) return nullptr;
        HIP_CHECK(hipStreamSynchronize(st));
        if (b.p) (void)hipFree(b.p);
        b.p = nullptr; b.n = 0;
        HIP_CHECK(hipMalloc((void**)&b.p, floats * 4));
        b.n = floats;
    }
    return b.p;
}
#define OP_TRY(...)                                                          \
    try { __VA_ARGS__; return RT_OK; }                                              \
    catch (const rt_error& ex) { g_op_error = ex.what(); return ex.code; }   \
    catch (const std::exception& ex) { g_op_error = ex.what(); return RT_E_INVALID; }

const char* rt_op_last_error(void) { return g_op_error.c_str(); }
// the stream ranges of the intra-image split (step_driver.inl) as a host-only query
int rt_op_split_range(int n_streams, int text_ref_stream, int inject, int part, int nparts, int* first, int* count) {
    OP_TRY({ RT_REQUIRE(first && count, "rt_op_split_range: null outputs"); region_split_range(n_streams, text_ref_stream, inject != 0, part, nparts, first, count); })
}
extern int g_store_legacy;
extern int g_store_apply_v1;
extern int g_c77_t1;
// host-only (no GPU): which streams of a self-attention launch share one softmax (csrc/attention.hip, launch_attention_units)
int rt_op_attention_units_plan(const int* q_src, const int* k_src, int B, int tokens, int DP, int mode, int* launch_of, int* unit_of, int* members_of) {
    if (!q_src || !k_src || !launch_of || !unit_of || !members_of || B < 1 || B > RT_MAXB) return -1;
    return attention_units_plan_host(q_src, k_src, B, tokens, DP, mode, launch_of, unit_of, members_of);
}
int rt_op_gemm_debug(int d) { gemm_set_debug(d); g_store_own_stats = (d >> 17) & 1; g_store_apply_v1 = (d >> 18) & 1; g_c77_t1 = (d >> 20) & 3; g_store_legacy = ((d & 32) ? 1 : 0) | ((d & 64) ? 2 : 0); attention_set_prio(((d >> 14) & 1) ^ 1); attention_set_units((d >> 24) & 7); gemm16_set_tall((d >> 27) & 1); groupnorm_set_fused(((d >> 23) & 1) ^ 1); groupnorm_set_chunk_div(((d >> 11) & 1) ? 128 : 64); gemm16_set_deep(((d >> 12) & 1) ^ 1); return RT_OK; }
int rt_op_probes_built(void) {
#ifdef RT_PROBES
    return 1;
#else
    return 0;
#endif
}
int rt_op_gemm_force_config(int cfg) {
    if (cfg < -1 || cfg > 8) return RT_E_INVALID;
    gemm_force_config(cfg);
    return RT_OK;
}

int rt_op_gemm(const void* A, const void* W, const float* bias, void* out, const void* res, const float* temb, int mode,
               int epi, int M, int N, int K, int lda, int ldw, int ldo, int ldres, int temb_ld, int rows_per_batch, int Hin,
               int Win, int Cin, int Hout, int Wout, void* stream) {
    OP_TRY({
        GemmArgs g{}; g.A = (const bf16_t*)A; g.W = (const bf16_t*)W; g.bias = bias; g.out = out; g.res = res; g.temb = temb;
        g.zero = op_zero_page(); g.mode = mode; g.epi = epi; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw; g.ldo = ldo;
        g.ldres = ldres; g.temb_ld = temb_ld; g.rows_per_batch = rows_per_batch; g.Hin = Hin; g.Win = Win; g.Cin = Cin;
        g.Hout = Hout; g.Wout = Wout;
        launch_gemm(g, (hipStream_t)stream);
    })
}
int rt_op_ln_partials(const void* x_f16, void* xb_bf16, float* partials, int rows, int C, int tile_cols, void* stream) {
    OP_TRY({ launch_ln_partials((const f16_t*)x_f16, (bf16_t*)xb_bf16, partials, rows, C, tile_cols, (hipStream_t)stream); })
}
int rt_op_gemm_emit_partials(const void* A, const void* W, const float* bias, void* out_f16, const void* res_f16, int M, int N, int K,
                             int rows_per_stream, void* xb_bf16, float* partials, int* tile_cols, void* stream) {
    OP_TRY({
        GemmArgs g{}; g.A = (const bf16_t*)A; g.W = (const bf16_t*)W; g.bias = bias; g.out = out_f16; g.res = res_f16; g.zero = op_zero_page();
        g.mode = A_DENSE; g.epi = EPI_F16; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldo = N; g.ldres = N;
        g.rows_per_stream = rows_per_stream; if (rows_per_stream > 0) g.split_tiles = cdiv(rows_per_stream, 128) * cdiv(N, 128);
        const int bn = gemm_ln_emit_bn(g);
        if (!bn) throw rt_error(RT_E_UNSUPPORTED, "rt_op_gemm_emit_partials: this shape's route has no partial-emitting epilogue");
        if (tile_cols) *tile_cols = bn;
        g.ln_emit = partials; g.ln_copy = (bf16_t*)xb_bf16;
        launch_gemm(g, (hipStream_t)stream);
    })
}
int rt_op_ln_gemm(const void* x_f16, const float* gamma, const float* beta, const void* W_bf16, const float* bias, void* out, int tokens,
                  int N, int C, int epi, int weights_on_rows, int rows_per_stream, const void* xb_bf16, const float* partials, int tile_cols, void* stream) {
    OP_TRY({
        RT_REQUIRE((C == 640 || C == 1280) && tokens > 0 && N > 0 && (epi == EPI_BF16 || epi == EPI_GEGLU) && !(weights_on_rows && epi != EPI_BF16), "rt_op_ln_gemm: shape");
        RT_REQUIRE((xb_bf16 != nullptr) == (partials != nullptr) && (xb_bf16 || x_f16), "rt_op_ln_gemm: a producer's (xb, partials) pair, or the fp16 trunk");
        RT_REQUIRE((tile_cols == 160 || tile_cols == 320) && C % (2 * tile_cols) == 0, "rt_op_ln_gemm: column tiles of 160 / 320, an even number per row");
        hipStream_t st = (hipStream_t)stream;
        GemmArgs g{}; g.zero = op_zero_page(); g.mode = A_DENSE; g.epi = epi; g.K = C; g.out = out;
        if (weights_on_rows) { g.M = N; g.N = tokens; g.lda = C; g.ldw = C; g.ldo = tokens; g.weights_on_rows = 1; }
        else { g.M = tokens; g.N = N; g.lda = C; g.ldw = C; g.ldo = epi == EPI_GEGLU ? N / 2 : N; }
        g.rows_per_stream = rows_per_stream;
        if (rows_per_stream > 0) g.split_tiles = weights_on_rows ? cdiv(N, 128) * cdiv(rows_per_stream, 128) : cdiv(rows_per_stream, 128) * cdiv(N, 128);
        g.ln_npair = C / tile_cols / 2; g.ln_ld = tokens; g.ln_inv_c = 1.f / (float)C; g.ln_eps = 1e-5f;
        if (!gemm_ln_fold_ok(g)) throw rt_error(RT_E_UNSUPPORTED, "rt_op_ln_gemm: this shape's route has no folded instantiation");
        // scratch of the op (the engine keeps these in its own arenas): W', (s, c), xb, partials
        bf16_t* Wf = nullptr; float* sc = nullptr; float* part = nullptr; bf16_t* xb = nullptr;
        HIP_CHECK(hipMalloc((void**)&Wf, (size_t)N * C * 2));
        HIP_CHECK(hipMalloc((void**)&sc, (size_t)N * 8));
        try {
            launch_ln_fold_derive((const bf16_t*)W_bf16, C, bias, gamma, beta, N, C, Wf, sc, st);
            if (!partials) {
                HIP_CHECK(hipMalloc((void**)&part, (size_t)tokens * 16 * 4));
                HIP_CHECK(hipMalloc((void**)&xb, (size_t)tokens * C * 2));
                launch_ln_partials((const f16_t*)x_f16, xb, part, tokens, C, tile_cols, st);
            }
            const bf16_t* xop = partials ? (const bf16_t*)xb_bf16 : xb;
            g.ln_part = partials ? partials : part; g.ln_s = sc; g.bias = nullptr;
            if (weights_on_rows) { g.A = Wf; g.W = xop; } else { g.A = xop; g.W = Wf; }
            launch_gemm(g, st);
            HIP_CHECK(hipStreamSynchronize(st));
        } catch (...) { (void)hipFree(Wf); (void)hipFree(sc); (void)hipFree(part); (void)hipFree(xb); throw; }
        (void)hipFree(Wf); (void)hipFree(sc); (void)hipFree(part); (void)hipFree(xb);
    })
}
int rt_op_attention(const void* Q, int ldq, const void* K, int ldk, const void* VT, int ldvt, void* O, int ldo, const int* q_src,
                    const int* k_src, const int* v_src, const int* wset, const float* wabs, const float* wsgn, int B, int H,
                    int N, int NK, int nk_valid, int DP, int cross, void* stream) {
    OP_TRY({
        RT_REQUIRE(B >= 1 && B <= RT_MAXB, "rt_op_attention: batch");
        AttnArgs a{}; a.Q = (const bf16_t*)Q; a.ldq = ldq; a.K = (const bf16_t*)K; a.ldk = ldk; a.VT = (const bf16_t*)VT; a.ldvt = ldvt;
        a.O = (bf16_t*)O; a.ldo = ldo; a.wabs = wabs; a.wsgn = wsgn;
        for (int b = 0; b < B; ++b) { a.q_src[b] = q_src ? q_src[b] : b; a.k_src[b] = k_src ? k_src[b] : b; a.v_src[b] = v_src ? v_src[b] : b; a.wset[b] = wset ? wset[b] : 0; }
        a.B = B; a.H = H; a.N = N; a.NK = NK; a.nk_valid = nk_valid; a.DP = DP; a.cross = cross;
        bool same_kv = true;
        for (int b = 0; b < B; ++b) same_kv = same_kv && a.k_src[b] == a.v_src[b];
        if (cross && same_kv && gemm_cross77_enabled() && cross77_supported(H, DP, N, NK, nk_valid)) launch_cross77(a, (hipStream_t)stream);     // the engine's kernel for these shapes
        else launch_attention(a, (hipStream_t)stream);
    })
}
// The shape rule of csrc/gemm16.hip as a host-only query (no device needed): which tile variant a problem takes (-1: stays on gemm.hip /
// the patch convolution) and whether it uses the W-stationary tile -> XCD order.  `streams` images / streams of rows_per_stream rows each.
int rt_op_split_plan(int conv3x3, int epi, int streams, int rows_per_stream, int N, int K_or_Cin, int* route, int* slices) {
    try {
        GemmArgs g{};
        g.epi = epi; g.N = N;
        if (conv3x3) {
            int side = 1; while (side * side < rows_per_stream) ++side;
            if (side * side != rows_per_stream) return RT_E_INVALID;                // square maps only in this query
            g.mode = conv3x3 == 3 ? A_CONV3_UP2 : A_CONV3; g.Cin = K_or_Cin; g.K = 9 * K_or_Cin; g.ldw = g.K; g.ldo = N;
            g.Hout = g.Wout = side; g.Hin = g.Win = conv3x3 == 3 ? side / 2 : side; g.rows_per_batch = rows_per_stream; g.M = streams * rows_per_stream;
        } else {
            g.mode = A_DENSE; g.K = K_or_Cin; g.lda = g.K; g.ldw = g.K; g.rows_per_stream = rows_per_stream; g.M = streams * rows_per_stream;
        }
        g.split_tiles = cdiv(rows_per_stream, 128) * cdiv(N, 128);                  // what the engine passes (engine.hip: conv3 / gemm)
        int r = 0, sl = 1;
        gemm_split_plan(g, &r, &sl);
        if (route) *route = r;
        if (slices) *slices = sl;
        return RT_OK;
    } catch (...) { return RT_E_INVALID; }
}
int rt_op_gemm16_pick(int conv3x3, int epi, int streams, int rows_per_stream, int N, int K_or_Cin, int weights_on_rows, int* w_stationary) {
    try {
        GemmArgs g{};
        g.epi = epi; g.N = N; g.weights_on_rows = weights_on_rows;
        if (conv3x3) {
            int side = 1; while (side * side < rows_per_stream) ++side;
            if (side * side != rows_per_stream) return -2;                      // square maps only in this query
            g.mode = A_CONV3; g.Cin = K_or_Cin; g.K = 9 * K_or_Cin; g.ldw = g.K; g.ldo = N;
            g.Hin = g.Hout = g.Win = g.Wout = side; g.rows_per_batch = rows_per_stream; g.M = streams * rows_per_stream;
        } else {
            g.mode = A_DENSE; g.K = K_or_Cin; g.lda = g.K; g.ldw = g.K; g.rows_per_stream = rows_per_stream;
            if (weights_on_rows) { g.M = N; g.N = streams * rows_per_stream; } else g.M = streams * rows_per_stream;
        }
        int ws = 0;
        const int v = gemm16_pick(g, weights_on_rows, &ws);
        if (w_stationary) *w_stationary = ws;
        return v;
    } catch (...) { return -3; }
}
// One tile variant of the 16x16x32 family (csrc/gemm16.hip) on a dense problem - tests and micro-benchmarks; rt_op_gemm picks by shape.
int rt_op_gemm16_variant(const void* A, const void* W, const float* bias, void* out, const void* res, int epi, int M, int N, int K, int lda,
                         int ldw, int ldo, int ldres, int weights_on_rows, int variant, int wstat, void* stream) {
    OP_TRY({
        GemmArgs g{}; g.A = (const bf16_t*)A; g.W = (const bf16_t*)W; g.bias = bias; g.out = out; g.res = res; g.zero = op_zero_page();
        g.mode = A_DENSE; g.epi = epi; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw; g.ldo = ldo; g.ldres = ldres; g.weights_on_rows = weights_on_rows;
        if (variant < 0) { int ws = 0; variant = gemm16_pick(g, weights_on_rows, &ws); wstat = ws; RT_REQUIRE(variant >= 0, "rt_op_gemm16_variant: the family has no tile for this shape"); }
        launch_gemm16_variant(g, variant, wstat, (hipStream_t)stream);
    })
}
// Host-only query of the grouped-launch rule (tests): which grouped instantiation attn1's two projections take for `streams_qk` / `streams`
// streams of `rows_per_stream` tokens - 0: 224x320 + 160x224, 1: 224x256 + 160x224, 2: 128x256 + 160x64, 3: 224x256 + 160x128 (Q|K tile +
// V^T tile) - or -1: two launches.
int rt_op_gemm_pair_pick(int streams_qk, int streams, int rows_per_stream, int Nqk, int Nv, int K) {
    try {
        GemmArgs a{}; a.mode = A_DENSE; a.epi = EPI_BF16; a.M = streams_qk * rows_per_stream; a.N = Nqk; a.K = K; a.lda = K; a.ldw = K; a.ldo = Nqk;
        GemmArgs b{}; b.mode = A_DENSE; b.epi = EPI_BF16; b.M = Nv; b.N = streams * rows_per_stream; b.K = K; b.lda = K; b.ldw = K; b.ldo = b.N; b.weights_on_rows = 1;
        a.rows_per_stream = b.rows_per_stream = rows_per_stream;
        a.split_tiles = cdiv(rows_per_stream, 128) * cdiv(Nqk, 128); b.split_tiles = cdiv(Nv, 128) * cdiv(rows_per_stream, 128);
        return gemm_pair_is_grouped(a, b) ? gemm16_pair_variant(a, b) : -1;
    } catch (...) { return -3; }
}
// attn1's two projections of one LayerNorm output X [M, K] as the engine launches them (launch_gemm_pair): qk[Mqk, Nqk] = X[:Mqk] Wqk^T + bqk
// and vt[Nv, M] = Wv X^T.  *grouped = 1 when they went out as ONE grouped launch (gemm16_dual_kernel), 0 when as two launches.
int rt_op_gemm_qk_vt(const void* X, int ldx, int K, int rows_per_stream, const void* Wqk, const float* bqk, int Mqk, int Nqk, void* qk, int ldqk,
                     const void* Wv, int Nv, int M, void* vt, int ldvt, int* grouped, void* stream) {
    OP_TRY({
        GemmArgs a{}; a.A = (const bf16_t*)X; a.W = (const bf16_t*)Wqk; a.bias = bqk; a.out = qk; a.zero = op_zero_page();
        a.mode = A_DENSE; a.epi = EPI_BF16; a.M = Mqk; a.N = Nqk; a.K = K; a.lda = ldx; a.ldw = K; a.ldo = ldqk; a.temb_ld = Nqk;
        GemmArgs b{}; b.A = (const bf16_t*)Wv; b.W = (const bf16_t*)X; b.out = vt; b.zero = a.zero;
        b.mode = A_DENSE; b.epi = EPI_BF16; b.M = Nv; b.N = M; b.K = K; b.lda = K; b.ldw = ldx; b.ldo = ldvt; b.weights_on_rows = 1;
        if (rows_per_stream > 0) {
            a.rows_per_stream = b.rows_per_stream = rows_per_stream;
            a.split_tiles = cdiv(rows_per_stream, 128) * cdiv(Nqk, 128); b.split_tiles = cdiv(Nv, 128) * cdiv(rows_per_stream, 128);
        }
        if (grouped) *grouped = gemm_pair_is_grouped(a, b) ? 1 : 0;
        launch_gemm_pair(a, b, (hipStream_t)stream);
    })
}
// The block BASELINE.json's north star names: attn2 of a BasicTransformerBlock (models/attention.py:169-189) with the reference
// processor's arithmetic (models/attention_processor.py:476-545; font-size softmax :386-401) as ONE call:
//   trunk_out = trunk_in + to_out( softmax_fs( to_q(x) K[prompt]^T ) V[prompt] ) + b_out
// K / V^T come from the per-prompt cache (77 keys padded to 96; step invariant), Q and O live in caller-provided scratch.  Three
// launches (to_q GEMM, 96-key attention, to_out GEMM + fp16 residual): why the fused form is not faster is measured in LABNOTES.md R4.1 / 4.8.
int rt_op_cross_attn_block(const void* x, const void* wq, const void* wo, const float* bo, const void* kcache, const void* vtcache, int ldvt,
                           const int* prompt_host, const int* wset_host, const float* wabs, const float* wsgn, const void* trunk_in,
                           void* trunk_out, void* q_scratch, void* o_scratch, int B, int N, int C, int H, int DP, void* stream) {
    OP_TRY({
        RT_REQUIRE(B >= 1 && B <= RT_MAXB && N > 0 && C % 8 == 0 && H > 0 && DP % 32 == 0, "rt_op_cross_attn_block: shape");
        hipStream_t st = (hipStream_t)stream;
        const int M = B * N, HD = H * DP;
        if (gemm_xblock_enabled() && xblock_supported(C, H, DP, N)) {
            // the engine's path for the 640-channel level: the whole block in one launch (neither scratch buffer is written)
            XBlockArgs xa{}; xa.x = (const bf16_t*)x; xa.wq = (const bf16_t*)wq; xa.wo = (const bf16_t*)wo; xa.bo = bo; xa.kc = (const bf16_t*)kcache;
            xa.vt = (const bf16_t*)vtcache; xa.res = (const f16_t*)trunk_in; xa.out = (f16_t*)trunk_out; xa.wabs = wabs; xa.wsgn = wsgn;
            xa.ldk = HD; xa.ldvt = ldvt; xa.ldres = C; xa.ldo = C; xa.M = M; xa.tokens = N; xa.nk_valid = 77; xa.C = C; xa.H = H;
            for (int b = 0; b < B; ++b) { xa.prompt[b] = prompt_host ? prompt_host[b] : 0; xa.wset[b] = wset_host ? wset_host[b] : 0; }
            launch_xblock(xa, st);
            return RT_OK;
        }
        const bool c77 = gemm_cross77_enabled() && cross77_supported(H, DP, N, 96, 77);
        if (!c77 && gemm_xattn_enabled() && xattn_fused_supported(C, H, DP, N)) {
            // round 4's path for these shapes: to_q + attention in one launch (Q stays in LDS; q_scratch is not written)
            GemmArgs g{}; g.A = (const bf16_t*)x; g.W = (const bf16_t*)wq; g.out = o_scratch; g.zero = op_zero_page(); g.mode = A_DENSE; g.epi = EPI_XATTN;
            g.M = M; g.N = HD; g.K = C; g.lda = C; g.ldw = C; g.ldo = HD; g.rows_per_stream = N;
            g.xa_k = (const bf16_t*)kcache; g.xa_vt = (const bf16_t*)vtcache; g.xa_ldk = HD; g.xa_ldvt = ldvt; g.xa_tokens = N; g.xa_nk_valid = 77;
            g.xa_wabs = wabs; g.xa_wsgn = wsgn;
            for (int b = 0; b < B; ++b) { g.xa_prompt[b] = prompt_host ? prompt_host[b] : 0; g.xa_wset[b] = wset_host ? wset_host[b] : 0; }
            launch_xattn_fused(g, st);
        } else {
        GemmArgs g{}; g.A = (const bf16_t*)x; g.W = (const bf16_t*)wq; g.out = q_scratch; g.zero = op_zero_page(); g.mode = A_DENSE; g.epi = EPI_BF16;
        g.M = M; g.N = HD; g.K = C; g.lda = C; g.ldw = C; g.ldo = HD; g.rows_per_stream = N; g.split_tiles = cdiv(N, 128) * cdiv(HD, 128);
        launch_gemm(g, st);
        AttnArgs a{}; a.Q = (const bf16_t*)q_scratch; a.ldq = HD; a.K = (const bf16_t*)kcache; a.ldk = HD; a.VT = (const bf16_t*)vtcache; a.ldvt = ldvt;
        a.O = (bf16_t*)o_scratch; a.ldo = HD; a.wabs = wabs; a.wsgn = wsgn;
        for (int b = 0; b < B; ++b) { a.q_src[b] = b; a.k_src[b] = prompt_host ? prompt_host[b] : 0; a.v_src[b] = a.k_src[b]; a.wset[b] = wset_host ? wset_host[b] : 0; }
        a.B = B; a.H = H; a.N = N; a.NK = 96; a.nk_valid = 77; a.DP = DP; a.cross = 1;
        if (c77) launch_cross77(a, st); else launch_attention(a, st);
        }
        GemmArgs o{}; o.A = (const bf16_t*)o_scratch; o.W = (const bf16_t*)wo; o.bias = bo; o.out = trunk_out; o.res = trunk_in; o.zero = op_zero_page();
        o.mode = A_DENSE; o.epi = EPI_F16; o.M = M; o.N = C; o.K = HD; o.lda = HD; o.ldw = HD; o.ldo = C; o.ldres = C; o.rows_per_stream = N;
        o.split_tiles = cdiv(N, 128) * cdiv(C, 128);
        launch_gemm(o, st);
    })
}
int rt_op_groupnorm(const void* x1, const void* x2, int in_bf16, int C1, int C2, int G, int B, int HW, const float* gamma,
                    const float* beta, float eps, int silu, void* out, void* raw, void* stream) {
    OP_TRY({
        GroupNormArgs a{}; a.x1 = x1; a.x2 = x2; a.in_bf16 = in_bf16; a.C1 = C1; a.C2 = C2; a.G = G; a.B = B; a.HW = HW;
        a.gamma = gamma; a.beta = beta; a.eps = eps; a.silu = silu; a.out = (bf16_t*)out; a.raw_out = (bf16_t*)raw;
        a.nchunk = groupnorm_nchunk(HW); a.rows_per_chunk = groupnorm_rows_per_chunk(HW);
        float* partial = nullptr;
        HIP_CHECK(hipMalloc((void**)&partial, (size_t)B * a.nchunk * G * 2 * 4));
        a.partial = partial; a.fuse_finalize = 1;               // the path the UNet forward takes
        launch_groupnorm(a, (hipStream_t)stream);
        HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
        HIP_CHECK(hipFree(partial));
    })
}
int rt_op_layernorm(const float* x, const float* gamma, const float* beta, void* out, int rows, int C, float eps, void* stream) {
    OP_TRY({ launch_layernorm(x, 0, gamma, beta, (bf16_t*)out, rows, C, eps, (hipStream_t)stream); })
}
int rt_op_layernorm_f16(const void* x_f16, const float* gamma, const float* beta, void* out, int rows, int C, float eps, void* stream) {
    OP_TRY({ launch_layernorm(x_f16, 1, gamma, beta, (bf16_t*)out, rows, C, eps, (hipStream_t)stream); })
}
int rt_op_small_linear(const float* a, int lda, const void* W, int ldw, const float* bias, float* out, int ldo, int B, int N,
                       int K, int silu_in, int accumulate, void* stream) {
    OP_TRY({ launch_small_linear(a, lda, (const bf16_t*)W, ldw, bias, out, ldo, B, N, K, silu_in, accumulate, (hipStream_t)stream); })
}
int rt_op_timestep_embed(const float* t, int n, int dim, float* out, int ldo, void* stream) {
    OP_TRY({ launch_timestep_embed(t, n, dim, out, ldo, (hipStream_t)stream); })
}
int rt_op_embed(const int* ids, const float* tok, const float* pos, float* out, int rows, int N, int C, int vocab, void* stream) {
    OP_TRY({ launch_embed(ids, tok, pos, out, rows, N, C, vocab, (hipStream_t)stream); })
}
int rt_op_activation(const void* x_bf16, void* out_bf16, long long n, int kind, void* stream) {
    OP_TRY({ launch_activation((const bf16_t*)x_bf16, (bf16_t*)out_bf16, (size_t)n, kind, (hipStream_t)stream); })
}
int rt_op_causal_attention(const void* q, const void* k, const void* v, int ld, void* out, int ldo, int B, int H, int N, int d, float scale,
                           void* stream) {
    OP_TRY({ launch_causal_attention((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ld, (bf16_t*)out, ldo, B, H, N, d, scale, (hipStream_t)stream); })
}
int rt_op_cast_bf16(const float* x, void* out_bf16, long long n, void* stream) {
    OP_TRY({ RT_REQUIRE(n > 0, "cast: empty"); launch_cast_f32_bf16(x, (bf16_t*)out_bf16, (size_t)n, (hipStream_t)stream); })
}
int rt_op_attention_probs_avg(const void* Q, int ldq, long long q_row0, const void* K, int ldk, long long k_row0, float* out,
                              int H, int N, int NK, int NKpad, int NKrows, int DP, int accumulate, void* stream) {
    OP_TRY({
        AttnStoreArgs a{}; a.Q = (const bf16_t*)Q; a.ldq = ldq; a.q_row0 = (long)q_row0; a.K = (const bf16_t*)K; a.ldk = ldk; a.k_row0 = (long)k_row0;
        a.out = out; a.H = H; a.N = N; a.NK = NK; a.NKpad = NKpad; a.NKrows = NKrows; a.DP = DP; a.overwrite = accumulate ? 0 : 1;
        a.stats = op_store_stats((size_t)H * N * 2, (hipStream_t)stream);
        launch_attn_store(a, (hipStream_t)stream);
    })
}

}  // extern "C"
