// VAE decoder forward + input-gradient (backward-data only) for the colour-guidance step, and plain decode.
//
// Replaces `self.vae.decode(...)` + `loss_total.backward()` of the reference's guidance block
// (models/region_diffusion.py:151-168, models/region_diffusion_sdxl.py:849-867) and the final decode
// (rd.py:227-236, xl.py:916-944).  AutoencoderKL itself is third-party (diffusers 0.18.2, not on disk): the graph
// below follows the oracle restatement oracle/vae.py (parity unpinned against diffusers, pinned against the oracle
// incl. its torch-autograd gradient).  Only d(loss)/d(latents) is needed: no weight gradients, and the UNet is not
// differentiated (noise_pred is a constant in predict_x0).
// All contractions (3x3 convs, their backward-data = 3x3 conv with flipped/transposed weights, linears, attention
// score/PV GEMMs and their adjoints) run on the MFMA GEMMs of gemm16.hip / gemm.hip with bf16 operands and fp32 accumulation.
//
// PRECISE mode (rt_vae_config.precise, round 3): the SDXL pipeline of the reference decodes this VAE in fp32
// (models/region_diffusion_sdxl.py:856 `.to(dtype=torch.float32)`).  gfx950 has no fast fp32 matrix path (the f32-input MFMA runs
// at 1/16 of the bf16 rate), so fp32-class products are built from bf16 MFMAs: every operand is kept as a PAIR
// (hi = bf16(v), lo = bf16(v - hi), together 16 mantissa bits) and every contraction is three passes, hi*hi + lo*hi + hi*lo,
// accumulated in fp32 in the same output (the lo*lo term is 2^-18 relative and dropped).  Everything between contractions
// (GroupNorm, SiLU, softmax, residual adds, the loss) already runs in fp32.  Three times the MFMA work of the default mode.
#include "common.h"
#include "../../include/rtdiff.h"
#include "vae.h"
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {
struct Arena2 {
    char* base = nullptr; size_t off = 0, cap = 0;
    void* alloc(size_t bytes) {
        off = (off + 255) & ~(size_t)255;
        void* p = base ? base + off : (void*)(uintptr_t)(off + 256);
        off += bytes;
        if (base && off > cap) throw rt_error(RT_E_STATE, "vae: arena overflow");
        return p;
    }
};
struct NormW { float* g = nullptr; float* b = nullptr; int C = 0; };
struct MatW { bf16_t* w = nullptr; bf16_t* w_lo = nullptr; float* b = nullptr; int N = 0, K = 0; };   // w_lo: precise mode only
struct BT { bf16_t* hi = nullptr; bf16_t* lo = nullptr; };                                                // a bf16 operand (+ its low part)
struct VConv { MatW f, b; int cin = 0, cout = 0, cinP = 0, coutP = 0; };
struct VLin { MatW f, b; };
struct VRes { std::string name; NormW n1, n2; VConv c1, c2; bool has_sc = false; VLin sc; int cin = 0, cout = 0; };
struct VAttn { NormW gn; VLin q, k, v, o; int C = 0; };
struct Slot { std::string name; std::vector<int64_t> shape; std::vector<PackArgs> packs; bool bound = false; };

struct ResSaved { const float* x; float* part1; const void* h2; bool h2_bf16; float* part2; int H, W; };
struct AttnSaved { const float* x; float* part; BT q, k, v, P; int N; };
}  // namespace

struct rt_vae {
    rt_vae_config cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    Arena2 arena;
    char* arena_base = nullptr; size_t arena_bytes = 0;
    char* ws_base = nullptr; size_t ws_cap = 0, ws_off = 0, ws_peak = 0;
    bool dry = false;
    bool precise = false;
    bf16_t* zero = nullptr;
    std::vector<Slot> slots;
    std::map<std::string, int> slot_index;
    // plan
    float* pq_w = nullptr; float* pq_b = nullptr;
    VConv conv_in, conv_out;
    VRes mid0, mid1;
    VAttn attn;
    std::vector<std::vector<VRes>> up_res;
    std::vector<VConv> up_conv;      // size n-1
    NormW norm_out;
    int G = 32;

    // ---------------------------------------------------------------- memory
    void* walloc(size_t bytes) {
        ws_off = (ws_off + 255) & ~(size_t)255;
        void* p = ws_base ? ws_base + ws_off : (void*)(uintptr_t)(ws_off + 256);
        ws_off += bytes;
        if (ws_off > ws_peak) ws_peak = ws_off;
        if (!dry && ws_off > ws_cap) throw rt_error(RT_E_STATE, "vae: workspace overflow");
        return p;
    }
    float* f32(size_t n) { return (float*)walloc(n * 4); }
    bf16_t* b16(size_t n) { return (bf16_t*)walloc(n * 2); }
    BT bt(size_t n) { BT t; t.hi = b16(n); t.lo = precise ? b16(n) : nullptr; return t; }

    // ---------------------------------------------------------------- plan
    void add_slot(const std::string& name, std::vector<int64_t> shape, std::vector<PackArgs> packs) {
        Slot s; s.name = name; s.shape = std::move(shape); s.packs = std::move(packs);
        slot_index[name] = (int)slots.size(); slots.push_back(s);
    }
    static PackArgs pk_vec(void* dst, int n) {
        PackArgs p{}; p.dst = dst; p.dst_f32 = 1; p.rows = n; p.cols = 1; p.ld_dst = 1; p.row_map = PACK_ROWS_ID; p.c_inner = 1; p.ci_valid = 1;
        p.s_r = 1; p.scale = 1.f; return p;
    }
    NormW mk_norm(const std::string& n, int C) {
        NormW w; w.C = C; w.g = (float*)arena.alloc((size_t)C * 4); w.b = (float*)arena.alloc((size_t)C * 4);
        add_slot(n + ".weight", {C}, {pk_vec(w.g, C)}); add_slot(n + ".bias", {C}, {pk_vec(w.b, C)});
        return w;
    }
    // 3x3 conv: forward [CoutN][tap][CinP] and backward-data [CinN][flipped tap][CoutP]
    VConv mk_conv3(const std::string& n, int Cin, int Cout) {
        VConv c; c.cin = Cin; c.cout = Cout; c.cinP = (Cin + 7) & ~7; c.coutP = (Cout + 7) & ~7;
        const int CoutN = (Cout + 3) & ~3, CinN = (Cin + 3) & ~3;      // GEMM N must be a multiple of 4
        c.f.N = CoutN; c.f.K = 9 * c.cinP; c.f.w = (bf16_t*)arena.alloc((size_t)CoutN * c.f.K * 2);
        c.f.b = (float*)arena.alloc((size_t)CoutN * 4);
        c.b.N = CinN; c.b.K = 9 * c.coutP; c.b.w = (bf16_t*)arena.alloc((size_t)CinN * c.b.K * 2);
        PackArgs pf{}; pf.dst = c.f.w; pf.rows = Cout; pf.cols = c.f.K; pf.ld_dst = c.f.K; pf.row_map = PACK_ROWS_ID; pf.c_inner = c.cinP;
        pf.ci_valid = Cin; pf.s_r = (long)Cin * 9; pf.s_co = 1; pf.s_ci = 9; pf.scale = 1.f;
        PackArgs pb{}; pb.dst = c.b.w; pb.rows = Cin; pb.cols = c.b.K; pb.ld_dst = c.b.K; pb.row_map = PACK_ROWS_ID; pb.c_inner = c.coutP;
        pb.ci_valid = Cout; pb.s_r = 9; pb.s_co = -1; pb.s_ci = (long)Cin * 9; pb.s_base = 8; pb.scale = 1.f;
        std::vector<PackArgs> packs{pf, pb};
        if (precise) {
            c.f.w_lo = (bf16_t*)arena.alloc((size_t)CoutN * c.f.K * 2); c.b.w_lo = (bf16_t*)arena.alloc((size_t)CinN * c.b.K * 2);
            PackArgs lf = pf; lf.dst = c.f.w_lo; lf.lo_part = 1; PackArgs lb = pb; lb.dst = c.b.w_lo; lb.lo_part = 1;
            packs.push_back(lf); packs.push_back(lb);
        }
        add_slot(n + ".weight", {Cout, Cin, 3, 3}, packs);
        add_slot(n + ".bias", {Cout}, {pk_vec(c.f.b, Cout)});
        return c;
    }
    // Linear / 1x1 conv [N, K]: forward as is, backward-data transposed [K, N]
    VLin mk_lin(const std::string& n, int K, int N, bool conv1x1, bool bias = true) {
        VLin l; l.f.N = N; l.f.K = K; l.f.w = (bf16_t*)arena.alloc((size_t)N * K * 2);
        l.b.N = K; l.b.K = N; l.b.w = (bf16_t*)arena.alloc((size_t)N * K * 2);
        PackArgs pf{}; pf.dst = l.f.w; pf.rows = N; pf.cols = K; pf.ld_dst = K; pf.row_map = PACK_ROWS_ID; pf.c_inner = K; pf.ci_valid = K;
        pf.s_r = K; pf.s_ci = 1; pf.scale = 1.f;
        PackArgs pb{}; pb.dst = l.b.w; pb.rows = K; pb.cols = N; pb.ld_dst = N; pb.row_map = PACK_ROWS_ID; pb.c_inner = N; pb.ci_valid = N;
        pb.s_r = 1; pb.s_ci = K; pb.scale = 1.f;
        std::vector<int64_t> shp = conv1x1 ? std::vector<int64_t>{N, K, 1, 1} : std::vector<int64_t>{N, K};
        std::vector<PackArgs> packs{pf, pb};
        if (precise) {
            l.f.w_lo = (bf16_t*)arena.alloc((size_t)N * K * 2); l.b.w_lo = (bf16_t*)arena.alloc((size_t)N * K * 2);
            PackArgs lf = pf; lf.dst = l.f.w_lo; lf.lo_part = 1; PackArgs lb = pb; lb.dst = l.b.w_lo; lb.lo_part = 1;
            packs.push_back(lf); packs.push_back(lb);
        }
        add_slot(n + ".weight", shp, packs);
        if (bias) { l.f.b = (float*)arena.alloc((size_t)N * 4); add_slot(n + ".bias", {N}, {pk_vec(l.f.b, N)}); }
        return l;
    }
    VRes mk_res(const std::string& n, int cin, int cout) {
        RT_REQUIRE(cin % 8 == 0 && cout % 8 == 0, "vae: channel counts must be multiples of 8");
        VRes r; r.name = n; r.cin = cin; r.cout = cout;
        r.n1 = mk_norm(n + ".norm1", cin); r.c1 = mk_conv3(n + ".conv1", cin, cout);
        r.n2 = mk_norm(n + ".norm2", cout); r.c2 = mk_conv3(n + ".conv2", cout, cout);
        r.has_sc = cin != cout;
        if (r.has_sc) r.sc = mk_lin(n + ".conv_shortcut", cin, cout, true);
        return r;
    }
    void build_plan() {
        slots.clear(); slot_index.clear(); up_res.clear(); up_conv.clear();
        G = cfg.norm_groups;
        precise = cfg.precise != 0;
        const int n = cfg.n_blocks;
        zero = (bf16_t*)arena.alloc(256);
        pq_w = (float*)arena.alloc(16 * 4); pq_b = (float*)arena.alloc(4 * 4);
        { PackArgs p{}; p.dst = pq_w; p.dst_f32 = 1; p.rows = 4; p.cols = 4; p.ld_dst = 4; p.row_map = PACK_ROWS_ID; p.c_inner = 4; p.ci_valid = 4;
          p.s_r = 4; p.s_ci = 1; p.scale = 1.f; add_slot("post_quant_conv.weight", {4, 4, 1, 1}, {p}); }
        add_slot("post_quant_conv.bias", {4}, {pk_vec(pq_b, 4)});
        const int top = cfg.block_out_channels[n - 1];
        conv_in = mk_conv3("decoder.conv_in", 4, top);
        mid0 = mk_res("decoder.mid_block.resnets.0", top, top);
        const std::string a = "decoder.mid_block.attentions.0";
        attn.C = top;
        attn.gn = mk_norm(a + ".group_norm", top);
        attn.q = mk_lin(a + ".to_q", top, top, false); attn.k = mk_lin(a + ".to_k", top, top, false);
        attn.v = mk_lin(a + ".to_v", top, top, false); attn.o = mk_lin(a + ".to_out.0", top, top, false);
        mid1 = mk_res("decoder.mid_block.resnets.1", top, top);
        int out_c = top;
        for (int i = 0; i < n; ++i) {
            const int prev = out_c; out_c = cfg.block_out_channels[n - 1 - i];
            std::vector<VRes> rs;
            for (int j = 0; j < cfg.layers_per_block + 1; ++j)
                rs.push_back(mk_res("decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), j == 0 ? prev : out_c, out_c));
            up_res.push_back(rs);
            if (i != n - 1) up_conv.push_back(mk_conv3("decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv", out_c, out_c));
        }
        norm_out = mk_norm("decoder.conv_norm_out", cfg.block_out_channels[0]);
        conv_out = mk_conv3("decoder.conv_out", cfg.block_out_channels[0], 3);
    }
    void require_bound() { for (auto& s : slots) if (!s.bound) throw rt_error(RT_E_MISSING_WEIGHT, "vae weight not bound: " + s.name); }

    // ---------------------------------------------------------------- launch helpers
    void gemm1(const bf16_t* A, int lda, const bf16_t* W, int ldw, const float* bias, int M, int N, int K, void* out, int ldo, int epi,
               const float* res, int ldres, const bf16_t* A_lo = nullptr, const bf16_t* W_lo = nullptr) {
        GemmArgs g{}; g.A = A; g.W = W; g.bias = bias; g.out = out; g.res = res; g.zero = zero; g.mode = A_DENSE; g.epi = epi;
        g.A_lo = A_lo; g.W_lo = W_lo;
        g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw; g.ldo = ldo; g.ldres = ldres;
        launch_gemm(g, stream);
    }
    // pair_lo != null: try to get the fp32 result as the bf16 (hi, lo) pair straight from the kernel (out = hi plane); returns false -
    // nothing launched - when the route launch_gemm takes for this problem cannot (the caller then asks for fp32 and splits)
    bool conv1(const bf16_t* in, int mode, const bf16_t* W, const float* bias, int N, int K, int H, int Wd, int CinP, void* out, int epi, const float* res,
               const bf16_t* in_lo = nullptr, const bf16_t* W_lo = nullptr, bf16_t* pair_lo = nullptr) {
        int Ho = H, Wo = Wd;
        if (mode == A_CONV3_UP2) { Ho = 2 * H; Wo = 2 * Wd; }
        GemmArgs g{}; g.A = in; g.W = W; g.bias = bias; g.out = out; g.res = res; g.zero = zero; g.mode = mode; g.epi = epi;
        g.A_lo = in_lo; g.W_lo = W_lo; g.pair_lo = pair_lo;
        g.M = Ho * Wo; g.N = N; g.K = K; g.ldw = K; g.ldo = N; g.ldres = N; g.rows_per_batch = Ho * Wo; g.Hin = H; g.Win = Wd; g.Cin = CinP;
        g.Hout = Ho; g.Wout = Wo;
        RT_REQUIRE(K == 9 * CinP, "vae conv: weight/input channel mismatch");
        // one image: the 224 x 256 tiles of the GEMM-loop convolution fill the chip only from 256^2 x 256 channels upwards (SD 64^2
        // latent: 13.1 ms per guidance call with every layer on it, 11.3 ms on the patch kernel; SDXL 128^2: 39.7 vs 41.0 the other way)
        // ... and every hi / lo (precise) contraction: as ONE launch of three passes the patch kernel runs the 128-channel layers at 1.6 PF
        // where the GEMM-loop form of the 256-channel layers reaches 0.86 (profiles/r4_vae_precise_kernel_stats.csv)
        g.prefer_patch_conv = in_lo != nullptr || (long)g.M * N < 192L * 224 * 256;
        if (pair_lo && !gemm_pair_output_ok(g)) return false;
        launch_gemm(g, stream);
        return true;
    }
    void split(const float* x, BT o, size_t n) { if (!dry) launch_cast_f32_bf16(x, o.hi, n, stream, o.lo); }
    BT castb(const float* x, size_t n) { BT o = bt(n); split(x, o, n); return o; }
    // fp32 temporaries of the precise mode live only until they are split into a pair: the bump pointer goes back on scope exit
    // (everything is ordered on one stream, so the next user of the bytes runs after the split)
    struct Scratch { rt_vae* v; size_t mark; explicit Scratch(rt_vae* v_) : v(v_), mark(v_->ws_off) {} ~Scratch() { v->ws_off = mark; } };
    // ---- dense  out(fp32) = A W^T (+bias) (+res): one pass, or hi*hi + lo*hi + hi*lo accumulated in place (precise)
    void gemm_f32(BT A, int lda, const MatW& W, int M, float* out, int ldo, const float* res = nullptr, int ldres = 0, bool bias = true) {
        if (dry) return;
        // precise: hi*hi + lo*hi + hi*lo as ONE contraction call (launch_gemm fuses the passes where the kernel can, else three launches)
        gemm1(A.hi, lda, W.w, W.K, bias ? W.b : nullptr, M, W.N, W.K, out, ldo, EPI_F32, res, ldres, precise ? A.lo : nullptr, precise ? W.w_lo : nullptr);
    }
    // ---- dense with a bf16 operand as the result (precise: fp32 result, then split into the pair)
    BT gemm_b(BT A, int lda, const MatW& W, int M, bool bias = true) {
        const size_t n = (size_t)M * W.N;
        if (precise) { BT o = bt(n); Scratch sc(this); float* t = f32(n); gemm_f32(A, lda, W, M, t, W.N, nullptr, 0, bias); split(t, o, n); return o; }
        BT o = bt(n);
        if (!dry) gemm1(A.hi, lda, W.w, W.K, bias ? W.b : nullptr, M, W.N, W.K, o.hi, W.N, EPI_BF16, nullptr, 0);
        return o;
    }
    // ---- raw operands (attention score / PV products): both sides are pairs
    void raw_f32(BT A, int lda, BT W, int ldw, int M, int N, int K, float* out, int ldo) {
        if (dry) return;
        gemm1(A.hi, lda, W.hi, ldw, nullptr, M, N, K, out, ldo, EPI_F32, nullptr, 0, precise ? A.lo : nullptr, precise ? W.lo : nullptr);
    }
    BT raw_b(BT A, int lda, BT W, int ldw, int M, int N, int K) {
        const size_t n = (size_t)M * N;
        if (precise) { BT o = bt(n); Scratch sc(this); float* t = f32(n); raw_f32(A, lda, W, ldw, M, N, K, t, N); split(t, o, n); return o; }
        BT o = bt(n);
        if (!dry) gemm1(A.hi, lda, W.hi, ldw, nullptr, M, N, K, o.hi, N, EPI_BF16, nullptr, 0);
        return o;
    }
    // ---- 3x3 convolutions
    void conv_f32(BT in, int mode, const MatW& W, int H, int Wd, int CinP, float* out, const float* res = nullptr, bool bias = true) {
        if (dry) return;
        conv1(in.hi, mode, W.w, bias ? W.b : nullptr, W.N, W.K, H, Wd, CinP, out, EPI_F32, res, precise ? in.lo : nullptr, precise ? W.w_lo : nullptr);
    }
    BT conv_b(BT in, int mode, const MatW& W, int H, int Wd, int CinP, bool bias = true) {
        const size_t n = (size_t)H * Wd * W.N;                     // (only stride-1 same-size convolutions produce operands)
        if (precise) {
            // the pair straight from the convolution's epilogue where the patch kernel takes the problem (every 3x3 layer of the SDXL
            // decoder at 1024^2); the fp32 temporary + split kernel otherwise
            BT o = bt(n);
            if (!dry && conv1(in.hi, mode, W.w, bias ? W.b : nullptr, W.N, W.K, H, Wd, CinP, o.hi, EPI_F32, nullptr, in.lo, W.w_lo, o.lo)) return o;
            Scratch sc(this); float* t = f32(n); conv_f32(in, mode, W, H, Wd, CinP, t, nullptr, bias); split(t, o, n); return o;
        }
        BT o = bt(n);
        if (!dry) conv1(in.hi, mode, W.w, bias ? W.b : nullptr, W.N, W.K, H, Wd, CinP, o.hi, EPI_BF16, nullptr);
        return o;
    }
    float* gn_fwd(const void* x, bool x_bf16, int C, int HW, const NormW& n, bool silu, BT out, BT raw) {
        const int rpc = groupnorm_bwd_rows_per_chunk(HW), nchunk = (HW + rpc - 1) / rpc;      // the VAE's own chunk rule (norm.hip)
        float* part = f32((size_t)nchunk * G * 2);
        if (dry) return part;
        GroupNormArgs a{}; a.x1 = x; a.in_bf16 = x_bf16; a.C1 = C; a.C2 = 0; a.G = G; a.B = 1; a.HW = HW; a.gamma = n.g; a.beta = n.b; a.eps = 1e-6f;
        a.silu = silu; a.out = out.hi; a.out_lo = out.lo; a.raw_out = raw.hi; a.raw_lo = raw.lo; a.partial = part; a.nchunk = nchunk;
        a.rows_per_chunk = rpc;
        launch_groupnorm(a, stream);
        return part;
    }
    void gn_bwd(const void* x, bool x_bf16, BT dA, const float* fwd_part, int C, int HW, const NormW& n, bool silu, const float* add,
                float* out, BT out_b) {
        const int rpc = groupnorm_bwd_rows_per_chunk(HW), nchunk = (HW + rpc - 1) / rpc;      // (B = 1: the forward's statistics are at fwd_part[2 g] whatever its chunking)
        float* bp = f32((size_t)nchunk * G * 2);
        if (dry) return;
        GroupNormBwdArgs a{}; a.x = x; a.x_bf16 = x_bf16; a.dA = dA.hi; a.dA_lo = dA.lo; a.fwd_partial = fwd_part; a.bwd_partial = bp; a.gamma = n.g; a.beta = n.b;
        a.eps = 1e-6f; a.silu = silu; a.C = C; a.G = G; a.B = 1; a.HW = HW; a.nchunk = nchunk; a.rows_per_chunk = rpc; a.add = add;
        a.out = out; a.out_bf16 = out_b.hi; a.out_bf16_lo = out_b.lo;
        launch_groupnorm_bwd(a, stream);
    }
    BT transp(BT in, int R, int C) {
        BT o = bt((size_t)R * C);
        if (!dry) { launch_transpose_bf16(in.hi, o.hi, R, C, stream); if (in.lo) launch_transpose_bf16(in.lo, o.lo, R, C, stream); }
        return o;
    }

    // ---------------------------------------------------------------- blocks
    float* res_fwd(const VRes& r, const float* x, int H, int W, ResSaved* sv) {
        const size_t HW = (size_t)H * W;
        BT h1 = bt(HW * r.cin);
        BT raw = r.has_sc ? bt(HW * r.cin) : BT{};
        float* p1 = gn_fwd(x, false, r.cin, (int)HW, r.n1, true, h1, raw);
        BT h3 = bt(HW * r.cout);
        const void* h2; float* p2;
        if (precise) {              // conv1 output stays fp32: it is the input of a GroupNorm, not of a contraction
            float* h2f = f32(HW * r.cout);
            conv_f32(h1, A_CONV3, r.c1.f, H, W, r.cin, h2f);
            p2 = gn_fwd(h2f, false, r.cout, (int)HW, r.n2, true, h3, BT{});
            h2 = h2f;
        } else {
            BT h2b = conv_b(h1, A_CONV3, r.c1.f, H, W, r.cin);
            p2 = gn_fwd(h2b.hi, true, r.cout, (int)HW, r.n2, true, h3, BT{});
            h2 = h2b.hi;
        }
        float* out = f32(HW * r.cout);
        const float* resid = x;
        if (r.has_sc) { gemm_f32(raw, r.cin, r.sc.f, (int)HW, out, r.cout); resid = out; }
        conv_f32(h3, A_CONV3, r.c2.f, H, W, r.cout, out, resid);
        if (sv) { sv->x = x; sv->part1 = p1; sv->h2 = h2; sv->h2_bf16 = !precise; sv->part2 = p2; sv->H = H; sv->W = W; }
        return out;
    }
    // A gradient tensor as the backward pass hands it on: fp32 (the skip / residual additions read it) and, where the producing
    // GroupNorm-backward kernel could write it in the same pass, the bf16 pair the next contraction consumes (b.hi == null: not yet split)
    struct Grad { float* f = nullptr; BT b{}; };
    BT pair_of(const Grad& g, size_t n) { return g.b.hi ? g.b : castb(g.f, n); }
    Grad res_bwd(const VRes& r, const ResSaved& sv, const Grad& dOutG) {
        const int H = sv.H, W = sv.W; const size_t HW = (size_t)H * W;
        const float* dOut = dOutG.f;
        BT dOb = pair_of(dOutG, HW * r.cout);
        BT dH3 = conv_b(dOb, A_CONV3, r.c2.b, H, W, r.cout, false);
        BT dH2 = bt(HW * r.cout);
        gn_bwd(sv.h2, sv.h2_bf16, dH3, sv.part2, r.cout, (int)HW, r.n2, true, nullptr, nullptr, dH2);
        BT dH1 = conv_b(dH2, A_CONV3, r.c1.b, H, W, r.cout, false);
        const float* skip = dOut;
        if (r.has_sc) { float* s = f32(HW * r.cin); gemm_f32(dOb, r.cout, r.sc.b, (int)HW, s, r.cin, nullptr, 0, false); skip = s; }
        Grad dX; dX.f = f32(HW * r.cin); dX.b = bt(HW * r.cin);
        gn_bwd(sv.x, false, dH1, sv.part1, r.cin, (int)HW, r.n1, true, skip, dX.f, dX.b);
        return dX;
    }
    float* attn_fwd(const float* x, int N, AttnSaved* sv) {
        const int C = attn.C;
        BT g = bt((size_t)N * C);
        float* part = gn_fwd(x, false, C, N, attn.gn, false, g, BT{});
        BT q = gemm_b(g, C, attn.q.f, N), k = gemm_b(g, C, attn.k.f, N), v = gemm_b(g, C, attn.v.f, N);
        BT vT = transp(v, N, C);
        float* S = f32((size_t)N * N);
        raw_f32(q, C, k, C, N, N, C, S, N);
        BT P = bt((size_t)N * N);
        if (!dry) launch_softmax_rows(S, P.hi, N, N, 1.f / std::sqrt((float)C), stream, P.lo);
        BT O = raw_b(P, N, vT, N, N, C, N);
        float* out = f32((size_t)N * C);
        gemm_f32(O, C, attn.o.f, N, out, C, x, C);
        if (sv) { sv->x = x; sv->part = part; sv->q = q; sv->k = k; sv->v = v; sv->P = P; sv->N = N; }
        return out;
    }
    Grad attn_bwd(const AttnSaved& sv, const Grad& dOutG) {
        const int C = attn.C, N = sv.N;
        const float* dOut = dOutG.f;
        BT dOb = pair_of(dOutG, (size_t)N * C);
        BT dO = gemm_b(dOb, C, attn.o.b, N, false);
        // dV = P^T dO
        BT PT = transp(sv.P, N, N), dOT = transp(dO, N, C);
        BT dV = raw_b(PT, N, dOT, N, N, C, N);
        // dP = dO V^T ; dS = scale * P o (dP - rowsum(dP o P))
        float* dP = f32((size_t)N * N);
        raw_f32(dO, C, sv.v, C, N, N, C, dP, N);
        BT dS = PT;                                               // P^T is dead: reuse
        if (!dry) launch_softmax_bwd(sv.P.hi, dP, dS.hi, N, N, 1.f / std::sqrt((float)C), stream, sv.P.lo, dS.lo);
        // dQ = dS K ; dK = dS^T Q
        BT kT = transp(sv.k, N, C), qT = transp(sv.q, N, C), dST = transp(dS, N, N);
        BT dQ = raw_b(dS, N, kT, N, N, C, N), dK = raw_b(dST, N, qT, N, N, C, N);
        float* dg = f32((size_t)N * C);
        gemm_f32(dQ, C, attn.q.b, N, dg, C, nullptr, 0, false);
        gemm_f32(dK, C, attn.k.b, N, dg, C, dg, C, false);
        gemm_f32(dV, C, attn.v.b, N, dg, C, dg, C, false);
        BT dgb = castb(dg, (size_t)N * C);
        Grad dX; dX.f = f32((size_t)N * C); dX.b = bt((size_t)N * C);
        gn_bwd(sv.x, false, dgb, sv.part, C, N, attn.gn, false, dOut, dX.f, dX.b);
        return dX;
    }

    // ---------------------------------------------------------------- decoder forward (+ tape)
    struct Tape { std::vector<ResSaved> res; AttnSaved attn; std::vector<std::pair<int, int>> up_hw; const float* last_x; float* part_out; int Hi, Wi; };
    float* forward(const float* lat, const float* eps, float c_lat, float c_eps, int h, int w, Tape* tp) {
        RT_REQUIRE(h <= cfg.latent_h && w <= cfg.latent_w, "vae: latent larger than configured");
        const size_t hw = (size_t)h * w;
        BT z8 = bt(hw * 8);
        if (!dry) launch_pq_conv_fwd(lat, eps, c_lat, c_eps, pq_w, pq_b, z8.hi, (int)hw, stream, z8.lo);
        float* x = f32(hw * conv_in.f.N);
        conv_f32(z8, A_CONV3, conv_in.f, h, w, 8, x);
        ResSaved rs; AttnSaved as;
        x = res_fwd(mid0, x, h, w, &rs); if (tp) tp->res.push_back(rs);
        x = attn_fwd(x, (int)hw, &as); if (tp) tp->attn = as;
        x = res_fwd(mid1, x, h, w, &rs); if (tp) tp->res.push_back(rs);
        int H = h, W = w;
        for (size_t i = 0; i < up_res.size(); ++i) {
            for (auto& r : up_res[i]) { x = res_fwd(r, x, H, W, &rs); if (tp) tp->res.push_back(rs); }
            if (i + 1 < up_res.size()) {
                const int C = up_conv[i].cin;
                BT xb = castb(x, (size_t)H * W * C);
                float* y = f32((size_t)4 * H * W * C);
                conv_f32(xb, A_CONV3_UP2, up_conv[i].f, H, W, C, y);
                if (tp) tp->up_hw.push_back({H, W});
                H *= 2; W *= 2; x = y;
            }
        }
        const int C0 = cfg.block_out_channels[0];
        BT hn = bt((size_t)H * W * C0);
        float* part = gn_fwd(x, false, C0, H * W, norm_out, true, hn, BT{});
        float* img = f32((size_t)H * W * conv_out.f.N);          // [HWi, 4], channel 3 is padding
        conv_f32(hn, A_CONV3, conv_out.f, H, W, C0, img);
        if (tp) { tp->last_x = x; tp->part_out = part; tp->Hi = H; tp->Wi = W; }
        return img;
    }
    // d(loss)/d(z1) where z1 = post_quant_conv output, given dimg (bf16 [HWi, 8])
    float* backward(const Tape& tp, BT dimg, int h, int w) {
        int H = tp.Hi, W = tp.Wi;
        const int C0 = cfg.block_out_channels[0];
        BT dHn = conv_b(dimg, A_CONV3, conv_out.b, H, W, 8, false);
        Grad dX; dX.f = f32((size_t)H * W * C0); dX.b = bt((size_t)H * W * C0);
        gn_bwd(tp.last_x, false, dHn, tp.part_out, C0, H * W, norm_out, true, nullptr, dX.f, dX.b);
        int ri = (int)tp.res.size() - 1;
        for (int i = (int)up_res.size() - 1; i >= 0; --i) {
            if (i + 1 < (int)up_res.size()) {
                const int C = up_conv[i].cin;
                BT dYb = pair_of(dX, (size_t)H * W * C);
                float* dUp = f32((size_t)H * W * C);
                conv_f32(dYb, A_CONV3, up_conv[i].b, H, W, C, dUp, nullptr, false);
                H /= 2; W /= 2;
                float* d = f32((size_t)H * W * C);
                if (!dry) launch_sumpool2x2(dUp, d, 1, H, W, C, stream);
                dX = Grad{}; dX.f = d;                                // (the pooled gradient exists as fp32 only: split by its consumer)
            }
            for (int j = (int)up_res[i].size() - 1; j >= 0; --j) dX = res_bwd(up_res[i][j], tp.res[ri--], dX);
        }
        dX = res_bwd(mid1, tp.res[ri--], dX);
        dX = attn_bwd(tp.attn, dX);
        dX = res_bwd(mid0, tp.res[ri--], dX);
        BT dXb = pair_of(dX, (size_t)h * w * conv_in.cout);
        float* dz = f32((size_t)h * w * 4);
        conv_f32(dXb, A_CONV3, conv_in.b, h, w, conv_in.coutP, dz, nullptr, false);
        return dz;
    }
};

// ================================================================================================ C ABI
static thread_local std::string g_vae_create_error;
#define VAE_TRY(v, ...)                                                          \
    try { __VA_ARGS__; return RT_OK; }                                          \
    catch (const rt_error& ex) { (v)->err = ex.what(); return ex.code; }        \
    catch (const std::exception& ex) { (v)->err = ex.what(); return RT_E_INVALID; }

static void vae_need_device(rt_vae* v) { if (!v->arena_base) throw rt_error(RT_E_STATE, "vae engine has no device (weight-table-only)"); }

// workspace estimate by a dry run of decode + guidance at the largest latent
static size_t vae_measure(rt_vae* v) {
    v->dry = true; v->ws_off = 0; v->ws_peak = 0;
    rt_vae::Tape tp;
    float* img = v->forward(nullptr, nullptr, 1.f, 0.f, v->cfg.latent_h, v->cfg.latent_w, &tp);
    (void)img;
    BT dimg = v->bt((size_t)tp.Hi * tp.Wi * 8); v->f32(4096 * 16 * 4 + 64);
    v->backward(tp, dimg, v->cfg.latent_h, v->cfg.latent_w);
    v->dry = false;
    const size_t peak = v->ws_peak; v->ws_off = 0;
    return peak + (1 << 20);
}

extern "C" {
int rt_vae_create(const rt_vae_config* cfg, int device, rt_vae** out) {
    rt_vae* v = nullptr;
    try {
        RT_REQUIRE(cfg && out, "rt_vae_create: null argument");
        RT_REQUIRE(cfg->n_blocks >= 2 && cfg->n_blocks <= 4 && cfg->layers_per_block >= 1, "rt_vae_create: bad config");
        v = new rt_vae(); v->cfg = *cfg; v->device = device;
        v->arena = Arena2(); v->build_plan();
        v->arena_bytes = v->arena.off + 256;
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0 || device < 0) { *out = v; return RT_OK; }
        HIP_CHECK(hipSetDevice(device));
        HIP_CHECK(hipStreamCreate(&v->stream));
        HIP_CHECK(hipMalloc((void**)&v->arena_base, v->arena_bytes));
        HIP_CHECK(hipMemset(v->arena_base, 0, v->arena_bytes));
        v->arena = Arena2(); v->arena.base = v->arena_base; v->arena.cap = v->arena_bytes; v->build_plan();
        v->ws_cap = vae_measure(v);
        HIP_CHECK(hipMalloc((void**)&v->ws_base, v->ws_cap));
        *out = v; return RT_OK;
    } catch (const std::exception& ex) { g_vae_create_error = ex.what(); delete v; return RT_E_INVALID; }
}
int rt_vae_destroy(rt_vae* v) {
    if (!v) return RT_OK;
    if (v->arena_base) { (void)hipSetDevice(v->device); (void)hipStreamSynchronize(v->stream); (void)hipFree(v->arena_base); (void)hipFree(v->ws_base); (void)hipStreamDestroy(v->stream); }
    delete v; return RT_OK;
}
const char* rt_vae_last_error(rt_vae* v) { return v ? v->err.c_str() : g_vae_create_error.c_str(); }
int rt_vae_weight_count(rt_vae* v) { return (int)v->slots.size(); }
int rt_vae_weight_info(rt_vae* v, int idx, char* name, int cap, int64_t* shape4, int* ndim) {
    VAE_TRY(v, {
        RT_REQUIRE(idx >= 0 && idx < (int)v->slots.size(), "rt_vae_weight_info: index");
        const Slot& s = v->slots[idx];
        RT_REQUIRE((int)s.name.size() < cap, "rt_vae_weight_info: name buffer too small");
        std::strcpy(name, s.name.c_str()); *ndim = (int)s.shape.size();
        for (size_t i = 0; i < s.shape.size(); ++i) shape4[i] = s.shape[i];
    })
}
int rt_vae_bind_weight(rt_vae* v, const char* name, const void* ptr, int dtype, const int64_t* shape, int ndim) {
    VAE_TRY(v, {
        vae_need_device(v);
        std::string nm = name;
        // pre-0.18 AttentionBlock names of older checkpoints
        const char* alias[4][2] = {{".query.", ".to_q."}, {".key.", ".to_k."}, {".value.", ".to_v."}, {".proj_attn.", ".to_out.0."}};
        for (auto& a : alias) { const size_t p = nm.find(a[0]); if (p != std::string::npos) nm.replace(p, std::strlen(a[0]), a[1]); }
        auto it = v->slot_index.find(nm);
        if (it == v->slot_index.end()) throw rt_error(RT_E_INVALID, std::string("unknown vae weight: ") + name);
        Slot& s = v->slots[it->second];
        long n_in = 1, n_exp = 1;
        for (int i = 0; i < ndim; ++i) n_in *= shape[i];
        for (auto d : s.shape) n_exp *= d;
        RT_REQUIRE(n_in == n_exp, "rt_vae_bind_weight: element count mismatch");
        RT_REQUIRE(dtype >= 0 && dtype <= 2, "rt_vae_bind_weight: dtype");
        for (PackArgs p : s.packs) { p.src = ptr; p.src_dtype = dtype; launch_pack(p, v->stream); }
        s.bound = true;
    })
}
int rt_vae_synchronize(rt_vae* v) { VAE_TRY(v, { vae_need_device(v); HIP_CHECK(hipStreamSynchronize(v->stream)); }) }
int rt_vae_arena_info(rt_vae* v, void** p, uint64_t* bytes) { VAE_TRY(v, { vae_need_device(v); *p = v->arena_base; *bytes = v->arena_bytes; }) }
int rt_vae_arena_mark_bound(rt_vae* v) { for (auto& s : v->slots) s.bound = true; return RT_OK; }

int rt_vae_decode(rt_vae* v, const float* latents, int h, int w, int divide_by_scaling, float* img_out) {
    VAE_TRY(v, {
        vae_need_device(v); v->require_bound();
        v->ws_off = 0;
        const float c = divide_by_scaling ? 1.f / v->cfg.scaling_factor : 1.f;
        float* img = v->forward(latents, nullptr, c, 0.f, h, w, nullptr);
        launch_nhwc4_to_nchw3(img, img_out, 8 * h * 8 * w, v->stream);
        HIP_CHECK(hipStreamSynchronize(v->stream));
    })
}

int rt_vae_color_guidance(rt_vae* v, float* latents, const float* noise_pred, float alpha_t, int h, int w, const float* masks_img,
                          const float* target_rgb_host, int n_regions, float weight, const float* mask_all, float* grad_out, float* loss_out_host) {
    VAE_TRY(v, {
        vae_need_device(v); v->require_bound();
        RT_REQUIRE(alpha_t > 0.f && alpha_t < 1.f, "rt_vae_color_guidance: alpha_t");
        v->ws_off = 0;
        // x0 = (lat - eps*sqrt(1-a))/sqrt(a); z0 = x0 / scaling   (predict_x0: rd.py:176-178, xl.py:955-957)
        const float sa = std::sqrt(alpha_t), s1 = std::sqrt(1.f - alpha_t), sc = v->cfg.scaling_factor;
        rt_vae::Tape tp;
        float* img = v->forward(latents, noise_pred, 1.f / (sa * sc), -s1 / (sa * sc), h, w, &tp);
        const int HWi = tp.Hi * tp.Wi;
        float* tgt = v->f32((size_t)n_regions * 3 + 4);
        HIP_CHECK(hipMemcpyAsync(tgt, target_rgb_host, (size_t)n_regions * 12, hipMemcpyHostToDevice, v->stream));
        ColorLossArgs c{}; c.img = img; c.ldi = v->conv_out.f.N; c.masks = masks_img; c.target = tgt; c.n = n_regions; c.HWi = HWi;
        BT dimg = v->bt((size_t)HWi * 8);
        c.nblk = 1024; c.partial = v->f32((size_t)c.nblk * n_regions * 4); c.dimg = dimg.hi; c.dimg_lo = dimg.lo; c.loss_out = v->f32(4);
        launch_color_loss_grad(c, v->stream);
        float* dz = v->backward(tp, dimg, h, w);
        launch_pq_conv_bwd_update(dz, 4, v->pq_w, 1.f / (sa * sc), weight, mask_all, latents, grad_out, h * w, v->stream);
        if (loss_out_host) HIP_CHECK(hipMemcpyAsync(loss_out_host, c.loss_out, 4, hipMemcpyDeviceToHost, v->stream));
        HIP_CHECK(hipStreamSynchronize(v->stream));
    })
}
}  // extern "C"
