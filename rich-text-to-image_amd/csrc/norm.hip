// GroupNorm (+SiLU, + virtual channel concat) and LayerNorm for NHWC / token-major activations.
//
// Replaces nn.GroupNorm + SiLU in ResnetBlock2D (models/resnet.py:594-624), Transformer2DModel.norm
// (models/transformer_2d.py:137,274; eps 1e-6), conv_norm_out (models/unet_2d_condition.py:555-557,975-977)
// and the three nn.LayerNorm of BasicTransformerBlock (models/attention.py:84,103,119).  The skip
// concatenation of the up blocks (models/unet_2d_blocks.py:2107-2109, 2210-2212) is never materialised
// in fp32: GroupNorm reads the two fp32 sources as one virtual [C1 | C2] tensor and writes the bf16
// operand(s) of the following convolution.  HBM-bound kernels: 16-B vector loads, fp32 statistics,
// deterministic two-level reduction (no atomics in global memory).
#include "common.h"

// rows per chunk = HW / 64 clamped to [16, 256] (rounds 1 - 5: HW / 128 clamped to [16, 128]): a workgroup of the two-launch form owns 64 rows at
// 4096-row maps (7 x 4096 x 640 fp16: 48.5 -> 40.1 us; 128 rows: 41.5) and 256 at 16384-row maps (7 x 16384 x 320: 65.3 -> 56.9 us; 512 rows: 60.1);
// config-3 step -0.18 ms same-box.  A function of HW only.
static int g_gn_chunk_div = 64;     // (A/B: rt_op_gemm_debug bit 11 restores 128)
static int g_gn_chunk_max = 256;
void groupnorm_set_chunk_div(int d) { g_gn_chunk_div = d; g_gn_chunk_max = d == 128 ? 128 : 256; }
static inline int gn_rows_per_chunk(int HW) { int r = HW / g_gn_chunk_div; return r < 16 ? 16 : (r > g_gn_chunk_max ? g_gn_chunk_max : r); }
int groupnorm_rows_per_chunk(int HW) { return gn_rows_per_chunk(HW); }
// the VAE (vae.hip: forward and backward GroupNorm of ONE image) keeps the chunks of rounds 1 - 5: with the rule above its backward kernels got slower (guidance call
// SD 10.7 -> 12.4 ms, SDXL precise 82.9 -> 86.6 ms), and with it on the forward pair alone the calls read 83.25 vs 82.87 ms although the stand-alone
// 1024^2 x 128 fp32 map gains (471 -> 277 us)
int groupnorm_bwd_rows_per_chunk(int HW) { int r = HW / 128; return r < 16 ? 16 : (r > 128 ? 128 : r); }
int groupnorm_nchunk(int HW) { return cdiv(HW, gn_rows_per_chunk(HW)); }

// IT: element type of the input(s): 0 fp32, 1 bf16 (single source), 2 fp16 (the UNet trunk; virtual concat allowed)
template <int IT>
__device__ __forceinline__ void load4(const void* x1, const void* x2, int C1, int C2, size_t row, int c, float v[4]) {
    if (IT == 1) {
        const uint2 u = *(const uint2*)((const bf16_t*)x1 + row * C1 + c);
        v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
        v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    } else if (IT == 2) {
        const uint2 u = (c < C1) ? *(const uint2*)((const f16_t*)x1 + row * C1 + c) : *(const uint2*)((const f16_t*)x2 + row * C2 + (c - C1));
        const f16_t* h = (const f16_t*)&u;
        v[0] = (float)h[0]; v[1] = (float)h[1]; v[2] = (float)h[2]; v[3] = (float)h[3];
    } else {
        const float4 f = (c < C1) ? *(const float4*)((const float*)x1 + row * C1 + c)
                                  : *(const float4*)((const float*)x2 + row * C2 + (c - C1));
        v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
    }
}

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// Thread decomposition shared by both kernels: nv = C/4 channel vectors per row; the block is tcols x nrl threads
// (gn_block_shape): tcols = ceil(nv / ceil(nv/512)) vector columns, nrl = 512 / tcols row lanes; a thread owns the vectors
// v0, v0 + tcols, ... and every nrl-th row.  (With a fixed 256-thread block 38 % of the threads idled at C = 640 / 1280.)
#define GN_MAXC 2560
struct GnShape { int tcols, nrl; };
__host__ __device__ inline GnShape gn_block_shape(int nv) {
    const int iters = (nv + 511) / 512;
    GnShape s; s.tcols = (nv + iters - 1) / iters; s.nrl = 512 / s.tcols; if (s.nrl < 1) s.nrl = 1;
    return s;
}

// grid (nchunk, B); partial[b][chunk][g][2] = (sum, sumsq) over this chunk's rows
template <int BF16IN>
__global__ __launch_bounds__(512) void gn_stats_kernel(GroupNormArgs p) {
    __shared__ __attribute__((aligned(16))) float sh_s[GN_MAXC], sh_q[GN_MAXC];
    const int C = p.C1 + p.C2, cpg = C / p.G, nv = C >> 2;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int r0 = chunk * p.rows_per_chunk;
    const int r1 = min(r0 + p.rows_per_chunk, p.HW);
    const GnShape shp = gn_block_shape(nv);
    const int nrl = shp.nrl, tcols = shp.tcols;
    const int rl = threadIdx.x / tcols;
    const int v0 = threadIdx.x - rl * tcols;
    if (rl < nrl) {
        for (int vec = v0; vec < nv; vec += tcols) {
            const int c = vec * 4;
            float s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
            // four rows per trip: the loads are issued together (one row at a time left ~8 KB in flight per block and the
            // kernel at 3.7 TB/s); the accumulation order over rows is unchanged
            for (int r = r0 + rl; r < r1; r += 4 * nrl) {
                float v[4][4];
                // rows past the end of the chunk are re-read from row r (clamped address) and masked out of the sums: a branch
                // around each load compiled to load ; s_waitcnt vmcnt(0) four times over
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int rr = r + u * nrl < r1 ? r + u * nrl : r;
                    load4<BF16IN>(p.x1, p.x2, p.C1, p.C2, (size_t)b * p.HW + rr, c, v[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool ok = r + u * nrl < r1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float t = ok ? v[u][e] : 0.f; s[e] += t; ss[e] += t * t; }
                }
            }
            // one 16-B store per array (four 4-B stores at a 16-B lane stride were 4-way bank conflicts: LDS conflict fraction 0.66 - 0.69 in
            // profiles/r5_pmc_sq.json)
            *(float4*)&sh_s[rl * C + c] = make_float4(s[0], s[1], s[2], s[3]);
            *(float4*)&sh_q[rl * C + c] = make_float4(ss[0], ss[1], ss[2], ss[3]);
        }
    }
    __syncthreads();
    // Group sums by ALL 512 threads: 16 lanes per group, lane j adds the group's elements j, j + 16, ... of every row lane in a fixed
    // order (consecutive lanes read consecutive floats: conflict-free), then a fixed 16-lane tree (quad DPP + two row rotations): deterministic.
    // (Round 5: 32 threads walked cpg x nrl elements each at a stride of cpg floats - 8-way conflicts at 80 channels per group, and a serial
    //  tail of ~160 LDS reads per block.)
    {
        const int nslot = blockDim.x >> 4, slot = threadIdx.x >> 4, j = threadIdx.x & 15;      // whole 16-lane DPP rows of the block (320 / 480 / 512 threads: 20 / 30 / 32)
        for (int g0 = 0; g0 < p.G; g0 += nslot) {                                               // (uniform trip count: the DPP steps run converged)
            const int g = g0 + slot;
            float s = 0.f, q = 0.f;
            if (slot < nslot && g < p.G) {
                for (int k = 0; k < nrl; ++k)
                    for (int c = g * cpg + j; c < (g + 1) * cpg; c += 16) { s += sh_s[k * C + c]; q += sh_q[k * C + c]; }
            }
            s = dpp_add<0xB1>(s); s = dpp_add<0x4E>(s); s = dpp_add<0x124>(s); s = dpp_add<0x128>(s);      // quad_perm [1,0,3,2], [2,3,0,1], row_ror 4, row_ror 8
            q = dpp_add<0xB1>(q); q = dpp_add<0x4E>(q); q = dpp_add<0x124>(q); q = dpp_add<0x128>(q);
            if (slot < nslot && g < p.G && j == 0) {
                float* dst = p.partial + ((size_t)b * p.nchunk + chunk) * 2 * p.G + 2 * g;
                dst[0] = s; dst[1] = q;
            }
        }
    }
}

// grid (G, B): reduce the per-chunk partials ONCE per (batch entry, group) (fixed order => deterministic; fp64) and leave the result in
// chunk 0's slot: partial[b][0][g] = (A, B).  mode 0: (mean, rstd) from (sum, sumsq); mode 1: (s/n, q/n) (backward means).
// Before this kernel existed every block of the apply kernels re-reduced all nchunk partials itself: at the VAE's 1024^2
// maps (8192 chunks) that was 4.8 ms per GroupNorm backward instead of ~0.3 ms.
__global__ __launch_bounds__(256) void gn_finalize_kernel(float* partial, int nchunk, int G, double n, float eps, int mode) {
    __shared__ double sh[2][256];
    const int b = blockIdx.y, g = blockIdx.x, t = threadIdx.x;          // one block per (batch entry, group)
    double s = 0.0, q = 0.0;
    for (int k = t; k < nchunk; k += 256) {
        s += (double)partial[((size_t)b * nchunk + k) * 2 * G + 2 * g];
        q += (double)partial[((size_t)b * nchunk + k) * 2 * G + 2 * g + 1];
    }
    sh[0][t] = s; sh[1][t] = q;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {                                 // fixed tree => deterministic
        if (t < w) { sh[0][t] += sh[0][t + w]; sh[1][t] += sh[1][t + w]; }
        __syncthreads();
    }
    if (t == 0) {
        s = sh[0][0]; q = sh[1][0];
        float A, Bv;
        if (mode == 0) {
            const double mu = s / n;
            double var = q / n - mu * mu;
            if (var < 0) var = 0;
            A = (float)mu; Bv = (float)(1.0 / sqrt(var + (double)eps));
        } else {
            A = (float)(s / n); Bv = (float)(q / n);
        }
        // chunk 0's raw partial of THIS group was read above by this block only (thread 0, k = 0): safe to overwrite
        partial[(size_t)b * nchunk * 2 * G + 2 * g] = A;
        partial[(size_t)b * nchunk * 2 * G + 2 * g + 1] = Bv;
    }
}
void launch_gn_finalize(float* partial, int B, int nchunk, int G, double n, float eps, int mode, hipStream_t st) {
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(G, B), dim3(256), 0, st, partial, nchunk, G, n, eps, mode);
}

// grid (nchunk, B); reads the finalized (mean, rstd) of partial[b][0][g].  A thread owns VW = 8 consecutive channels (two 16-B
// loads, one 16-B store per output: the 4-channel form spent its time issuing 8-B stores, 3.0 TB/s) or 4 when the channel counts
// are not multiples of 8.
template <int BF16IN, int VW>
__device__ __forceinline__ void loadv(const void* x1, const void* x2, int C1, int C2, size_t row, int c, float (&v)[VW]) {
    if constexpr (BF16IN == 2 && VW == 8) {          // fp16 trunk: one 16-B load (C1 % 8 == 0: the vector never straddles x1 | x2)
        const uint4 u = (c < C1) ? *(const uint4*)((const f16_t*)x1 + row * C1 + c) : *(const uint4*)((const f16_t*)x2 + row * C2 + (c - C1));
        const f16_t* h = (const f16_t*)&u;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)h[e];
        return;
    }
#pragma unroll
    for (int h = 0; h < VW / 4; ++h) {
        float t[4];
        load4<BF16IN>(x1, x2, C1, C2, row, c + 4 * h, t);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * h + e] = t[e];
    }
}
template <int VW>
__device__ __forceinline__ void storev(bf16_t* dst, const float (&y)[VW]) {
    if constexpr (VW == 8) {
        uint4 o; o.x = pack_bf16x2(y[0], y[1]); o.y = pack_bf16x2(y[2], y[3]); o.z = pack_bf16x2(y[4], y[5]); o.w = pack_bf16x2(y[6], y[7]);
        *(uint4*)dst = o;
    } else {
        uint2 o; o.x = pack_bf16x2(y[0], y[1]); o.y = pack_bf16x2(y[2], y[3]);
        *(uint2*)dst = o;
    }
}
// low parts of the same values (precise VAE mode)
template <int VW>
__device__ __forceinline__ void storev_lo(bf16_t* dst, const float (&y)[VW]) {
    float r[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e) r[e] = bf16_residual(y[e]);
    storev<VW>(dst, r);
}
template <int BF16IN, int VW>
__global__ __launch_bounds__(512) void gn_apply_kernel(GroupNormArgs p) {
    __shared__ float mean[32], rstd[32];
    const int C = p.C1 + p.C2, cpg = C / p.G, nv = C / VW;
    const int b = blockIdx.y, chunk = blockIdx.x;
    // the rows this thread starts with are requested BEFORE the statistics are finalized (they do not depend on them): the first
    // memory round trip of the block runs beside the finalize below instead of behind it (round 6)
    const int r0 = chunk * p.apply_rows;
    const int r1 = min(r0 + p.apply_rows, p.HW);
    const GnShape shp = gn_block_shape(nv);
    const int nrl = shp.nrl, tcols = shp.tcols;
    const int rl = threadIdx.x / tcols;
    const int v0 = threadIdx.x - rl * tcols;
    constexpr int RIF = 2;                                          // rows in flight per thread (round 6: 4 measured SLOWER: 31.6 -> 36.6 us at 7 x 1024 x 2560; the VAE's fp32 maps of 10^6 rows: no difference, 81.8 vs 81.9 ms per precise guidance call)
    float vpre[RIF][VW];
    const bool active = rl < nrl && v0 < nv && r0 + rl < r1;
    if (active) {
#pragma unroll
        for (int u = 0; u < RIF; ++u) {
            const int rr = r0 + rl + u * nrl < r1 ? r0 + rl + u * nrl : r0 + rl;
            loadv<BF16IN, VW>(p.x1, p.x2, p.C1, p.C2, (size_t)b * p.HW + rr, v0 * VW, vpre[u]);
        }
    }
    if (p.fuse_finalize) {
        // raw per-chunk (sum, sumsq): 16 threads per group sum every 16th chunk in fp64, thread g adds the 16 pieces in a fixed
        // order => deterministic and independent of the batch size; same (mean, rstd) arithmetic as gn_finalize_kernel mode 0
        __shared__ double fs[32 * 16], fq[32 * 16];
        for (int idx = threadIdx.x; idx < p.G * 16; idx += blockDim.x) {
            const int g = idx >> 4, j = idx & 15;
            double s = 0.0, q = 0.0;
            // nchunk <= 128 (launch_groupnorm): at most 8 partials per thread, requested TOGETHER (the rolled loop was up to 8 dependent
            // round trips in front of every block's first row), added in the same ascending order
            float2 sq[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = j + 16 * u;
                sq[u] = k < p.nchunk ? *(const float2*)(p.partial + ((size_t)b * p.nchunk + k) * 2 * p.G + 2 * g) : make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (j + 16 * u < p.nchunk) { s += (double)sq[u].x; q += (double)sq[u].y; }
            fs[idx] = s; fq[idx] = q;
        }
        __syncthreads();
        if (threadIdx.x < p.G) {
            double s = 0.0, q = 0.0;
            for (int j = 0; j < 16; ++j) { s += fs[threadIdx.x * 16 + j]; q += fq[threadIdx.x * 16 + j]; }
            const double mu = s / p.fin_n;
            double var = q / p.fin_n - mu * mu;
            if (var < 0) var = 0;
            mean[threadIdx.x] = (float)mu; rstd[threadIdx.x] = (float)(1.0 / sqrt(var + (double)p.eps));
        }
        __syncthreads();
    }
    // finalized statistics (fuse_finalize == 0: (mean, rstd) of group g sit in p.stats[b][g] - left there by the statistics kernel's last
    // workgroup or by gn_finalize_kernel): every thread reads the one or two groups its channels belong to itself - no LDS, no barrier
    const float2* gstat = p.fuse_finalize ? nullptr : (const float2*)p.stats + (size_t)b * (p.stats_ld >> 1);
    if (rl >= nrl) return;
    for (int vec = v0; vec < nv; vec += tcols) {
        const int c = vec * VW;
        float sc[VW], sh[VW];                                      // y = x * sc + sh  ==  (x - mean) * rstd * gamma + beta
        float gm[VW], bt[VW];
#pragma unroll
        for (int h = 0; h < VW / 4; ++h) {
            const float4 g4 = *(const float4*)(p.gamma + c + 4 * h), b4 = *(const float4*)(p.beta + c + 4 * h);
            gm[4 * h] = g4.x; gm[4 * h + 1] = g4.y; gm[4 * h + 2] = g4.z; gm[4 * h + 3] = g4.w;
            bt[4 * h] = b4.x; bt[4 * h + 1] = b4.y; bt[4 * h + 2] = b4.z; bt[4 * h + 3] = b4.w;
        }
        if (cpg >= VW) {
            const int g0 = c / cpg, split = (g0 + 1) * cpg - c;     // channels c .. c + split - 1 belong to group g0, the rest to g0 + 1
            const int g1 = g0 + 1 < p.G ? g0 + 1 : g0;
            float m0, r0s, m1, r1s;
            if (p.fuse_finalize) { m0 = mean[g0]; r0s = rstd[g0]; m1 = mean[g1]; r1s = rstd[g1]; }
            else { const float2 a0 = gstat[g0], a1 = gstat[g1]; m0 = a0.x; r0s = a0.y; m1 = a1.x; r1s = a1.y; }
#pragma unroll
            for (int e = 0; e < VW; ++e) {
                const float mg = e < split ? m0 : m1, rg = e < split ? r0s : r1s;
                sc[e] = rg * gm[e]; sh[e] = bt[e] - mg * sc[e];
            }
        } else {                                                   // narrow groups (tiny configurations): a vector spans several of them
#pragma unroll
            for (int e = 0; e < VW; ++e) {
                const int g = (c + e) / cpg;
                float mg, rg;
                if (p.fuse_finalize) { mg = mean[g]; rg = rstd[g]; } else { const float2 a = gstat[g]; mg = a.x; rg = a.y; }
                sc[e] = rg * gm[e]; sh[e] = bt[e] - mg * sc[e];
            }
        }
        for (int r = r0 + rl; r < r1; r += RIF * nrl) {
            float v[RIF][VW];
            // rows past the end of the chunk re-read row r (clamped address, not branched around: a branch per load compiles to
            // load ; s_waitcnt vmcnt(0) RIF times over) and are skipped below
            if (vec == v0 && r == r0 + rl) {                        // the trip that was requested in front of the finalize
#pragma unroll
                for (int u = 0; u < RIF; ++u)
#pragma unroll
                    for (int e = 0; e < VW; ++e) v[u][e] = vpre[u][e];
            } else {
#pragma unroll
                for (int u = 0; u < RIF; ++u) {
                    const int rr = r + u * nrl < r1 ? r + u * nrl : r;
                    loadv<BF16IN, VW>(p.x1, p.x2, p.C1, p.C2, (size_t)b * p.HW + rr, c, v[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < RIF; ++u) {
                if (r + u * nrl >= r1) continue;
                const size_t row = (size_t)b * p.HW + r + u * nrl;
                float y[VW];
#pragma unroll
                for (int e = 0; e < VW; ++e) {
                    y[e] = v[u][e] * sc[e] + sh[e];
                    // SiLU with v_exp_f32 (2^x) + v_rcp_f32: the IEEE division alone was ~10 VALU ops per element, half of this kernel
                    if (p.silu) y[e] = y[e] * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * y[e]));
                }
                storev<VW>(p.out + row * C + c, y);
                if (p.raw_out) storev<VW>(p.raw_out + row * C + c, v[u]);
                if (p.out_lo) storev_lo<VW>(p.out_lo + row * C + c, y);
                if (p.raw_lo) storev_lo<VW>(p.raw_lo + row * C + c, v[u]);
            }
        }
    }
}

// ---------------------------------------------------------------- GroupNorm in ONE launch (round 6): one workgroup per (batch entry, group)
// The two-launch form above is a latency chain (LABNOTES R6.2): statistics kernel, kernel boundary, then an apply kernel whose every workgroup
// first re-reduces the per-chunk partials - 15 us for the 2 MB tensors of SD-v1.5's 16x16 maps, 45 us for 7 x 32^2 x 1280.  Where ONE
// workgroup can own a whole (batch entry, group) - HW x C/G elements, read twice: the second pass hits the L2 the first one filled - the
// statistics never leave the workgroup: pass 1 sums (x, x^2) (per-thread fp32 over <= 8 rows in flight, then fp64: 64-lane butterfly +
// 8-wave LDS step in a fixed order => deterministic and independent of the batch), pass 2 normalises, applies SiLU and writes the bf16
// operand.  A thread owns ONE VW-channel vector column of the group (gamma / beta loaded once) and every nrl-th row.  Workgroup -> (b, g)
// order: each XCD takes G / 8 NEIGHBOURING groups of every batch entry, so the 128-B lines that 2 - 3 groups share (40 - 160 B per row and
// group) are fetched into one L2.  Leaves (mean, rstd) in partial[b][0][g] like gn_finalize_kernel (the VAE backward reads them there).
template <int IT, int VW>
__global__ __launch_bounds__(512) void gn_fused_kernel(GroupNormArgs p) {
    __shared__ double sh_s[8], sh_q[8];
    __shared__ float sh_stat[2];
    const int C = p.C1 + p.C2, cpg = C / p.G, nvg = cpg / VW;
    int g, b;
    {
        const int bid = blockIdx.x;
        if ((p.G & 7) == 0) { const int gpx = p.G >> 3, x = bid & 7, i = bid >> 3; g = x * gpx + i % gpx; b = i / gpx; }
        else { g = bid % p.G; b = bid / p.G; }
    }
    const int nrl = 512 / nvg;                                        // row lanes; threads >= nvg * nrl idle in the passes
    const int tid = threadIdx.x, rl = tid / nvg, v = tid - rl * nvg;
    const bool active = rl < nrl;
    const int c = g * cpg + v * VW;
    const size_t row0 = (size_t)b * p.HW;
    constexpr int U = 8;
    float s = 0.f, q = 0.f;
    if (active) {
        for (int r = rl; r < p.HW; r += U * nrl) {
            float x[U][VW];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rr = r + u * nrl < p.HW ? r + u * nrl : r;      // clamped, masked below (a branch per load serialises the loads)
                loadv<IT, VW>(p.x1, p.x2, p.C1, p.C2, row0 + rr, c, x[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ok = r + u * nrl < p.HW;
#pragma unroll
                for (int e = 0; e < VW; ++e) { const float t = ok ? x[u][e] : 0.f; s += t; q += t * t; }
            }
        }
    }
    double ds = (double)s, dq = (double)q;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { ds += __shfl_xor(ds, m); dq += __shfl_xor(dq, m); }
    if ((tid & 63) == 0) { sh_s[tid >> 6] = ds; sh_q[tid >> 6] = dq; }
    __syncthreads();
    if (tid == 0) {
        double ts = 0.0, tq = 0.0;
        for (int w = 0; w < 8; ++w) { ts += sh_s[w]; tq += sh_q[w]; }
        const double n = (double)cpg * p.HW, mu = ts / n;
        double var = tq / n - mu * mu;
        if (var < 0) var = 0;
        const float mf = (float)mu, rf = (float)(1.0 / sqrt(var + (double)p.eps));
        sh_stat[0] = mf; sh_stat[1] = rf;
        float* dst = p.partial + (size_t)b * p.nchunk * 2 * p.G + 2 * g;
        dst[0] = mf; dst[1] = rf;
    }
    __syncthreads();
    if (!active) return;
    float sc[VW], sf[VW];
    {
        const float mean = sh_stat[0], rstd = sh_stat[1];
#pragma unroll
        for (int e = 0; e < VW; ++e) { sc[e] = rstd * p.gamma[c + e]; sf[e] = p.beta[c + e] - mean * sc[e]; }
    }
    for (int r = rl; r < p.HW; r += U * nrl) {
        float x[U][VW];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u * nrl < p.HW ? r + u * nrl : r;
            loadv<IT, VW>(p.x1, p.x2, p.C1, p.C2, row0 + rr, c, x[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (r + u * nrl >= p.HW) continue;
            const size_t row = row0 + r + u * nrl;
            float y[VW];
#pragma unroll
            for (int e = 0; e < VW; ++e) {
                y[e] = x[u][e] * sc[e] + sf[e];
                if (p.silu) y[e] = y[e] * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * y[e]));
            }
            storev<VW>(p.out + row * C + c, y);
            if (p.raw_out) storev<VW>(p.raw_out + row * C + c, x[u]);
        }
    }
}
// The same launch when a thread's share of the (batch entry, group) is at most U rows (fp16 trunk; every SD-v1.5 GroupNorm below its 64^2
// level, SDXL's 1280-channel GroupNorms): the rows stay in registers between the two passes - ONE memory round trip instead of two - and
// gamma / beta are requested beside them.  Same summation order as gn_fused_kernel (rows u = 0 .. U-1 ascending, then the same reduction).
template <int VW, int U>
__global__ __launch_bounds__(512) void gn_fused_single_kernel(GroupNormArgs p) {
    __shared__ double sh_s[8], sh_q[8];
    __shared__ float sh_stat[2];
    const int C = p.C1 + p.C2, cpg = C / p.G, nvg = cpg / VW;
    int g, b;
    {
        const int bid = blockIdx.x;
        if ((p.G & 7) == 0) { const int gpx = p.G >> 3, x = bid & 7, i = bid >> 3; g = x * gpx + i % gpx; b = i / gpx; }
        else { g = bid % p.G; b = bid / p.G; }
    }
    const int nrl = 512 / nvg;
    const int tid = threadIdx.x, rl = tid / nvg, v = tid - rl * nvg;
    const bool active = rl < nrl && rl < p.HW;
    const int c = g * cpg + v * VW;
    const size_t row0 = (size_t)b * p.HW;
    float x[U][VW], gm[VW], bt[VW];
    float s = 0.f, q = 0.f;
    if (active) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = rl + u * nrl < p.HW ? rl + u * nrl : rl;
            loadv<2, VW>(p.x1, p.x2, p.C1, p.C2, row0 + rr, c, x[u]);
        }
#pragma unroll
        for (int h = 0; h < VW / 4; ++h) {
            const float4 g4 = *(const float4*)(p.gamma + c + 4 * h), b4 = *(const float4*)(p.beta + c + 4 * h);
            gm[4 * h] = g4.x; gm[4 * h + 1] = g4.y; gm[4 * h + 2] = g4.z; gm[4 * h + 3] = g4.w;
            bt[4 * h] = b4.x; bt[4 * h + 1] = b4.y; bt[4 * h + 2] = b4.z; bt[4 * h + 3] = b4.w;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = rl + u * nrl < p.HW;
#pragma unroll
            for (int e = 0; e < VW; ++e) { const float t = ok ? x[u][e] : 0.f; s += t; q += t * t; }
        }
    }
    double ds = (double)s, dq = (double)q;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { ds += __shfl_xor(ds, m); dq += __shfl_xor(dq, m); }
    if ((tid & 63) == 0) { sh_s[tid >> 6] = ds; sh_q[tid >> 6] = dq; }
    __syncthreads();
    if (tid == 0) {
        double ts = 0.0, tq = 0.0;
        for (int w = 0; w < 8; ++w) { ts += sh_s[w]; tq += sh_q[w]; }
        const double n = (double)cpg * p.HW, mu = ts / n;
        double var = tq / n - mu * mu;
        if (var < 0) var = 0;
        const float mf = (float)mu, rf = (float)(1.0 / sqrt(var + (double)p.eps));
        sh_stat[0] = mf; sh_stat[1] = rf;
        float* dst = p.partial + (size_t)b * p.nchunk * 2 * p.G + 2 * g;
        dst[0] = mf; dst[1] = rf;
    }
    __syncthreads();
    if (!active) return;
    float sc[VW], sf[VW];
    {
        const float mean = sh_stat[0], rstd = sh_stat[1];
#pragma unroll
        for (int e = 0; e < VW; ++e) { sc[e] = rstd * gm[e]; sf[e] = bt[e] - mean * sc[e]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (rl + u * nrl >= p.HW) continue;
        const size_t row = row0 + rl + u * nrl;
        float y[VW];
#pragma unroll
        for (int e = 0; e < VW; ++e) {
            y[e] = x[u][e] * sc[e] + sf[e];
            if (p.silu) y[e] = y[e] * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * y[e]));
        }
        storev<VW>(p.out + row * C + c, y);
        if (p.raw_out) storev<VW>(p.raw_out + row * C + c, x[u]);
    }
}
static int g_gn_fused = 1;
void groupnorm_set_fused(int on) { g_gn_fused = on; }
// 0: the two-launch form; else the vector width of the one-launch form.  A function of the shape of ONE batch entry.
static int gn_fused_vw(const GroupNormArgs& a) {
    if (!g_gn_fused || a.out_lo || a.raw_lo) return 0;
    const int C = a.C1 + a.C2, cpg = C / a.G;
    // measured (profiles/r6_gn_one_launch_bench.txt): wins while a thread's chain is <= ~20 rows (7 x 32^2 x 1280: 27.1 vs 43.1 us, x 2560: 33.8 vs
    // 40.1); at 4096 rows (40 rows per thread, five dependent round trips per pass) the two-launch form's 448 workgroups are faster (53 vs 46 us)
    if (a.HW > 1024 || (long)a.HW * cpg > 98304) return 0;
    const int vw = (cpg % 8 == 0 && a.C1 % 8 == 0 && a.C2 % 8 == 0) ? 8 : ((cpg % 4 == 0) ? 4 : 0);
    if (!vw || cpg / vw > 512) return 0;
    return vw;
}

void launch_groupnorm(const GroupNormArgs& a, hipStream_t st) {
    const int C = a.C1 + a.C2;
    RT_REQUIRE(a.G >= 1 && a.G <= 32 && C % a.G == 0, "groupnorm: bad group count");
    RT_REQUIRE(a.C1 % 4 == 0 && a.C2 % 4 == 0, "groupnorm: channels must be multiples of 4");
    RT_REQUIRE(a.in_bf16 >= 0 && a.in_bf16 <= 2 && !(a.in_bf16 == 1 && a.x2), "groupnorm: input type 0 fp32 / 1 bf16 (no concat) / 2 fp16");
    RT_REQUIRE(a.rows_per_chunk >= 1 && a.nchunk == cdiv(a.HW, a.rows_per_chunk), "groupnorm: nchunk mismatch");      // (the UNet engine passes groupnorm_rows_per_chunk, the VAE its own rule)
    RT_REQUIRE(C <= GN_MAXC, "groupnorm: too many channels");
    const GnShape shp = gn_block_shape(C >> 2);
    RT_REQUIRE(shp.nrl * C <= GN_MAXC || shp.nrl == 1, "groupnorm: LDS staging too small");
    dim3 grid(a.nchunk, a.B), block(shp.tcols * shp.nrl);
    const double n = (double)(C / a.G) * a.HW;
    const bool wide = a.C1 % 8 == 0 && a.C2 % 8 == 0;               // 8 channels per thread in the apply pass
    const GnShape sh8 = gn_block_shape(C >> 3);
    dim3 block8(sh8.tcols * sh8.nrl);
    if (const int vw = gn_fused_vw(a)) {
        dim3 gridf(a.G * a.B), blockf(512);
        if (a.in_bf16 == 2) {
            const int nvg = (C / a.G) / vw, rpt = cdiv(a.HW, 512 / nvg);      // rows per thread
#define RT_GNS(VW_, U_) { hipLaunchKernelGGL((gn_fused_single_kernel<VW_, U_>), gridf, blockf, 0, st, a); HIP_CHECK(hipGetLastError()); return; }
            if (vw == 8) { if (rpt <= 4) RT_GNS(8, 4) else if (rpt <= 8) RT_GNS(8, 8) else if (rpt <= 12) RT_GNS(8, 12) }
            else { if (rpt <= 8) RT_GNS(4, 8) else if (rpt <= 16) RT_GNS(4, 16) else if (rpt <= 24) RT_GNS(4, 24) }
#undef RT_GNS
        }
#define RT_GNF(IT) { if (vw == 8) hipLaunchKernelGGL((gn_fused_kernel<IT, 8>), gridf, blockf, 0, st, a); else hipLaunchKernelGGL((gn_fused_kernel<IT, 4>), gridf, blockf, 0, st, a); }
        if (a.in_bf16 == 1) RT_GNF(1) else if (a.in_bf16 == 2) RT_GNF(2) else RT_GNF(0)
#undef RT_GNF
        HIP_CHECK(hipGetLastError());
        return;
    }
    GroupNormArgs aa = a;
    // (round 6, measured and dropped - profiles/r6_groupnorm_experiments.txt: a finer apply grid, four rows in flight, the finalize as its own
    //  launch, and the statistics kernel's last workgroup finalizing behind a ticket counter)
    aa.fuse_finalize = a.fuse_finalize && a.nchunk <= 128;
    aa.fin_n = n;
    aa.stats = a.partial; aa.stats_ld = a.nchunk * 2 * a.G;        // non-fused form: gn_finalize_kernel leaves (mean, rstd) in chunk 0's slot of every batch entry
    aa.apply_rows = a.rows_per_chunk;
    dim3 grid_apply(cdiv(a.HW, aa.apply_rows), a.B);
#define RT_GN_LAUNCH(IT)                                                                                  \
    {                                                                                                         \
        hipLaunchKernelGGL(gn_stats_kernel<IT>, grid, block, 0, st, aa);                                      \
        if (!aa.fuse_finalize) launch_gn_finalize(a.partial, a.B, a.nchunk, a.G, n, a.eps, 0, st);            \
        if (wide) hipLaunchKernelGGL((gn_apply_kernel<IT, 8>), grid_apply, block8, 0, st, aa);                \
        else hipLaunchKernelGGL((gn_apply_kernel<IT, 4>), grid_apply, block, 0, st, aa);                      \
    }
    if (a.in_bf16 == 1) RT_GN_LAUNCH(1) else if (a.in_bf16 == 2) RT_GN_LAUNCH(2) else RT_GN_LAUNCH(0)
#undef RT_GN_LAUNCH
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------- LayerNorm: one wave per row
// A lane owns 8 consecutive channels per 512-channel pass: two 16-B loads, ONE 16-B store (with 4 channels per lane the kernel sat
// at 4.0 TB/s, bound by the number of 8-B store instructions: guide T21).
#define LN_MAXP 3   // C <= 1536 (8 channels per lane per 512-channel pass)
template <bool F16IN>
__device__ __forceinline__ void ln_load8(const void* xr, int c, float4 (&v)[2]) {
    if (F16IN) {
        const uint4 u = *(const uint4*)((const f16_t*)xr + c);
        const f16_t* h = (const f16_t*)&u;
        v[0] = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]); v[1] = make_float4((float)h[4], (float)h[5], (float)h[6], (float)h[7]);
    } else {
        v[0] = *(const float4*)((const float*)xr + c); v[1] = *(const float4*)((const float*)xr + c + 4);
    }
}
// wave-wide sum, result in every lane: four DPP steps inside each row of 16 lanes (quad_perm [1,0,3,2] / [2,3,0,1], row_ror 4 / 8),
// then the two cross-row exchanges through ds_bpermute (six dependent LDS round trips per sum before)
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add<0xB1>(v); v = dpp_add<0x4E>(v); v = dpp_add<0x124>(v); v = dpp_add<0x128>(v);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
// NP = ceil(C / 512) passes.  Every load of the row (and of gamma / beta) is issued before the first wait: the addresses of lanes
// past the end of the row are clamped instead of branched around (the branchy form compiled to load ; s_waitcnt vmcnt(0) ; load ...,
// i.e. three dependent memory round trips per row plus the parameter loads behind the statistics: 11 us for 7168 x 1280).
template <bool F16IN, int NP>
__global__ __launch_bounds__(256) void layernorm_kernel(const void* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, bf16_t* __restrict__ out,
                                                        int rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const char* xr = (const char*)x + (size_t)row * C * (F16IN ? 2 : 4);
    float4 v[NP][2], g[NP][2], bb[NP][2];
    bool act[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int c = i * 512 + lane * 8;
        act[i] = c < C;
        const int cc = act[i] ? c : 0;
        ln_load8<F16IN>(xr, cc, v[i]);
        g[i][0] = *(const float4*)(gamma + cc); g[i][1] = *(const float4*)(gamma + cc + 4);
        bb[i][0] = *(const float4*)(beta + cc); bb[i][1] = *(const float4*)(beta + cc + 4);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const float t = (v[i][0].x + v[i][0].y + v[i][0].z + v[i][0].w) + (v[i][1].x + v[i][1].y + v[i][1].z + v[i][1].w);
        s += act[i] ? t : 0.f;
    }
    const float mu = wave_sum(s) / C;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        float t = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float a = v[i][h].x - mu, b = v[i][h].y - mu, d = v[i][h].z - mu, e = v[i][h].w - mu;
            t += a * a + b * b + d * d + e * e;
        }
        ss += act[i] ? t : 0.f;
    }
    const float rs = rsqrtf(wave_sum(ss) / C + eps);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        uint4 o;
        uint32_t* ow = (uint32_t*)&o;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            ow[2 * h] = pack_bf16x2((v[i][h].x - mu) * rs * g[i][h].x + bb[i][h].x, (v[i][h].y - mu) * rs * g[i][h].y + bb[i][h].y);
            ow[2 * h + 1] = pack_bf16x2((v[i][h].z - mu) * rs * g[i][h].z + bb[i][h].z, (v[i][h].w - mu) * rs * g[i][h].w + bb[i][h].w);
        }
        if (act[i]) *(uint4*)(out + (size_t)row * C + i * 512 + lane * 8) = o;
    }
}

// Channel counts of the form 40 * LPR (320 / 640 / 1280: every transformer width of SD-v1.5 and SDXL): LPR lanes per row, 64 / LPR
// rows per wave, five 8-element chunks per lane - no clamped duplicate loads (the 512-column passes above leave the last pass of a
// 640 / 1280-wide row 3/4 / 1/2 empty: 15.1 us for 28672 x 640, 9.6 us for 7168 x 1280, the same 36.7 MB) and the row reductions
// stay inside DPP rows for LPR <= 16.  gamma / beta are read chunk by chunk behind the statistics (5 KB, L1 resident).
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {                // sum over aligned groups of LPR lanes, result in every lane
    v = dpp_add<0xB1>(v); v = dpp_add<0x4E>(v);                       // 4 lanes
    if (LPR >= 8) v = dpp_add<0x124>(v);                              // row_ror 4 -> 8 lanes
    if (LPR >= 16) v = dpp_add<0x128>(v);                             // row_ror 8 -> 16 lanes
    if (LPR >= 32) v += __shfl_xor(v, 16);
    if (LPR >= 64) v += __shfl_xor(v, 32);
    return v;
}
template <>
__device__ __forceinline__ float group_sum<8>(float v) {             // row_ror 4 would mix the two 8-lane groups of a DPP row
    v = dpp_add<0xB1>(v); v = dpp_add<0x4E>(v);
    return v + __shfl_xor(v, 4);
}
template <bool F16IN, int LPR>
__global__ __launch_bounds__(256) void layernorm40_kernel(const void* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, bf16_t* __restrict__ out,
                                                          int rows, float eps) {
    constexpr int C = 40 * LPR, RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, sub = lane / LPR, l = lane % LPR;
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + sub;
    const int rowc = row < rows ? row : rows - 1;                     // whole groups stay converged: clamp, do not branch
    const char* xr = (const char*)x + (size_t)rowc * C * (F16IN ? 2 : 4);
    float4 v[5][2];
#pragma unroll
    for (int i = 0; i < 5; ++i) ln_load8<F16IN>(xr, (i * LPR + l) * 8, v[i]);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) s += (v[i][0].x + v[i][0].y + v[i][0].z + v[i][0].w) + (v[i][1].x + v[i][1].y + v[i][1].z + v[i][1].w);
    const float mu = group_sum<LPR>(s) * (1.f / C);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float a = v[i][h].x - mu, b = v[i][h].y - mu, d = v[i][h].z - mu, e = v[i][h].w - mu;
            ss += a * a + b * b + d * d + e * e;
        }
    const float rs = rsqrtf(group_sum<LPR>(ss) * (1.f / C) + eps);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int c = (i * LPR + l) * 8;
        const float4 g0 = *(const float4*)(gamma + c), g1 = *(const float4*)(gamma + c + 4);
        const float4 b0 = *(const float4*)(beta + c), b1 = *(const float4*)(beta + c + 4);
        uint4 o;
        o.x = pack_bf16x2((v[i][0].x - mu) * rs * g0.x + b0.x, (v[i][0].y - mu) * rs * g0.y + b0.y);
        o.y = pack_bf16x2((v[i][0].z - mu) * rs * g0.z + b0.z, (v[i][0].w - mu) * rs * g0.w + b0.w);
        o.z = pack_bf16x2((v[i][1].x - mu) * rs * g1.x + b1.x, (v[i][1].y - mu) * rs * g1.y + b1.y);
        o.w = pack_bf16x2((v[i][1].z - mu) * rs * g1.z + b1.z, (v[i][1].w - mu) * rs * g1.w + b1.w);
        if (row < rows) *(uint4*)(out + (size_t)row * C + c) = o;
    }
}

void launch_layernorm(const void* x, int x_f16, const float* gamma, const float* beta, bf16_t* out, int rows, int C,
                      float eps, hipStream_t st) {
    RT_REQUIRE(C % 8 == 0 && C >= 8 && C <= LN_MAXP * 512, "layernorm: C must be a multiple of 8 and <= 1536");
    if (C == 320 || C == 640 || C == 1280) {                        // 40 * LPR
        const int lpr = C / 40, rpb = 4 * (64 / lpr);
        const dim3 grid40(cdiv(rows, rpb)), block40(256);
#define RT_LN40(F16_, LPR_) hipLaunchKernelGGL((layernorm40_kernel<F16_, LPR_>), grid40, block40, 0, st, x, gamma, beta, out, rows, eps)
        if (x_f16) { if (lpr == 8) RT_LN40(true, 8); else if (lpr == 16) RT_LN40(true, 16); else RT_LN40(true, 32); }
        else { if (lpr == 8) RT_LN40(false, 8); else if (lpr == 16) RT_LN40(false, 16); else RT_LN40(false, 32); }
#undef RT_LN40
        HIP_CHECK(hipGetLastError());
        return;
    }
    const dim3 grid(cdiv(rows, 4)), block(256);
    const int np = cdiv(C, 512);
#define RT_LN(F16_, NP_) hipLaunchKernelGGL((layernorm_kernel<F16_, NP_>), grid, block, 0, st, x, gamma, beta, out, rows, C, eps)
    if (x_f16) { if (np == 1) RT_LN(true, 1); else if (np == 2) RT_LN(true, 2); else RT_LN(true, 3); }
    else { if (np == 1) RT_LN(false, 1); else if (np == 2) RT_LN(false, 2); else RT_LN(false, 3); }
#undef RT_LN
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------- LayerNorm fold (gemm16.hip, "LNF")
// W' = bf16(gamma_k W_nk) from the packed bf16 weight, s_n = sum_k W'_nk (of the ROUNDED values: the correction must cancel what the
// MFMA adds up), c_n = b_n + sum_k beta_k W_nk  (models/attention.py:150,168,181: norm1 -> attn1, norm2 -> attn2, norm3 -> ff).
// One workgroup per weight row, fixed-order tree reduction: a pure function of the checkpoint (every rank derives the same bits).
__global__ __launch_bounds__(256) void ln_fold_derive_kernel(const bf16_t* __restrict__ W, int ldw, const float* __restrict__ bias,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, int K,
                                                             bf16_t* __restrict__ Wf, float2* __restrict__ sc_out) {
    __shared__ float red[2][256];
    const int n = blockIdx.x, tid = threadIdx.x;
    float s = 0.f, c = 0.f;
    for (int k = tid; k < K; k += 256) {
        const float w = bf16_to_f32(W[(size_t)n * ldw + k]);
        const bf16_t wf = f32_to_bf16(gamma[k] * w);
        Wf[(size_t)n * ldw + k] = wf;
        s += bf16_to_f32(wf);
        c = fmaf(beta[k], w, c);
    }
    red[0][tid] = s; red[1][tid] = c;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (tid < h) { red[0][tid] += red[0][tid + h]; red[1][tid] += red[1][tid + h]; }
        __syncthreads();
    }
    if (tid == 0) sc_out[n] = make_float2(red[0][0], red[1][0] + (bias ? bias[n] : 0.f));
}
void launch_ln_fold_derive(const bf16_t* W, int ldw, const float* bias, const float* gamma, const float* beta, int N, int K,
                           bf16_t* Wf, float* sc, hipStream_t st) {
    RT_REQUIRE(N > 0 && K > 0 && ldw >= K, "ln_fold_derive: shape");
    hipLaunchKernelGGL(ln_fold_derive_kernel, dim3(N), dim3(256), 0, st, W, ldw, bias, gamma, beta, K, Wf, (float2*)sc);
    HIP_CHECK(hipGetLastError());
}

// Stand-alone producer of what the fp16-trunk epilogues of gemm16.hip leave (LNF = 2) for a trunk some other kernel wrote: xb = bf16 of
// the fp16 trunk values and one (sum, sum of squares) of xb per row and column tile of `bn` (160 / 320) columns, pair-major
// [tile pair][row] float4 = two tiles, added in the same order as there (v_dot2c_f32_bf16 pair by pair inside an 8-value item; items
// j, j + 4, j + 8 per quarter of an 80-column block; the quarters as (q0 + q1) + (q2 + q3); the blocks of a tile in ascending order).
// (The epilogue rounds xb from the fp32 value it also rounds the trunk from; here only the fp16 trunk exists: same statistics of its own
//  xb, not the same bits as an emitting producer - which of the two a layer uses is a function of its shape alone.)
__global__ __launch_bounds__(256) void ln_partials_kernel(const f16_t* __restrict__ x, bf16_t* __restrict__ xb, float* __restrict__ part, int rows, int C, int bn) {
    const int ntn = C / bn, nb = bn / 80;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= (long)rows * ntn) return;
    const long row = id / ntn; const int tn = (int)(id - row * ntn);
    float t1 = 0.f, t2 = 0.f;
    for (int b = 0; b < nb; ++b) {
        const long off = row * C + (long)tn * bn + b * 80;
        const f16_t* p = x + off;
        float i1[10], i2[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const uint4 v = *(const uint4*)(p + j * 8);
            const f16_t* h = (const f16_t*)&v;
            uint4 cb;
            cb.x = pack_bf16x2((float)h[0], (float)h[1]); cb.y = pack_bf16x2((float)h[2], (float)h[3]);
            cb.z = pack_bf16x2((float)h[4], (float)h[5]); cb.w = pack_bf16x2((float)h[6], (float)h[7]);
            *(uint4*)(xb + off + j * 8) = cb;
            float s1 = 0.f, s2 = 0.f;
            asm volatile("v_dot2c_f32_bf16 %0, %2, %6\n\tv_dot2c_f32_bf16 %1, %2, %2\n\t"
                         "v_dot2c_f32_bf16 %0, %3, %6\n\tv_dot2c_f32_bf16 %1, %3, %3\n\t"
                         "v_dot2c_f32_bf16 %0, %4, %6\n\tv_dot2c_f32_bf16 %1, %4, %4\n\t"
                         "v_dot2c_f32_bf16 %0, %5, %6\n\tv_dot2c_f32_bf16 %1, %5, %5\n\t"
                         "s_nop 2"      /* a DOT result read by another VALU opcode needs 3 wait states, and hipcc's hazard recogniser does not look into asm */
                         : "+v"(s1), "+v"(s2) : "v"(cb.x), "v"(cb.y), "v"(cb.z), "v"(cb.w), "v"(0x3f803f80u));
            i1[j] = s1; i2[j] = s2;
        }
        float q1[4], q2[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            q1[jj] = i1[jj] + i1[jj + 4]; q2[jj] = i2[jj] + i2[jj + 4];
            if (jj < 2) { q1[jj] += i1[jj + 8]; q2[jj] += i2[jj + 8]; }
        }
        const float b1 = (q1[0] + q1[1]) + (q1[2] + q1[3]), b2 = (q2[0] + q2[1]) + (q2[2] + q2[3]);
        if (b == 0) { t1 = b1; t2 = b2; } else { t1 += b1; t2 += b2; }
    }
    *((float2*)part + ((long)(tn >> 1) * rows + row) * 2 + (tn & 1)) = make_float2(t1, t2);
}
void launch_ln_partials(const f16_t* x, bf16_t* xb, float* part, int rows, int C, int bn, hipStream_t st) {
    RT_REQUIRE(rows > 0 && (bn == 160 || bn == 320) && C % (2 * bn) == 0 && C / bn <= 8, "ln_partials: column tiles of 160 / 320, an even number (at most 8) per row");
    const long n = (long)rows * (C / bn);
    hipLaunchKernelGGL(ln_partials_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, xb, part, rows, C, bn);
    HIP_CHECK(hipGetLastError());
}
