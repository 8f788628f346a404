// VERDICT r5 item 1, form (i) in its smallest instance (csrc/gemm16.hip, gemm16_chain_kernel; LABNOTES R6.5): attn1.to_out (fp16 trunk + residual,
// LayerNorm partials) -> attn2.to_q (LayerNorm-folded) at 7 x 1024 tokens x 1280 channels as ONE launch with per-panel flag counters, against the
// two launches of the engine.  Checks the chained outputs bit for bit against the two launches, times interleaved rounds, prints the stamps.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -DRT_PROBE tools/probes/chain_bench.hip \
//         rich-text-to-image_amd/csrc/gemm16.hip -o tools/probes/chain_bench        (gemm16.hip also with -DRT_PROBE, WITHOUT -amdgpu-mfma-vgpr-form as in the Makefile)
#include "../../rich-text-to-image_amd/csrc/gemm.hip"
#include <vector>
#include <cstring>
#include <cmath>
void launch_gemm16_chain(const GemmArgs& a, const GemmArgs& b, unsigned target, int mode, hipStream_t st);
void gemm16_chain_read_times(long long* dst);
void gemm16_chain_reset();
#ifdef RT_G16_TIMING
void gemm16_read_times(long long* dst, int n);
#endif

int main() {
    const int M = 7168, C = 1280;
    bf16_t *A, *Wo, *Wq, *zero, *xb, *xb2, *q1, *q2; f16_t *trunk, *out1, *out2; float *bias, *part1, *part2, *svec;
    hipMalloc(&A, (size_t)M * C * 2); hipMalloc(&Wo, (size_t)C * C * 2); hipMalloc(&Wq, (size_t)C * C * 2); hipMalloc(&zero, 256);
    hipMalloc(&xb, (size_t)M * C * 2); hipMalloc(&xb2, (size_t)M * C * 2); hipMalloc(&q1, (size_t)M * C * 2); hipMalloc(&q2, (size_t)M * C * 2);
    hipMalloc(&trunk, (size_t)M * C * 2); hipMalloc(&out1, (size_t)M * C * 2); hipMalloc(&out2, (size_t)M * C * 2);
    hipMalloc(&bias, C * 4); hipMalloc(&part1, (size_t)M * 16 * 4); hipMalloc(&part2, (size_t)M * 16 * 4); hipMalloc(&svec, C * 8);
    {
        uint32_t x = 4242;
        std::vector<uint16_t> h((size_t)M * C);
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 | ((x >> 9) & 0x83ff) | ((x >> 3) & 0x8000)); }
        hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3000 | ((x >> 9) & 0x8fff)); }       // fp16 residual
        hipMemcpy(trunk, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        std::vector<uint16_t> w((size_t)C * C);
        for (auto& v : w) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3800 | ((x >> 9) & 0x83ff) | ((x >> 3) & 0x8000)); }
        hipMemcpy(Wo, w.data(), w.size() * 2, hipMemcpyHostToDevice);
        for (auto& v : w) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3800 | ((x >> 9) & 0x83ff) | ((x >> 3) & 0x8000)); }
        hipMemcpy(Wq, w.data(), w.size() * 2, hipMemcpyHostToDevice);
        std::vector<float> hb(2 * C); for (auto& v : hb) { x = x * 1664525u + 1013904223u; v = ((x >> 8) & 0xffff) / 65536.f - 0.5f; }
        hipMemcpy(bias, hb.data(), C * 4, hipMemcpyHostToDevice); hipMemcpy(svec, hb.data(), 2 * C * 4, hipMemcpyHostToDevice);
    }
    hipMemset(zero, 0, 256);
    auto producer = [&](f16_t* out, bf16_t* copy, float* part) {
        GemmArgs g{}; g.A = A; g.W = Wo; g.zero = zero; g.mode = A_DENSE; g.epi = EPI_F16; g.bias = bias; g.out = out; g.res = trunk; g.ldres = C;
        g.M = M; g.N = C; g.K = C; g.lda = C; g.ldw = C; g.ldo = C; g.ln_emit = part; g.ln_copy = copy; return g;
    };
    auto consumer = [&](const bf16_t* copy, const float* part, bf16_t* out) {
        GemmArgs g{}; g.A = copy; g.W = Wq; g.zero = zero; g.mode = A_DENSE; g.epi = EPI_BF16; g.out = out;
        g.M = M; g.N = C; g.K = C; g.lda = C; g.ldw = C; g.ldo = C; g.ln_part = part; g.ln_npair = 4; g.ln_ld = M; g.ln_s = svec; g.ln_inv_c = 1.f / C; g.ln_eps = 1e-5f; return g;
    };
    const GemmArgs p1 = producer(out1, xb, part1), c1 = consumer(xb, part1, q1), p2 = producer(out2, xb2, part2), c2 = consumer(xb2, part2, q2);
    // correctness: two launches vs the chain (fresh outputs)
    unsigned gen = 0;
    gemm16_chain_reset();
    hipMemset(q1, 0, (size_t)M * C * 2);
    launch_gemm16_variant(p1, 0, 0, 0); launch_gemm16_variant(c1, 0, 0, 0);
    for (int mode : {0, 4}) {                                                           // as designed (buffer_inv sc1) / L1-only invalidate
        hipMemset(q2, 0xff, (size_t)M * C * 2); hipMemset(out2, 0xff, (size_t)M * C * 2); hipMemset(xb2, 0xff, (size_t)M * C * 2);
        launch_gemm16_chain(p2, c2, 8 * ++gen, mode, 0);
        hipDeviceSynchronize();
        std::vector<uint16_t> a((size_t)M * C), b((size_t)M * C);
        hipMemcpy(a.data(), q1, a.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(b.data(), q2, b.size() * 2, hipMemcpyDeviceToHost);
        size_t bad = 0; for (size_t i = 0; i < a.size(); ++i) bad += a[i] != b[i];
        hipMemcpy(a.data(), out1, a.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(b.data(), out2, b.size() * 2, hipMemcpyDeviceToHost);
        size_t bad2 = 0; for (size_t i = 0; i < a.size(); ++i) bad2 += a[i] != b[i];
        printf("chained launch (mode %d) vs two launches: to_q outputs differing %zu of %zu, trunk outputs differing %zu\n", mode, bad, a.size(), bad2);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 20;
    float best[6] = {1e30f, 1e30f, 1e30f, 1e30f, 1e30f, 1e30f};
    const int modes[6] = {-1, 0, 1, 2, 3, 4};                                          // -1: two launches; chain modes: 0 as designed, 1 no wait, 2 no invalidate, 3 neither, 4 buffer_inv sc0
    for (int round = 0; round < 5; ++round)
        for (int w = 0; w < 6; ++w) {
            auto go = [&]() { if (w == 0) { launch_gemm16_variant(p1, 0, 0, 0); launch_gemm16_variant(c1, 0, 0, 0); } else launch_gemm16_chain(p2, c2, 8 * ++gen, modes[w], 0); };
            go(); go();
            hipEventRecord(e0, 0);
            for (int r = 0; r < reps; ++r) go();
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best[w] = std::min(best[w], ms / reps);
        }
    printf("to_out -> to_q (7168 x 1280 x 1280 each): two launches %.1f us | one chained launch %.1f us | no flag wait (wrong results) %.1f | no L1 invalidate %.1f | neither %.1f | buffer_inv sc0 %.1f us\n",
           best[0] * 1e3, best[1] * 1e3, best[2] * 1e3, best[3] * 1e3, best[4] * 1e3, best[5] * 1e3);
    for (int mode = 0; mode < 4; ++mode) {
        launch_gemm16_chain(p2, c2, 8 * ++gen, mode, 0); hipDeviceSynchronize();
        std::vector<long long> t(256 * 4); gemm16_chain_read_times(t.data());
        double ph1 = 0, wait = 0, ph2 = 0, wmax = 0; long long tmin = t[0], tmax = t[3];
        for (int i = 0; i < 256; ++i) { ph1 += t[i * 4 + 1] - t[i * 4]; wait += t[i * 4 + 2] - t[i * 4 + 1]; ph2 += t[i * 4 + 3] - t[i * 4 + 2]; wmax = std::max(wmax, (double)(t[i * 4 + 2] - t[i * 4 + 1]));
                                       tmin = std::min(tmin, t[i * 4]); tmax = std::max(tmax, t[i * 4 + 3]); }
        printf("  mode %d: phase 1 (to_out incl. store drain) %6.0f | flag wait %6.0f (max %6.0f) | phase 2 (to_q) %6.0f cycles per workgroup\n", mode, ph1 / 256, wait / 256, wmax, ph2 / 256);
#ifdef RT_G16_TIMING
        { std::vector<long long> g(256 * 8); gemm16_read_times(g.data(), 256 * 8); double seg[4] = {0, 0, 0, 0};
          for (int i = 0; i < 256; ++i) for (int k = 0; k < 4; ++k) seg[k] += (double)(g[i * 8 + k + 1] - g[i * 8 + k]);
          printf("          phase 2 inside: prologue %6.0f | loop %7.0f | drain + exchange %6.0f | epilogue %6.0f cycles\n", seg[0] / 256, seg[1] / 256, seg[2] / 256, seg[3] / 256); }
#endif
    }
    return 0;
}
