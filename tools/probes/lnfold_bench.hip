// LayerNorm fold (gemm16.hip "LNF"): the folded consumers / the partial-emitting producer against the plain kernels of the same variant,
// interleaved timing rounds + (with -DRT_G16_TIMING) the in-kernel stamps of every workgroup.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -DRT_PROBE -DRT_G16_TIMING tools/probes/lnfold_bench.hip \
//         rich-text-to-image_amd/csrc/gemm16.hip -o tools/probes/lnfold_bench        (gemm16.hip also with -DRT_PROBE -DRT_G16_TIMING)
#include "../../rich-text-to-image_amd/csrc/gemm.hip"
#include <vector>
#include <cstring>
#include <cmath>
#ifdef RT_G16_TIMING
void gemm16_read_times(long long* dst, int n);
#endif
static const struct { int BM, BN; } kVarP[RT_G16_NVAR] = {{224, 160}, {128, 160}, {224, 256}, {256, 256}, {224, 320}, {256, 320}, {160, 224}, {160, 128}, {128, 256}, {64, 160}, {128, 320}, {64, 320}, {160, 64}, {128, 160}};

int main() {
    const size_t A_ELEMS = (size_t)28672 * 5120, O_BYTES = (size_t)28672 * 5120 * 2;
    bf16_t *A, *W, *zero, *xb; void *out1, *resid; float *bias, *svec, *part;
    hipMalloc(&A, A_ELEMS * 2); hipMalloc(&W, A_ELEMS * 2); hipMalloc(&out1, O_BYTES); hipMalloc(&zero, 256); hipMalloc(&xb, (size_t)28672 * 1280 * 2);
    hipMalloc(&resid, (size_t)28672 * 1280 * 2); hipMalloc(&bias, 10240 * 4); hipMalloc(&svec, 28672 * 8); hipMalloc(&part, (size_t)28672 * 16 * 8);
    {
        std::vector<uint16_t> h(1 << 24); uint32_t x = 12345;
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 | ((x >> 9) & 0x83ff) | ((x >> 3) & 0x8000)); }
        for (size_t off = 0; off < A_ELEMS * 2; off += h.size() * 2) hipMemcpy((char*)A + off, h.data(), std::min(h.size() * 2, A_ELEMS * 2 - off), hipMemcpyHostToDevice);
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3800 | ((x >> 9) & 0x83ff) | ((x >> 3) & 0x8000)); }
        for (size_t off = 0; off < A_ELEMS * 2; off += h.size() * 2) hipMemcpy((char*)W + off, h.data(), std::min(h.size() * 2, A_ELEMS * 2 - off), hipMemcpyHostToDevice);
        std::vector<float> hb(28672); for (auto& v : hb) { x = x * 1664525u + 1013904223u; v = ((x >> 8) & 0xffff) / 65536.f - 0.5f; }
        hipMemcpy(bias, hb.data(), 10240 * 4, hipMemcpyHostToDevice); hipMemcpy(svec, hb.data(), 28672 * 4, hipMemcpyHostToDevice); hipMemcpy(svec + 28672, hb.data(), 28672 * 4, hipMemcpyHostToDevice);
        std::vector<uint16_t> hr((size_t)28672 * 1280); for (auto& v : hr) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3000 | ((x >> 9) & 0x8fff)); }
        hipMemcpy(resid, hr.data(), hr.size() * 2, hipMemcpyHostToDevice);
        std::vector<float> hp((size_t)28672 * 16 * 2); for (size_t i = 0; i < hp.size(); i += 2) { hp[i] = 3.f; hp[i + 1] = 100.f; }
        hipMemcpy(part, hp.data(), hp.size() * 4, hipMemcpyHostToDevice);
    }
    hipMemset(zero, 0, 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Case { const char* name; int M, N, K, epi, res, vt, v, wstat, mode; };      // mode: 1 consumer, 2 producer
    const Case cases[] = {
        {"GEGLU 1280", 7168, 10240, 1280, EPI_GEGLU, 0, 0, 2, 1, 1},
        {"GEGLU 640", 28672, 5120, 640, EPI_GEGLU, 0, 0, 3, 0, 1},
        {"attn2.to_q 1280", 7168, 1280, 1280, EPI_BF16, 0, 0, 0, 0, 1},
        {"attn1 Q|K 1280", 7168, 2560, 1280, EPI_BF16, 0, 0, 4, 0, 1},
        {"V^T 1280", 1280, 7168, 1280, EPI_BF16, 0, 1, 6, 0, 1},
        {"to_out 1280 (f16 + res)", 7168, 1280, 1280, EPI_F16, 1, 0, 0, 0, 2},
        {"ff.net.2 1280 (f16 + res)", 7168, 1280, 5120, EPI_F16, 1, 0, 0, 0, 2},
        {"to_out 640 (f16 + res)", 28672, 640, 640, EPI_F16, 1, 0, 4, 0, 2},
    };
    for (const Case& c : cases) {
        GemmArgs g{}; g.A = A; g.W = W; g.zero = zero; g.mode = A_DENSE; g.epi = c.epi; g.bias = c.vt ? nullptr : bias; g.out = out1;
        g.M = c.M; g.N = c.N; g.K = c.K; g.lda = c.K; g.ldw = c.K; g.ldo = c.epi == EPI_GEGLU ? c.N / 2 : c.N;
        if (c.res) { g.res = resid; g.ldres = c.N; }
        g.weights_on_rows = c.vt;
        GemmArgs f = g;
        if (c.mode == 1) { f.ln_part = part; f.ln_npair = c.K == 1280 ? 4 : 1; f.ln_ld = c.vt ? c.N : c.M; f.ln_s = svec; f.ln_inv_c = 1.f / c.K; f.ln_eps = 1e-5f; f.bias = nullptr; }
        else { f.ln_emit = part; f.ln_copy = xb; }
        printf("%-28s %5dx%5dx%4d v%d\n", c.name, c.M, c.N, c.K, c.v);
        float best[2] = {1e30f, 1e30f};
        const int reps = 20;
        for (int round = 0; round < 4; ++round)
            for (int w = 0; w < 2; ++w) {
                const GemmArgs& q = w ? f : g;
                launch_gemm16_variant(q, c.v, c.wstat, 0); launch_gemm16_variant(q, c.v, c.wstat, 0);
                hipEventRecord(e0, 0);
                for (int r = 0; r < reps; ++r) launch_gemm16_variant(q, c.v, c.wstat, 0);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); best[w] = std::min(best[w], ms / reps);
            }
        printf("    plain %7.1f us   LNF %7.1f us   (%+.1f us)\n", best[0] * 1e3, best[1] * 1e3, (best[1] - best[0]) * 1e3);
#ifdef RT_G16_TIMING
        for (int w = 0; w < 2; ++w) {
            launch_gemm16_variant(w ? f : g, c.v, c.wstat, 0); hipDeviceSynchronize();
            const int nwg = cdiv(c.M, kVarP[c.v].BM) * cdiv(c.N, kVarP[c.v].BN);
            std::vector<long long> t((size_t)nwg * 8);
            gemm16_read_times(t.data(), nwg * 8);
            double seg[4] = {0, 0, 0, 0}, e5 = 0, e6 = 0, e7 = 0;
            for (int i = 0; i < nwg; ++i) {
                for (int k = 0; k < 4; ++k) seg[k] += (double)(t[i * 8 + k + 1] - t[i * 8 + k]);
                e5 += (double)(t[i * 8 + 5] - t[i * 8 + 3]); e6 += (double)(t[i * 8 + 6] - t[i * 8 + 5]); e7 += (double)(t[i * 8 + 7] - t[i * 8 + 3]);
            }
            printf("    %s: prologue %6.0f | loop %7.0f | drain+exchange %6.0f | epilogue %6.0f cycles [bias wait %5.0f | first row tile %5.0f | all stores issued %6.0f] (%d WGs)\n",
                   w ? "LNF  " : "plain", seg[0] / nwg, seg[1] / nwg, seg[2] / nwg, seg[3] / nwg, e5 / nwg, e6 / nwg, e7 / nwg, nwg);
        }
#endif
        fflush(stdout);
    }
    return 0;
}
