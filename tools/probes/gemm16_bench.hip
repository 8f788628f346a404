// gemm16.hip (16x16x32 family) against gemm.hip (32x32x16 family) on the engine's GEMM shapes: correctness (max |diff| relative to
// max |ref|, outputs of the old kernels as reference) and interleaved timing rounds (guide 5.4 rules 24 / 25: one process, random data).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -DRT_PROBE tools/probes/gemm16_bench.hip \
//         rich-text-to-image_amd/csrc/gemm16.hip -o tools/probes/gemm16_bench        (two translation units)
#include "../../rich-text-to-image_amd/csrc/gemm.hip"
static const struct { int BM, BN; } kVar[RT_G16_NVAR] = {{224, 160}, {128, 160}, {224, 256}, {256, 256}, {224, 320}, {256, 320}, {160, 224}, {160, 128}, {128, 256}, {64, 160}, {128, 320}, {64, 320}, {160, 64}, {128, 160}};
#include <vector>
#include <cstring>
#include <cmath>
#ifdef RT_G16_TIMING
void gemm16_read_times(long long* dst, int n);
#endif

__global__ void diff_kernel(const void* a, const void* b, size_t n, int kind /*0 bf16, 1 f32, 2 f16*/, float* out /* [maxdiff, maxref] as uint bits */) {
    float md = 0.f, mr = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float x, y;
        if (kind == 0) { x = bf16_to_f32(((const bf16_t*)a)[i]); y = bf16_to_f32(((const bf16_t*)b)[i]); }
        else if (kind == 1) { x = ((const float*)a)[i]; y = ((const float*)b)[i]; }
        else { x = (float)((const f16_t*)a)[i]; y = (float)((const f16_t*)b)[i]; }
        const float d = fabsf(x - y);
        if (!(d <= md)) md = d;        // NaN propagates
        mr = fmaxf(mr, fabsf(x));
    }
    atomicMax((unsigned*)out, __float_as_uint(md != md ? 1e30f : md));
    atomicMax((unsigned*)out + 1, __float_as_uint(mr));
}

struct Case { const char* name; int M, N, K, epi, res, vt; int old_cfgs[3]; int new_vars[6]; int wstat_both; };

int main(int argc, char** argv) {
    const size_t A_ELEMS = (size_t)28672 * 5120, W_ELEMS = (size_t)28672 * 5120, O_BYTES = (size_t)28672 * 5120 * 4;
    bf16_t *A, *W, *zero; void *out0, *out1, *resid; float *bias, *dstat;
    hipMalloc(&A, A_ELEMS * 2); hipMalloc(&W, W_ELEMS * 2); hipMalloc(&out0, O_BYTES); hipMalloc(&out1, O_BYTES); hipMalloc(&zero, 256);
    hipMalloc(&resid, (size_t)28672 * 1280 * 4); hipMalloc(&bias, 10240 * 4); hipMalloc(&dstat, 8);
    {
        std::vector<uint16_t> h(1 << 24); uint32_t x = 12345;
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 | ((x >> 9) & 0x83ff) | ((x >> 3) & 0x8000)); }   // ~ +-[0.5, 2)
        for (size_t off = 0; off < A_ELEMS * 2; off += h.size() * 2) hipMemcpy((char*)A + off, h.data(), std::min(h.size() * 2, A_ELEMS * 2 - off), hipMemcpyHostToDevice);
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3800 | ((x >> 9) & 0x83ff) | ((x >> 3) & 0x8000)); }
        for (size_t off = 0; off < W_ELEMS * 2; off += h.size() * 2) hipMemcpy((char*)W + off, h.data(), std::min(h.size() * 2, W_ELEMS * 2 - off), hipMemcpyHostToDevice);
        std::vector<float> hb(10240); for (auto& v : hb) { x = x * 1664525u + 1013904223u; v = ((x >> 8) & 0xffff) / 65536.f - 0.5f; }
        hipMemcpy(bias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
        // residual: f16 values (also valid when read as f32 garbage? no: F32 cases use a separately filled region) - fill as f16 pattern
        std::vector<uint16_t> hr((size_t)28672 * 1280 * 2); for (auto& v : hr) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3000 | ((x >> 9) & 0x8fff)); }
        hipMemcpy(resid, hr.data(), hr.size() * 2, hipMemcpyHostToDevice);
    }
    hipMemset(zero, 0, 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const bool quick = argc > 1 && !strcmp(argv[1], "q");
    const bool small_only = argc > 1 && !strcmp(argv[1], "s");      // only the small-batch cases (plain pass, SD-v1.5)

    // name, M, N, K, epi, residual, weights_on_rows, {old configs}, {new variants}, try wstat 0 and 1
    const Case cases[] = {
        {"attn.to_out / to_q 1280 (f16 trunk + res)", 7168, 1280, 1280, EPI_F16, 1, 0, {2, -1, -1}, {0, 13, 1, -1, -1, -1}, 0},
        {"to_out 1280 (f16 trunk, NO residual)", 7168, 1280, 1280, EPI_F16, 0, 0, {2, -1, -1}, {0, -1, -1, -1, -1, -1}, 0},
        {"attn2.to_q 1280 (bf16)", 7168, 1280, 1280, EPI_BF16, 0, 0, {2, -1, -1}, {0, 13, -1, -1, -1, -1}, 0},
        {"ff.net.2 1280 (K = 5120, f16 + res)", 7168, 1280, 5120, EPI_F16, 1, 0, {2, -1, -1}, {0, 13, -1, -1, -1, -1}, 0},
        {"attn1 Q|K 1280 (N = 2560)", 7168, 2560, 1280, EPI_BF16, 0, 0, {2, -1, -1}, {0, -1, 4, -1, 2, -1}, 0},
        {"attn1 Q|K 1280, 4 streams", 4096, 2560, 1280, EPI_BF16, 0, 0, {2, -1, -1}, {4, 1, 3, -1, -1, -1}, 0},
        {"GEGLU 1280", 7168, 10240, 1280, EPI_GEGLU, 0, 0, {3, -1, -1}, {2, -1, 3, -1, -1, -1}, 1},
        {"GEGLU 640", 28672, 5120, 640, EPI_GEGLU, 0, 0, {3, -1, -1}, {2, -1, 3, -1, -1, -1}, 0},
        {"to_out / to_q 640 (f16 + res)", 28672, 640, 640, EPI_F16, 1, 0, {8, -1, -1}, {0, 4, 13, -1, -1, -1}, 0},
        {"to_q 640 (bf16)", 28672, 640, 640, EPI_BF16, 0, 0, {8, -1, -1}, {0, 4, -1, -1, -1, -1}, 0},
        {"ff.net.2 640 (K = 2560)", 28672, 640, 2560, EPI_F16, 1, 0, {8, -1, -1}, {0, 4, -1, -1, -1, -1}, 0},
        {"attn1 Q|K 640 (N = 1280)", 28672, 1280, 640, EPI_BF16, 0, 0, {8, -1, -1}, {0, 4, 2, -1, -1, -1}, 0},
        {"V^T 1280", 1280, 7168, 1280, EPI_BF16, 0, 1, {2, -1, -1}, {6, 7, -1, -1, -1, -1}, 0},
        {"V^T 640", 640, 28672, 640, EPI_BF16, 0, 1, {0, -1, -1}, {6, 7, -1, 4, -1, -1}, 0},
        {"ragged rows (M = 5000), f32 + res", 5000, 1280, 1280, EPI_F32, 1, 0, {2, -1, -1}, {0, 1, -1, -1, -1, -1}, 0},
        {"5 streams 1280", 5120, 1280, 1280, EPI_F16, 1, 0, {2, 0, -1}, {0, 1, -1, -1, -1, -1}, 0},
        {"plain pass (2 streams) to_out 1280", 2048, 1280, 1280, EPI_F16, 1, 0, {2, -1, -1}, {1, 9, -1, -1, -1, -1}, 0},
        {"plain pass (2 streams) ff.net.2 1280", 2048, 1280, 5120, EPI_F16, 1, 0, {2, -1, -1}, {1, 9, -1, -1, -1, -1}, 0},
        {"plain pass (2 streams) to_out 640", 8192, 640, 640, EPI_F16, 1, 0, {8, -1, -1}, {4, 10, 11, 1, -1, -1}, 0},
        {"plain pass (2 streams) ff.net.2 640", 8192, 640, 2560, EPI_F16, 1, 0, {8, -1, -1}, {4, 10, 11, -1, -1, -1}, 0},
        {"plain pass (2 streams) Q|K 640", 8192, 1280, 640, EPI_BF16, 0, 0, {8, -1, -1}, {4, 10, 11, -1, -1, -1}, 0},
        {"plain pass (2 streams) Q|K 1280", 2048, 2560, 1280, EPI_BF16, 0, 0, {2, -1, -1}, {8, 4, 10, 11, -1, -1}, 0},
        {"plain pass (2 streams) V^T 1280", 1280, 2048, 1280, EPI_BF16, 0, 1, {2, -1, -1}, {6, 7, 12, -1, -1, -1}, 0},
        {"SD-v1.5 3 streams V^T 640", 640, 3072, 640, EPI_BF16, 0, 1, {0, -1, -1}, {7, 12, -1, -1, -1, -1}, 0},
        {"SD-v1.5 3 streams V^T 1280 (16^2)", 1280, 768, 1280, EPI_BF16, 0, 1, {2, -1, -1}, {7, 12, -1, -1, -1, -1}, 0},
        {"SD-v1.5 3 streams 32^2 x 640 to_out", 3072, 640, 640, EPI_F16, 1, 0, {8, -1, -1}, {1, 9, -1, -1, -1, -1}, 0},
        {"SD-v1.5 3 streams 32^2 x 640 ff.net.2", 3072, 640, 2560, EPI_F16, 1, 0, {8, -1, -1}, {1, 9, -1, -1, -1, -1}, 0},
        {"SD-v1.5 3 streams 16^2 x 1280 to_out", 768, 1280, 1280, EPI_F16, 1, 0, {2, -1, -1}, {1, 9, -1, -1, -1, -1}, 0},
        {"SD-v1.5 5 streams 32^2 x 640 to_out", 5120, 640, 640, EPI_F16, 1, 0, {8, -1, -1}, {1, 9, -1, -1, -1, -1}, 0},
        {"4096^3 bf16", 4096, 4096, 4096, EPI_BF16, 0, 0, {3, 7, -1}, {3, -1, -1, -1, -1, -1}, 0},
        {"8192x4096x4096 bf16", 8192, 4096, 4096, EPI_BF16, 0, 0, {7, -1, -1}, {3, -1, -1, -1, -1, -1}, 0},
    };
    for (const Case& c : cases) {
        if (quick && c.M * (long)c.N > 40000000L) continue;
        if (small_only && !(strstr(c.name, "plain pass") || strstr(c.name, "SD-v1.5"))) continue;
        GemmArgs g{}; g.A = A; g.W = W; g.zero = zero; g.mode = A_DENSE; g.epi = c.epi; g.bias = bias;
        g.M = c.M; g.N = c.N; g.K = c.K; g.lda = c.K; g.ldw = c.K; g.ldo = c.epi == EPI_GEGLU ? c.N / 2 : c.N;
        if (c.vt) g.bias = nullptr;
        if (c.res) { g.res = resid; g.ldres = c.N; }
        g.weights_on_rows = c.vt;
        const size_t nout = (size_t)c.M * g.ldo;
        const int kind = (c.epi == EPI_F32) ? 1 : (c.epi == EPI_F16 ? 2 : 0);
        printf("%-44s %5dx%5dx%4d  (%.1f GFLOP)\n", c.name, c.M, c.N, c.K, 2.0 * c.M * c.N * c.K * 1e-9);
        // reference output: first old config
        g.out = out0;
        hipMemset(out0, 0, nout * 4);
        launch_with_cfg(g, c.old_cfgs[0], 0);
        struct V { int is_new, id, wstat; float best, sum; int n; } vs[16]; int nv = 0;
        for (int i = 0; i < 3; ++i) if (c.old_cfgs[i] >= 0) vs[nv++] = V{0, c.old_cfgs[i], 0, 1e30f, 0.f, 0};
        for (int i = 0; i < 6; ++i) if (c.new_vars[i] >= 0) {
            vs[nv++] = V{1, c.new_vars[i], 0, 1e30f, 0.f, 0};
            if (c.wstat_both) vs[nv++] = V{1, c.new_vars[i], 1, 1e30f, 0.f, 0};
        }
        auto run = [&](const V& v, void* o) {
            GemmArgs q = g; q.out = o;
            if (v.is_new) launch_gemm16_variant(q, v.id, v.wstat, 0); else launch_with_cfg(q, v.id, 0);
        };
        // correctness of every new variant against the reference
        for (int i = 0; i < nv; ++i) if (vs[i].is_new) {
            hipMemset(out1, 0xff, nout * (kind == 1 ? 4 : 2));
            try { run(vs[i], out1); } catch (const std::exception& ex) { printf("    new v%d: %s\n", vs[i].id, ex.what()); vs[i].id = -1; continue; }
            hipMemset(dstat, 0, 8);
            hipLaunchKernelGGL(diff_kernel, dim3(1024), dim3(256), 0, 0, out0, out1, nout, kind, dstat);
            float hs[2]; hipMemcpy(hs, dstat, 8, hipMemcpyDeviceToHost);
            const hipError_t err = hipDeviceSynchronize();
            printf("    check v%d%s: max|diff| %.4g  max|ref| %.4g  rel %.2e %s%s\n", vs[i].id, vs[i].wstat ? "w" : "", hs[0], hs[1], hs[0] / (hs[1] + 1e-30f),
                   hs[0] / (hs[1] + 1e-30f) < (kind == 1 ? 2e-5f : 1e-2f) ? "OK" : "MISMATCH", err != hipSuccess ? hipGetErrorString(err) : "");
        }
        const int reps = c.M * (long)c.N * c.K > 3e11 ? 8 : 20;
        for (int round = 0; round < 3; ++round)
            for (int i = 0; i < nv; ++i) {
                if (vs[i].id < 0) continue;
                run(vs[i], out1); run(vs[i], out1);
                hipEventRecord(e0, 0);
                for (int r = 0; r < reps; ++r) run(vs[i], out1);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
                vs[i].best = std::min(vs[i].best, ms); vs[i].sum += ms; vs[i].n++;
            }
#ifdef RT_G16_TIMING
        for (int i = 0; i < nv; ++i) {
            if (vs[i].id < 0 || !vs[i].is_new) continue;
            run(vs[i], out1); hipDeviceSynchronize();
            const int nwg = cdiv(c.M, kVar[vs[i].id].BM) * cdiv(c.N, kVar[vs[i].id].BN);
            std::vector<long long> t((size_t)nwg * 8);
            gemm16_read_times(t.data(), nwg * 8);
            double seg[4] = {0, 0, 0, 0}, e5 = 0, e6 = 0, e7 = 0;
            for (int w = 0; w < nwg; ++w) {
                for (int k = 0; k < 4; ++k) seg[k] += (double)(t[w * 8 + k + 1] - t[w * 8 + k]);
                e5 += (double)(t[w * 8 + 5] - t[w * 8 + 3]); e6 += (double)(t[w * 8 + 6] - t[w * 8 + 5]); e7 += (double)(t[w * 8 + 7] - t[w * 8 + 3]);
            }
            printf("    timing v%d%s: prologue %6.0f | loop %7.0f | drain+exchange %6.0f | epilogue %6.0f cycles (mean over %d WGs) [epilogue: bias wait %5.0f | first row tile %5.0f | all stores issued %6.0f]\n",
                   vs[i].id, vs[i].wstat ? "w" : "", seg[0] / nwg, seg[1] / nwg, seg[2] / nwg, seg[3] / nwg, nwg, e5 / nwg, e6 / nwg, e7 / nwg);
        }
#endif
        for (int i = 0; i < nv; ++i) {
            if (vs[i].id < 0) continue;
            const double fl = 2.0 * c.M * c.N * c.K;
            printf("    %s %d%s  %4dx%-3d  min %7.1f us (%5.0f TF)  mean %7.1f us\n", vs[i].is_new ? "g16 v" : "old c", vs[i].id, vs[i].wstat ? "w" : " ",
                   vs[i].is_new ? kVar[vs[i].id].BM : kCfg[vs[i].id].BM, vs[i].is_new ? kVar[vs[i].id].BN : kCfg[vs[i].id].BN,
                   vs[i].best * 1e3, fl / (vs[i].best * 1e-3) / 1e12, vs[i].sum / vs[i].n * 1e3);
        }
        fflush(stdout);
    }
    if (!small_only) {   // 3x3 stride-1 convolutions: patch kernel (gemm.hip conv3p_kernel) against the implicit GEMM of gemm16.hip
        struct Cv { int B, H, W, Cin, Cout, epi, v; } cs[] = {{7, 32, 32, 1280, 1280, EPI_BF16, 0}, {7, 32, 32, 2560, 1280, EPI_F16, 0}, {7, 32, 32, 1920, 1280, EPI_BF16, 0},
            {7, 64, 64, 640, 640, EPI_BF16, 4}, {7, 64, 64, 1280, 640, EPI_F16, 4}, {7, 64, 64, 960, 640, EPI_BF16, 4}, {7, 128, 128, 320, 320, EPI_BF16, 4},
            {7, 128, 128, 640, 320, EPI_F16, 4}, {3, 20, 24, 128, 320, EPI_BF16, 4}};
        for (auto c : cs) {
            GemmArgs g{}; g.A = A; g.W = W; g.zero = zero; g.mode = A_CONV3; g.epi = c.epi; g.bias = bias;
            g.M = c.B * c.H * c.W; g.N = c.Cout; g.K = 9 * c.Cin; g.ldw = g.K; g.ldo = c.Cout; g.rows_per_batch = c.H * c.W;
            g.Hin = g.Hout = c.H; g.Win = g.Wout = c.W; g.Cin = c.Cin;
            if (c.epi == EPI_F16) { g.res = resid; g.ldres = c.Cout; }
            const size_t nout = (size_t)g.M * g.ldo;
            const int kind = c.epi == EPI_F16 ? 2 : 0;
            printf("conv3x3 %dx%dx%dx%d -> %d  (%.1f GFLOP)\n", c.B, c.H, c.W, c.Cin, c.Cout, 2.0 * g.M * g.N * g.K * 1e-9);
            auto run = [&](int which, void* o) {
                GemmArgs q = g; q.out = o;
                if (which == 0) {
                    if (c.H % 16 || c.W % 16) launch_with_cfg(q, 2, 0);
                    else if (c.epi == EPI_F16) launch_conv3p<EPI_F16, false>(q, 0); else launch_conv3p<EPI_BF16, false>(q, 0);
                } else launch_gemm16_variant(q, c.v, 0, 0);
            };
            hipMemset(out0, 0, nout * 2); hipMemset(out1, 0xff, nout * 2);
            run(0, out0); run(1, out1);
            hipMemset(dstat, 0, 8);
            hipLaunchKernelGGL(diff_kernel, dim3(1024), dim3(256), 0, 0, out0, out1, nout, kind, dstat);
            float hs[2]; hipMemcpy(hs, dstat, 8, hipMemcpyDeviceToHost);
            printf("    check v%d: max|diff| %.4g  max|ref| %.4g  rel %.2e %s\n", c.v, hs[0], hs[1], hs[0] / (hs[1] + 1e-30f), hs[0] / (hs[1] + 1e-30f) < 1e-2f ? "OK" : "MISMATCH");
            for (int which = 0; which < 2; ++which) {
                float best = 1e30f;
                for (int round = 0; round < 3; ++round) {
                    run(which, out1);
                    hipEventRecord(e0, 0);
                    for (int r = 0; r < 8; ++r) run(which, out1);
                    hipEventRecord(e1, 0); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms / 8);
                }
                printf("    %s  min %7.1f us (%5.0f TF)\n", which == 0 ? "patch kernel " : "gemm16 conv  ", best * 1e3, 2.0 * g.M * g.N * g.K / (best * 1e-3) / 1e12);
            }
            fflush(stdout);
        }
    }
    // what the shape-based rule picks
    printf("pick:");
    for (const Case& c : cases) {
        GemmArgs g{}; g.mode = A_DENSE; g.epi = c.epi; g.M = c.M; g.N = c.N; g.K = c.K; g.lda = c.K; g.ldw = c.K; g.weights_on_rows = c.vt;
        int ws = 0; const int v = gemm16_pick(g, c.vt, &ws);
        printf(" %d%s", v, ws ? "w" : "");
    }
    printf("\n");
    return 0;
}
