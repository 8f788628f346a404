// Stand-alone GEMM timing (hipEvents) for tile-configuration / main-loop experiments.  Build variants with -D flags.
#include "../../rich-text-to-image_amd/csrc/gemm.hip"
#include <vector>
#include <algorithm>
#include <cstring>
int main(int argc, char** argv) {
    struct Shape { int M, N, K; } shapes[] = {{7168, 2560, 1280}, {28672, 1280, 640}, {7168, 10240, 1280}, {28672, 5120, 640}, {7168, 1280, 1280}, {7168, 1280, 5120}, {28672, 640, 640}, {28672, 640, 2560}, {4096, 4096, 4096}, {8192, 8192, 8192}};
    bf16_t *A, *W, *out, *zero;
    hipMalloc(&A, (size_t)28672 * 8192 * 2); hipMalloc(&W, (size_t)10240 * 8192 * 2); hipMalloc(&out, (size_t)28672 * 10240 * 2); hipMalloc(&zero, 256);
    // pseudo-random bf16 fill (values ~ +-1): data-dependent power/clock effects matter (guide 5.4 rule 25)
    { std::vector<uint16_t> h(1 << 24); uint32_t x = 12345; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 | ((x >> 9) & 0x83ff) | ((x >> 3) & 0x8000)); }
      for (size_t off = 0; off < (size_t)28672 * 8192 * 2; off += h.size() * 2) hipMemcpy((char*)A + off, h.data(), std::min(h.size() * 2, (size_t)28672 * 8192 * 2 - off), hipMemcpyHostToDevice);
      for (size_t off = 0; off < (size_t)10240 * 8192 * 2; off += h.size() * 2) hipMemcpy((char*)W + off, h.data(), std::min(h.size() * 2, (size_t)10240 * 8192 * 2 - off), hipMemcpyHostToDevice); }
    hipMemset(zero, 0, 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float *resid, *bias; hipMalloc(&resid, (size_t)28672 * 1280 * 4); hipMemset(resid, 0, (size_t)28672 * 1280 * 4); hipMalloc(&bias, 10240 * 4); hipMemset(bias, 0, 10240 * 4);
    const bool quick = argc > 1 && !strcmp(argv[1], "q");      // "q": only the per-shape configuration table below
    {   // GEGLU epilogue: the 16-wave kernel (cfg 3) against the 8-phase kernel (cfg 7), interleaved rounds
        struct E { int M, N, K; } es[] = {{7168, 10240, 1280}, {28672, 5120, 640}};
        for (auto e : es) {
            printf("GEGLU %5dx%5dx%4d:", e.M, e.N, e.K);
            for (int cfg : {3, 7, 3, 7}) {
                GemmArgs g{}; g.A = A; g.W = W; g.out = out; g.zero = zero; g.mode = A_DENSE; g.epi = EPI_GEGLU; g.bias = bias;
                g.M = e.M; g.N = e.N; g.K = e.K; g.lda = e.K; g.ldw = e.K; g.ldo = e.N / 2;
                for (int r = 0; r < 3; ++r) launch_with_cfg(g, cfg, 0);
                hipEventRecord(e0, 0);
                for (int r = 0; r < 30; ++r) launch_with_cfg(g, cfg, 0);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("  cfg%d %6.1f us %5.0f TF", cfg, ms / 30 * 1e3, 2.0 * e.M * e.N * e.K / (ms / 30 * 1e-3) / 1e12);
            }
            printf("\n");
        }
    }
    if (!quick) {   // epilogue cost on the two FF shapes: GEGLU (erf) vs plain bf16, fp32+residual vs plain bf16
        struct E { int M, N, K, epi, cfg; const char* name; } es[] = {{7168, 10240, 1280, EPI_BF16, 3, "geglu-shape bf16"}, {7168, 10240, 1280, EPI_GEGLU, 3, "geglu-shape GEGLU"},
            {28672, 5120, 640, EPI_BF16, 3, "geglu640 bf16"}, {28672, 5120, 640, EPI_GEGLU, 3, "geglu640 GEGLU"},
            {7168, 1280, 1280, EPI_BF16, 2, "out-shape bf16"}, {7168, 1280, 1280, EPI_F32, 2, "out-shape f32+res"}, {7168, 1280, 5120, EPI_F32, 2, "ff2 f32+res"}, {7168, 1280, 5120, EPI_F32, 6, "ff2 f32+res pp"}};
        for (auto e : es) {
            GemmArgs g{}; g.A = A; g.W = W; g.out = out; g.zero = zero; g.mode = A_DENSE; g.epi = e.epi; g.bias = bias;
            g.M = e.M; g.N = e.N; g.K = e.K; g.lda = e.K; g.ldw = e.K; g.ldo = e.epi == EPI_GEGLU ? e.N / 2 : e.N;
            if (e.epi == EPI_F32) { g.res = resid; g.ldres = e.N; }
            for (int r = 0; r < 3; ++r) launch_with_cfg(g, e.cfg, 0);
            hipEventRecord(e0, 0);
            for (int r = 0; r < 50; ++r) launch_with_cfg(g, e.cfg, 0);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-22s cfg%d %7.1f us\n", e.name, e.cfg, ms / 50 * 1e3);
        }
    }
    if (!quick) {   // 3x3 convolutions: patch kernel vs the implicit-GEMM configuration 2
        struct Cv { int B, H, W, Cin, Cout; } cs[] = {{7, 32, 32, 1280, 1280}, {7, 32, 32, 2560, 1280}, {7, 64, 64, 640, 640}, {7, 64, 64, 1920, 640}, {7, 128, 128, 320, 320}, {7, 128, 128, 960, 320}};
        for (auto c : cs) {
            GemmArgs g{}; g.A = A; g.W = W; g.out = out; g.zero = zero; g.mode = A_CONV3; g.epi = EPI_BF16; g.bias = bias;
            g.M = c.B * c.H * c.W; g.N = c.Cout; g.K = 9 * c.Cin; g.ldw = g.K; g.ldo = c.Cout; g.rows_per_batch = c.H * c.W;
            g.Hin = g.Hout = c.H; g.Win = g.Wout = c.W; g.Cin = c.Cin;
            printf("conv %dx%dx%dx%d->%d:", c.B, c.H, c.W, c.Cin, c.Cout);
            for (int variant = 0; variant < 2; ++variant) {
                auto run = [&]() { if (variant == 0) launch_conv3p<EPI_BF16, false>(g, 0); else launch_with_cfg(g, 2, 0); };
                for (int r = 0; r < 3; ++r) run();
                hipEventRecord(e0, 0);
                for (int r = 0; r < 20; ++r) run();
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("  %s %7.1f us %6.0f TF", variant == 0 ? "patch" : "cfg2 ", ms / 20 * 1e3, 2.0 * g.M * g.N * g.K / (ms / 20 * 1e-3) / 1e12);
            }
            printf("\n");
        }
    }
    {   // L2-channel test: the same problems with the operand rows padded by 128 B (row stride no longer a multiple of 2 KB)
        struct Shape { int M, N, K, pad; } ps[] = {{8192, 4096, 4096, 0}, {8192, 4096, 4096, 64}, {8192, 4096, 4096, 192}, {7168, 1280, 1280, 0}, {7168, 1280, 1280, 64},
                                                   {7168, 10240, 1280, 0}, {7168, 10240, 1280, 64}, {7168, 1280, 5120, 0}, {7168, 1280, 5120, 64}};
        for (auto sh : ps) {
            printf("pad %3d %5dx%5dx%4d:", sh.pad, sh.M, sh.N, sh.K);
            for (int cfg : {2, 3, 7}) {
                GemmArgs g{}; g.A = A; g.W = W; g.out = out; g.zero = zero; g.mode = A_DENSE; g.epi = EPI_BF16;
                g.M = sh.M; g.N = sh.N; g.K = sh.K; g.lda = sh.K + sh.pad; g.ldw = sh.K + sh.pad; g.ldo = sh.N;
                for (int r = 0; r < 3; ++r) launch_with_cfg(g, cfg, 0);
                hipEventRecord(e0, 0);
                for (int r = 0; r < 30; ++r) launch_with_cfg(g, cfg, 0);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("  cfg%d %6.1f us %5.0f TF", cfg, ms / 30 * 1e3, 2.0 * sh.M * sh.N * sh.K / (ms / 30 * 1e-3) / 1e12);
            }
            printf("\n");
        }
    }
#ifdef RT_G8_TIMING
    for (auto sh : shapes) {
        GemmArgs g{}; g.A = A; g.W = W; g.out = out; g.zero = zero; g.mode = A_DENSE; g.epi = EPI_BF16;
        g.M = sh.M; g.N = sh.N; g.K = sh.K; g.lda = sh.K; g.ldw = sh.K; g.ldo = sh.N;
        launch_with_cfg(g, 7, 0); launch_with_cfg(g, 7, 0); hipDeviceSynchronize();
        long long t[32]; hipMemcpyFromSymbol(t, HIP_SYMBOL(g_g8_times), sizeof(t));
        const double nph = 4.0 * ((sh.K + 63) / 64);
        printf("g8 timing %5dx%5dx%4d (cycles per phase):\n", sh.M, sh.N, sh.K);
        for (int w : {0, 3, 4, 7}) printf("    wave %d: load section %6.0f | barrier-in %5.0f | lgkm+mfma %5.0f | barrier-out %5.0f\n", w, t[w * 4] / nph, t[w * 4 + 1] / nph, t[w * 4 + 2] / nph, t[w * 4 + 3] / nph);
    }
#endif
    for (auto sh : shapes) {
        printf("%5dx%5dx%4d:", sh.M, sh.N, sh.K);
        for (int cfg : {2, 9, 3, 7, 0, 1, 6}) {
            GemmArgs g{}; g.A = A; g.W = W; g.out = out; g.zero = zero; g.mode = A_DENSE; g.epi = EPI_BF16;
            g.M = sh.M; g.N = sh.N; g.K = sh.K; g.lda = sh.K; g.ldw = sh.K; g.ldo = sh.N;
            for (int r = 0; r < 3; ++r) launch_with_cfg(g, cfg, 0);
            hipEventRecord(e0, 0);
            const int it = 50;
            for (int r = 0; r < it; ++r) launch_with_cfg(g, cfg, 0);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // every configuration must reproduce the first one bit for bit (same k order, same MFMA shape)
            const size_t nb = std::min((size_t)sh.M * sh.N * 2, (size_t)64 << 20);
            static std::vector<char> ref, cur; cur.resize(nb);
            hipMemcpy(cur.data(), out, nb, hipMemcpyDeviceToHost);
            const char* tag = "";
            if (cfg == 2) ref = cur; else tag = memcmp(ref.data(), cur.data(), nb) ? " !!MISMATCH" : " ok";
            printf("  cfg%d %6.1f us %6.0f TF%s", cfg, ms / it * 1e3, 2.0 * sh.M * sh.N * sh.K / (ms / it * 1e-3) / 1e12, tag);
        }
        printf("\n");
#ifdef RT_PP_TIMING
        { GemmArgs g{}; g.A = A; g.W = W; g.out = out; g.zero = zero; g.mode = A_DENSE; g.epi = EPI_BF16;
          g.M = sh.M; g.N = sh.N; g.K = sh.K; g.lda = sh.K; g.ldw = sh.K; g.ldo = sh.N;
          launch_with_cfg(g, 6, 0); hipDeviceSynchronize();
          long long t[64]; hipMemcpyFromSymbol(t, HIP_SYMBOL(g_pp_times), sizeof(t));
          const double nph = 2.0 * ((sh.K + 63) / 64);
          for (int w = 0; w < 8; ++w) printf("    wave %d: dma issue %6.0f | barrier1 %5.0f | mfma+read issue %5.0f | ds wait %5.0f | barrier2 %5.0f   cycles per phase\n", w,
              t[w * 8] / nph, t[w * 8 + 1] / nph, t[w * 8 + 2] / nph, t[w * 8 + 3] / nph, t[w * 8 + 4] / nph); }
#endif
    }
    return 0;
}
