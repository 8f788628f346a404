// Stand-alone GEMM timing (hipEvents) for tile-configuration / main-loop experiments.  Build variants with -D flags.
#include "../../rich-text-to-image_amd/csrc/gemm.hip"
#include <vector>
#include <algorithm>
int main(int argc, char** argv) {
    struct Shape { int M, N, K; } shapes[] = {{256, 160, 64}, {7168, 1280, 64}, {7168, 1280, 128}, {7168, 1280, 320}, {7168, 1280, 640}, {7168, 1280, 1280}, {7168, 1280, 5120}, {7168, 10240, 1280}, {7168, 2560, 1280}, {28672, 640, 2560}, {8192, 8192, 8192}};
    bf16_t *A, *W, *out, *zero;
    hipMalloc(&A, (size_t)28672 * 8192 * 2); hipMalloc(&W, (size_t)10240 * 8192 * 2); hipMalloc(&out, (size_t)28672 * 10240 * 2); hipMalloc(&zero, 256);
    // pseudo-random bf16 fill (values ~ +-1): data-dependent power/clock effects matter (guide 5.4 rule 25)
    { std::vector<uint16_t> h(1 << 24); uint32_t x = 12345; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 | ((x >> 9) & 0x83ff) | ((x >> 3) & 0x8000)); }
      for (size_t off = 0; off < (size_t)28672 * 8192 * 2; off += h.size() * 2) hipMemcpy((char*)A + off, h.data(), std::min(h.size() * 2, (size_t)28672 * 8192 * 2 - off), hipMemcpyHostToDevice);
      for (size_t off = 0; off < (size_t)10240 * 8192 * 2; off += h.size() * 2) hipMemcpy((char*)W + off, h.data(), std::min(h.size() * 2, (size_t)10240 * 8192 * 2 - off), hipMemcpyHostToDevice); }
    hipMemset(zero, 0, 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (auto sh : shapes) {
        printf("%5dx%5dx%4d:", sh.M, sh.N, sh.K);
        for (int cfg : {0, 2, 3, 4}) {
            GemmArgs g{}; g.A = A; g.W = W; g.out = out; g.zero = zero; g.mode = A_DENSE; g.epi = EPI_BF16;
            g.M = sh.M; g.N = sh.N; g.K = sh.K; g.lda = sh.K; g.ldw = sh.K; g.ldo = sh.N;
            for (int r = 0; r < 3; ++r) launch_with_cfg(g, cfg, 0);
            hipEventRecord(e0, 0);
            const int it = 50;
            for (int r = 0; r < it; ++r) launch_with_cfg(g, cfg, 0);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("  cfg%d %6.1f us %6.0f TF", cfg, ms / it * 1e3, 2.0 * sh.M * sh.N * sh.K / (ms / it * 1e-3) / 1e12);
        }
        printf("\n");
    }
    return 0;
}
