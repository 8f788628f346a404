// Build with -DRT_PROBE.  Patch-convolution kernel (conv3p) with 160 / 96 / 64-channel column tiles on the SD-v1.5 and SDXL shapes:
// time per launch and a bit-for-bit comparison of the outputs (the k order does not depend on the column tiling).
#include "../../rich-text-to-image_amd/csrc/gemm.hip"
#include <vector>
#include <cstring>
int main() {
    bf16_t *A, *W, *out[3], *zero; float* bias;
    const size_t na = (size_t)7 * 128 * 128 * 960, nw = (size_t)1280 * 9 * 2560, no = (size_t)7 * 128 * 128 * 320;
    hipMalloc(&A, na * 2); hipMalloc(&W, nw * 2); hipMalloc(&zero, 256); hipMemset(zero, 0, 256);
    for (auto& o : out) hipMalloc(&o, no * 2);
    hipMalloc(&bias, 1280 * 4); hipMemset(bias, 0, 1280 * 4);
    { std::vector<uint16_t> h(1 << 24); uint32_t x = 4242;
      for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 | ((x >> 9) & 0x83ff) | ((x >> 3) & 0x8000)); }
      for (size_t off = 0; off < na * 2; off += h.size() * 2) hipMemcpy((char*)A + off, h.data(), std::min(h.size() * 2, na * 2 - off), hipMemcpyHostToDevice);
      for (size_t off = 0; off < nw * 2; off += h.size() * 2) hipMemcpy((char*)W + off, h.data(), std::min(h.size() * 2, nw * 2 - off), hipMemcpyHostToDevice); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Cv { int B, H, W, Cin, Cout; } cs[] = {{3, 64, 64, 320, 320}, {5, 64, 64, 320, 320}, {3, 64, 64, 640, 320}, {3, 32, 32, 640, 640}, {5, 32, 32, 640, 640}, {3, 32, 32, 1280, 640},
        {7, 64, 64, 320, 320}, {7, 32, 32, 640, 640}, {3, 16, 16, 1280, 1280}, {5, 16, 16, 1280, 1280},
        {7, 32, 32, 1280, 1280}, {5, 32, 32, 1280, 1280}, {7, 64, 64, 640, 640}, {5, 64, 64, 640, 640}, {7, 128, 128, 320, 320}};
    std::vector<uint16_t> h0(no), h1(no);
    for (auto c : cs) {
        GemmArgs g{}; g.A = A; g.W = W; g.zero = zero; g.mode = A_CONV3; g.epi = EPI_BF16; g.bias = bias;
        g.M = c.B * c.H * c.W; g.N = c.Cout; g.K = 9 * c.Cin; g.ldw = g.K; g.ldo = c.Cout; g.rows_per_batch = c.H * c.W;
        g.Hin = g.Hout = c.H; g.Win = g.Wout = c.W; g.Cin = c.Cin;
        const int ntm = c.B * (c.H / 16) * (c.W / 16);
        printf("conv %dx%dx%dx%d->%d:", c.B, c.H, c.W, c.Cin, c.Cout);
        int vi = 0;
        for (int tn : {5, 3, 2}) {
            g_conv3p_tn = tn; g.out = out[vi++];
            for (int r = 0; r < 3; ++r) launch_conv3p<EPI_BF16, false>(g, 0);
            hipEventRecord(e0, 0);
            for (int r = 0; r < 20; ++r) launch_conv3p<EPI_BF16, false>(g, 0);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("  TN%d %4d wg %7.1f us %5.0f TF", tn, ntm * cdiv(c.Cout, 32 * tn), ms / 20 * 1e3, 2.0 * g.M * g.N * g.K / (ms / 20 * 1e-3) / 1e12);
        }
        const size_t nb = (size_t)g.M * g.N;
        hipMemcpy(h0.data(), out[0], nb * 2, hipMemcpyDeviceToHost);
        bool same = true;
        for (int v = 1; v < 3; ++v) { hipMemcpy(h1.data(), out[v], nb * 2, hipMemcpyDeviceToHost); same = same && !memcmp(h0.data(), h1.data(), nb * 2); }
        printf("  %s\n", same ? "bit-identical" : "DIFFERENT");
    }
    // 16x16 maps (SD-v1.5 level 2): the engine routes these through the split-K implicit GEMM (launch_gemm); compare with the patch kernel
    struct Cv2 { int B, H, W, Cin, Cout; } c2[] = {{3, 16, 16, 1280, 1280}, {5, 16, 16, 1280, 1280}, {3, 16, 16, 2560, 1280}, {5, 16, 16, 2560, 1280}, {3, 16, 16, 640, 1280}};
    float* of; hipMalloc(&of, (size_t)5 * 256 * 1280 * 4);
    for (auto c : c2) {
        GemmArgs g{}; g.A = A; g.W = W; g.zero = zero; g.mode = A_CONV3; g.epi = EPI_F32; g.bias = bias; g.out = of;
        g.M = c.B * c.H * c.W; g.N = c.Cout; g.K = 9 * c.Cin; g.ldw = g.K; g.ldo = c.Cout; g.rows_per_batch = c.H * c.W;
        g.Hin = g.Hout = c.H; g.Win = g.Wout = c.W; g.Cin = c.Cin;
        g.split_tiles = cdiv(c.H * c.W, 128) * cdiv(c.Cout, 128);
        printf("conv %dx%dx%dx%d->%d:", c.B, c.H, c.W, c.Cin, c.Cout);
        for (int variant = 0; variant < 3; ++variant) {
            g_conv3p_tn = variant == 1 ? 2 : 3;
            auto run = [&]() { if (variant == 0) launch_gemm(g, 0); else launch_conv3p<EPI_F32, false>(g, 0); };
            for (int r = 0; r < 3; ++r) run();
            hipEventRecord(e0, 0);
            for (int r = 0; r < 20; ++r) run();
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("  %s %7.1f us %5.0f TF", variant == 0 ? "split-K (engine route)" : variant == 1 ? "patch TN2" : "patch TN3", ms / 20 * 1e3, 2.0 * g.M * g.N * g.K / (ms / 20 * 1e-3) / 1e12);
        }
        printf("\n");
    }
    return 0;
}
