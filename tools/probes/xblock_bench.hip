// The one-launch cross-attention block (xblock.hip) at SDXL shape A (7 streams x 4096 tokens x 640 channels) with the in-kernel stamp
// breakdown (-DRT_XB_TIMING): entry | x + tile 0 landed | to_q (20 tiles) | attention (10 heads) | to_out chunk 0 | epilogue 0 + to_out
// chunk 1 | epilogue 1 | stores retired.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRT_XB_TIMING tools/probes/xblock_bench.hip rich-text-to-image_amd/csrc/xblock.hip -o tools/probes/xblock_bench
#include "../../rich-text-to-image_amd/csrc/common.h"
#include <vector>
#include <cstdio>
#include <algorithm>
void xblock_read_times(long long* dst, int n);

int main() {
    const int B = 7, N = 4096, C = 640, H = 10, M = B * N;
    bf16_t *x, *wq, *wo, *Kc, *VT; f16_t *res, *out; float* bo;
    hipMalloc(&x, (size_t)M * C * 2); hipMalloc(&wq, (size_t)C * C * 2); hipMalloc(&wo, (size_t)C * C * 2); hipMalloc(&Kc, (size_t)8 * 96 * C * 2);
    hipMalloc(&VT, (size_t)C * 8 * 96 * 2); hipMalloc(&res, (size_t)M * C * 2); hipMalloc(&out, (size_t)M * C * 2); hipMalloc(&bo, C * 4);
    hipMemset(bo, 0, C * 4); hipMemset(res, 0, (size_t)M * C * 2);
    {
        std::vector<uint16_t> h(1 << 22); uint32_t s = 777;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x3800 | ((s >> 9) & 0x83ff) | ((s >> 3) & 0x8000)); }
        for (size_t off = 0; off < (size_t)M * C * 2; off += h.size() * 2) hipMemcpy((char*)x + off, h.data(), std::min(h.size() * 2, (size_t)M * C * 2 - off), hipMemcpyHostToDevice);
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x3000 | ((s >> 9) & 0x83ff) | ((s >> 3) & 0x8000)); }
        hipMemcpy(wq, h.data(), (size_t)C * C * 2, hipMemcpyHostToDevice); hipMemcpy(wo, h.data() + 999, (size_t)C * C * 2, hipMemcpyHostToDevice);
        hipMemcpy(Kc, h.data() + 5000, (size_t)8 * 96 * C * 2, hipMemcpyHostToDevice); hipMemcpy(VT, h.data() + 70000, (size_t)C * 8 * 96 * 2, hipMemcpyHostToDevice);
    }
    XBlockArgs a{}; a.x = x; a.wq = wq; a.wo = wo; a.bo = bo; a.kc = Kc; a.vt = VT; a.res = res; a.out = out; a.ldk = C; a.ldvt = 8 * 96; a.ldres = C; a.ldo = C;
    a.M = M; a.tokens = N; a.nk_valid = 77; a.C = C; a.H = H;
    const int pr[7] = {0, 4, 0, 4, 1, 2, 3};
    for (int b = 0; b < B; ++b) { a.prompt[b] = pr[b]; a.wset[b] = -1; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) launch_xblock(a, 0);
    hipEventRecord(e0);
    for (int i = 0; i < 50; ++i) launch_xblock(a, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int nwg = M / 128;
    std::vector<long long> t((size_t)nwg * 8);
    xblock_read_times(t.data(), nwg * 8);
    double seg[7] = {0, 0, 0, 0, 0, 0, 0};
    long long first = t[0], last = 0;
    for (int w = 0; w < nwg; ++w) {
        for (int i = 0; i < 7; ++i) seg[i] += (double)(t[w * 8 + i + 1] - t[w * 8 + i]);
        first = std::min(first, t[w * 8]); last = std::max(last, t[w * 8 + 7]);
    }
    printf("xblock A (7 x 4096 x 640): %.1f us per launch, %d workgroups; 52.6 GFLOP executed -> %.0f TFLOP/s\n", ms / 50 * 1e3, nwg, 52.57e9 / (ms / 50 * 1e-3) / 1e12);
    const char* names[7] = {"prologue (x + tile 0)", "to_q (20 tiles)", "attention (10 heads)", "to_out chunk 0 (10 tiles)", "epilogue 0 + to_out chunk 1", "epilogue 1", "stores retire"};
    double tot = 0;
    for (int i = 0; i < 7; ++i) { printf("  %-32s %8.0f cycles avg per workgroup (s_memtime)\n", names[i], seg[i] / nwg); tot += seg[i] / nwg; }
    printf("  total %.0f; first entry -> last retire %lld\n", tot, last - first);
    return 0;
}
