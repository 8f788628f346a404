// Fused to_q + cross-attention kernel (gemm16.hip, EPI_XATTN) against the plain to_q GEMM on the two SDXL shapes, with the in-kernel
// stamp breakdown (-DRT_G16_TIMING): entry | first tile | K loop | Q -> LDS | 5 attention phases | barrier | stores issued | retired.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRT_G16_TIMING tools/probes/xattn_bench.hip rich-text-to-image_amd/csrc/gemm16.hip -o tools/probes/xattn_bench
#include "../../rich-text-to-image_amd/csrc/common.h"
#include <vector>
#include <cstdio>
#include <algorithm>
void gemm16_read_times(long long* dst, int n);
void gemm16_read_xa_segments(long long* dst, int n);

int main() {
    const size_t AE = (size_t)28672 * 1280;
    bf16_t *A, *W, *Kc, *VT, *O, *zero;
    hipMalloc(&A, AE * 2); hipMalloc(&W, (size_t)1280 * 1280 * 2); hipMalloc(&Kc, (size_t)8 * 96 * 1280 * 2); hipMalloc(&VT, (size_t)1280 * 8 * 96 * 2);
    hipMalloc(&O, AE * 2); hipMalloc(&zero, 256); hipMemset(zero, 0, 256);
    {
        std::vector<uint16_t> h(1 << 22); uint32_t x = 777;
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3800 | ((x >> 9) & 0x83ff) | ((x >> 3) & 0x8000)); }
        for (size_t off = 0; off < AE * 2; off += h.size() * 2) hipMemcpy((char*)A + off, h.data(), std::min(h.size() * 2, AE * 2 - off), hipMemcpyHostToDevice);
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3400 | ((x >> 9) & 0x83ff) | ((x >> 3) & 0x8000)); }
        hipMemcpy(W, h.data(), (size_t)1280 * 1280 * 2, hipMemcpyHostToDevice);
        hipMemcpy(Kc, h.data(), (size_t)8 * 96 * 1280 * 2, hipMemcpyHostToDevice);
        hipMemcpy(VT, h.data() + 4096, (size_t)1280 * 8 * 96 * 2, hipMemcpyHostToDevice);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct { const char* name; int tokens, C, H; } shapes[] = {{"B 1024 x 1280", 1024, 1280, 20}, {"A 4096 x 640", 4096, 640, 10}};
    for (auto& sh : shapes) {
        const int B = 7, M = B * sh.tokens, HD = sh.H * 64;
        GemmArgs g{}; g.A = A; g.W = W; g.out = O; g.zero = zero; g.mode = A_DENSE; g.epi = EPI_XATTN; g.M = M; g.N = HD; g.K = sh.C; g.lda = sh.C; g.ldw = sh.C; g.ldo = HD;
        g.rows_per_stream = sh.tokens; g.xa_k = Kc; g.xa_vt = VT; g.xa_ldk = HD; g.xa_ldvt = 8 * 96; g.xa_tokens = sh.tokens; g.xa_nk_valid = 77;
        const int pr[7] = {0, 4, 0, 4, 1, 2, 3};
        for (int b = 0; b < B; ++b) { g.xa_prompt[b] = pr[b]; g.xa_wset[b] = -1; }
        GemmArgs q = g; q.epi = EPI_BF16;
        int wst = 0; const int v = gemm16_pick(q, 0, &wst);
        auto timeit = [&](auto fn) {
            for (int i = 0; i < 5; ++i) fn();
            hipEventRecord(e0);
            for (int i = 0; i < 50; ++i) fn();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 50 * 1e3;
        };
        const double us_q = timeit([&] { launch_gemm16_variant(q, v, wst, 0); });
        const double us_f = timeit([&] { launch_xattn_fused(g, 0); });
        const int nwg = (M / 128) * (HD / 320);
        std::vector<long long> t((size_t)nwg * 8);
        gemm16_read_times(t.data(), nwg * 8);
        // slots: 0 entry, 1 first tile, 2 loop end, 3 Q in LDS, 5 phases done, 6 final barrier, 7 stores issued, 4 stores retired
        const int order[8] = {0, 1, 2, 3, 5, 6, 7, 4};
        double seg[7] = {0, 0, 0, 0, 0, 0, 0}, tot = 0;
        long long first = t[0], last = 0;
        for (int w = 0; w < nwg; ++w) {
            for (int i = 0; i < 7; ++i) seg[i] += (double)(t[w * 8 + order[i + 1]] - t[w * 8 + order[i]]);
            tot += (double)(t[w * 8 + 4] - t[w * 8 + 0]);
            first = std::min(first, t[w * 8 + 0]); last = std::max(last, t[w * 8 + 4]);
        }
        printf("%s: to_q alone (variant %d) %.1f us | fused to_q + attention %.1f us, %d workgroups\n", sh.name, v, us_q, us_f, nwg);
        {
            std::vector<long long> sg((size_t)nwg * 8);
            gemm16_read_xa_segments(sg.data(), nwg * 8);
            double a[5] = {0, 0, 0, 0, 0};
            for (int w = 0; w < nwg; ++w) for (int i = 0; i < 5; ++i) a[i] += (double)sg[w * 8 + i];
            printf("   attention phases, wave 0, cycles summed over the 5 heads (mean): wait + refill issue %.0f | Q/K reads + QK^T %.0f | softmax %.0f | V^T reads + PV %.0f | O write %.0f\n",
                   a[0] / nwg, a[1] / nwg, a[2] / nwg, a[3] / nwg, a[4] / nwg);
        }
        printf("   cycles per workgroup (mean): prologue %.0f | K loop %.0f | Q->LDS + KV2 issue %.0f | 5 phases %.0f | barrier %.0f | store issue %.0f | retire %.0f | total %.0f ; first entry -> last retire %lld\n",
               seg[0] / nwg, seg[1] / nwg, seg[2] / nwg, seg[3] / nwg, seg[4] / nwg, seg[5] / nwg, seg[6] / nwg, tot / nwg, last - first);
    }
    return 0;
}
