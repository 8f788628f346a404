// Build with -DRT_PROBE.  Stand-alone self-attention timing (+ per-segment s_memtime breakdown with -DRT_ATTN_TIMING) for the two SDXL shapes.
#include "../../rich-text-to-image_amd/csrc/attention.hip"
#include <vector>
int main() {
    struct Sh { int B, H, N; } shs[] = {{7, 10, 4096}, {7, 20, 1024}};
    const int DP = 64;
    bf16_t *Q, *K, *VT, *O;
    const size_t rows = 7 * 4096, ld = 20 * DP;
    hipMalloc(&Q, rows * ld * 2); hipMalloc(&K, rows * ld * 2); hipMalloc(&VT, rows * ld * 2); hipMalloc(&O, rows * ld * 2);
    { std::vector<uint16_t> h(rows * ld); uint32_t x = 777;
      for (bf16_t* dst : {Q, K, VT}) { for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 | ((x >> 9) & 0x83ff) | ((x >> 3) & 0x8000)); }
        hipMemcpy(dst, h.data(), h.size() * 2, hipMemcpyHostToDevice); } }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode : {1, 1, 3, 4, 5, 1, 3, 4, 5, 1, 3, 4, 5})        // 0: round-2 softmax (v_sub), 1: -m folded into the MFMAs (FOLD), 2: FOLD, 8 waves,
                                                                    // 3: FOLD + s_setprio 2 in the MFMA phases, 4: ... in the softmax phases instead, 5: fixed priority per wave
    for (auto sh : shs) {
        const int nw = mode == 2 ? 8 : 4;
        g_attn_nw = nw; g_attn_nofold = mode == 0; g_attn_prio = mode >= 3 ? mode - 2 : 0;
        AttnArgs a{}; a.Q = Q; a.ldq = sh.H * DP; a.K = K; a.ldk = sh.H * DP; a.VT = VT; a.ldvt = sh.B * sh.N; a.O = O; a.ldo = sh.H * DP;
        for (int b = 0; b < sh.B; ++b) { a.q_src[b] = a.k_src[b] = a.v_src[b] = b; a.wset[b] = 0; }
        a.B = sh.B; a.H = sh.H; a.N = sh.N; a.NK = sh.N; a.nk_valid = sh.N; a.DP = DP; a.cross = 0;
        for (int r = 0; r < 3; ++r) launch_attention(a, 0);
        hipEventRecord(e0, 0);
        for (int r = 0; r < 20; ++r) launch_attention(a, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("self-attn mode %d %d waves/WG B%d H%d N%d d64: %7.1f us %6.0f TF\n", mode, nw, sh.B, sh.H, sh.N, ms / 20 * 1e3, 4.0 * sh.B * sh.H * (double)sh.N * sh.N * 64 / (ms / 20 * 1e-3) / 1e12);
#ifdef RT_ATTN_TIMING
        long long t[32]; hipMemcpyFromSymbol(t, HIP_SYMBOL(g_attn_times), sizeof(t));
        const double nt = sh.N / 64.0;
        for (int w = 0; w < 4; ++w) printf("   wave %d: stage %5.0f | QK %5.0f | softmax %5.0f | PV %5.0f | vmcnt %5.0f | barrier %5.0f  cycles per key tile\n", w,
            t[w * 8] / nt, t[w * 8 + 1] / nt, t[w * 8 + 2] / nt, t[w * 8 + 3] / nt, t[w * 8 + 4] / nt, t[w * 8 + 5] / nt);
#endif
    }
    return 0;
}
