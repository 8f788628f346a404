// Occupancy staircase of the self-attention kernel: time vs number of workgroups (N = 1024: 8 workgroups per (stream, head)).
// Round 3 result (profiles/r3_attn_occupancy_staircase.txt): there is NO staircase - 256 WGs (one per CU) take 20.5 us, 1024 (four per CU,
// the VGPR limit) 50.5 us, 2048 83.6 us: a workgroup alone on a CU runs 2.5x faster than with three neighbours, so the dispatcher's
// dynamic refill absorbs "1.09 rounds"; cutting the last query blocks into key ranges to fill the tail was measured SLOWER
// (N = 4096: 338 -> 401 us) and removed.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -DRT_PROBE tools/probes/attn_occ.hip -o tools/probes/attn_occ
#include "../../rich-text-to-image_amd/csrc/attention.hip"
#include <vector>
int main() {
    const int DP = 64, N = 1024;
    bf16_t *Q, *K, *VT, *O;
    const size_t rows = 8 * N, ld = 40 * DP;
    hipMalloc(&Q, rows * ld * 2); hipMalloc(&K, rows * ld * 2); hipMalloc(&VT, rows * ld * 2); hipMalloc(&O, rows * ld * 2);
    { std::vector<uint16_t> h(rows * ld); uint32_t x = 777;
      for (bf16_t* dst : {Q, K, VT}) { for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 | ((x >> 9) & 0x83ff) | ((x >> 3) & 0x8000)); }
        hipMemcpy(dst, h.data(), h.size() * 2, hipMemcpyHostToDevice); } }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int H : {4, 8, 12, 14, 16, 17, 18, 20, 24, 28, 32, 36, 40}) {
        const int B = 8;
        AttnArgs a{}; a.Q = Q; a.ldq = H * DP; a.K = K; a.ldk = H * DP; a.VT = VT; a.ldvt = B * N; a.O = O; a.ldo = H * DP;
        for (int b = 0; b < B; ++b) { a.q_src[b] = a.k_src[b] = a.v_src[b] = b; a.wset[b] = 0; }
        a.B = B; a.H = H; a.N = N; a.NK = N; a.nk_valid = N; a.DP = DP; a.cross = 0;
        for (int r = 0; r < 3; ++r) launch_attention(a, 0);
        hipEventRecord(e0, 0);
        for (int r = 0; r < 20; ++r) launch_attention(a, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%5d WGs (%.2f per CU): %7.1f us  %6.0f TF\n", B * H * 8, B * H * 8 / 256.0, ms / 20 * 1e3, 4.0 * B * H * (double)N * N * 64 / (ms / 20 * 1e-3) / 1e12);
    }
    return 0;
}
