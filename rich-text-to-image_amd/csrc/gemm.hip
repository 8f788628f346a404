// bf16 MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950.
//
//   C[M,N] = A[M,K] * W[N,K]^T (+bias, +epilogue)       fp32 accumulate
//
// Replaces every nn.Linear / nn.Conv2d of the reference UNet (models/attention.py:209-304 GEGLU FF,
// models/attention_processor.py:137-152 q/k/v/out projections, models/resnet.py:505-560 conv1/conv2/
// shortcut, models/resnet.py:103-222 up/down-sample convs, models/transformer_2d.py:139-177 proj_in/out).
//
// Structure (CDNA4): 128x128x64 block tile, 4 wavefronts (2x2), each wave 64x64 = 2x2 tiles of
// v_mfma_f32_32x32x16_bf16; both operands are staged HBM -> LDS with global_load_lds_dwordx4
// (16 B/lane, no VGPR round trip) into a double buffer; the LDS image is XOR-swizzled on 16-B slots
// (slot ^= (row>>1)&7) by permuting the per-lane *source* address, so the ds_read_b128 fragment reads
// are bank-conflict free (cdna guide 5.4 rule 21 / T2).  For convolutions the A operand is gathered on
// the fly from the NHWC bf16 activation (im2col never materialised): K index = tap*Cin + c.
#include "common.h"

#define BM 128
#define BN 128
#define BK 64
#define NTHREADS 256

template <int MODE>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // stage s: A at smem + s*STAGE, B at smem + s*STAGE + BM*128
    constexpr int STAGE = (BM + BN) * BK * 2;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- XCD-aware bijective block remap (each XCD gets a contiguous run of tiles)
    const int ntn = (p.N + BN - 1) / BN;
    const int ntm = (p.M + BM - 1) / BM;
    const int nwg = ntm * ntn;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // grouped ordering: 8 tile-rows x all tile-columns per group, column-major inside the group, so
    // the ~64 tiles resident on one XCD share 8 A panels and 8 W panels (fits the 4 MiB L2)
    const int gsz = 8 * ntn;
    const int first_m = (bid / gsz) * 8;
    const int gm = (ntm - first_m) < 8 ? (ntm - first_m) : 8;
    const int tm = first_m + (bid % gsz) % gm, tn = (bid % gsz) / gm;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- loader geometry: instruction i, wave w moves LDS chunks [(i*4+w)*64, +64): 8 rows x 8 slots
    const int lrow = lane >> 3;                          // row within the 8-row group
    const int pslot = lane & 7;                          // physical 16-B slot
    // row = (i*4+wave)*8 + lrow ; (row>>1)&7 is independent of i
    const int lslot = pslot ^ ((((wave & 1) << 2) | (lrow >> 1)) & 7);   // logical slot (k offset /8)

    const bf16_t* a_ptr[BM / 32];
    int cy[BM / 32], cx[BM / 32];
    const bf16_t* b_ptr[BN / 32];
#pragma unroll
    for (int i = 0; i < BM / 32; ++i) {
        int row = m0 + (i * 4 + wave) * 8 + lrow;
        if (row >= p.M) row = p.M - 1;
        if (MODE == A_DENSE) {
            a_ptr[i] = p.A + (size_t)row * p.lda;
            cy[i] = cx[i] = 0;
        } else {
            const int b = row / p.rows_per_batch;
            const int pix = row - b * p.rows_per_batch;
            const int y = pix / p.Wout, x = pix - y * p.Wout;
            a_ptr[i] = p.A + (size_t)b * p.Hin * p.Win * p.Cin;
            cy[i] = y; cx[i] = x;
        }
    }
#pragma unroll
    for (int i = 0; i < BN / 32; ++i) {
        int row = n0 + (i * 4 + wave) * 8 + lrow;
        if (row >= p.N) row = p.N - 1;
        b_ptr[i] = p.W + (size_t)row * p.ldw;
    }

    auto stage = [&](int s, int k0) {
        char* sa = smem + s * STAGE;
        char* sb = sa + BM * BK * 2;
        const int k = k0 + lslot * 8;
        const bool kin = k < p.K;
        if (MODE == A_DENSE) {
#pragma unroll
            for (int i = 0; i < BM / 32; ++i) {
                const bf16_t* src = kin ? a_ptr[i] + k : p.zero;
                glds16(src, sa + (i * 4 + wave) * 1024);
            }
        } else {
            int tap = 0, c = 0, ky = 0, kx = 0;
            if (kin) { tap = k / p.Cin; c = k - tap * p.Cin; ky = tap / 3; kx = tap - ky * 3; }
#pragma unroll
            for (int i = 0; i < BM / 32; ++i) {
                int yy, xx;
                bool ok = kin;
                if (MODE == A_CONV3) {
                    yy = cy[i] + ky - 1; xx = cx[i] + kx - 1;
                    ok = ok && yy >= 0 && yy < p.Hin && xx >= 0 && xx < p.Win;
                } else if (MODE == A_CONV3_S2) {
                    yy = cy[i] * 2 + ky - 1; xx = cx[i] * 2 + kx - 1;
                    ok = ok && yy >= 0 && yy < p.Hin && xx >= 0 && xx < p.Win;
                } else {   // nearest 2x upsample fused into the gather: conv runs on the (2Hin x 2Win) grid
                    yy = cy[i] + ky - 1; xx = cx[i] + kx - 1;
                    ok = ok && yy >= 0 && yy < 2 * p.Hin && xx >= 0 && xx < 2 * p.Win;
                    yy >>= 1; xx >>= 1;
                }
                const bf16_t* src = ok ? a_ptr[i] + ((size_t)yy * p.Win + xx) * p.Cin + c : p.zero;
                glds16(src, sa + (i * 4 + wave) * 1024);
            }
        }
#pragma unroll
        for (int i = 0; i < BN / 32; ++i) {
            const bf16_t* src = kin ? b_ptr[i] + k : p.zero;
            glds16(src, sb + (i * 4 + wave) * 1024);
        }
    };

    // ---- compute geometry: wave (wm, wn) owns a 64x64 sub-tile
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int a_off[2], b_off[2], a_key[2], b_key[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra = wm * 64 + i * 32 + l31;
        const int rb = wn * 64 + i * 32 + l31;
        a_off[i] = ra * 128; a_key[i] = (ra >> 1) & 7;
        b_off[i] = BM * BK * 2 + rb * 128; b_key[i] = (rb >> 1) & 7;
    }

    const int nk = (p.K + BK - 1) / BK;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, (kt + 1) * BK);
        const char* sbase = smem + cur * STAGE;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = *(const bf16x8*)(sbase + a_off[i] + (((ks * 2 + hi) ^ a_key[i]) << 4));
                fb[i] = *(const bf16x8*)(sbase + b_off[i] + (((ks * 2 + hi) ^ b_key[i]) << 4));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // prefetched tile landed in LDS
        __syncthreads();                                    // ... for every wave; also fences the buffer swap
    }

    // ---- epilogue. C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int col_in_wave = l31;
    if (p.epi == EPI_GEGLU) {
        // packed weight rows: per 64-column block [32 value | 32 gate] (see engine pack_geglu)
        const int oc = ((n0 + wn * 64) >> 1) + col_in_wave;
        const int cv = n0 + wn * 64 + col_in_wave, cg = cv + 32;
        if (cg < p.N) {
            const float bv = p.bias ? p.bias[cv] : 0.f, bg = p.bias ? p.bias[cg] : 0.f;
            bf16_t* out = (bf16_t*)p.out;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (row < p.M) {
                        const float v = acc[i][0][r] + bv, g = acc[i][1][r] + bg;
                        const float ge = 0.5f * g * (1.f + erff(g * 0.70710678118654752440f));
                        out[(size_t)row * p.ldo + oc] = f32_to_bf16(v * ge);
                    }
                }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + col_in_wave;
        if (col >= p.N) continue;
        const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (row >= p.M) continue;
                float v = acc[i][j][r] + bv;
                if (p.epi == EPI_BF16) {
                    ((bf16_t*)p.out)[(size_t)row * p.ldo + col] = f32_to_bf16(v);
                } else if (p.epi == EPI_F32) {
                    if (p.res) v += p.res[(size_t)row * p.ldres + col];
                    ((float*)p.out)[(size_t)row * p.ldo + col] = v;
                } else {   // EPI_BF16_TEMB
                    v += p.temb[(size_t)(row / p.rows_per_batch) * p.temb_ld + col];
                    ((bf16_t*)p.out)[(size_t)row * p.ldo + col] = f32_to_bf16(v);
                }
            }
    }
}

void launch_gemm(const GemmArgs& a, hipStream_t st) {
    RT_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty problem");
    RT_REQUIRE(a.K % 8 == 0 && a.ldw % 8 == 0, "gemm: K and ldw must be multiples of 8");
    RT_REQUIRE(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W & 15) == 0, "gemm: operands must be 16-B aligned");
    if (a.mode == A_DENSE) RT_REQUIRE(a.lda % 8 == 0, "gemm: lda must be a multiple of 8");
    else RT_REQUIRE(a.Cin % 8 == 0 && a.K == 9 * a.Cin && a.rows_per_batch == a.Hout * a.Wout, "conv: bad geometry");
    if (a.epi == EPI_GEGLU) RT_REQUIRE(a.N % 64 == 0, "geglu: N must be a multiple of 64");
    const int ntm = cdiv(a.M, BM), ntn = cdiv(a.N, BN);
    const size_t lds = 2 * (BM + BN) * BK * 2;
    dim3 grid(ntm * ntn), block(NTHREADS);
    static bool attr_set = false;
    if (!attr_set) {
        HIP_CHECK(hipFuncSetAttribute((const void*)gemm_kernel<A_DENSE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_CHECK(hipFuncSetAttribute((const void*)gemm_kernel<A_CONV3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_CHECK(hipFuncSetAttribute((const void*)gemm_kernel<A_CONV3_S2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_CHECK(hipFuncSetAttribute((const void*)gemm_kernel<A_CONV3_UP2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    switch (a.mode) {
        case A_DENSE: hipLaunchKernelGGL(gemm_kernel<A_DENSE>, grid, block, lds, st, a); break;
        case A_CONV3: hipLaunchKernelGGL(gemm_kernel<A_CONV3>, grid, block, lds, st, a); break;
        case A_CONV3_S2: hipLaunchKernelGGL(gemm_kernel<A_CONV3_S2>, grid, block, lds, st, a); break;
        case A_CONV3_UP2: hipLaunchKernelGGL(gemm_kernel<A_CONV3_UP2>, grid, block, lds, st, a); break;
        default: throw rt_error(RT_E_INVALID, "gemm: bad mode");
    }
    HIP_CHECK(hipGetLastError());
}
