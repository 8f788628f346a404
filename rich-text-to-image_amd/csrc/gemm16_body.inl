// Body of ONE output tile of the gemm16 family: included (textually) by gemm16_kernel - the kernel form every variant was measured
// in - and by gemm16_tile, the device-function form the grouped launch (gemm16_dual_kernel) calls.  Two inclusions instead of one
// function because hipcc allocates registers differently for the two forms: with the kernels calling a shared always-inline function the
// fp16-trunk epilogues of the 224 x 160 K-split kernels spilled their residual window (196 B of scratch per lane), the kernel form does not.
// In scope: template parameters MODE, EPI, TMW, TNW, WM, WN, WK, S, LNF; `p` (GemmArgs), `wstat`, `bid_in` (the workgroup's linear id inside
// its problem's grid).  LNF: LayerNorm fold (gemm16.hip, "LNF"): 0 none, 1 / 3 consumer (token rows / token columns), 2 producer of the partials.
    constexpr int NW = WM * WN * WK;                       // 8 waves (two per SIMD) or 4 waves (one per SIMD, 512 registers each)
    static_assert(NW == 8 || NW == 4, "4 or 8 waves");
    static_assert(WK == 1 || (WK == 2 && (S == 3 || S == 5)), "K split over at most two waves (3-slot ring; 5 slots for the 64-row tiles)");
    constexpr int BM = WM * TMW * 16, BN = WN * TNW * 16;
    constexpr int STAGE = (BM + BN) * 128;                 // bytes per ring slot: A rows then W rows, 128 B (64 k) each
    constexpr int GA = BM / 8, GB = BN / 8, GT = GA + GB;  // 8-row groups = one wave-wide LDS-DMA each
    constexpr int PW = (GT + NW - 1) / NW;                 // LDS-DMA pieces per wave and K tile
    constexpr int KS = 2 / WK;                             // 32-deep k steps per K tile and wave
    static_assert(EPI != EPI_GEGLU || TNW % 4 == 0, "GEGLU: a wave owns whole packed 64-column blocks [32 value | 32 gate]");
    static_assert((S - 2) * PW <= 63 && S >= 2 && (S <= 3 || (S == 5 && WK == 2)), "ring depth");
    constexpr bool LNC = LNF == RT_LNF_ROWS || LNF == RT_LNF_COLS;    // consumer of a folded LayerNorm: fp16 operands, epilogue correction
    static_assert(!LNC || (MODE == A_DENSE && (EPI == EPI_BF16 || EPI == EPI_GEGLU)), "LayerNorm fold: dense bf16-output / GEGLU consumers");
    static_assert(LNF != RT_LNF_EMIT || (MODE == A_DENSE && EPI == EPI_F16 && TNW == 5), "LayerNorm partials: fp16-trunk epilogue, 80-column wave tiles");
    static_assert(LNF != RT_LNF_COLS || EPI == EPI_BF16, "column statistics: the V^T form");
    constexpr int LN_NST = LNF == RT_LNF_COLS ? BN : BM;   // token rows (columns) of the tile whose (mu, rstd) the workgroup tabulates
    static_assert(!LNC || (LN_NST * 8 <= 2560 && LN_NST <= NW * 64 && BM <= 320 && BN <= 320 && BM <= NW * 64 && BN <= NW * 64), "statistics table: [320] (mu, rstd) | [320] s | [320] c");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    G16_T(0)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = WK == 2 ? (wave & 1) : 0;
    const int wq = WK == 2 ? (wave >> 1) : wave;
    const int wm = wq / WN, wn = wq % WN;

    // ---- tile mapping.  Default: XCD-aware bijective remap (each XCD gets a contiguous run of tiles) + groups of 4 tile rows x all
    // tile columns.  wstat (wide N, e.g. GEGLU): every XCD owns ntn/8 tile COLUMNS for all tile rows, so its share of W stays in its
    // 4 MiB L2 for the whole launch and W is fetched from HBM exactly once (the grouped order re-fetched the 26 MB GEGLU weight 7 x).
    const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM, nwg = ntm * ntn;
    int tm, tn;
    if (wstat == 1) {
        const int cpx = ntn >> 3, xcd = bid_in & 7, idx = bid_in >> 3;           // host guarantees ntn % 8 == 0
        tn = xcd * cpx + idx % cpx; tm = idx / cpx;
    } else {
        int bid = bid_in;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        const int GRP = wstat == 2 ? 8 : 4;                         // (wstat 2, round 6 A/B: groups of 8 tile rows - the group's W panels are re-fetched half as often)
        const int gsz = GRP * ntn;
        const int first_m = (bid / gsz) * GRP;
        const int gm = (ntm - first_m) < GRP ? (ntm - first_m) : GRP;
        tm = first_m + (bid % gsz) % gm; tn = (bid % gsz) / gm;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    // EPI_XATTN: the tile's stream (a 128-row tile never straddles: xa_tokens % 128 == 0), its prompt / multiplier set, first head
    const int xa_b = EPI == EPI_XATTN ? m0 / p.xa_tokens : 0;
    const int xa_prompt = EPI == EPI_XATTN ? p.xa_prompt[xa_b] : 0, xa_wset = EPI == EPI_XATTN ? p.xa_wset[xa_b] : -1;
    const int xa_head0 = n0 >> 6;
    // LNF consumers: thread t requests the partial sums of token row (column) t of the tile NOW, ahead of the prologue's LDS-DMA pieces
    // (VMEM loads retire in order: the first counted wait of the prologue covers them); they become the (-mu rstd, rstd) table behind
    // the first barrier.  The producer leaves ONE (sum, sum of squares) per token and column TILE of its grid, two tiles per float4,
    // pair-major [pair][token]: 1, 2 or 4 coalesced 16-B loads per thread (a token-major 16 x float2 layout held the prologue's pieces
    // back by 2.1 k cycles per tile: every load instruction touched 64 lines; profiles/r6_lnfold_probe_v1.txt).
    // ... and (s, c) of weight row t (V^T form: of the tile's weight rows; else: of its columns): staged in the same table, so that the
    // epilogue needs no registers for them before the K-split exchange has freed half of the accumulators
    float4 lnp[LNC ? 4 : 1];
    constexpr int LN_NW = LNF == RT_LNF_COLS ? BM : BN;    // weight rows of the tile
    float2 ln_scw = {0.f, 0.f};
    if constexpr (LNC) {
        int trow = (LNF == RT_LNF_COLS ? n0 : m0) + tid;
        const int lim = LNF == RT_LNF_COLS ? p.N : p.M;
        if (trow >= lim) trow = lim - 1;
        const float4* pp = (const float4*)p.ln_part + trow;
        lnp[0] = pp[0];
#pragma unroll
        for (int j = 1; j < 4; ++j) lnp[j] = j < p.ln_npair ? pp[(size_t)j * p.ln_ld] : float4{0.f, 0.f, 0.f, 0.f};
        int wrow = (LNF == RT_LNF_COLS ? m0 : n0) + tid;
        const int wlim = LNF == RT_LNF_COLS ? p.M : p.N;
        if (wrow >= wlim) wrow = wlim - 1;
        ln_scw = ((const float2*)p.ln_s)[wrow];                          // (s, c) interleaved
    }
    // ---- loader: piece i of this wave copies 8-row group g = i*NW + wave of the (A rows | W rows) list.  Buffer-descriptor LDS-DMA
    // (buffer_load_dwordx4 ... offen lds, guide T8): per piece ONE 32-bit VGPR byte offset, the K-tile offset is a scalar (soffset)
    // and the operand base sits in an SGPR descriptor - the flat form kept a 64-bit address per piece alive and spilled in the loop.
    const int lrow = lane >> 3, pslot = lane & 7;
    int voff[PW];                       // byte offset of this lane's 16-B chunk at k0 = 0 (conv: of its pixel's channel vector)
    int ldst[PW];                       // LDS byte offset of the group inside a ring slot (wave-uniform)
    bool pisA[PW];                      // wave-uniform
    int tapmask[MODE == A_CONV3 ? PW : 1];   // conv, A pieces: bit t set <=> tap t of this lane's pixel lies inside the image
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        int g = i * NW + wave; if (g > GT - 1) g = GT - 1;          // tail duplicates copy the same bytes to the same place
        const bool isA = g < GA;
        const int gl = isA ? g : g - GA;
        int row = (isA ? m0 : n0) + gl * 8 + lrow;
        const int lim = isA ? p.M : p.N;
        if (row >= lim) row = lim - 1;
        const int key = ((gl << 2) | (lrow >> 1)) & 7;              // (tile row >> 1) & 7
        if (MODE == A_CONV3 && isA) {
            const int b = row / p.rows_per_batch, pix = row - b * p.rows_per_batch;
            const int y = pix / p.Win, x = pix - y * p.Win;
            voff[i] = (row * p.Cin + ((pslot ^ key) << 3)) * 2;     // NHWC, stride 1: output pixel index == input pixel index
            int m = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                if (yy >= 0 && yy < p.Hin && xx >= 0 && xx < p.Win) m |= 1 << t;
            }
            tapmask[MODE == A_CONV3 ? i : 0] = m;
        } else {
            voff[i] = (row * (isA ? p.lda : p.ldw) + ((pslot ^ key) << 3)) * 2;
            if (MODE == A_CONV3) tapmask[MODE == A_CONV3 ? i : 0] = 0;
        }
        pisA[i] = isA;
        ldst[i] = (isA ? 0 : BM * 128) + gl * 1024;
    }
    // conv, precise VAE: three passes (A W, A_lo W, A W_lo) as one K loop of 3 x 9 x Cin / 64 tiles
    // ... and, round 4, the dense fp32-output kernels (the VAE's 512-wide projections, 1x1 shortcuts and attention products - the UNet's
    // dense launches are bf16 / fp16 / GEGLU epilogues and do not carry this code): the pass is a scalar advanced with the issued tiles
    constexpr bool TRIPLE_DENSE = MODE == A_DENSE && EPI == EPI_F32;
    const bool triple = (MODE == A_CONV3 || TRIPLE_DENSE) && p.A_lo != nullptr;
    const int nkp = p.K / BK16;                                      // K tiles of ONE pass
    const int nk = nkp * (triple ? 3 : 1);                           // host guarantees K % 128 == 0 / conv: 9 * Cin / 64, nk >= S + 1
    int d_kt = 0, d_pass = 0;                                        // dense triple: K tile inside the pass / pass of the NEXT tile to be issued
    // conv: (tap, chunk) of the NEXT K tile to be issued, carried as scalars (tiles are issued strictly in order)
    const int nch = MODE == A_CONV3 ? p.Cin / BK16 : 1;
    const unsigned a_range = MODE == A_CONV3 ? (unsigned)p.M * (unsigned)p.Cin * 2u : 0x7fffffffu;
    int is_tap = 0, is_chunk = 0, is_ky = 0, is_kx = 0, is_pass = 0;
    int is_pix = -p.Win - 1;                                         // pixel shift of the tap, advanced incrementally (a (ky, kx) product in the
                                                                     // loop made hipcc build a 9-entry table in scratch memory)
    auto stage_piece = [&](int t, int slot_off, int i) {             // piece i of K tile t -> ring slot at slot_off
        if (MODE == A_CONV3 && pisA[i]) {
            const int shift = is_pix * p.Cin * 2;                                       // scalar: (ky - 1) * Win + (kx - 1) pixels
            const int vo = ((tapmask[MODE == A_CONV3 ? i : 0] >> is_tap) & 1) ? voff[i] + shift : RT_G16_OOB;
            glds16_buf(is_pass == 1 ? p.A_lo : p.A, vo, is_chunk * (BK16 * 2), smem + slot_off + ldst[i], a_range);
        } else {
            // conv weights are packed [Cout][tap][Cin]: the K offset of tile (tap, chunk) is tap * Cin + chunk * 64 (= t * 64 in the
            // tap-major order)
            const int wk = MODE == A_CONV3 ? (is_tap * p.Cin + is_chunk * BK16) * 2 : (TRIPLE_DENSE ? d_kt : t) * (BK16 * 2);
            const void* base = pisA[i] ? (const void*)(TRIPLE_DENSE && d_pass == 1 ? p.A_lo : p.A)
                                       : (const void*)(((MODE == A_CONV3 && is_pass == 2) || (TRIPLE_DENSE && d_pass == 2)) ? p.W_lo : p.W);
            // (round 6, measured and dropped: the activation pieces of the W-stationary GEGLU as non-temporal loads, so that the A stream
            //  would not evict the XCD's 3.3 MB of W between rounds - the step got 1.8 ms SLOWER: the five column workgroups of a row panel
            //  then fetch A from the fabric one by one instead of sharing it through L2; profiles/r6_ab_geglu_nt_activation_stream.txt)
            glds16_buf(base, voff[i], wk, smem + slot_off + ldst[i]);
        }
    };
    // K-tile order of the convolution.  Round 3 ran (tap, chunk): for each tap the whole channel range of the tile's pixels streams
    // by, so the nine re-reads of a pixel are a full A panel apart (224 rows x Cin x 2 B = 573 KB per workgroup, 18 MB per XCD against
    // 4 MiB of L2): 615 - 936 MB per launch came over the fabric for 221 MB of operands (profiles/r3_pmc_traffic.json).  Round 4:
    // (chunk, tap) - the nine taps of one 64-channel chunk follow each other, the shifted rows they re-read are 28 KB per workgroup
    // and still in the vL1D / L2.  Same k order as the patch kernel of gemm.hip.  -DRT_G16_CONV_TAP_MAJOR builds the old order (A/B
    // timing: `make tapmajor` + RTDIFF_LIB_PATH).
    auto tile_issued = [&]() {                                       // every piece of a K tile went out: advance (tap, chunk)
        if (TRIPLE_DENSE) { if (++d_kt == nkp) { d_kt = 0; ++d_pass; } }
        if (MODE == A_CONV3) {
#ifdef RT_G16_CONV_TAP_MAJOR
            if (++is_chunk == nch) { is_chunk = 0; ++is_tap; if (++is_kx == 3) { is_kx = 0; ++is_ky; is_pix += p.Win - 2; } else ++is_pix;
                                      if (is_tap == 9) { is_tap = 0; is_kx = 0; is_ky = 0; is_pix = -p.Win - 1; ++is_pass; } }
#else
            if (++is_tap == 9) { is_tap = 0; is_kx = 0; is_pix = -p.Win - 1; if (++is_chunk == nch) { is_chunk = 0; ++is_pass; } }
            else if (++is_kx == 3) { is_kx = 0; is_pix += p.Win - 2; }
            else ++is_pix;
#endif
            (void)is_ky;
        }
    };

    // ---- fragments: lane (l15, q): row l15 of a 16-row tile, 16-B chunk c = 4*khalf + q of the 128-B row
    const int l15 = lane & 15, q4 = lane >> 4;
    const int key = (l15 >> 1) & 7;                                  // tile bases are multiples of 16 rows: the key depends on l15 only
    int aoff[KS], boff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int c = (WK == 2 ? kh : ks) * 4 + q4;
        aoff[ks] = (wm * TMW * 16 + l15) * 128 + ((c ^ key) << 4);
        boff[ks] = BM * 128 + (wn * TNW * 16 + l15) * 128 + ((c ^ key) << 4);
    }
    f32x4_t acc[TMW][TNW];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: all S slots in flight; tile 0 landed; fragments of k step 0 in registers
#pragma unroll
    for (int s = 0; s < S; ++s) {
#pragma unroll
        for (int i = 0; i < PW; ++i) stage_piece(s, s * STAGE, i);
        tile_issued();
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 1) * PW) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    G16_T(1)
    if constexpr (LNC) {
        // (mu, rstd) of the tile's tokens -> LDS behind the ring; read in the epilogue, many barriers later.  Fixed summation order.
        // Written by inline ds_write_b64: in front of a compiler-visible LDS store hipcc's waitcnt pass would put s_waitcnt vmcnt(0),
        // i.e. wait for every ring slot in flight (LABNOTES R4.1).
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1 += lnp[j].x; s2 += lnp[j].y; s1 += lnp[j].z; s2 += lnp[j].w; }
        const float mu = s1 * p.ln_inv_c;
        const float var = fmaxf(s2 * p.ln_inv_c - mu * mu, 0.f);
        const float rs = rsqrtf(var + p.ln_eps);
        const unsigned tbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(smem + S * STAGE);
        if (tid < LN_NST) {
            const unsigned long long w64 = ((unsigned long long)__float_as_uint(rs) << 32) | __float_as_uint(-mu * rs);     // (-mu rstd, rstd)
            asm volatile("ds_write_b64 %0, %1" ::"v"(tbase + (unsigned)tid * 8u), "v"(w64) : "memory");
        }
        if (tid < LN_NW) {
            asm volatile("ds_write_b32 %0, %1 offset:2560\n\tds_write_b32 %0, %2 offset:3840" ::"v"(tbase + (unsigned)tid * 4u), "v"(ln_scw.x), "v"(ln_scw.y) : "memory");
        }
    }
    bf16x8 fa[TMW], fb[2][TNW];
#pragma unroll
    for (int j = 0; j < TNW; ++j) fb[0][j] = *(const bf16x8*)(smem + boff[0] + j * 2048);
#pragma unroll
    for (int i = 0; i < TMW; ++i) fa[i] = *(const bf16x8*)(smem + aoff[0] + i * 2048);

    // One k step.  CUR: which W fragment set it multiplies.  KSI: k step inside the tile.  NEXT: 0 = no further step,
    // 1 = next step is in the same K tile, 2 = next step opens K tile t+1: wait for it (BEHIND = K tiles that may stay in flight
    // behind it) + barrier, then - REFILL - copy tile t+S into tile t's slot.
    // cur_off / nxt_off: LDS byte offsets of the slots of tiles t and t+1 (scalars carried by the loop; no modulo arithmetic in
    // the loop: hipcc's strength reduction turned `% 3` into per-fragment VGPR induction variables that spilled).
    auto kstep = [&](int t, int cur_off, int nxt_off, auto cur_c, auto ksi_c, auto next_c, auto refill_c, auto behind_c) {
        constexpr int CUR = decltype(cur_c)::value, KSI = decltype(ksi_c)::value, NEXT = decltype(next_c)::value;
        constexpr bool REFILL = decltype(refill_c)::value != 0;
        constexpr int BEHIND = decltype(behind_c)::value;
        constexpr int NKS = NEXT == 1 ? KSI + 1 : 0;                 // k step index of the next step inside its tile
        const int rd = __builtin_amdgcn_readfirstlane(NEXT == 2 ? nxt_off : cur_off);
        const char* na = smem + rd + aoff[NKS];
        const char* nbb = smem + rd + boff[NKS];
        // row 0 first: its operands were read a whole step ago, and the last fragment read of the previous step (issued behind
        // that step's last MFMA row) retires behind these MFMAs instead of in front of the barrier
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
#if RT_G16_ABLATE == 2
            asm volatile("" ::"v"(fb[CUR][j]), "v"(fa[0]));
#else
            acc[0][j] = g16_mma<LNC>(fb[CUR][j], fa[0], acc[0][j]);
#endif
        }
        if constexpr (NEXT == 2) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // every fragment read of tile t is retired: its slot may be refilled
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BEHIND * PW) : "memory");   // this wave's pieces of tile t+1 landed
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NEXT != 0) {
#pragma unroll
            for (int j = 0; j < TNW; ++j) fb[CUR ^ 1][j] = *(const bf16x8*)(nbb + j * 2048);
            fa[0] = *(const bf16x8*)(na);
        }
#pragma unroll
        for (int i = 1; i < TMW; ++i) {
#pragma unroll
            for (int j = 0; j < TNW; ++j) {
#if RT_G16_ABLATE == 2
                asm volatile("" ::"v"(fb[CUR][j]), "v"(fa[i]));
#else
                acc[i][j] = g16_mma<LNC>(fb[CUR][j], fa[i], acc[i][j]);
#endif
            }
            if constexpr (NEXT != 0) fa[i] = *(const bf16x8*)(na + i * 2048);
            if constexpr (NEXT == 2 && REFILL) {
                // LDS-DMA issue slots spread behind the MFMA rows (a piece costs its wave 60-200 cycles of issue under load)
                constexpr int PPR = (PW + TMW - 2) / (TMW - 1);      // pieces per MFMA row
#if RT_G16_ABLATE != 1
#pragma unroll
                for (int pc = 0; pc < PPR; ++pc) { const int pi = (i - 1) * PPR + pc; if (pi < PW) stage_piece(t + S, cur_off, pi); }
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (NEXT == 2 && REFILL) tile_issued();
    };
    using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>; using C2 = std::integral_constant<int, 2>;
    using CS2 = std::integral_constant<int, S - 2>;
    int cur_off = 0, nxt_off = STAGE;
    auto advance = [&]() { cur_off = nxt_off; nxt_off = nxt_off + STAGE == S * STAGE ? 0 : nxt_off + STAGE; };
    // Tiles t <= nk-S-1 refill their slot (steady state); the last S tiles do not, and the wait in front of tile t+1 allows exactly
    // the min(S-2, nk-2-t) younger tiles that were issued: no dummy copies, nothing to drain behind the loop.
    if constexpr (KS == 1) {
        // K-split class: CUR alternates per tile, so tiles run in pairs; the host guarantees an even tile count, S = 3, nk >= 4
        // (S = 5, round 6: the 64-row tiles of small batches - their K tile is 0.13 us of MFMA work against an L2 / fabric round trip of
        //  ~1.5 us, so the loop runs at (round trip) / (tiles in flight): four in flight instead of two; nk >= 8, same k order and bits)
        int t = 0;
        if constexpr (S == 5) {
            using C3 = std::integral_constant<int, 3>;
            for (; t + 8 <= nk; t += 2) {
                kstep(t, cur_off, nxt_off, C0{}, C0{}, C2{}, C1{}, C3{}); advance();
                kstep(t + 1, cur_off, nxt_off, C1{}, C0{}, C2{}, C1{}, C3{}); advance();
            }
            kstep(t, cur_off, nxt_off, C0{}, C0{}, C2{}, C1{}, C3{}); advance();         // t = nk-6: the last refill (tile nk-1)
            kstep(t + 1, cur_off, nxt_off, C1{}, C0{}, C2{}, C0{}, C3{}); advance();     // nk-5: tiles nk-3 .. nk-1 may stay in flight
            kstep(t + 2, cur_off, nxt_off, C0{}, C0{}, C2{}, C0{}, C2{}); advance();     // nk-4
            kstep(t + 3, cur_off, nxt_off, C1{}, C0{}, C2{}, C0{}, C1{}); advance();     // nk-3
            kstep(t + 4, cur_off, nxt_off, C0{}, C0{}, C2{}, C0{}, C0{}); advance();     // nk-2
            kstep(t + 5, cur_off, nxt_off, C1{}, C0{}, C0{}, C0{}, C0{});                // nk-1
        } else {
        for (; t + 6 <= nk; t += 2) {
            kstep(t, cur_off, nxt_off, C0{}, C0{}, C2{}, C1{}, CS2{}); advance();
            kstep(t + 1, cur_off, nxt_off, C1{}, C0{}, C2{}, C1{}, CS2{}); advance();
        }
        kstep(t, cur_off, nxt_off, C0{}, C0{}, C2{}, C1{}, CS2{}); advance();            // t = nk-4: the last refill (tile nk-1)
        kstep(t + 1, cur_off, nxt_off, C1{}, C0{}, C2{}, C0{}, C1{}); advance();         // nk-3: tile nk-1 may stay in flight
        kstep(t + 2, cur_off, nxt_off, C0{}, C0{}, C2{}, C0{}, C0{}); advance();         // nk-2
        kstep(t + 3, cur_off, nxt_off, C1{}, C0{}, C0{}, C0{}, C0{});                    // nk-1
        }
    } else {
        int t = 0;
        for (; t + S < nk; ++t) {
            kstep(t, cur_off, nxt_off, C0{}, C0{}, C1{}, C0{}, C0{});
            kstep(t, cur_off, nxt_off, C1{}, C1{}, C2{}, C1{}, CS2{}); advance();
        }
        if constexpr (S == 3) {                                                          // t = nk-3
            kstep(t, cur_off, nxt_off, C0{}, C0{}, C1{}, C0{}, C0{});
            kstep(t, cur_off, nxt_off, C1{}, C1{}, C2{}, C0{}, C1{}); advance(); ++t;
        }
        kstep(t, cur_off, nxt_off, C0{}, C0{}, C1{}, C0{}, C0{});                        // t = nk-2
        kstep(t, cur_off, nxt_off, C1{}, C1{}, C2{}, C0{}, C0{}); advance(); ++t;
        if constexpr (EPI == EPI_XATTN) {
            // nothing of the ring is in flight any more: K / V^T of the tile's first two heads, into the LDS beyond the ring,
            // land behind the last K tile's MFMAs
            xattn_stage_head(p, smem, XA_KVA, xa_head0, xa_prompt, wave, lane);
            xattn_stage_head(p, smem, XA_KVB, xa_head0 + 1, xa_prompt, wave, lane);
        }
        kstep(t, cur_off, nxt_off, C0{}, C0{}, C1{}, C0{}, C0{});                        // t = nk-1
        kstep(t, cur_off, nxt_off, C1{}, C1{}, C0{}, C0{}, C0{});
    }
    G16_T(2)
    if constexpr (EPI == EPI_XATTN) {
        xattn_tail<TMW, TNW, WN>(p, smem, acc, m0, n0, wm, wn, wave, lane, tid, xa_prompt, xa_head0, xa_wset);
        return;
    }

    // ---- which 16-row tiles this wave finishes: K-split: kh = 0 owns row tiles [0, H0), kh = 1 owns [H0, TMW)
    constexpr int H0 = WK == 2 ? (TMW + 1) / 2 : TMW;
    auto owned = [&](int i) { return WK == 1 || ((i < H0) == (kh == 0)); };           // wave-uniform
    constexpr bool F16 = EPI == EPI_F16;
    constexpr bool TEMB = EPI == EPI_BF16_TEMB;                      // bf16 out + the per-image time-embedding row (resnet.py:611-613)
    constexpr bool F32 = EPI == EPI_F32 || F16;                      // fp32 slab; F16: fp16 in HBM (output and residual), rounded once
    constexpr int ES = F32 ? 4 : 2;
    constexpr int TNO = EPI == EPI_GEGLU ? TNW / 2 : TNW;            // 16-column output tiles per wave
    const int NO = EPI == EPI_GEGLU ? (p.N >> 1) : p.N;
    const int wcol0 = n0 + wn * TNW * 16;                            // first (packed) column of this wave
    const int ocol0 = EPI == EPI_GEGLU ? (wcol0 >> 1) : wcol0;
    // fp16 residual of the owned tiles: requested NOW, ahead of the K-split exchange / the first slab transposes, so that its HBM
    // round trip is paid once and in the shadow of that work (issued tile by tile inside the store loop it was paid TMW times:
    // 14.5 k of the 51 k cycles of a 7168 x 1280 x 1280 launch, profiles/r3_gemm16_probe_v1_timing.txt)
    constexpr int IPR = TNO * 16 / 8;                                // 8-column items per row
    constexpr int NIT = (16 * IPR + 63) / 64;
    // a rolling window of RW owned tiles (static register indices: the s-th owned tile of a wave is tile s or H0 + s)
    constexpr int NOWN = WK == 2 ? H0 : TMW;                         // owned tiles of the kh = 0 half (kh = 1 owns TMW - H0 <= H0)
    constexpr int RW = NOWN < 4 ? NOWN : ((LNF == RT_LNF_EMIT && WK == 2) ? 3 : 4);   // (producer of the LayerNorm partials, K-split form: a window of three tiles keeps it free of scratch)
    // (producer of the LayerNorm partials: the window as ext-vector values that an empty asm pins at their use - see the store loop)
    typedef unsigned int g16_u4 __attribute__((ext_vector_type(4)));
    using RresT = std::conditional_t<LNF == RT_LNF_EMIT, g16_u4, uint4>;
    RresT rres[F16 ? RW : 1][F16 ? NIT : 1];
    auto load_res = [&](int s_own, int slot) {                       // residual of the s_own-th owned tile -> window slot (static)
        const int i = (WK == 2 && kh == 1) ? H0 + s_own : s_own;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * 64 + lane;
            const int r = idx / IPR, c8 = idx - r * IPR;
            int row = m0 + (wm * TMW + i) * 16 + r; if (row >= p.M) row = p.M - 1;
            int col = ocol0 + c8 * 8; if (col >= NO) col = 0;
            if (F16) rres[F16 ? slot : 0][F16 ? it : 0] = (r < 16 && i < TMW) ? *(const RresT*)((const f16_t*)p.res + (size_t)row * p.ldres + col) : RresT{0, 0, 0, 0};
        }
    };
    // K-split forms: ahead of the exchange (its round trip, 7 - 8 k cycles for the chip-wide 18 MB burst, hides behind the LDS pass:
    // 29.4 vs 31.6 us per 7168 x 1280 x 1280 launch); class A has no exchange to hide behind and measured better with the request
    // after the post-loop barrier (ff.net.2 28672 x 640 x 2560: 74.1 -> 69.7 us, to_out 640: 30.2 -> 28.6 us)
    constexpr bool RES_EARLY = WK == 2;
    if constexpr (F16 && RES_EARLY) {
        if (p.res) {
#pragma unroll
            for (int s_own = 0; s_own < RW; ++s_own) load_res(s_own, s_own);
        }
    }
    // bias of this wave's columns: requested here as well (the fragment registers of the main loop are dead), so that its round trip
    // (1.0-1.3 k cycles when it opened the epilogue, profiles/r3_gemm16_probe_v5_epilogue_breakdown.txt) hides behind the barrier / exchange
    float bias_v[TNO][4], bias_g[EPI == EPI_GEGLU ? TNO : 1][4];
    auto load_bias = [&]() {
#pragma unroll
    for (int t = 0; t < TNO; ++t) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { bias_v[t][e] = 0.f; if (EPI == EPI_GEGLU) bias_g[t][e] = 0.f; }
        if (p.bias && !LNC) {
            if constexpr (EPI == EPI_GEGLU) {
                // packed columns: per 64-block [32 value | 32 gate]; output tile t (16 columns) = value tile 4*(t/2) + t%2, gate + 2
                const int col = wcol0 + ((t >> 1) * 4 + (t & 1)) * 16 + 4 * q4;
                if (col + 32 + 4 <= p.N) {
                    const float4 b0 = *(const float4*)(p.bias + col), b1 = *(const float4*)(p.bias + col + 32);
                    bias_v[t][0] = b0.x; bias_v[t][1] = b0.y; bias_v[t][2] = b0.z; bias_v[t][3] = b0.w;
                    bias_g[t][0] = b1.x; bias_g[t][1] = b1.y; bias_g[t][2] = b1.z; bias_g[t][3] = b1.w;
                }
            } else {
                const int col = wcol0 + t * 16 + 4 * q4;
                if (col < p.N) {
                    const float4 b0 = *(const float4*)(p.bias + col);
                    bias_v[t][0] = b0.x; bias_v[t][1] = b0.y; bias_v[t][2] = b0.z; bias_v[t][3] = b0.w;
                }
            }
        }
    }
    };
    constexpr bool BIAS_EARLY = !F16 && MODE == A_DENSE;                                // (the fp16-trunk forms hold the residual window here: hoisting the bias too spills)
    if constexpr (BIAS_EARLY) load_bias();
    // raw barriers from here on: a __syncthreads() carries s_waitcnt vmcnt(0) and would wait for the residual round trip
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                    // every wave left the ring: exchange buffers + transpose slabs

    // ---- K-split reduction through LDS in ONE pass: every wave parks the partial sums of the row tiles its partner owns, one
    // barrier, every wave adds its partner's partials to the tiles it owns (kh = 0 + kh = 1 in that order in both halves).
    if constexpr (WK == 2) {
        static_assert((NW / 2) * TMW * TNW * 1024 <= S * STAGE, "exchange region");
        char* xr = smem + (size_t)wq * (TMW * TNW * 1024) + lane * 16;
#pragma unroll
        for (int i = 0; i < TMW; ++i) {
            if (owned(i)) continue;
#pragma unroll
            for (int j = 0; j < TNW; ++j) *(f32x4_t*)(xr + (i * TNW + j) * 1024) = acc[i][j];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < TMW; ++i) {
            if (!owned(i)) continue;
#pragma unroll
            for (int j = 0; j < TNW; ++j) {
                const f32x4_t o = *(const f32x4_t*)(xr + (i * TNW + j) * 1024);
                acc[i][j] = kh == 0 ? acc[i][j] + o : o + acc[i][j];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                // the exchange region becomes the transpose slabs
    }
    G16_T(3)
    // LNF consumers: s / c of this wave's columns out of the table behind the ring (c takes the place of the bias)
    float lns_v[LNF == RT_LNF_ROWS ? TNO : 1][4], lns_g[(LNF == RT_LNF_ROWS && EPI == EPI_GEGLU) ? TNO : 1][4];
    if constexpr (LNF == RT_LNF_ROWS) {
        const char* tb = smem + S * STAGE;
#pragma unroll
        for (int t = 0; t < TNO; ++t) {
            const int cl = wn * TNW * 16 + (EPI == EPI_GEGLU ? ((t >> 1) * 4 + (t & 1)) * 16 : t * 16) + 4 * q4;     // column inside the tile (GEGLU: packed value column)
            const float4 sv = *(const float4*)(tb + 2560 + cl * 4), cv = *(const float4*)(tb + 3840 + cl * 4);
            lns_v[t][0] = sv.x; lns_v[t][1] = sv.y; lns_v[t][2] = sv.z; lns_v[t][3] = sv.w;
            bias_v[t][0] = cv.x; bias_v[t][1] = cv.y; bias_v[t][2] = cv.z; bias_v[t][3] = cv.w;
            if constexpr (EPI == EPI_GEGLU) {
                constexpr int tg = EPI == EPI_GEGLU ? 1 : 0;
                const float4 sg = *(const float4*)(tb + 2560 + (cl + 32) * 4), cg = *(const float4*)(tb + 3840 + (cl + 32) * 4);
                lns_g[tg * t][0] = sg.x; lns_g[tg * t][1] = sg.y; lns_g[tg * t][2] = sg.z; lns_g[tg * t][3] = sg.w;
                bias_g[tg * t][0] = cg.x; bias_g[tg * t][1] = cg.y; bias_g[tg * t][2] = cg.z; bias_g[tg * t][3] = cg.w;
            }
        }
    }
    if constexpr (F16 && !RES_EARLY) {
        if (p.res) {
#pragma unroll
            for (int s_own = 0; s_own < RW; ++s_own) load_res(s_own, s_own);
        }
    }

    // ---- epilogue: lane (l15, q4) holds row l15 and columns 4*q4 .. +3 of every 16x16 tile.  Each wave transposes one 16-row tile
    // at a time through its private slab and moves row-contiguous 16-B chunks to / from HBM (guide T21).
    constexpr int RS = TNO * 16 * ES + 16;                           // slab row stride (16-B pad: conflict-free 16-B column writes)
    static_assert(NW * 16 * RS <= S * STAGE, "slabs");
    char* slab = smem + (size_t)wave * 16 * RS;
    static_assert(LNF != RT_LNF_EMIT || (IPR == 10 && NW * 16 * RS + NW * 16 * IPR * 8 <= S * STAGE), "partials scratch behind the slabs");
    const unsigned ln_scr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(smem + NW * 16 * RS + wave * (16 * IPR * 8));
    static_assert(LNF != RT_LNF_EMIT || (NW * 16 * RS + NW * 16 * IPR * 8 + BM * WN * 8 <= S * STAGE), "partials table of the workgroup behind the scratch");
    const unsigned ln_wgt = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(smem + NW * 16 * RS + NW * (16 * IPR * 8));
    (void)ln_scr; (void)ln_wgt;
    if constexpr (!BIAS_EARLY) load_bias();
#ifdef RT_G16_TIMING
    { float bsum = 0.f; for (int t = 0; t < TNO; ++t) bsum += bias_v[t][0]; asm volatile("" ::"v"(bsum)); G16_T(5) }
#endif
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
        if (!owned(i)) continue;
        // LNF consumers: (mu, rstd) of the lane's token row / s, c of its weight row, from the table
        float ln_nm = 0.f, ln_rs = 1.f, ln_sr = 0.f, ln_cr = 0.f;
        if constexpr (LNF == RT_LNF_ROWS) {
            const float2 st = *(const float2*)(smem + S * STAGE + ((wm * TMW + i) * 16 + l15) * 8);
            ln_nm = st.x; ln_rs = st.y;                               // (the table holds -mu rstd)
        }
        if constexpr (LNF == RT_LNF_COLS) {
            ln_sr = *(const float*)(smem + S * STAGE + 2560 + ((wm * TMW + i) * 16 + l15) * 4);
            ln_cr = *(const float*)(smem + S * STAGE + 3840 + ((wm * TMW + i) * 16 + l15) * 4);
        }
        (void)ln_nm; (void)ln_rs; (void)ln_sr; (void)ln_cr;
        // registers -> slab
#pragma unroll
        for (int t = 0; t < TNO; ++t) {
            float v[4];
            if constexpr (EPI == EPI_GEGLU) {
                const int tv = (t >> 1) * 4 + (t & 1);
#pragma unroll
                for (int e = 0; e < 4; e += 2) {                     // two gates per packed-fp32 issue slot
                    f32x2 gt, vl;
                    if constexpr (LNF == RT_LNF_ROWS) {              // LN(x) W^T + b = rstd (x W'^T - mu s) + c, in front of the gelu
                        // rstd acc + (c - mu rstd s): two packed FMAs per pair of values
                        constexpr int tg = (LNF == RT_LNF_ROWS && EPI == EPI_GEGLU) ? 1 : 0;
                        const f32x2 rs2 = {ln_rs, ln_rs}, nm2 = {ln_nm, ln_nm};
                        gt = __builtin_elementwise_fma(rs2, f32x2{acc[i][tv + 2][e], acc[i][tv + 2][e + 1]},
                                                       __builtin_elementwise_fma(nm2, f32x2{lns_g[tg * t][e], lns_g[tg * t][e + 1]}, f32x2{bias_g[t][e], bias_g[t][e + 1]}));
                        vl = __builtin_elementwise_fma(rs2, f32x2{acc[i][tv][e], acc[i][tv][e + 1]},
                                                       __builtin_elementwise_fma(nm2, f32x2{lns_v[t][e], lns_v[t][e + 1]}, f32x2{bias_v[t][e], bias_v[t][e + 1]}));
                    } else {
                        gt = f32x2{acc[i][tv + 2][e] + bias_g[t][e], acc[i][tv + 2][e + 1] + bias_g[t][e + 1]};
                        vl = f32x2{acc[i][tv][e] + bias_v[t][e], acc[i][tv][e + 1] + bias_v[t][e + 1]};
                    }
                    const f32x2 o = vl * gelu_erf_x2(gt);
                    v[e] = o.x; v[e + 1] = o.y;
                }
            } else if constexpr (LNF == RT_LNF_ROWS) {
                const f32x2 rs2 = {ln_rs, ln_rs}, nm2 = {ln_nm, ln_nm};
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    constexpr int tr = LNF == RT_LNF_ROWS ? 1 : 0;
                    const f32x2 o2 = __builtin_elementwise_fma(rs2, f32x2{acc[i][t][e], acc[i][t][e + 1]},
                                                               __builtin_elementwise_fma(nm2, f32x2{lns_v[tr * t][e], lns_v[tr * t][e + 1]}, f32x2{bias_v[t][e], bias_v[t][e + 1]}));
                    v[e] = o2.x; v[e + 1] = o2.y;
                }
            } else if constexpr (LNF == RT_LNF_COLS) {
                const float4 ca = *(const float4*)(smem + S * STAGE + (wn * TNW * 16 + t * 16 + 4 * q4) * 8);            // (-mu rstd, rstd) of columns 4 q4, 4 q4 + 1
                const float4 cb = *(const float4*)(smem + S * STAGE + (wn * TNW * 16 + t * 16 + 4 * q4) * 8 + 16);       // ... + 2, + 3
                v[0] = fmaf(ca.y, acc[i][t][0], fmaf(ca.x, ln_sr, ln_cr)); v[1] = fmaf(ca.w, acc[i][t][1], fmaf(ca.z, ln_sr, ln_cr));
                v[2] = fmaf(cb.y, acc[i][t][2], fmaf(cb.x, ln_sr, ln_cr)); v[3] = fmaf(cb.w, acc[i][t][3], fmaf(cb.z, ln_sr, ln_cr));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][t][e] + bias_v[t][e];
            }
            if constexpr (TEMB) {
                int row = m0 + (wm * TMW + i) * 16 + l15; if (row >= p.M) row = p.M - 1;
                const int col = wcol0 + t * 16 + 4 * q4;
                if (col < p.N) {
                    const float4 tv = *(const float4*)(p.temb + (size_t)(row / p.rows_per_batch) * p.temb_ld + col);
                    v[0] += tv.x; v[1] += tv.y; v[2] += tv.z; v[3] += tv.w;
                }
            }
            char* dst = slab + l15 * RS + (t * 16 + 4 * q4) * ES;
            if constexpr (F32) *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
            else { uint2 w; w.x = pack_bf16x2(v[0], v[1]); w.y = pack_bf16x2(v[2], v[3]); *(uint2*)dst = w; }
        }
        // slab -> HBM (LDS operations of one wave execute in order: no barrier)
        const int row0 = m0 + (wm * TMW + i) * 16;
        if constexpr (F16) {
            // (producer of the LayerNorm partials: nothing of this tile's store section may move above this point - with the partial-sum
            //  code in the loop the scheduler otherwise converts the whole residual window to fp32 where it is loaded: twice the registers,
            //  scratch, and the wait for the residual's round trip in front of the K-split exchange instead of behind it)
            if constexpr (LNF == RT_LNF_EMIT) {
                __builtin_amdgcn_sched_barrier(0);
                if (p.res) {
#pragma unroll
                    for (int it = 0; it < NIT; ++it) asm volatile("" : "+v"(rres[F16 ? ((i < H0 ? i : i - H0) % RW) : 0][F16 ? it : 0]));
                }
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = it * 64 + lane;
                const int r = idx / IPR, c8 = idx - r * IPR;
                const int row = row0 + r, col = ocol0 + c8 * 8;
                if (r >= 16 || row >= p.M || col >= NO) continue;
                const float4 a0 = *(const float4*)(slab + r * RS + c8 * 32), a1 = *(const float4*)(slab + r * RS + c8 * 32 + 16);
                float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                if (p.res) {
                    const f16_t* rh = (const f16_t*)&rres[F16 ? ((i < H0 ? i : i - H0) % RW) : 0][it];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)rh[e];
                }
                uint4 o; f16_t* oh = (f16_t*)&o;
#pragma unroll
                for (int e = 0; e < 8; ++e) oh[e] = (f16_t)v[e];
                *(uint4*)((f16_t*)p.out + (size_t)row * p.ldo + col) = o;
                if constexpr (LNF == RT_LNF_EMIT) {
                    // xb = bf16 of the same eight fp32 values -> the consumers' MFMA operand; sum / sum of squares of xb (one v_dot2c_f32_bf16
                    // per pair and moment) -> the wave's scratch [16 rows][10 items] behind the slabs.  Inline asm for the LDS traffic:
                    // compiler-visible LDS operations behind a global store make hipcc's waitcnt pass wait for the store (vmcnt(0)); a
                    // wave's LDS operations execute in order, which is all that is needed here.
                    g16_u4 cb;
                    cb.x = pack_bf16x2(v[0], v[1]); cb.y = pack_bf16x2(v[2], v[3]); cb.z = pack_bf16x2(v[4], v[5]); cb.w = pack_bf16x2(v[6], v[7]);
                    *(g16_u4*)(p.ln_copy + (size_t)row * p.ldo + col) = cb;
                    float s1 = 0.f, s2 = 0.f;
                    asm volatile("v_dot2c_f32_bf16 %0, %2, %6\n\tv_dot2c_f32_bf16 %1, %2, %2\n\t"
                                 "v_dot2c_f32_bf16 %0, %3, %6\n\tv_dot2c_f32_bf16 %1, %3, %3\n\t"
                                 "v_dot2c_f32_bf16 %0, %4, %6\n\tv_dot2c_f32_bf16 %1, %4, %4\n\t"
                                 "v_dot2c_f32_bf16 %0, %5, %6\n\tv_dot2c_f32_bf16 %1, %5, %5\n\t"
                                 "s_nop 2"      /* a DOT result read by another VALU opcode needs 3 wait states, and hipcc's hazard recogniser does not look into asm */
                                 : "+v"(s1), "+v"(s2) : "v"(cb.x), "v"(cb.y), "v"(cb.z), "v"(cb.w), "v"(0x3f803f80u));
                    const unsigned long long w64 = ((unsigned long long)__float_as_uint(s2) << 32) | __float_as_uint(s1);
                    asm volatile("ds_write_b64 %0, %1" ::"v"(ln_scr + (unsigned)idx * 8u), "v"(w64) : "memory");
                }
            }
            if constexpr (LNF == RT_LNF_EMIT) {
                // lane (row rr, quarter jj) adds items jj, jj + 4, jj + 8 of its row, the four quarters meet by two quad-DPP exchanges
                // (fixed order: deterministic); quarter 0 stores the row's (sum, sum of squares) over these 80 columns
                const int rr = lane >> 2, jj = lane & 3;
                unsigned long long x0, x1, x2;
                asm volatile("ds_read_b64 %0, %3\n\tds_read_b64 %1, %3 offset:32\n\tds_read_b64 %2, %3 offset:64\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(x0), "=&v"(x1), "=&v"(x2) : "v"(ln_scr + (unsigned)(rr * IPR + jj) * 8u) : "memory");
                float a1 = __uint_as_float((unsigned)x0) + __uint_as_float((unsigned)x1);
                float a2 = __uint_as_float((unsigned)(x0 >> 32)) + __uint_as_float((unsigned)(x1 >> 32));
                if (jj < 2) { a1 += __uint_as_float((unsigned)x2); a2 += __uint_as_float((unsigned)(x2 >> 32)); }
                a1 += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(a1), 0xB1, 0xF, 0xF, true));     // quad_perm [1, 0, 3, 2]
                a2 += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(a2), 0xB1, 0xF, 0xF, true));
                a1 += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(a1), 0x4E, 0xF, 0xF, true));     // quad_perm [2, 3, 0, 1]
                a2 += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(a2), 0x4E, 0xF, 0xF, true));
                // -> the workgroup's table [BM rows][WN wave columns]; combined and stored once, behind the last tile
                if (jj == 0) {
                    const unsigned long long w64 = ((unsigned long long)__float_as_uint(a2) << 32) | __float_as_uint(a1);
                    asm volatile("ds_write_b64 %0, %1" ::"v"(ln_wgt + (unsigned)((((wm * TMW + i) * 16 + rr) * WN + wn) * 8)), "v"(w64) : "memory");
                }
            }
            // the window slot of this tile is free: request the residual of the owned tile RW positions further on
            const int s_own = i < H0 ? i : i - H0;                   // compile-time after unrolling
            if (s_own + RW < NOWN && p.res) load_res(s_own + RW, s_own % RW);
        } else {
            constexpr int CPR = TNO * 16 * ES / 16;                  // 16-B chunks per row
#pragma unroll
            for (int idx0 = 0; idx0 < 16 * CPR; idx0 += 64) {
                const int idx = idx0 + lane;
                const int r = idx / CPR, ch = idx - r * CPR;
                const int row = row0 + r, col = ocol0 + ch * (16 / ES);
                if (r >= 16 || row >= p.M || col >= NO) continue;
                const uint4 qv = *(const uint4*)(slab + r * RS + ch * 16);
                if constexpr (F32) {
                    float4 o = make_float4(__uint_as_float(qv.x), __uint_as_float(qv.y), __uint_as_float(qv.z), __uint_as_float(qv.w));
                    if (p.res) { const float4 rv = *(const float4*)((const float*)p.res + (size_t)row * p.ldres + col); o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w; }
                    *(float4*)((float*)p.out + (size_t)row * p.ldo + col) = o;
                } else {
                    *(uint4*)((bf16_t*)p.out + (size_t)row * p.ldo + col) = qv;
                }
            }
        }
#ifdef RT_G16_TIMING
        if (i == (WK == 2 && kh == 1 ? H0 : 0)) G16_T(6)
#endif
    }
    if constexpr (LNF == RT_LNF_EMIT) {
        // one (sum, sum of squares) per token row and column TILE: the WN wave columns added in order; two tiles share a float4 of the
        // pair-major array [tile pair][row] the consumers read
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tid < BM && m0 + tid < p.M) {
            const float2* wt = (const float2*)(smem + NW * 16 * RS + NW * (16 * IPR * 8)) + tid * WN;
            float a1 = wt[0].x, a2 = wt[0].y;
#pragma unroll
            for (int w = 1; w < WN; ++w) { a1 += wt[w].x; a2 += wt[w].y; }
            *((float2*)p.ln_emit + ((size_t)(tn >> 1) * p.M + m0 + tid) * 2 + (tn & 1)) = make_float2(a1, a2);
        }
    }
#ifdef RT_G16_TIMING
    G16_T(7)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G16_T(4)
#endif
