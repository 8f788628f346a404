// Helpers shared by cross77.hip (the engine's 77-key cross-attention) and xblock.hip (the one-launch cross-attention block, probe builds):
// the pi / rho row permutations that make MFMA accumulator tiles the next MFMA's B operand, accumulator -> bf16x8 packing, and row
// reductions over the four 16-lane groups of a wave.
#pragma once
#include "common.h"

#define XB_KVH 22528                           // one head inside a K / V^T tile: K [80 rows][128 B] + V^T [12 chunks][64 d][16 B]

static __device__ __forceinline__ int xb_pi(int rho) {             // within a 64-row block
    const int a = (rho >> 4) & 3, i = rho & 15;
    return (rho & ~63) + 32 * (a >> 1) + 8 * (i >> 2) + 4 * (a & 1) + (i & 3);
}
static __device__ __forceinline__ int xb_rho(int r) {              // within a 64-row block: LDS row 16 a + 4 q + r <- channel 16 q + 4 a + r
    const int a = (r >> 4) & 3, q = (r >> 2) & 3;
    return (r & ~63) + 16 * q + 4 * a + (r & 3);
}
static __device__ __forceinline__ bf16x8 xb_pack8(const f32x4& a0, const f32x4& a1) {
    union { uint32_t u[4]; bf16x8 v; } pk;
    pk.u[0] = pack_bf16x2(a0[0], a0[1]); pk.u[1] = pack_bf16x2(a0[2], a0[3]);
    pk.u[2] = pack_bf16x2(a1[0], a1[1]); pk.u[3] = pack_bf16x2(a1[2], a1[3]);
    return pk.v;
}

// Row reductions over the four 16-lane groups by gfx950's v_permlane16_swap / v_permlane32_swap (VALU) instead of two ds_bpermute
// round trips.  The maxima are plain fmaxf: this file is compiled with -fno-honor-nans (scores are finite or -inf), which drops the
// canonicalising self-max fmaxf's NaN semantics drag in.  (An inline-asm v_max3_f32 is NOT an option: hipcc's hazard recogniser does not
// see the registers an asm statement reads, so the wait states between an MFMA and a VALU read of its result were missing and the
// maxima were taken over stale registers - harmless for the mathematics, any reference works, but the last bits differed from run to
// run, which the determinism tests caught.)
static __device__ __forceinline__ float xb_max(float a, float b) { return fmaxf(a, b); }
static __device__ __forceinline__ float xb_max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
static __device__ __forceinline__ float xb_rowmax(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = xb_max(__uint_as_float(r[0]), __uint_as_float(r[1]));
    const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return xb_max(__uint_as_float(r2[0]), __uint_as_float(r2[1]));
}
static __device__ __forceinline__ float xb_rowsum(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
}

