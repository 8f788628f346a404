// a11 + a12 + a14 of SURVEY.md section 8a: mask-weighted region combine (models/region_diffusion.py:119-132,
// models/region_diffusion_sdxl.py:810-825), classifier-free guidance, scheduler update (third-party
// diffusers 0.18.2 PNDM/PLMS and Euler restated, see oracle/schedulers.py) and background blend
// (rd.py:171-173, xl.py:870-872) fused into one elementwise launch over 4*h*w elements.
// The arithmetic follows the reference's fp32 operation order so that the epilogue itself is exact.
#include "step.h"
#include "../../include/rtdiff.h"

__global__ void step_epilogue_kernel(StepArgs p) {
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= p.HW) return;
    auto ld4 = [&](int s) { return *(const float4*)(p.eps + ((size_t)s * p.HW + pix) * 4); };
    float e[4], er[4];
    const float4 eu = ld4(p.s_uncond), eb = ld4(p.s_base);
    const float euv[4] = {eu.x, eu.y, eu.z, eu.w}, ebv[4] = {eb.x, eb.y, eb.z, eb.w};
    if (p.plain) {
#pragma unroll
        for (int c = 0; c < 4; ++c) e[c] = euv[c] + p.g * (ebv[c] - euv[c]);
    } else {
        float nu[4], nt[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float ml = p.masks[((size_t)(p.R - 1) * 4 + c) * p.HW + pix];
            nu[c] = euv[c] * ml; nt[c] = ebv[c] * ml;
        }
        for (int r = 0; r < p.R - 1; ++r) {
            const float4 q = ld4(p.s_region[r]);
            const float qv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float m = p.masks[((size_t)r * 4 + c) * p.HW + pix];
                nu[c] = nu[c] + euv[c] * m; nt[c] = nt[c] + qv[c] * m;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) e[c] = nu[c] + p.g * (nt[c] - nu[c]);
    }
    if (p.noise_pred) {
#pragma unroll
        for (int c = 0; c < 4; ++c) p.noise_pred[(size_t)c * p.HW + pix] = e[c];
    }
    const bool has_ref = p.s_uref >= 0 && p.step_ref;
    if (has_ref) {
        const float4 a = ld4(p.s_uref), b = ld4(p.s_tref);
        const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) er[c] = av[c] + p.g * (bv[c] - av[c]);
    }
    const int nstream = has_ref ? 2 : 1;
    float newv[2][4];
    for (int s = 0; s < nstream; ++s) {
        float* x = s == 0 ? p.lat : p.lat_ref;
        const float* ee = s == 0 ? e : er;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const size_t li = (size_t)c * p.HW + pix;
            const size_t hi = (size_t)s * 4 * p.HW + li;
            float sample = x[li];
            float ep = ee[c];
            if (p.sched == RT_SCHED_EULER) {
                // pred_x0 = x - sigma*eps; derivative = (x - pred_x0)/sigma; x += derivative * dsigma  (gamma = 0)
                newv[s][c] = sample + ep * p.dsigma;
            } else {
                if (p.push) p.ets[0][hi] = ep;
                if (p.pndm_mode == 0) { p.cur_sample[hi] = sample; }
                else if (p.pndm_mode == 1) { ep = (ep + p.ets[1][hi]) / 2.f; sample = p.cur_sample[hi]; }
                else if (p.pndm_mode == 2) { ep = (3.f * ep - p.ets[1][hi]) / 2.f; }
                else if (p.pndm_mode == 3) { ep = (23.f * ep - 16.f * p.ets[1][hi] + 5.f * p.ets[2][hi]) / 12.f; }
                else { ep = (1.f / 24.f) * (55.f * ep - 59.f * p.ets[1][hi] + 37.f * p.ets[2][hi] - 9.f * p.ets[3][hi]); }
                newv[s][c] = p.ca * sample - p.cb * ep;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const size_t li = (size_t)c * p.HW + pix;
        float l = newv[0][c];
        const float lr = has_ref ? newv[1][c] : p.lat_ref[li];
        if (p.blend) {
            const float ml = p.masks[((size_t)(p.R - 1) * 4 + c) * p.HW + pix];
            l = lr * ml + l * (1.f - ml);
        }
        p.lat[li] = l;
        if (has_ref) p.lat_ref[li] = lr;
    }
}
void launch_step_epilogue(const StepArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(step_epilogue_kernel, dim3(cdiv(a.HW, 256)), dim3(256), 0, st, a);
    HIP_CHECK(hipGetLastError());
}

// emb[b][c] = base[c] + table[idx[b]][c]
__global__ void gather_add_rows_kernel(const float* base, const float* table, IdxList idx, float* out, int B, int C) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * C; i += gridDim.x * blockDim.x) {
        const int b = i / C, c = i % C;
        out[i] = base[c] + table[(size_t)idx.v[b] * C + c];
    }
}
void launch_gather_add_rows(const float* base, const float* table, const int* idx, float* out, int B, int C, hipStream_t st) {
    IdxList l{}; for (int b = 0; b < B; ++b) l.v[b] = idx[b];
    hipLaunchKernelGGL(gather_add_rows_kernel, dim3(cdiv(B * C, 256)), dim3(256), 0, st, base, table, l, out, B, C);
    HIP_CHECK(hipGetLastError());
}

// out[b] = sc[b] + hres[src[b]]   (out may alias sc).  sc / out are fp16 trunk tensors, hres is the fp32 residual branch: the sum is
// rounded once, exactly like the fused epilogue of a resnet without injection (a stream keeps its bits whether or not another
// stream of the batch injects).
__global__ void inject_add_kernel(f16_t* out, const f16_t* sc, const float* hres, IdxList src, size_t per_batch4) {
    const int b = blockIdx.y;
    const uint2* s4 = (const uint2*)sc + (size_t)b * per_batch4;
    const float4* h4 = (const float4*)hres + (size_t)src.v[b] * per_batch4;
    uint2* o4 = (uint2*)out + (size_t)b * per_batch4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < per_batch4; i += (size_t)gridDim.x * blockDim.x) {
        const uint2 a = s4[i]; const float4 h = h4[i];
        const f16_t* ah = (const f16_t*)&a;
        uint2 o; f16_t* oh = (f16_t*)&o;
        oh[0] = (f16_t)(h.x + (float)ah[0]); oh[1] = (f16_t)(h.y + (float)ah[1]); oh[2] = (f16_t)(h.z + (float)ah[2]); oh[3] = (f16_t)(h.w + (float)ah[3]);
        o4[i] = o;
    }
}
void launch_inject_add(f16_t* out, const f16_t* sc, const float* hres, const int* src, int B, size_t per_batch, hipStream_t st) {
    IdxList l{}; for (int b = 0; b < B; ++b) l.v[b] = src[b];
    const size_t n4 = per_batch / 4;
    int gx = (int)((n4 + 255) / 256); if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(inject_add_kernel, dim3(gx, B), dim3(256), 0, st, out, sc, hres, l, n4);
    HIP_CHECK(hipGetLastError());
}

// [B, HW, 4] -> [B, 4, HW]
__global__ void nhwc4_to_nchw_kernel(const float* in, float* out, int B, int HW) {
    const size_t n = (size_t)B * HW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / HW, pix = i % HW;
        const float4 v = ((const float4*)in)[i];
        float* o = out + b * 4 * HW + pix;
        o[0] = v.x; o[HW] = v.y; o[2 * (size_t)HW] = v.z; o[3 * (size_t)HW] = v.w;
    }
}
void launch_nhwc4_to_nchw(const float* in, float* out, int B, int HW, hipStream_t st) {
    hipLaunchKernelGGL(nhwc4_to_nchw_kernel, dim3(cdiv(B * HW, 256)), dim3(256), 0, st, in, out, B, HW);
    HIP_CHECK(hipGetLastError());
}

// text context [P, 77, D] f32 -> bf16 [P, 96, D] with zero rows 77..95
__global__ void pad_ctx_kernel(const float* ctx, bf16_t* out, int P, int D) {
    const size_t n = (size_t)P * 96 * D;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t c = i % D, k = (i / D) % 96, p_ = i / ((size_t)96 * D);
        out[i] = k < 77 ? f32_to_bf16(ctx[(p_ * 77 + k) * D + c]) : (bf16_t)0;
    }
}
void launch_pad_ctx(const float* ctx, bf16_t* out, int P, int D, hipStream_t st) {
    hipLaunchKernelGGL(pad_ctx_kernel, dim3(cdiv(P * 96 * D, 256)), dim3(256), 0, st, ctx, out, P, D);
    HIP_CHECK(hipGetLastError());
}

// lat = lat_ref * M + lat * (1 - M)   (rd.py:171-173, xl.py:870-872) as a separate launch when colour guidance has to run
// between the scheduler step and the blend
__global__ void background_blend_kernel(float* lat, const float* lat_ref, const float* m, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) lat[i] = lat_ref[i] * m[i] + lat[i] * (1.f - m[i]);
}
void launch_background_blend(float* lat, const float* lat_ref, const float* mask_last, int n, hipStream_t st) {
    hipLaunchKernelGGL(background_blend_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, lat, lat_ref, mask_last, n);
    HIP_CHECK(hipGetLastError());
}
