// RMSNorm row body shared by norm.hip (chitu_hip_rmsnorm) and kv.hip (the fused MLA q_norm / kv
// kernel): one 256-thread workgroup normalises one row.  See norm.hip for the contract.
#pragma once
#include "common.h"

namespace chitu {

constexpr int kNormThreads = 256;
constexpr int kNormMaxChunks = 4;  // dim <= 256 * 8 * 4 = 8192

// QMODE 0: no quant; 1: act_quant (no eps, no clamp); 2: per_token_group_quant (eps, clamp).
// ADD: residual input present.  Every load of the row (x, add, weights) is issued before the first
// use, straight-line (chunk indices are clamped, not branched on): one memory round trip, not one
// per chunk.  sum_out may alias x or add (in-place residual): it is only written after all loads.
template <int QMODE, bool ADD>
__device__ __forceinline__ void rmsnorm_row(
    const int row, const bf16_t* x, int64_t x_stride, const bf16_t* add, int64_t add_stride, bf16_t* sum_out,
    int64_t sum_stride, const bf16_t* __restrict__ w, bf16_t* y, int64_t y_stride,
    fp8_t* __restrict__ q, float* __restrict__ qs, int dim, float eps, float qeps) {
    __shared__ float red[kNormThreads / 64];
    const int tid = threadIdx.x;
    const bf16_t* xr = x + (int64_t)row * x_stride;
    const int n_chunks = dim >> 3;
    i32x4 xraw[kNormMaxChunks], araw[kNormMaxChunks], wreg[kNormMaxChunks];
#pragma unroll
    for (int i = 0; i < kNormMaxChunks; ++i) {
        const int c = min(tid + i * kNormThreads, n_chunks - 1);
        xraw[i] = *reinterpret_cast<const i32x4*>(xr + c * 8);
        if (ADD) araw[i] = *reinterpret_cast<const i32x4*>(add + (int64_t)row * add_stride + c * 8);
    }
#pragma unroll
    for (int i = 0; i < kNormMaxChunks; ++i) {
        const int c = min(tid + i * kNormThreads, n_chunks - 1);
        wreg[i] = *reinterpret_cast<const i32x4*>(w + c * 8);
    }
    float v[kNormMaxChunks][8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < kNormMaxChunks; ++i) {
        const bool act = tid + i * kNormThreads < n_chunks;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t u = (uint32_t)xraw[i][k];
            v[i][2 * k] = __uint_as_float(u << 16);
            v[i][2 * k + 1] = __uint_as_float(u & 0xffff0000u);
        }
        if (ADD) {
            // residual: x <- bf16(x + add), the reference's `x = x + attn(...)` in bf16
            // (model_deepseek_v3.py:1107-1113), folded into the norm that consumes it
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t u = (uint32_t)araw[i][k];
                const uint32_t s2 = f32x2_to_bf16x2(v[i][2 * k] + __uint_as_float(u << 16),
                                                    v[i][2 * k + 1] + __uint_as_float(u & 0xffff0000u));
                v[i][2 * k] = __uint_as_float(s2 << 16);
                v[i][2 * k + 1] = __uint_as_float(s2 & 0xffff0000u);
                araw[i][k] = (int)s2;
            }
        }
        if (act) {
#pragma unroll
            for (int k = 0; k < 8; ++k) ss += v[i][k] * v[i][k];
        }
    }
    if (ADD && sum_out) {
#pragma unroll
        for (int i = 0; i < kNormMaxChunks; ++i) {
            const int c = tid + i * kNormThreads;
            if (c < n_chunks) *reinterpret_cast<i32x4*>(sum_out + (int64_t)row * sum_stride + c * 8) = araw[i];
        }
    }
    ss = wave_reduce_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    ss = red[0] + red[1] + red[2] + red[3];
    const float rr = rsqrtf(ss / (float)dim + eps);
#pragma unroll
    for (int i = 0; i < kNormMaxChunks; ++i) {
        const int c = tid + i * kNormThreads;
        const bool act = c < n_chunks;
        float o[8];
        i32x4 out;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t u = (uint32_t)wreg[i][k];
            const uint32_t h2 = f32x2_to_bf16x2((v[i][2 * k] * rr) * __uint_as_float(u << 16),
                                                (v[i][2 * k + 1] * rr) * __uint_as_float(u & 0xffff0000u));
            out[k] = (int)h2;
            o[2 * k] = act ? __uint_as_float(h2 << 16) : 0.f;
            o[2 * k + 1] = act ? __uint_as_float(h2 & 0xffff0000u) : 0.f;
        }
        if (y && act) *reinterpret_cast<i32x4*>(y + (int64_t)row * y_stride + c * 8) = out;
        if (QMODE != 0) {
            // dim % 128 == 0 => a 16-lane group is either fully active or fully idle
            float amax = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) amax = __builtin_fmaxf(amax, __builtin_fabsf(o[k]));
            amax = row16_reduce_max(amax);
            if (QMODE == 2) amax = __builtin_fmaxf(amax, qeps);
            const float sc = amax / 448.0f;
            const i32x2 packed = quant8_fp8<QMODE == 2>(o, act ? sc : 1.0f);
            if (act) {
                *reinterpret_cast<i32x2*>(q + (int64_t)row * dim + c * 8) = packed;
                if ((tid & 15) == 0) qs[(int64_t)row * (dim >> 7) + (c >> 4)] = sc;
            }
        }
    }
}


// ---- the WIDE row form: 1024 threads per row, ONE 8-element chunk per thread (dim <= 8192) -------------
// Used wherever a full hidden-state row is normalised behind a residual add: chitu_hip_rmsnorm with `add`
// (norm.hip) and the fused all-reduce launch (comm.hip).  Both call the two functions below, so the library
// transport and the in-graph xGMI transport give bit-identical steps.
constexpr int kNormWideThreads = 1024;
constexpr int kNormMaxTerms = 16;

// 4 packed bf16 pairs + 4 packed bf16 pairs -> v[8] = bf16(a + b) as floats, sraw = the packed sums.
__device__ __forceinline__ void add_bf16x8(const i32x4& a, const i32x4& b, float (&v)[8], i32x4& sraw) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t ua = (uint32_t)a[i], ub = (uint32_t)b[i];
        const uint32_t s2 = f32x2_to_bf16x2(__uint_as_float(ua << 16) + __uint_as_float(ub << 16),
                                            __uint_as_float(ua & 0xffff0000u) + __uint_as_float(ub & 0xffff0000u));
        v[2 * i] = __uint_as_float(s2 << 16);
        v[2 * i + 1] = __uint_as_float(s2 & 0xffff0000u);
        sraw[i] = (int)s2;
    }
}
__device__ __forceinline__ void unpack_bf16x8(const i32x4& a, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(((uint32_t)a[i]) << 16);
        v[2 * i + 1] = __uint_as_float(((uint32_t)a[i]) & 0xffff0000u);
    }
}
// fp32 sum of `terms` packed rows (k < terms of MAXT loaded registers), ONE rounding: chitu_hip_moe_sum's arithmetic.
template <int MAXT>
__device__ __forceinline__ i32x4 sum_terms_bf16x8(const i32x4 (&t)[MAXT], int terms) {
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < MAXT; ++k) {
        if (k < terms) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t u = (uint32_t)t[k][i];
                a[2 * i] += __uint_as_float(u << 16);
                a[2 * i + 1] += __uint_as_float(u & 0xffff0000u);
            }
        }
    }
    i32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (int)f32x2_to_bf16x2(a[2 * i], a[2 * i + 1]);
    return r;
}

// v[8]: this thread's chunk `tid` of the (already residual-added, bf16-valued) row; act: chunk inside the row.
// Mean square over the row (per-thread sequential, wave butterfly, the 16 wave sums in order), y = (v * rr) * w
// with one rounding, optional fp8 quantisation of the rounded y (16 lanes = one 128-wide group).
// tile_major: the fp8 output in the layout the small-batch GEMMs read with fully coalesced loads (gemm_common.h,
// "tile-major activations"): q[tile = row / 16][dim / 16][row % 16][16 B], qs[tile][dim / 128][row % 16].
// QMODE 3 (round 6): the per-token INT8 quantisation of the rounded y (quant_act of the reference's W8A8Linear, quantize/w8a8.py:18-26:
// scale = max(|y|, 1e-5) / 127 over the whole row, code = clamp(rint(y / scale)) -- w8a8_int8.hip::quant_act_int8_vec_kernel's arithmetic on the
// same bf16 values): q holds int8 codes [rows, dim], qs one scale per row; `red2` = 16 more floats of LDS for the row maximum.
template <int QMODE>
__device__ __forceinline__ void rmsnorm_wide_finish(const float (&v)[8], bool act, int row, const i32x4& wraw, bf16_t* y,
                                                    int64_t y_stride, fp8_t* __restrict__ q, float* __restrict__ qs,
                                                    int dim, float eps, float qeps, float* red, int tile_major = 0,
                                                    float* red2 = nullptr) {
    const int tid = threadIdx.x;
    float ss = 0.f;
    if (act) {
#pragma unroll
        for (int i = 0; i < 8; ++i) ss += v[i] * v[i];
    }
    ss = wave_reduce_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    ss = 0.f;
#pragma unroll
    for (int i = 0; i < kNormWideThreads / 64; ++i) ss += red[i];
    const float rr = rsqrtf(ss / (float)dim + eps);
    float o[8];
    i32x4 out;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t u = (uint32_t)wraw[i];
        const uint32_t h2 = f32x2_to_bf16x2((v[2 * i] * rr) * __uint_as_float(u << 16),
                                            (v[2 * i + 1] * rr) * __uint_as_float(u & 0xffff0000u));
        out[i] = (int)h2;
        o[2 * i] = act ? __uint_as_float(h2 << 16) : 0.f;
        o[2 * i + 1] = act ? __uint_as_float(h2 & 0xffff0000u) : 0.f;
    }
    if (y && act) *reinterpret_cast<i32x4*>(y + (int64_t)row * y_stride + tid * 8) = out;
    if (QMODE == 3) {
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) amax = __builtin_fmaxf(amax, __builtin_fabsf(o[i]));  // (o = 0 outside the row)
        amax = wave_reduce_max(amax);
        if ((tid & 63) == 0) red2[tid >> 6] = amax;
        __syncthreads();
        amax = 0.f;
#pragma unroll
        for (int i = 0; i < kNormWideThreads / 64; ++i) amax = __builtin_fmaxf(amax, red2[i]);
        const float sc = __builtin_fmaxf(amax, 1e-5f) / 127.0f;
        if (act) {
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float r = rintf(o[k] / sc);
                r = __builtin_fminf(__builtin_fmaxf(r, -128.f), 127.f);
                const uint32_t b = (uint32_t)(uint8_t)(int8_t)r;
                if (k < 4) lo |= b << (8 * k);
                else hi |= b << (8 * (k - 4));
            }
            i32x2 pk;
            pk[0] = (int)lo;
            pk[1] = (int)hi;
            *reinterpret_cast<i32x2*>(q + (int64_t)row * dim + tid * 8) = pk;
        }
        if (tid == 0) qs[row] = sc;
    } else if (QMODE != 0) {
        // dim % 128 == 0 => a 16-lane group is either fully active or fully idle
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) amax = __builtin_fmaxf(amax, __builtin_fabsf(o[i]));
        amax = row16_reduce_max(amax);
        if (QMODE == 2) amax = __builtin_fmaxf(amax, qeps);
        const float sc = amax / 448.0f;
        const i32x2 packed = quant8_fp8<QMODE == 2>(o, act ? sc : 1.0f);
        if (act) {
            if (tile_major) {
                const int64_t tile = row >> 4;
                const int m = row & 15;
                *reinterpret_cast<i32x2*>(q + ((tile * (dim >> 4) + (tid >> 1)) * 16 + m) * 16 + (tid & 1) * 8) = packed;
                if ((tid & 15) == 0) qs[(tile * (dim >> 7) + (tid >> 4)) * 16 + m] = sc;
            } else {
                *reinterpret_cast<i32x2*>(q + (int64_t)row * dim + tid * 8) = packed;
                if ((tid & 15) == 0) qs[(int64_t)row * (dim >> 7) + (tid >> 4)] = sc;
            }
        }
    }
}

}  // namespace chitu
