// kv_norm + RoPE(k_pe) + page append of ONE token of the MLA KV stream, shared by kv.hip (mla_qkv_post_kernel),
// mla_q_proj.hip (the launch that also runs the wq_b GEMM) and absorb.hip (the W_UK absorb launch, q_lora_rank == 0).
#pragma once
#include "common.h"

namespace chitu {

// The page row of token b's new KV entry (nullptr when its table entry is out of range: nothing is written).
__device__ __forceinline__ bf16_t* mla_kv_row_ptr(int b, bf16_t* __restrict__ cache, int64_t num_pages, int page_size,
                                                  const int32_t* __restrict__ table, int pages_per_seq,
                                                  const int32_t* __restrict__ old_lens) {
    const int L = old_lens[b];
    const int pidx = L / page_size;
    if (L >= 0 && pidx < pages_per_seq) {
        const int64_t page = table[(int64_t)b * pages_per_seq + pidx];
        if (page >= 0 && page < num_pages) return cache + (page * page_size + (L % page_size)) * 576;
    }
    return nullptr;
}

// kv_norm(kv_c): one full wave, 8 of the 512 values per lane.
__device__ __forceinline__ void mla_kv_row_norm(int lane, const bf16_t* src, const bf16_t* __restrict__ kv_norm_w,
                                                float kv_eps, bf16_t* row) {
#pragma clang fp contract(off)
    const i32x4 raw = *reinterpret_cast<const i32x4*>(src + lane * 8);
    const i32x4 wraw = *reinterpret_cast<const i32x4*>(kv_norm_w + lane * 8);
    float v[8], ss = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t u = (uint32_t)raw[k];
        v[2 * k] = __uint_as_float(u << 16);
        v[2 * k + 1] = __uint_as_float(u & 0xffff0000u);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) ss += v[k] * v[k];
    ss = wave_reduce_sum(ss);
    const float rr = rsqrtf(ss / 512.0f + kv_eps);
    i32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t u = (uint32_t)wraw[k];
        o[k] = (int)f32x2_to_bf16x2((v[2 * k] * rr) * __uint_as_float(u << 16),
                                    (v[2 * k + 1] * rr) * __uint_as_float(u & 0xffff0000u));
    }
    if (row) *reinterpret_cast<i32x4*>(row + lane * 8) = o;
}

// RoPE(k_pe): lanes 0..31 of a wave, one pair each.
__device__ __forceinline__ void mla_kv_row_rope(int lane, int b, const bf16_t* src, const float* __restrict__ cos,
                                                const float* __restrict__ sin, bf16_t* row) {
#pragma clang fp contract(off)
    if (lane < 32 && row) {
        const float x0 = bf16_to_f32(src[512 + 2 * lane]), x1 = bf16_to_f32(src[512 + 2 * lane + 1]);
        const float c = cos[(int64_t)b * 32 + lane], s = sin[(int64_t)b * 32 + lane];
        *reinterpret_cast<uint32_t*>(row + 512 + 2 * lane) = f32x2_to_bf16x2(x0 * c - x1 * s, x1 * c + x0 * s);
    }
}

// kv_norm(kv_c) + RoPE(k_pe) of one token written straight into its page row (waves 0 and 1 of a
// workgroup); src = [kv_c (512) | k_pe (64)].
__device__ __forceinline__ void mla_kv_row(int b, const bf16_t* src, const bf16_t* __restrict__ kv_norm_w, float kv_eps,
                                           const float* __restrict__ cos, const float* __restrict__ sin,
                                           bf16_t* __restrict__ cache, int64_t num_pages, int page_size,
                                           const int32_t* __restrict__ table, int pages_per_seq,
                                           const int32_t* __restrict__ old_lens) {
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    if (wave > 1) return;
    bf16_t* row = mla_kv_row_ptr(b, cache, num_pages, page_size, table, pages_per_seq, old_lens);
    if (wave == 0) mla_kv_row_norm(lane, src, kv_norm_w, kv_eps, row);
    else mla_kv_row_rope(lane, b, src, cos, sin, row);
}

// The same row from ONE wave (both parts in turn): for launches whose workgroups are a single wave (absorb.hip).
__device__ __forceinline__ void mla_kv_row_one_wave(int b, const bf16_t* src, const bf16_t* __restrict__ kv_norm_w,
                                                    float kv_eps, const float* __restrict__ cos,
                                                    const float* __restrict__ sin, bf16_t* __restrict__ cache,
                                                    int64_t num_pages, int page_size, const int32_t* __restrict__ table,
                                                    int pages_per_seq, const int32_t* __restrict__ old_lens) {
    const int lane = threadIdx.x & 63;
    bf16_t* row = mla_kv_row_ptr(b, cache, num_pages, page_size, table, pages_per_seq, old_lens);
    mla_kv_row_norm(lane, src, kv_norm_w, kv_eps, row);
    mla_kv_row_rope(lane, b, src, cos, sin, row);
}

}  // namespace chitu
