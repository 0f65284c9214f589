// Paged-KV append and rotary embedding, gfx950.  Tiny byte movers: one launch each,
// 16-B accesses, everything read from persistent device buffers (graph-safe).
//
// Replaces (reference, read-only):
//   chitu/triton_kernels.py:18-48     append_to_paged_kv_cache_kernel  (page size 64 hard-coded)
//   chitu/triton_kernels.py:51-190    rotary_embedding_kernel_hf_llama / _llama
//   chitu/ops.py:51-91, 124-326       launchers + torch RoPE
#include "common.h"

namespace chitu {

// cache[(table[b][L/page] * page + L % page)] = this_kv[b]; rows are `row_bytes` long.
__global__ __launch_bounds__(128) void append_paged_kv_kernel(
    uint8_t* __restrict__ cache, int64_t num_pages, int page_size, int64_t row_bytes,
    const int32_t* __restrict__ table, int pages_per_seq, const uint8_t* __restrict__ this_kv,
    const int32_t* __restrict__ old_lens) {
    const int b = blockIdx.x;
    const int L = old_lens[b];
    const int pidx = L / page_size;
    if (L < 0 || pidx >= pages_per_seq) return;  // out-of-table: drop instead of corrupting memory
    const int64_t page = table[(int64_t)b * pages_per_seq + pidx];
    if (page < 0 || page >= num_pages) return;
    uint8_t* dst = cache + (page * page_size + (L % page_size)) * row_bytes;
    const uint8_t* src = this_kv + (int64_t)b * row_bytes;
    if ((row_bytes & 15) == 0) {
        for (int64_t i = threadIdx.x * 16; i < row_bytes; i += blockDim.x * 16)
            *reinterpret_cast<i32x4*>(dst + i) = *reinterpret_cast<const i32x4*>(src + i);
    } else {
        for (int64_t i = threadIdx.x; i < row_bytes; i += blockDim.x) dst[i] = src[i];
    }
}

template <typename T>
__device__ __forceinline__ float ld_f32(const T* p);
template <>
__device__ __forceinline__ float ld_f32<float>(const float* p) { return *p; }
struct bf16_e { uint16_t v; };
struct f16_e { uint16_t v; };
template <>
__device__ __forceinline__ float ld_f32<bf16_e>(const bf16_e* p) { return bf16_to_f32(p->v); }
template <>
__device__ __forceinline__ float ld_f32<f16_e>(const f16_e* p) { return f16_to_f32(p->v); }
__device__ __forceinline__ void st_f32(float* p, float v) { *p = v; }
__device__ __forceinline__ void st_f32(bf16_e* p, float v) { p->v = f32_to_bf16(v); }
__device__ __forceinline__ void st_f32(f16_e* p, float v) { p->v = f32_to_f16(v); }

// layout 0: interleaved (re, im) pairs ("llama"); 1: half-split ("hf-llama").
template <typename T>
__global__ __launch_bounds__(256) void rope_kernel(
    const T* __restrict__ q, const T* __restrict__ k, T* __restrict__ oq, T* __restrict__ ok,
    const float* __restrict__ cos, const float* __restrict__ sin, int bs, int qh, int kh, int d,
    int64_t q_sb, int64_t q_sh, int64_t k_sb, int64_t k_sh, int64_t oq_sb, int64_t oq_sh,
    int64_t ok_sb, int64_t ok_sh, int layout) {
#pragma clang fp contract(off)  // products rounded separately, like torch's q*cos + rot(q)*sin
    const int half = d >> 1;
    const int64_t total = (int64_t)bs * (qh + kh) * half;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx % half);
        const int64_t rest = idx / half;
        const int h = (int)(rest % (qh + kh));
        const int b = (int)(rest / (qh + kh));
        const T* src;
        T* dst;
        if (h < qh) {
            src = q + b * q_sb + h * q_sh;
            dst = oq + b * oq_sb + h * oq_sh;
        } else {
            src = k + b * k_sb + (h - qh) * k_sh;
            dst = ok + b * ok_sb + (h - qh) * ok_sh;
        }
        const int i0 = layout == 0 ? 2 * i : i;
        const int i1 = layout == 0 ? 2 * i + 1 : i + half;
        const float x0 = ld_f32<T>(src + i0), x1 = ld_f32<T>(src + i1);
        const float c = cos[(int64_t)b * half + i], s = sin[(int64_t)b * half + i];
        st_f32(dst + i0, x0 * c - x1 * s);
        st_f32(dst + i1, x1 * c + x0 * s);
    }
}

}  // namespace chitu

extern "C" int chitu_hip_append_paged_kv(void* kv_cache, int64_t num_pages, int32_t page_size,
                                         int64_t row_bytes, const int32_t* page_table,
                                         int32_t pages_per_seq, const void* this_kv,
                                         const int32_t* old_seq_lens, int32_t batch,
                                         void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(kv_cache && page_table && this_kv && old_seq_lens);
    CHITU_REQUIRE(num_pages >= 0 && page_size >= 1 && row_bytes >= 1 && pages_per_seq >= 1 && batch >= 0);
    if (batch == 0) return CHITU_OK;
    hipLaunchKernelGGL(append_paged_kv_kernel, dim3(batch), dim3(128), 0, (hipStream_t)stream,
                       (uint8_t*)kv_cache, num_pages, (int)page_size, row_bytes, page_table,
                       (int)pages_per_seq, (const uint8_t*)this_kv, old_seq_lens);
    CHITU_RETURN_LAUNCH_STATUS();
}

extern "C" int chitu_hip_rope(const void* q, const void* k, void* out_q, void* out_k,
                              const float* cos, const float* sin, int act_dtype, int32_t batch,
                              int32_t q_heads, int32_t k_heads, int32_t head_dim, int64_t q_sb,
                              int64_t q_sh, int64_t k_sb, int64_t k_sh, int64_t oq_sb,
                              int64_t oq_sh, int64_t ok_sb, int64_t ok_sh, int32_t layout,
                              void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(q && k && out_q && out_k && cos && sin);
    CHITU_REQUIRE(batch >= 0 && q_heads >= 0 && k_heads >= 0 && head_dim >= 2 && (head_dim & 1) == 0);
    CHITU_REQUIRE(layout == 0 || layout == 1);
    const int64_t total = (int64_t)batch * (q_heads + k_heads) * (head_dim / 2);
    if (total == 0) return CHITU_OK;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(T)                                                                              \
    hipLaunchKernelGGL(rope_kernel<T>, dim3(blocks), dim3(256), 0, st, (const T*)q, (const T*)k, \
                       (T*)out_q, (T*)out_k, cos, sin, (int)batch, (int)q_heads, (int)k_heads,   \
                       (int)head_dim, q_sb, q_sh, k_sb, k_sh, oq_sb, oq_sh, ok_sb, ok_sh, (int)layout)
    if (act_dtype == 0) LAUNCH(bf16_e);
    else if (act_dtype == 1) LAUNCH(f16_e);
    else if (act_dtype == 2) LAUNCH(float);
    else return CHITU_ERR_UNSUPPORTED;
#undef LAUNCH
    CHITU_RETURN_LAUNCH_STATUS();
}
