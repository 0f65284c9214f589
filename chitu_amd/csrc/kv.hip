// Paged-KV append and rotary embedding, gfx950.  Tiny byte movers: one launch each,
// 16-B accesses, everything read from persistent device buffers (graph-safe).
//
// Replaces (reference, read-only):
//   chitu/triton_kernels.py:18-48     append_to_paged_kv_cache_kernel  (page size 64 hard-coded)
//   chitu/triton_kernels.py:51-190    rotary_embedding_kernel_hf_llama / _llama
//   chitu/ops.py:51-91, 124-326       launchers + torch RoPE
#include "common.h"
#include "norm_common.h"
#include "mla_kv_row.h"

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

// MLA decode "KV prep": for each sequence, kv_norm(kv_c) and RoPE(k_pe) are written straight into
// the token's page row (no intermediate [kv | pe] tensor, no separate append), and RoPE is applied
// to q_pe in place.  Replaces four launches of the reference's decode_forward_paged
// (chitu/models/model_deepseek_v3.py:493-500 apply_rotary_pos_emb, :684 kv_norm, :686 torch.cat,
// attn_backend.py:720-722 append_to_paged_kv_cache) with identical arithmetic:
// RMSNorm in fp32 with one bf16 rounding, RoPE products rounded separately.
// One workgroup (256 threads) per sequence: wave 0 = norm (512 = 64 lanes x 8), wave 1 = k_pe,
// waves 2-3 = q_pe.  kv_lora_rank 512 / rope 64 (the MLA cache row is 576 wide).
__global__ __launch_bounds__(256) void mla_kv_prep_kernel(
    const bf16_t* __restrict__ kv_in, int64_t kv_stride, bf16_t* __restrict__ q_pe, int64_t q_sb, int64_t q_sh,
    int H, const float* __restrict__ cos, const float* __restrict__ sin, const bf16_t* __restrict__ norm_w,
    float eps, bf16_t* __restrict__ cache, int64_t num_pages, int page_size, const int32_t* __restrict__ table,
    int pages_per_seq, const int32_t* __restrict__ old_lens) {
#pragma clang fp contract(off)
    const int b = blockIdx.x, tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const bf16_t* src = kv_in + (int64_t)b * kv_stride;
    const float* cb = cos + (int64_t)b * 32;
    const float* sb = sin + (int64_t)b * 32;
    bf16_t* row = nullptr;
    {
        const int L = old_lens[b];
        const int pidx = L / page_size;
        if (L >= 0 && pidx < pages_per_seq) {
            const int64_t page = table[(int64_t)b * pages_per_seq + pidx];
            if (page >= 0 && page < num_pages) row = cache + (page * page_size + (L % page_size)) * 576;
        }
    }
    if (wave == 0) {
        const i32x4 raw = *reinterpret_cast<const i32x4*>(src + lane * 8);
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
        const float rr = rsqrtf(ss / 512.0f + eps);
        const i32x4 wraw = *reinterpret_cast<const i32x4*>(norm_w + lane * 8);
        i32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t u = (uint32_t)wraw[k];
            const uint16_t lo = f32_to_bf16((v[2 * k] * rr) * __uint_as_float(u << 16));
            const uint16_t hi = f32_to_bf16((v[2 * k + 1] * rr) * __uint_as_float(u & 0xffff0000u));
            o[k] = (int)((uint32_t)lo | ((uint32_t)hi << 16));
        }
        if (row) *reinterpret_cast<i32x4*>(row + lane * 8) = o;
    } else if (wave == 1) {
        if (lane < 32 && row) {
            const float x0 = bf16_to_f32(src[512 + 2 * lane]), x1 = bf16_to_f32(src[512 + 2 * lane + 1]);
            const float c = cb[lane], s = sb[lane];
            const uint32_t pk = (uint32_t)f32_to_bf16(x0 * c - x1 * s) | ((uint32_t)f32_to_bf16(x1 * c + x0 * s) << 16);
            *reinterpret_cast<uint32_t*>(row + 512 + 2 * lane) = pk;
        }
    } else {
        for (int idx = tid - 128; idx < H * 32; idx += 128) {
            const int h = idx >> 5, i = idx & 31;
            bf16_t* p = q_pe + b * q_sb + h * q_sh + 2 * i;
            const float x0 = bf16_to_f32(p[0]), x1 = bf16_to_f32(p[1]);
            const float c = cb[i], s = sb[i];
            const uint32_t pk = (uint32_t)f32_to_bf16(x0 * c - x1 * s) | ((uint32_t)f32_to_bf16(x1 * c + x0 * s) << 16);
            *reinterpret_cast<uint32_t*>(p) = pk;
        }
    }
}

// GQA / MHA decode "QKV post": RoPE on q (in place in the merged qkv row) and on k, with the rotated
// k and the v row written straight into the token's page (Attention.decode_forward_paged,
// models/model.py:167-198: apply_rotary_pos_emb + the in-place append that attn_with_kvcache does,
// attn_backend.py:108-115) -- three launches of the op-level path in one, same arithmetic as
// rope_kernel / append_paged_kv_kernel.  grid (batch); qkv row = [hq | hkv | hkv] heads of d bf16.
__global__ __launch_bounds__(256) void gqa_qkv_post_kernel(
    bf16_e* __restrict__ qkv, int64_t row_stride, int hq, int hkv, int d, const float* __restrict__ cos,
    const float* __restrict__ sin, int layout, bf16_e* __restrict__ k_cache, bf16_e* __restrict__ v_cache,
    int64_t num_pages, int page_size, const int32_t* __restrict__ table, int pages_per_seq,
    const int32_t* __restrict__ old_lens) {
#pragma clang fp contract(off)
    const int b = blockIdx.x, half = d >> 1;
    bf16_e* row = qkv + (int64_t)b * row_stride;
    int64_t dst_row = -1;
    {
        const int L = old_lens[b];
        const int pidx = L / page_size;
        if (L >= 0 && pidx < pages_per_seq) {
            const int64_t page = table[(int64_t)b * pages_per_seq + pidx];
            if (page >= 0 && page < num_pages) dst_row = (page * page_size + (L % page_size)) * (int64_t)hkv * d;
        }
    }
    for (int idx = threadIdx.x; idx < (hq + hkv) * half; idx += 256) {
        const int h = idx / half, i = idx % half;
        const int i0 = layout == 0 ? 2 * i : i, i1 = layout == 0 ? 2 * i + 1 : i + half;
        bf16_e* src = row + h * d;
        const float x0 = ld_f32<bf16_e>(src + i0), x1 = ld_f32<bf16_e>(src + i1);
        const float c = cos[(int64_t)b * half + i], s = sin[(int64_t)b * half + i];
        const float r0 = x0 * c - x1 * s, r1 = x1 * c + x0 * s;
        if (h < hq) {
            st_f32(src + i0, r0);
            st_f32(src + i1, r1);
        } else if (dst_row >= 0) {
            bf16_e* dst = k_cache + dst_row + (h - hq) * d;
            st_f32(dst + i0, r0);
            st_f32(dst + i1, r1);
        }
    }
    if (dst_row >= 0) {
        const bf16_e* vsrc = row + (hq + hkv) * d;
        for (int idx = threadIdx.x; idx < hkv * d / 8; idx += 256)
            *reinterpret_cast<i32x4*>(v_cache + dst_row + idx * 8) = *reinterpret_cast<const i32x4*>(vsrc + idx * 8);
    }
}

// Everything that consumes wqkv_a's output row [q_a (q_lora) | kv_c (512) | k_pe (64)] in ONE launch:
//   blockIdx.y == 0: q_norm + act_quant of q_a  -> the fp8 input of the wq_b GEMM (rmsnorm_row)
//   blockIdx.y == 1: kv_norm(kv_c) and RoPE(k_pe) written straight into the token's page row
// (the two parts of mla_kv_prep_kernel that do not need wq_b's output; q_pe is rotated by the
// W_UK absorb launch, absorb.hip).  Same arithmetic as chitu_hip_rmsnorm + chitu_hip_mla_kv_prep.
// num_partials > 0: `qkv` is not the bf16 GEMM output but the fp32 split-K planes [num_partials, batch,
// row_stride] of chitu_hip_fp8_gemm_blockscale_partials: the role's slice of row b is summed over the planes in
// order, rounded to bf16 once (= the GEMM's own output rounding) into LDS, and processed from there.
constexpr int kQkvPostMaxStage = 2048;  // q_lora_rank <= 2048 on the partial path
__global__ __launch_bounds__(256) void mla_qkv_post_kernel(
    const void* qkv_any, int num_partials, int batch, int64_t row_stride, int q_lora, const bf16_t* __restrict__ q_norm_w,
    float q_eps, fp8_t* __restrict__ q_fp8, float* __restrict__ q_scales, const bf16_t* __restrict__ kv_norm_w, float kv_eps,
    const float* __restrict__ cos, const float* __restrict__ sin, bf16_t* __restrict__ cache, int64_t num_pages,
    int page_size, const int32_t* __restrict__ table, int pages_per_seq, const int32_t* __restrict__ old_lens) {
    __shared__ __attribute__((aligned(16))) bf16_t stage[kQkvPostMaxStage];
    const int b = blockIdx.x;
    const bf16_t* qkv = reinterpret_cast<const bf16_t*>(qkv_any);
    const bf16_t* src = qkv + (int64_t)b * row_stride + (blockIdx.y == 0 ? 0 : q_lora);
    int64_t src_stride = row_stride;
    if (num_partials > 0) {
        const float* planes = reinterpret_cast<const float*>(qkv_any);
        const int col0 = blockIdx.y == 0 ? 0 : q_lora, n = blockIdx.y == 0 ? q_lora : 576;
        const int64_t plane = (int64_t)batch * row_stride;
        for (int c = threadIdx.x; c < (n >> 3); c += 256) {
            const float* p = planes + (int64_t)b * row_stride + col0 + c * 8;
            f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 4);
            for (int s = 1; s < num_partials; ++s) {
                const f32x4 l2 = *reinterpret_cast<const f32x4*>(p + s * plane), h2 = *reinterpret_cast<const f32x4*>(p + s * plane + 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) lo[r] += l2[r], hi[r] += h2[r];
            }
            i32x4 o;
            o[0] = (int)f32x2_to_bf16x2(lo[0], lo[1]), o[1] = (int)f32x2_to_bf16x2(lo[2], lo[3]);
            o[2] = (int)f32x2_to_bf16x2(hi[0], hi[1]), o[3] = (int)f32x2_to_bf16x2(hi[2], hi[3]);
            *reinterpret_cast<i32x4*>(stage + c * 8) = o;
        }
        __syncthreads();
        src = stage;
        src_stride = 0;
    }
    if (blockIdx.y == 0)
        rmsnorm_row<1, false>(b, src - (int64_t)b * src_stride, src_stride, nullptr, 0, nullptr, 0, q_norm_w, nullptr, 0, q_fp8,
                              q_scales, q_lora, q_eps, 0.f);
    else
        mla_kv_row(b, src, kv_norm_w, kv_eps, cos, sin, cache, num_pages, page_size, table, pages_per_seq, old_lens);
}

}  // namespace chitu

extern "C" int chitu_hip_mla_kv_prep(const void* kv_in_bf16, int64_t kv_row_stride, void* q_pe_bf16,
                                     int64_t q_stride_b, int64_t q_stride_h, int32_t heads,
                                     const float* cos, const float* sin, const void* kv_norm_weight_bf16,
                                     float eps, void* kv_cache, int64_t num_pages, int32_t page_size,
                                     const int32_t* page_table, int32_t pages_per_seq,
                                     const int32_t* old_seq_lens, int32_t batch, int32_t kv_lora_rank,
                                     int32_t rope_dim, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(kv_in_bf16 && q_pe_bf16 && cos && sin && kv_norm_weight_bf16 && kv_cache && page_table && old_seq_lens);
    CHITU_REQUIRE(batch >= 0 && heads >= 1 && num_pages >= 1 && page_size >= 1 && pages_per_seq >= 1);
    if (kv_lora_rank != 512 || rope_dim != 64) return CHITU_ERR_UNSUPPORTED;
    CHITU_REQUIRE(kv_row_stride % 8 == 0 && q_stride_b % 2 == 0 && q_stride_h % 2 == 0);
    if (batch == 0) return CHITU_OK;
    hipLaunchKernelGGL(mla_kv_prep_kernel, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)kv_in_bf16, kv_row_stride, (bf16_t*)q_pe_bf16, q_stride_b, q_stride_h,
                       (int)heads, cos, sin, (const bf16_t*)kv_norm_weight_bf16, eps, (bf16_t*)kv_cache, num_pages,
                       (int)page_size, page_table, (int)pages_per_seq, old_seq_lens);
    CHITU_RETURN_LAUNCH_STATUS();
}

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

extern "C" int chitu_hip_mla_qkv_post(const void* qkv_a, int32_t num_partials, int64_t row_stride, int32_t q_lora_rank,
                                      const void* q_norm_weight_bf16, float q_eps, void* q_fp8, float* q_scales,
                                      const void* kv_norm_weight_bf16, float kv_eps, const float* cos,
                                      const float* sin, void* kv_cache, int64_t num_pages, int32_t page_size,
                                      const int32_t* page_table, int32_t pages_per_seq,
                                      const int32_t* old_seq_lens, int32_t batch, int32_t kv_lora_rank,
                                      int32_t rope_dim, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(qkv_a && q_norm_weight_bf16 && q_fp8 && q_scales && kv_norm_weight_bf16 && cos && sin);
    CHITU_REQUIRE(kv_cache && page_table && old_seq_lens);
    CHITU_REQUIRE(batch >= 0 && num_pages >= 1 && page_size >= 1 && pages_per_seq >= 1 && q_lora_rank >= 128);
    CHITU_REQUIRE(num_partials >= 0 && num_partials <= 16);
    if (kv_lora_rank != 512 || rope_dim != 64) return CHITU_ERR_UNSUPPORTED;
    if (q_lora_rank % 128 != 0 || q_lora_rank > kNormThreads * 8 * kNormMaxChunks) return CHITU_ERR_UNSUPPORTED;
    if (num_partials > 0 && q_lora_rank > kQkvPostMaxStage) return CHITU_ERR_UNSUPPORTED;
    CHITU_REQUIRE(row_stride % 8 == 0 && row_stride >= q_lora_rank + 576);
    if (batch == 0) return CHITU_OK;
    hipLaunchKernelGGL(mla_qkv_post_kernel, dim3((unsigned)batch, 2), dim3(256), 0, (hipStream_t)stream, qkv_a,
                       (int)num_partials, (int)batch, row_stride, (int)q_lora_rank, (const bf16_t*)q_norm_weight_bf16, q_eps,
                       (fp8_t*)q_fp8, q_scales, (const bf16_t*)kv_norm_weight_bf16, kv_eps, cos, sin, (bf16_t*)kv_cache,
                       num_pages, (int)page_size, page_table, (int)pages_per_seq, old_seq_lens);
    CHITU_RETURN_LAUNCH_STATUS();
}

// ---- the decode step's prologue in one launch: vocabulary-parallel embedding lookup (ids outside this rank's
// row range give a zero row, chitu/tensor_parallel.py:199-208) and the gather of every sequence's rotary row
// (prepare_freqs_cis_decode, chitu/models/model.py:429-448).  One workgroup per token.
namespace chitu {
__global__ __launch_bounds__(256) void embed_rope_gather_kernel(
    const int64_t* __restrict__ tokens, const bf16_t* __restrict__ table, int64_t vocab_start, int64_t vocab_local, int dim,
    bf16_t* __restrict__ h, const int32_t* __restrict__ positions, const float* __restrict__ cos_table,
    const float* __restrict__ sin_table, int64_t table_rows, int half, float* __restrict__ cos_out, float* __restrict__ sin_out) {
    const int b = blockIdx.x;
    const int64_t local = tokens[b] - vocab_start;
    const bool mine = local >= 0 && local < vocab_local;
    const bf16_t* src = table + (mine ? local : 0) * dim;
    for (int c = threadIdx.x; c < (dim >> 3); c += 256) {
        i32x4 v = {0, 0, 0, 0};
        if (mine) v = *reinterpret_cast<const i32x4*>(src + c * 8);
        *reinterpret_cast<i32x4*>(h + (int64_t)b * dim + c * 8) = v;
    }
    if (cos_out) {
        const int64_t pos = min((int64_t)max(positions[b], 0), table_rows - 1);
        for (int i = threadIdx.x; i < half; i += 256) {
            cos_out[(int64_t)b * half + i] = cos_table[pos * half + i];
            sin_out[(int64_t)b * half + i] = sin_table[pos * half + i];
        }
    }
}
}  // namespace chitu

extern "C" int chitu_hip_embed_rope_gather(const int64_t* tokens, const void* embed_bf16, int64_t vocab_start,
                                           int64_t vocab_local, int32_t dim, void* h_bf16, const int32_t* positions,
                                           const float* cos_table, const float* sin_table, int64_t table_rows,
                                           int32_t half, float* cos_out, float* sin_out, int32_t batch, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(tokens && embed_bf16 && h_bf16 && vocab_local >= 1 && dim >= 8 && batch >= 0);
    if (dim % 8 != 0) return CHITU_ERR_UNSUPPORTED;
    CHITU_REQUIRE(!cos_out || (positions && cos_table && sin_table && sin_out && table_rows >= 1 && half >= 1));
    if (batch == 0) return CHITU_OK;
    hipLaunchKernelGGL(embed_rope_gather_kernel, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, tokens,
                       (const bf16_t*)embed_bf16, vocab_start, vocab_local, (int)dim, (bf16_t*)h_bf16, positions, cos_table,
                       sin_table, table_rows, (int)half, cos_out, sin_out);
    CHITU_RETURN_LAUNCH_STATUS();
}

extern "C" int chitu_hip_gqa_qkv_post(void* qkv_bf16, int64_t row_stride, int32_t q_heads, int32_t kv_heads,
                                      int32_t head_dim, const float* cos, const float* sin, int32_t layout,
                                      void* k_cache, void* v_cache, int64_t num_pages, int32_t page_size,
                                      const int32_t* page_table, int32_t pages_per_seq,
                                      const int32_t* old_seq_lens, int32_t batch, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(qkv_bf16 && cos && sin && k_cache && v_cache && page_table && old_seq_lens);
    CHITU_REQUIRE(batch >= 0 && q_heads >= 1 && kv_heads >= 1 && head_dim >= 8 && num_pages >= 1 && page_size >= 1);
    CHITU_REQUIRE(pages_per_seq >= 1 && (layout == 0 || layout == 1));
    if (head_dim % 8 != 0 || row_stride % 8 != 0 || row_stride < (int64_t)(q_heads + 2 * kv_heads) * head_dim)
        return CHITU_ERR_UNSUPPORTED;
    if (batch == 0) return CHITU_OK;
    hipLaunchKernelGGL(gqa_qkv_post_kernel, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, (bf16_e*)qkv_bf16,
                       row_stride, (int)q_heads, (int)kv_heads, (int)head_dim, cos, sin, (int)layout, (bf16_e*)k_cache,
                       (bf16_e*)v_cache, num_pages, (int)page_size, page_table, (int)pages_per_seq, old_seq_lens);
    CHITU_RETURN_LAUNCH_STATUS();
}
