// GQA / MHA paged decode attention (head_dim 128) for gfx950 -- the non-MLA decode path.
//
// Replaces (reference, read-only) the third-party call behind
//   FlashAttnBackend.attn_with_kvcache   chitu/attn_backend.py:208-243  (flash_attn.flash_attn_with_kvcache)
//   contract text                        chitu/attn_backend.py:92-164
// as used by Attention.decode_forward_paged (chitu/models/model.py:167-198, model_hf_llama.py:218-252):
//   out[b,h,:] = softmax_t(scale * q[b,h,:] . K[t, h/g, :]) . V[t, h/g, :],  t < seqlens[b]
// over a paged cache [pages, page_size, kv_heads, 128] (page_size % 16 == 0; the reference uses 256).
// The in-place append of this step's k/v (contract :108-115) is chitu_hip_append_paged_kv, run first.
//
// One wave per (KV split, sequence, kv head); the q heads of the group (<= 16) ride in the MFMA N
// dimension.  Per 16 tokens: K rows are loaded straight into A fragments (16 B per lane, 64 B
// contiguous per token per instruction), S^T = K Q^T by 4 x v_mfma_f32_16x16x32_bf16, so a lane holds
// S[4 tokens][one head] and P is already the A fragment of v_mfma_f32_16x16x16_bf16; V's 16 x 128
// sub-tile goes through a wave-private 4 KB LDS slab and is read back transposed
// (ds_read_b64_tr_b16).  Wave-local online softmax with deferred max; splits merged by a second tiny
// kernel (LSE), like the MLA path.
#include "common.h"

namespace chitu {

constexpr int kHd = 128;
constexpr int kVRowB = 272;  // LDS row stride of the V slab (256 B + 16 pad)
constexpr float kGqaDefer = 6.0f;

typedef short s16x4g __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4g;

// grid (num_splits, batch * kv_heads); block 64.
__global__ __launch_bounds__(64) void gqa_decode_kernel(
    const bf16_t* __restrict__ q, int64_t q_sb, int64_t q_sh, const bf16_t* __restrict__ kc,
    const bf16_t* __restrict__ vc, int64_t num_pages, int page_size, int Hkv, const int32_t* __restrict__ table,
    int table_stride, const int32_t* __restrict__ seqlens, float scale, float* __restrict__ part_o,
    float* __restrict__ part_lse, bf16_t* __restrict__ out, int Hq, int num_splits) {
    __shared__ __attribute__((aligned(16))) uint8_t vlds[16 * kVRowB];
    const int lane = threadIdx.x, j = lane & 15, g = lane >> 4;
    const int split = blockIdx.x, b = blockIdx.y / Hkv, kvh = blockIdx.y % Hkv;
    const int G = Hq / Hkv;  // q heads per kv head (<= 16)
    const int L = max(seqlens[b], 0);  // a corrupt negative length is an empty sequence, not a huge unsigned range
    const int n16 = (L + 15) >> 4;
    // 32-bit unsigned quotients: a 64-bit division is a software loop on the kernel's critical chain (the launcher bounds
    // 16-token steps x splits below 2^31)
    const int s0i = (int)((unsigned)n16 * (unsigned)split / (unsigned)num_splits);
    const int s1i = (int)((unsigned)n16 * (unsigned)(split + 1) / (unsigned)num_splits);
    const int32_t* tbl = table + (int64_t)b * table_stride;
    const int64_t tok_stride = (int64_t)Hkv * kHd;

    // Q^T fragments (B operand): lane holds q[head j][kk*32 + g*8 ..], zero for j >= G
    s16x8 qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        qf[kk] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (j < G) qf[kk] = *reinterpret_cast<const s16x8*>(q + b * q_sb + (kvh * G + j) * q_sh + kk * 32 + g * 8);
    }
    f32x4 o[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    s16x8 kf[4];
    i32x4 vr[4];
    auto issue = [&](int step) {
        const int t0 = step * 16;
        int64_t page = tbl[t0 / page_size];
        if (page < 0 || page >= num_pages) page = 0;
        const int64_t base = (page * page_size + (t0 % page_size)) * tok_stride + (int64_t)kvh * kHd;
        const int tk = min(j, L - 1 - t0);  // rows past the end re-read the last valid row (masked below)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            kf[kk] = *reinterpret_cast<const s16x8*>(kc + base + (int64_t)max(tk, 0) * tok_stride + kk * 32 + g * 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane + i * 64, row = c >> 4, col = c & 15;
            vr[i] = i32x4{0, 0, 0, 0};  // V rows past the end are staged as zeros (0 * garbage must stay 0)
            if (t0 + row < L) vr[i] = *reinterpret_cast<const i32x4*>(vc + base + (int64_t)row * tok_stride + col * 8);
        }
    };
    if (s0i < s1i) issue(s0i);
    for (int step = s0i; step < s1i; ++step) {
        const int t0 = step * 16;
        s16x8 kcur[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) kcur[kk] = kf[kk];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane + i * 64;
            *reinterpret_cast<i32x4*>(vlds + (c >> 4) * kVRowB + (c & 15) * 16) = vr[i];
        }
        if (step + 1 < s1i) issue(step + 1);
        // ---- S^T = K Q^T : lane holds S[token t0 + 4g + r][head j]
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kcur[kk], qf[kk], s, 0, 0, 0);
        float sv[4], mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sv[r] = (t0 + g * 4 + r) < L ? s[r] * scale : -INFINITY;
            mx = __builtin_fmaxf(mx, sv[r]);
        }
        float al[4] = {1.f, 1.f, 1.f, 1.f};
        const bool rescale = __any(mx > m_run + kGqaDefer);
        if (rescale) {
            mx = __builtin_fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = __builtin_fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = __builtin_fmaxf(m_run, mx);
            const float alpha = m_new == -INFINITY ? 1.f : __expf(m_run - m_new);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 4; ++r) al[r] = __shfl(alpha, g * 4 + r, 64);
        }
        const float m_safe = m_run == -INFINITY ? 0.f : m_run;
        s16x4g pa;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p = __expf(sv[r] - m_safe);
            l_run += p;
            pa[r] = (short)f32_to_bf16(p);
        }
        // ---- O += P V : B fragment = 4 token rows at one head-dim column (transpose read)
        const uint8_t* vbase = vlds + (g * 4 + (j >> 2)) * kVRowB + (j & 3) * 8;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const s16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4g*)(vbase + c * 32));
            if (rescale) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[c][r] *= al[r];
            }
            o[c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa, vb, o[c], 0, 0, 0);
        }
    }
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    // lane holds O[head 4g+r][dim c*16 + j]; its (m, l) are for head j -> fetch those of heads 4g+r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int hh = g * 4 + r;
        const float l = __shfl(l_run, hh, 64), m = __shfl(m_run, hh, 64);
        if (hh >= G) continue;
        const int h = kvh * G + hh;
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        if (num_splits == 1) {
#pragma unroll
            for (int c = 0; c < 8; ++c) out[((int64_t)b * Hq + h) * kHd + c * 16 + j] = f32_to_bf16(o[c][r] * inv);
        } else {
            float* dst = part_o + (((int64_t)b * Hq + h) * num_splits + split) * kHd;
#pragma unroll
            for (int c = 0; c < 8; ++c) dst[c * 16 + j] = o[c][r] * inv;
            if (j == 0) part_lse[((int64_t)b * Hq + h) * num_splits + split] = l > 0.f ? m + __logf(l) : -INFINITY;
        }
    }
}

// out[b,h,:] = sum_s w_s part_o[b,h,s,:] / sum_s w_s, w_s = exp(lse_s - max lse).  grid (batch*heads); one wave.
// Nothing in here is a chain of dependent loads: the lse values are read one per lane, the weights travel by
// shuffle, and the rows of part_o are read with data-independent addresses (8 in flight per lane); the two
// halves of the wave take the even and the odd splits and are added at the end.
__global__ __launch_bounds__(64) void gqa_merge_kernel(const float* __restrict__ part_o,
                                                       const float* __restrict__ part_lse,
                                                       bf16_t* __restrict__ out, int num_splits) {
    const int64_t bh = blockIdx.x;
    const int lane = threadIdx.x, half = lane >> 5, col = (lane & 31) * 4;
    const float* lse = part_lse + bh * num_splits;
    float ls[4], m = -INFINITY;  // num_splits <= 256
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = lane + 64 * i;
        ls[i] = s < num_splits ? lse[s] : -INFINITY;
        m = __builtin_fmaxf(m, ls[i]);
    }
    m = wave_reduce_max(m);
    float w[4], wsum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        w[i] = ls[i] == -INFINITY ? 0.f : __expf(ls[i] - m);
        wsum += w[i];
    }
    wsum = wave_reduce_sum(wsum);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* rows = part_o + bh * num_splits * kHd + col;
    // The trip count is the SAME for both halves (s0 is wave-uniform, the half enters through `s`): the __shfl below
    // is a ds_bpermute, which returns 0 for a source lane that has left the loop, so a half that exits one
    // iteration early (num_splits % 16 == 1) would drop the weights the other half still fetches from its lanes.
    for (int s0 = 0; s0 < num_splits; s0 += 16) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(rows + (int64_t)min(s0 + half + 2 * u, num_splits - 1) * kHd);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int s = s0 + half + 2 * u;  // uniform per half
            const float ws_all = s < 64 ? w[0] : s < 128 ? w[1] : s < 192 ? w[2] : w[3];
            const float ws = __shfl(ws_all, s & 63, 64);
            if (s < num_splits && ws > 0.f) {  // an empty split's row is never used (it may hold anything)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] += ws * v[u][i];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] += __shfl_xor(acc[i], 32, 64);
    if (half) return;
    const float inv = wsum > 0.f ? 1.0f / wsum : 0.f;
    i32x2 o2;
    o2[0] = (int)f32x2_to_bf16x2(acc[0] * inv, acc[1] * inv);
    o2[1] = (int)f32x2_to_bf16x2(acc[2] * inv, acc[3] * inv);
    *reinterpret_cast<i32x2*>(out + bh * kHd + col) = o2;
}

}  // namespace chitu

extern "C" int chitu_hip_gqa_decode(const void* q_bf16, int64_t q_stride_b, int64_t q_stride_h,
                                    const void* k_cache, const void* v_cache, int64_t num_pages,
                                    int32_t page_size, int32_t kv_heads, const int32_t* block_table,
                                    int32_t table_stride, const int32_t* seqlens, float softmax_scale,
                                    void* out_bf16, int32_t batch, int32_t q_heads, int32_t head_dim,
                                    int32_t num_splits, void* workspace, int64_t workspace_bytes,
                                    void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(q_bf16 && k_cache && v_cache && block_table && seqlens && out_bf16);
    CHITU_REQUIRE(batch >= 0 && q_heads >= 1 && kv_heads >= 1 && num_pages >= 1 && table_stride >= 1);
    CHITU_REQUIRE(q_heads % kv_heads == 0 && num_splits >= 1 && num_splits <= 256);
    if (head_dim != kHd || q_heads / kv_heads > 16) return CHITU_ERR_UNSUPPORTED;
    if (page_size < 16 || page_size % 16 != 0) return CHITU_ERR_UNSUPPORTED;
    CHITU_REQUIRE(q_stride_b % 8 == 0 && q_stride_h % 8 == 0);
    CHITU_REQUIRE((int64_t)table_stride * (page_size / 16) * (num_splits + 1) < (1ll << 31));  // 32-bit split arithmetic
    if (batch == 0) return CHITU_OK;
    float* part_o = nullptr;
    float* part_lse = nullptr;
    if (num_splits > 1) {
        const int64_t need = (int64_t)batch * q_heads * num_splits * (kHd + 1) * 4;
        CHITU_REQUIRE(workspace && workspace_bytes >= need);
        part_o = (float*)workspace;
        part_lse = part_o + (int64_t)batch * q_heads * num_splits * kHd;
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gqa_decode_kernel, dim3((unsigned)num_splits, (unsigned)(batch * kv_heads)), dim3(64), 0, st,
                       (const bf16_t*)q_bf16, q_stride_b, q_stride_h, (const bf16_t*)k_cache, (const bf16_t*)v_cache,
                       num_pages, (int)page_size, (int)kv_heads, block_table, (int)table_stride, seqlens,
                       softmax_scale, part_o, part_lse, (bf16_t*)out_bf16, (int)q_heads, (int)num_splits);
    if (num_splits > 1)
        hipLaunchKernelGGL(gqa_merge_kernel, dim3((unsigned)(batch * q_heads)), dim3(64), 0, st, part_o, part_lse,
                           (bf16_t*)out_bf16, (int)num_splits);
    CHITU_RETURN_LAUNCH_STATUS();
}
