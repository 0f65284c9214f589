// MLA (absorb mode) causal prefill attention for gfx950: MQA with head dims 576 / 512 over the
// prompt's own keys -- the attn_varlen_func call of AttentionDeepSeekV3.prefill_forward
// (chitu/models/model_deepseek_v3.py:589-599; interface chitu/attn_backend.py:39-90; the reference
// runs it on third-party flash_attn or on RefAttnBackend's O(L^2)-memory torch path, :394-455).
//
//   out[t,h,:] = softmax_{s <= t, same sequence}( scale * q[t,h,:] . kv[s,:] ) . kv[s,:512]
//
// Same tile machinery as mla_decode.hip (one 64-key tile staged in LDS and used for QK^T and PV,
// register-staged one tile ahead, Q in LDS, PV B-operand via ds_read_b64_tr_b16) with BQ = 4 query
// tokens per workgroup sharing every staged tile: KV is read T/4 times per sequence instead of T
// times.  Per query token the arithmetic and its order are the decode kernel's (num_splits = 1), so
// prefill and token-by-token decode agree bit for bit on the attention output.
// grid (ceil(max_seqlen / 4), n_seq, heads/16); block 256; ~152 KB LDS, one workgroup per CU.
#include "common.h"

namespace chitu {

namespace pf {
constexpr int kC = 512, kR = 64;
constexpr int kTile = 64;
constexpr int kRowB = 1184;   // LDS row stride in bytes (1152 + 32 pad): conflict-free ds_read_b128 rows, see mla_decode.hip
constexpr int kPStride = 72;  // P row stride in bf16 elements
constexpr int kBQ = 4;        // query tokens per workgroup
}  // namespace pf

typedef __attribute__((address_space(3))) s16x4 lds_s16x4_pf;

__global__ __launch_bounds__(256, 1) void mla_prefill_kernel(
    const bf16_t* __restrict__ q, int64_t q_st, int64_t q_sh, const bf16_t* __restrict__ kv, int64_t kv_st,
    const int32_t* __restrict__ cu_seqlens, float scale, bf16_t* __restrict__ out, int H) {
    using namespace pf;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* kv_lds = smem;                                    // [64][kRowB]
    uint8_t* q_lds = smem + kTile * kRowB;                     // [kBQ][16][kRowB]
    bf16_t* p_lds = reinterpret_cast<bf16_t*>(q_lds + kBQ * 16 * kRowB);  // [16][72]
    float* red_max = reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(p_lds) + 16 * kPStride * 2);  // [4][16]
    float* red_sum = red_max + 64;                                                                      // [4][16]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: SGPR index math
    const int j = lane & 15, g = lane >> 4;
    const int seq = blockIdx.y, hb = blockIdx.z;
    const int s0 = cu_seqlens[seq], s1 = cu_seqlens[seq + 1];
    const int L = s1 - s0;
    const int p0 = blockIdx.x * kBQ;  // first query position of this block
    if (p0 >= L) return;
    const int nq = min(kBQ, L - p0);
    const int h0 = hb * 16;
    const int n_tiles = (p0 + nq + kTile - 1) / kTile;  // keys 0 .. p0+nq-1
    const bf16_t* kbase = kv + (int64_t)s0 * kv_st;

    i32x4 pfr[18];
    auto issue = [&](int tile) {
        const int t0 = tile * kTile;
        const int valid = min(kTile, p0 + nq - t0);
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const int c = tid + i * 256;
            const int row = c / 72, col = c % 72;
            pfr[i] = i32x4{0, 0, 0, 0};  // rows past the block's last key are staged as zeros
            if (row < valid) pfr[i] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(kbase + (int64_t)(t0 + row) * kv_st + col * 8));
        }
    };
    issue(0);
    // Q -> LDS: kBQ tokens x 16 heads x 576
    for (int c = tid; c < kBQ * 16 * 72; c += 256) {
        const int qi = c / (16 * 72), rem = c % (16 * 72);
        const int row = rem / 72, col = rem % 72;
        const int t = s0 + p0 + min(qi, nq - 1);
        const int h = min(h0 + row, H - 1);
        *reinterpret_cast<i32x4*>(q_lds + (qi * 16 + row) * kRowB + col * 16) =
            *reinterpret_cast<const i32x4*>(q + (int64_t)t * q_st + h * q_sh + col * 8);
    }

    f32x4 o[kBQ][8];
    float m_run[kBQ][4], l_run[kBQ][4];
#pragma unroll
    for (int qi = 0; qi < kBQ; ++qi) {
#pragma unroll
        for (int c = 0; c < 8; ++c) o[qi][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            m_run[qi][r] = -INFINITY;
            l_run[qi][r] = 0.f;
        }
    }

    for (int tile = 0; tile < n_tiles; ++tile) {
        const int t0 = tile * kTile;
        __syncthreads();  // previous tile fully consumed (and Q staged, first time round)
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const int c = tid + i * 256;
            *reinterpret_cast<i32x4*>(kv_lds + (c / 72) * kRowB + (c % 72) * 16) = pfr[i];
        }
        __syncthreads();
        if (tile + 1 < n_tiles) issue(tile + 1);

#pragma unroll
        for (int qi = 0; qi < kBQ; ++qi) {
            const int pq = p0 + qi;  // this query's position: keys 0..pq
            if (qi >= nq || t0 > pq) continue;  // workgroup-uniform
            // ---- S = Q K^T for this wave's 16 keys
            f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, sb = f32x4{0.f, 0.f, 0.f, 0.f};
            {
                const uint8_t* krow = kv_lds + (wave * 16 + j) * kRowB + g * 16;
                const uint8_t* qrow = q_lds + (qi * 16 + j) * kRowB + g * 16;
#pragma unroll
                for (int kk = 0; kk < 18; kk += 2) {
                    const s16x8 q0 = *reinterpret_cast<const s16x8*>(qrow + kk * 64);
                    const s16x8 k0 = *reinterpret_cast<const s16x8*>(krow + kk * 64);
                    const s16x8 q1 = *reinterpret_cast<const s16x8*>(qrow + kk * 64 + 64);
                    const s16x8 k1 = *reinterpret_cast<const s16x8*>(krow + kk * 64 + 64);
                    sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q0, k0, sa, 0, 0, 0);
                    sb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q1, k1, sb, 0, 0, 0);
                }
            }
            const bool tok_ok = (t0 + wave * 16 + j) <= pq;  // causal
            float sv[4], mx[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sv[r] = tok_ok ? (sa[r] + sb[r]) * scale : -INFINITY;
                mx[r] = row16_reduce_max(sv[r]);
            }
            if (j == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) red_max[wave * 16 + g * 4 + r] = mx[r];
            }
            __syncthreads();
            float alpha[4], psum[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int hh = g * 4 + r;
                const float mt = __builtin_fmaxf(__builtin_fmaxf(red_max[hh], red_max[16 + hh]),
                                                 __builtin_fmaxf(red_max[32 + hh], red_max[48 + hh]));
                const float m_new = __builtin_fmaxf(m_run[qi][r], mt);  // finite: the tile's first key is <= pq
                alpha[r] = __expf(m_run[qi][r] - m_new);
                m_run[qi][r] = m_new;
                const float p = __expf(sv[r] - m_new);
                psum[r] = row16_reduce_sum(p);
                p_lds[hh * kPStride + wave * 16 + j] = f32_to_bf16(p);
            }
            if (j == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) red_sum[wave * 16 + g * 4 + r] = psum[r];
            }
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[qi][c][r] *= alpha[r];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int hh = g * 4 + r;
                l_run[qi][r] = l_run[qi][r] * alpha[r] + (red_sum[hh] + red_sum[16 + hh] + red_sum[32 + hh] + red_sum[48 + hh]);
            }
            // ---- O += P V : this wave owns latent columns [wave*128, wave*128+128)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const s16x8 pfrag = *reinterpret_cast<const s16x8*>(p_lds + j * kPStride + ks * 32 + g * 8);
                const uint8_t* vbase = kv_lds + (ks * 32 + g * 8 + (j >> 2)) * kRowB + (wave * 128 + (j & 3) * 4) * 2;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_pf*)(vbase + c * 32));
                    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_pf*)(vbase + 4 * kRowB + c * 32));
                    s16x8 vf;
                    vf[0] = v0[0]; vf[1] = v0[1]; vf[2] = v0[2]; vf[3] = v0[3];
                    vf[4] = v1[0]; vf[5] = v1[1]; vf[6] = v1[2]; vf[7] = v1[3];
                    o[qi][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pfrag, vf, o[qi][c], 0, 0, 0);
                }
            }
            __syncthreads();  // P and the reduction scratch are reused by the next query token
        }
    }

    // ---- epilogue: lane holds O[qi][head 4g+r][col wave*128 + c*16 + j]
#pragma unroll
    for (int qi = 0; qi < kBQ; ++qi) {
        if (qi >= nq) break;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int h = h0 + g * 4 + r;
            if (h >= H) continue;
            const float inv = 1.0f / l_run[qi][r];
            bf16_t* dst = out + ((int64_t)(s0 + p0 + qi) * H + h) * kC + wave * 128 + j;
#pragma unroll
            for (int c = 0; c < 8; ++c) dst[c * 16] = f32_to_bf16(o[qi][c][r] * inv);
        }
    }
}

}  // namespace chitu

extern "C" int chitu_hip_mla_prefill(const void* q_bf16, int64_t q_stride_t, int64_t q_stride_h, const void* kv_bf16,
                                     int64_t kv_stride_t, const int32_t* cu_seqlens, int32_t n_seq, int32_t max_seqlen,
                                     float softmax_scale, void* out_bf16, int32_t heads, int32_t kv_lora_rank,
                                     int32_t rope_dim, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(q_bf16 && kv_bf16 && cu_seqlens && out_bf16 && n_seq >= 0 && max_seqlen >= 0 && heads >= 1);
    if (kv_lora_rank != pf::kC || rope_dim != pf::kR) return CHITU_ERR_UNSUPPORTED;
    CHITU_REQUIRE(q_stride_t % 8 == 0 && q_stride_h % 8 == 0 && kv_stride_t % 8 == 0);
    if (n_seq == 0 || max_seqlen == 0) return CHITU_OK;
    const size_t lds = (size_t)(pf::kTile + pf::kBQ * 16) * pf::kRowB + 16 * pf::kPStride * 2 + 2 * 64 * sizeof(float);
    // (set on every call: the opt-in is per device, a process-wide "done" flag would skip the second GPU of a multi-device process)
    (void)hipFuncSetAttribute((const void*)mla_prefill_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const dim3 grid((unsigned)((max_seqlen + pf::kBQ - 1) / pf::kBQ), (unsigned)n_seq, (unsigned)((heads + 15) / 16));
    hipLaunchKernelGGL(mla_prefill_kernel, grid, dim3(256), lds, (hipStream_t)stream, (const bf16_t*)q_bf16, q_stride_t,
                       q_stride_h, (const bf16_t*)kv_bf16, kv_stride_t, cu_seqlens, softmax_scale, (bf16_t*)out_bf16,
                       (int)heads);
    CHITU_RETURN_LAUNCH_STATUS();
}
