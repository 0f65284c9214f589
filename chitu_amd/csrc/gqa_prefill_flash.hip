// Causal GQA / MHA prefill attention for gfx950, head_dim 128, flash form -- the attn_varlen_func call of
// Attention.prefill_forward (chitu/models/model.py:104-132; interface chitu/attn_backend.py:39-90; the reference runs it on
// third-party flash_attn, attn_backend.py:167-206, or on RefAttnBackend's O(L^2)-memory torch path, :394-455).
//
//   out[t,h,:] = softmax_{s <= t, same sequence}( scale * q[t,h,:] . k[s,h / G,:] ) . v[s,h / G,:]        G = Hq / Hkv
//
// The structure of mla_prefill_flash.hip at head_dim 128: the G query heads of a KV head are rows of one Q matrix, a
// workgroup owns 128 Q rows (128 / G consecutive tokens x G heads) of one KV head, each of its 4 waves 32 of them;
//   * Q fragments in registers (8 x 4 VGPRs), S^T = K Q^T by v_mfma_f32_32x32x16_bf16 so a Q row's scores of a 32-key block
//     sit in one lane pair: in-lane softmax, running maximum moved only when a block outgrows it by 2^8 (deferred rescale);
//   * O^T = V^T P^T: bf16 P is already the B fragment, V^T fragments by ds_read_b64_tr_b16, the 32 x 128 accumulator (64
//     registers) per Q row in-lane;
//   * 64-key K and V tiles by LDS-DMA (lds_dma.h) into a 2-deep ring, one barrier per tile; K rows [64][256 B] with chunk c of
//     key r at c ^ (r & 15) (ds_read_b128 of 16 keys x one chunk column: 16 bank groups), V rows with chunk c at
//     c ^ ((r & 3) << 2) (four consecutive keys of a transposed read: four 64-byte bank quarters);
//   * 64 KB of LDS and <= 256 registers: two workgroups per CU, so one workgroup's softmax runs beside the other's MFMAs.
// KV is read once per 128 / G query tokens (the composition this replaces -- one decode launch row per query token over
// staged pages, attn_backend._gqa_varlen_causal of rounds 2-4 -- read it once per token).  Causal work grows with the block
// index: blocks are issued heaviest first.
#include <type_traits>

#include "common.h"
#include "lds_dma.h"

namespace chitu {

namespace gpf {
constexpr int kD = 128;                // head_dim
constexpr int kTile = 64;              // keys per staged tile
constexpr int kRow = kD * 2;           // bytes per K / V row
constexpr int kTileB = kTile * kRow;   // 16 KB
constexpr int kBuf = 2 * kTileB;       // K + V of one ring slot
constexpr float kDefer = 8.0f;
}  // namespace gpf

typedef float f32x16g __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_gp;

template <int G>  // query heads per KV head (1 .. 32, a power of two)
__global__ __launch_bounds__(256, 2) void gqa_prefill_flash_kernel(
    const bf16_t* __restrict__ q, int64_t q_st, int64_t q_sh, const bf16_t* __restrict__ k, int64_t k_st, int64_t k_sh,
    const bf16_t* __restrict__ v, int64_t v_st, int64_t v_sh, const int32_t* __restrict__ cu_seqlens, float scale,
    bf16_t* __restrict__ out, int Hq) {
    using namespace gpf;
    constexpr int TPW = 32 / G;   // query tokens per wave
    constexpr int BQ = 4 * TPW;   // per workgroup
    __shared__ __attribute__((aligned(16))) uint8_t smem[2 * kBuf];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = lane & 31, hi = lane >> 5;
    const int seq = blockIdx.y, kvh = blockIdx.z;
    const int s0 = cu_seqlens[seq], L = cu_seqlens[seq + 1] - s0;
    const int p0 = ((int)gridDim.x - 1 - (int)blockIdx.x) * BQ;  // heaviest (latest) block first
    if (p0 >= L) return;
    const int nq = min(BQ, L - p0);
    const int n_keys = p0 + nq;
    const int n_tiles = (n_keys + kTile - 1) / kTile;
    const int tq = wave * TPW + row / G;    // this lane's query token within the block (one past the end repeats the last, stores nothing)
    const int pq = p0 + min(tq, nq - 1);    // its position: keys 0 .. pq
    const int head = kvh * G + row % G;
    const bf16_t* kbase = k + (int64_t)s0 * k_st + (int64_t)kvh * k_sh;
    const bf16_t* vbase = v + (int64_t)s0 * v_st + (int64_t)kvh * v_sh;
    const uint32_t lds0 = lds_offset_of(&smem[0]);

    // ---- tile DMA: wave w brings keys 16 w .. 16 w + 15 of K and of V, four 4-row pieces each
    const int dr = lane >> 4, dp = lane & 15;  // row inside a piece, chunk position
    auto issue = [&](int tile) {
        const int t0 = tile * kTile;
        const uint32_t dst = lds0 + (uint32_t)((tile & 1) * kBuf);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = wave * 16 + i * 4 + dr;
            const int64_t grow = min(t0 + r, L - 1);  // rows past the sequence repeat its last key (finite, masked by causality)
            glds16_vaddr(kbase + grow * k_st + ((dp ^ (r & 15)) << 3), dst + (uint32_t)((wave * 4 + i) * 1024));
            glds16_vaddr(vbase + grow * v_st + ((dp ^ ((r & 3) << 2)) << 3), dst + (uint32_t)(kTileB + (wave * 4 + i) * 1024));
        }
    };
    issue(0);

    // ---- Q fragments (B operand of S^T = K Q^T): lane (row, hi) holds q[token][head][16 kk + 8 hi .. + 8]
    s16x8 qf[8];
    {
        const bf16_t* qp = q + (int64_t)(s0 + pq) * q_st + (int64_t)head * q_sh + hi * 8;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) qf[kk] = *reinterpret_cast<const s16x8*>(qp + kk * 16);
        // hipcc must wait for these loads HERE, not inside the tile loop (its vmcnt ladder would drain the LDS-DMA it cannot see)
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) asm volatile("" : "+v"(qf[kk]));
    }
    f32x16g o[4];  // O^T[dim 32 cb + crow(reg, hi)][this lane's Q row]
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[cb][r] = 0.f;
    float m = -INFINITY, l = 0.f;
    const float c2 = scale * 1.4426950408889634f;

    // lane-constant LDS offsets
    int koff[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) koff[kk] = row * kRow + (((2 * kk + hi) ^ (lane & 15)) << 4);
    const int rr = (lane & 15) >> 2, cl = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
    int voff[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) voff[cb] = (4 * hi + rr) * kRow + ((cb ^ rr) << 6) + (cl << 4) + ((lane & 1) << 3);

    for (int tile = 0; tile < n_tiles; ++tile) {
        const int t0 = tile * kTile;
        glds_wait_all();   // this wave's pieces of the tile have landed
        __syncthreads();   // everyone's have; the other ring slot is no longer being read
        if (tile + 1 < n_tiles) issue(tile + 1);
        const uint8_t* bufK = smem + (tile & 1) * kBuf;
        const uint8_t* bufV = bufK + kTileB;
        const bool last = t0 + kTile > p0;  // the tile reaches into the block's own token range (two tiles at G = 1): mask by position
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (t0 + kb * 32 >= n_keys) break;  // the block's last key is before this half tile (workgroup-uniform)
            // ---- S^T[key 32 kb + crow(reg, hi)][Q row]
            f32x16g s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const s16x8 kf = *reinterpret_cast<const s16x8*>(bufK + kb * 32 * kRow + koff[kk]);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], s, 0, 0, 0);
            }
            float pmax = -INFINITY;
            if (last) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    s[r] = key <= pq ? s[r] * c2 : -INFINITY;
                    pmax = __builtin_fmaxf(pmax, s[r]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[r] *= c2;
                    pmax = __builtin_fmaxf(pmax, s[r]);
                }
            }
            if (__builtin_amdgcn_ballot_w64(pmax > m + kDefer) != 0) {  // m = -inf (first block): every lane votes
                const float mx = __builtin_fmaxf(pmax, __shfl_xor(pmax, 32, 64));
                const float m_new = __builtin_fmaxf(m, mx);  // finite: key 0 is visible to every row
                const float alpha = __builtin_amdgcn_exp2f(m - m_new);
                m = m_new;
                l *= alpha;
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[cb][r] *= alpha;
            }
            s16x8 pb[2];
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float a = __builtin_amdgcn_exp2f(s[r] - m);
                const float b = __builtin_amdgcn_exp2f(s[r + 1] - m);
                psum += a + b;
                const uint32_t pk = f32x2_to_bf16x2(a, b);
                pb[r >> 3][r & 7] = (short)(pk & 0xffffu);
                pb[r >> 3][(r & 7) + 1] = (short)(pk >> 16);
            }
            l += psum;
            // ---- O^T += V^T P^T: k-slot 8 hi + e of step t <-> key 32 kb + 16 t + 8 (e >> 2) + 4 hi + (e & 3)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const uint8_t* vb = bufV + (kb * 32 + t * 16) * kRow;
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_gp*)(vb + voff[cb]));
                    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_gp*)(vb + 8 * kRow + voff[cb]));
                    s16x8 vf;
                    vf[0] = v0[0]; vf[1] = v0[1]; vf[2] = v0[2]; vf[3] = v0[3];
                    vf[4] = v1[0]; vf[5] = v1[1]; vf[6] = v1[2]; vf[7] = v1[3];
                    o[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb[t], o[cb], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: lane holds dims 32 cb + 8 rq + 4 hi + {0..3} of its row: 8-byte stores (l and l ^ 32 fill 16 bytes)
    const float inv = 1.0f / (l + __shfl_xor(l, 32, 64));
    if (tq >= nq || head >= Hq) return;
    bf16_t* dst = out + ((int64_t)(s0 + p0 + tq) * Hq + head) * kD + 4 * hi;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            i32x2 pk;
            pk[0] = (int)f32x2_to_bf16x2(o[cb][4 * rq] * inv, o[cb][4 * rq + 1] * inv);
            pk[1] = (int)f32x2_to_bf16x2(o[cb][4 * rq + 2] * inv, o[cb][4 * rq + 3] * inv);
            *reinterpret_cast<i32x2*>(dst + 32 * cb + 8 * rq) = pk;
        }
    }
}

}  // namespace chitu

extern "C" int chitu_hip_gqa_prefill(const void* q_bf16, int64_t q_stride_t, int64_t q_stride_h, const void* k_bf16,
                                     int64_t k_stride_t, int64_t k_stride_h, const void* v_bf16, int64_t v_stride_t,
                                     int64_t v_stride_h, const int32_t* cu_seqlens, int32_t n_seq, int32_t max_seqlen,
                                     float softmax_scale, void* out_bf16, int32_t q_heads, int32_t kv_heads, int32_t head_dim,
                                     void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(q_bf16 && k_bf16 && v_bf16 && cu_seqlens && out_bf16 && n_seq >= 0 && max_seqlen >= 0);
    CHITU_REQUIRE(q_heads >= 1 && kv_heads >= 1 && q_heads % kv_heads == 0);
    if (head_dim != gpf::kD) return CHITU_ERR_UNSUPPORTED;
    const int G = q_heads / kv_heads;
    if (G > 32 || (G & (G - 1)) != 0) return CHITU_ERR_UNSUPPORTED;
    CHITU_REQUIRE((q_stride_t | q_stride_h | k_stride_t | k_stride_h | v_stride_t | v_stride_h) % 8 == 0);
    CHITU_REQUIRE(((uintptr_t)q_bf16 | (uintptr_t)k_bf16 | (uintptr_t)v_bf16 | (uintptr_t)out_bf16) % 16 == 0);
    if (n_seq == 0 || max_seqlen == 0) return CHITU_OK;
    const int BQ = 128 / G;
    const dim3 grid((unsigned)((max_seqlen + BQ - 1) / BQ), (unsigned)n_seq, (unsigned)kv_heads);
#define LAUNCH_G(GV)                                                                                                         \
    hipLaunchKernelGGL((gqa_prefill_flash_kernel<GV>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q_bf16, q_stride_t, \
                       q_stride_h, (const bf16_t*)k_bf16, k_stride_t, k_stride_h, (const bf16_t*)v_bf16, v_stride_t, v_stride_h,  \
                       cu_seqlens, softmax_scale, (bf16_t*)out_bf16, (int)q_heads)
    switch (G) {
        case 1: LAUNCH_G(1); break;
        case 2: LAUNCH_G(2); break;
        case 4: LAUNCH_G(4); break;
        case 8: LAUNCH_G(8); break;
        case 16: LAUNCH_G(16); break;
        default: LAUNCH_G(32); break;
    }
#undef LAUNCH_G
    CHITU_RETURN_LAUNCH_STATUS();
}
