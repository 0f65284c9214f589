// MLA (absorb mode) causal prefill attention for gfx950, flash form: the same contract as mla_prefill.hip
// (attn_varlen_func of AttentionDeepSeekV3.prefill_forward, chitu/models/model_deepseek_v3.py:589-599; interface
// chitu/attn_backend.py:39-90, torch path :394-455), built for the matrix pipe instead of for bit-equality with decode.
//
//   out[t,h,:] = softmax_{s <= t, same sequence}( scale * q[t,h,:] . kv[s,:] ) . kv[s,:512]
//
// MLA in absorb mode is MQA: the 16 local heads of a token share one 576-wide key row, so (token, head) pairs are just
// rows of one Q matrix.  A workgroup owns 8 query tokens x 16 heads = 128 Q rows; each of its 4 waves (one per SIMD, the
// whole 512-register file) owns 32 of them:
//   * Q fragments stay in registers for the whole kernel (36 x 4 VGPRs: the B operand of S^T = K Q^T), so a
//     v_mfma_f32_32x32x16_bf16 takes ONE 1 KB operand from LDS, not two (mla_prefill.hip fed two per 16x16x32);
//   * S^T = K Q^T puts a Q row's 32 scores of a key block in ONE lane pair (l, l^32): the softmax is in-lane, its running
//     maximum is exchanged across the pair only when it has to move (deferred rescale, threshold 8 in the exp2 domain);
//   * O^T = V^T P^T: rounded to bf16, a lane's P values are already the B fragment (k-slot 8*hi+e <-> key 16t + 8*(e>>2) +
//     4*hi + (e&3)); V^T fragments come from the staged tile by ds_read_b64_tr_b16 in that key order; the 32 x 512 fp32
//     accumulator (256 registers) is per Q row in-lane, so the rescale and the final 1/l are lane-local too;
//   * 64-key tiles arrive by LDS-DMA (global_load_lds_dwordx4, no staging VGPRs) into a 2-deep ring, one tile ahead of the
//     MFMAs, one workgroup barrier per tile.
// LDS image of a tile (conflict-free for both readers): latent part [64][1088 B] (1024 + 64 pad: four consecutive keys
// sit in four different 64-B bank quarters for the transposed reads), 16-B chunk c of key r stored at c ^ ((r >> 2) & 3)
// (sixteen keys x one chunk column = sixteen different bank groups for the ds_read_b128 K fragments); rope part [64][128 B],
// chunk c of key r at c ^ ((r >> 1) & 7).  The DMA writes lane-linear, so the XOR is applied to the per-lane SOURCE address.
// Causal work per workgroup grows with the block index: blocks are issued heaviest first.
// Not bit-equal to the decode kernel (other summation order, deferred max): the attention bar, 1e-2 of the peak.
#include <type_traits>

#include "common.h"
#include "lds_dma.h"

namespace chitu {

namespace pff {
constexpr int kC = 512, kR = 64;
constexpr int kTile = 64;                // keys per staged tile
constexpr int kBQ = 8;                   // query tokens per workgroup (x 16 heads = 128 Q rows, 32 per wave)
constexpr int kRowA = 1088;              // latent part: LDS row stride in bytes
constexpr int kRowB = 128;               // rope part
constexpr int kBufA = kTile * kRowA;     // 69632
constexpr int kBufB = kTile * kRowB;     // 8192
constexpr int kBuf = kBufA + kBufB;      // 77824 per ring slot
constexpr int kORow = 1040;              // epilogue: O rows staged as bf16 [128][1040 B] over the ring
constexpr float kDefer = 8.0f;           // deferred-rescale threshold (exp2 domain): P <= 2^8 before the next rescale
}  // namespace pff

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_ff;

// ---- the 32 x 512 fp32 accumulator is pinned to the AGPR file through "+a" asm operands.  With the MFMA builtins hipcc's
// allocator, given 400 live registers in a 512-register kernel, shuttles the accumulator between the two register files every
// tile (1040 v_accvgpr moves and 184 scratch accesses per tile in the builtin form of this kernel).  Here every PV MFMA is an
// asm statement whose accumulator operand is constrained to "a": the 16 tiles stay where they are for the whole loop and the
// compiler is left with a 256-VGPR problem (Q 144, S 16, fragments, addresses) that it solves without a spill;
// tests/test_flash_asm_audit.py compiles this file and checks exactly that (no scratch, no spill, no v_accvgpr_* in the tile
// loop outside the rescale branch).  Volatile asm pins memory operations, so the V^T fragment reads are pipelined by hand.
// Hazards hipcc does not pad around asm (guide 5.7): VALU-written operand -> MFMA (s_nop 1), MFMA result -> v_accvgpr_read /
// VALU read (acc_settle, s_settle), v_accvgpr_write -> MFMA SrcC (s_nop 3 after a rescale).
__device__ __forceinline__ void acc_settle(f32x16& o) { asm volatile("s_nop 15\n\ts_nop 15" : "+a"(o)); }
// S accumulation in the VGPR form of the instruction (hipcc's builtin insists on an AGPR destination and evicts an O tile
// for it every key block); plain (non-volatile) asm: scheduled like any pure value computation.  s_settle: the 8-pass MFMA
// result -> VALU read wait states hipcc does not insert for asm.
__device__ __forceinline__ void s_mfma0(f32x16& s, const s16x8& a, const s16x8& b) {
    asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(s) : "v"(a), "v"(b));
}
__device__ __forceinline__ void s_mfma(f32x16& s, const s16x8& a, const s16x8& b) {
    asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s) : "v"(a), "v"(b));
}
__device__ __forceinline__ void s_settle(f32x16& s) { asm("s_nop 15\n\ts_nop 3" : "+v"(s)); }
// o += A(vf) x B(pb)   (o pinned to the accumulator file)
template <bool PAD>
__device__ __forceinline__ void acc_mfma(f32x16& o, const s16x8& vf, const s16x8& pb) {
    if (PAD)
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o) : "v"(vf), "v"(pb));
    else
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o) : "v"(vf), "v"(pb));
}
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

__global__ __launch_bounds__(256, 1) void mla_prefill_flash_kernel(
    const bf16_t* __restrict__ q, int64_t q_st, int64_t q_sh, const bf16_t* __restrict__ kv, int64_t kv_st,
    const int32_t* __restrict__ cu_seqlens, float scale, bf16_t* __restrict__ out, int H) {
    using namespace pff;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = lane & 31, hi = lane >> 5;
    const int seq = blockIdx.y, h0 = blockIdx.z * 16;
    const int s0 = cu_seqlens[seq], L = cu_seqlens[seq + 1] - s0;
    const int p0 = ((int)gridDim.x - 1 - (int)blockIdx.x) * kBQ;  // heaviest (latest) block first
    if (p0 >= L) return;
    const int nq = min(kBQ, L - p0);
    const int n_keys = p0 + nq;  // keys 0 .. n_keys - 1 are visible to this block's last token
    const int n_tiles = (n_keys + kTile - 1) / kTile;
    const int tq = 2 * wave + (row >> 4);   // this lane's query token within the block (a token past the end repeats the last one, stores nothing)
    const int pq = p0 + min(tq, nq - 1);    // its position: keys 0 .. pq
    const int head = min(h0 + (row & 15), H - 1);
    const bf16_t* kbase = kv + (int64_t)s0 * kv_st;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;

    // ---- tile DMA: wave w brings rows 16w .. 16w+15 (one 1 KiB piece per latent row, two pieces of 8 rope rows)
    uint32_t voffx[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) voffx[x] = (uint32_t)((lane ^ x) * 16);
    const int rope_chunk[2] = {64 + ((lane & 7) ^ (lane >> 4)), 64 + ((lane & 7) ^ (4 + (lane >> 4)))};
    auto issue = [&](int tile) {
        const int t0 = tile * kTile;
        const uint32_t dst = lds0 + (uint32_t)((tile & 1) * kBuf);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = wave * 16 + i;
            const int grow = min(t0 + r, L - 1);  // rows past the sequence repeat its last key (finite, masked by causality)
            glds16_sbase(kbase + (int64_t)grow * kv_st, voffx[(i >> 2) & 3], dst + (uint32_t)(r * kRowA));
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int r = wave * 16 + e * 8 + (lane >> 3);
            const int grow = min(t0 + r, L - 1);
            glds16_vaddr(kbase + (int64_t)grow * kv_st + rope_chunk[e] * 8, dst + (uint32_t)(kBufA + (wave * 2 + e) * 1024));
        }
    };
    issue(0);

    // ---- Q fragments (B operand of S^T = K Q^T): lane (row, hi) holds q[token][head][16 kk + 8 hi .. + 8]
    s16x8 qf[36];
    {
        const bf16_t* qp = q + (int64_t)(s0 + pq) * q_st + (int64_t)head * q_sh + hi * 8;
#pragma unroll
        for (int kk = 0; kk < 36; ++kk) qf[kk] = *reinterpret_cast<const s16x8*>(qp + kk * 16);
        // make hipcc wait for these loads HERE: left to the first use, its counted s_waitcnt vmcnt(35 .. 0) ladder sits inside
        // the tile loop and, on every later tile, drains the LDS-DMA pieces of the next tile it knows nothing about
#pragma unroll
        for (int kk = 0; kk < 36; ++kk) asm volatile("" : "+v"(qf[kk]));
    }

    // O^T[column 32 cb + crow(reg, hi)][this lane's Q row] = a[16 cb + reg]
    f32x16 o[16];
#pragma unroll
    for (int cb = 0; cb < 16; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[cb][r] = 0.f;
    float m = -INFINITY, l = 0.f;  // running maximum (exp2 domain; equal in lanes l and l^32) and this lane's share of the row sum
    const float c2 = scale * 1.4426950408889634f;

    // lane-constant LDS offsets
    const int xl = (lane >> 2) & 3;           // chunk XOR of this lane's K row (row & 15 = lane & 15)
    const int gbl = (lane >> 1) & 7;          // rope-part chunk XOR of this lane's K row
    const int k_off = row * kRowA + ((hi ^ (xl & 1)) * 16);
    const int k_sw = xl >> 1;
    const int kb_off = row * kRowB;
    // V^T fragment (transposed read): 16-lane group g16 = lane >> 4 reads [4 keys][16 columns]; this lane's 8 bytes
    const int v_row = 4 * hi + ((lane & 15) >> 2);
    const int v_cl = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
    int v_off[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) v_off[u] = (v_row + 8 * u) * kRowA + ((v_cl ^ (hi | (2 * u))) * 16) + (lane & 1) * 8;

    for (int tile = 0; tile < n_tiles; ++tile) {
        const int t0 = tile * kTile;
        glds_wait_all();                                   // this wave's pieces of the tile have landed
        __syncthreads();                                   // everyone's have; the other ring slot is no longer being read
        if (tile + 1 < n_tiles) issue(tile + 1);
        const uint8_t* bufA = smem + (tile & 1) * kBuf;
        const uint8_t* bufB = bufA + kBufA;
        const bool last = tile == n_tiles - 1;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (last && t0 + kb * 32 >= n_keys) break;  // the block's last key is before this half tile (workgroup-uniform)
            // ---- S^T[key 32 kb + crow(reg, hi)][Q row] = K Q^T
            f32x16 s;
            {
                const uint8_t* ka = bufA + kb * 32 * kRowA + k_off;
#pragma unroll
                for (int kk = 0; kk < 32; ++kk) {
                    const s16x8 kf = *reinterpret_cast<const s16x8*>(ka + (((kk & 1) ^ k_sw) * 32) + (kk >> 1) * 64);
                    if (kk == 0) s_mfma0(s, kf, qf[0]);
                    else s_mfma(s, kf, qf[kk]);
                }
                const uint8_t* kr = bufB + kb * 32 * kRowB + kb_off;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const s16x8 kf = *reinterpret_cast<const s16x8*>(kr + (((2 * j + hi) ^ gbl) * 16));
                    s_mfma(s, kf, qf[32 + j]);
                }
                s_settle(s);
            }
            // ---- scale, causal mask (diagonal tile only), the lane's maximum
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
            // ---- deferred rescale: move the maximum only when some row's scores outgrow it by more than kDefer
            if (__builtin_amdgcn_ballot_w64(pmax > m + kDefer) != 0) {  // m = -inf (first block): every lane votes
                const float mx = __builtin_fmaxf(pmax, __shfl_xor(pmax, 32, 64));
                const float m_new = __builtin_fmaxf(m, mx);  // finite: key 0 is visible to every row
                const float alpha = __builtin_amdgcn_exp2f(m - m_new);
                m = m_new;
                l *= alpha;
#pragma unroll
                for (int cb = 0; cb < 16; ++cb) {
                    acc_settle(o[cb]);
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[cb][r] *= alpha;
                    asm volatile("s_nop 3" : "+a"(o[cb]));
                }
            }
            // ---- P = exp2(S - m); bf16 P is the B fragment of O^T += V^T P^T
            s16x8 pb[2];
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float a = __builtin_amdgcn_exp2f(s[r] - m);  // masked keys: exp2(-inf) = 0
                const float b = __builtin_amdgcn_exp2f(s[r + 1] - m);
                psum += a + b;
                const uint32_t pk = f32x2_to_bf16x2(a, b);
                pb[r >> 3][r & 7] = (short)(pk & 0xffffu);
                pb[r >> 3][(r & 7) + 1] = (short)(pk >> 16);
            }
            l += psum;
            // ---- O^T += V^T P^T over the block's 32 keys: 32 MFMAs n = 16 t + cb (key step t, column block cb), V^T fragments
            // read kAhead MFMAs ahead in SOURCE order (memory operations do not move across the asm statements)
            {
                const uint8_t* vb = bufA + kb * 32 * kRowA;
                constexpr int kAhead = 4;
                s16x8 vf[kAhead];
                auto vread = [&](auto n_) {
                    constexpr int n = decltype(n_)::value;
                    const uint8_t* p = vb + (n >> 4) * 16 * kRowA + (n & 15) * 64;
                    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ff*)(p + v_off[0]));
                    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ff*)(p + v_off[1]));
                    s16x8 r;
                    r[0] = v0[0]; r[1] = v0[1]; r[2] = v0[2]; r[3] = v0[3];
                    r[4] = v1[0]; r[5] = v1[1]; r[6] = v1[2]; r[7] = v1[3];
                    vf[n % kAhead] = r;
                };
                static_for<0, kAhead>(vread);
                static_for<0, 32>([&](auto n_) {
                    constexpr int n = decltype(n_)::value;
                    acc_mfma<(n & 15) == 0>(o[n & 15], vf[n % kAhead], pb[n >> 4]);
                    if constexpr (n + kAhead < 32) vread(std::integral_constant<int, n + kAhead>{});
                });
            }
        }
    }

    // ---- epilogue: normalise, stage this wave's 32 rows through LDS (the ring is free after the barrier), store whole rows
    __syncthreads();
    const float inv = 1.0f / (l + __shfl_xor(l, 32, 64));
    uint8_t* ostage = smem + wave * 32 * kORow;
    static_for<0, 64>([&](auto g) {
        constexpr int cb = decltype(g)::value >> 2, rq = decltype(g)::value & 3, b = rq * 4;
        if constexpr (rq == 0) acc_settle(o[cb]);
        i32x2 pk;
        pk[0] = (int)f32x2_to_bf16x2(o[cb][b] * inv, o[cb][b + 1] * inv);
        pk[1] = (int)f32x2_to_bf16x2(o[cb][b + 2] * inv, o[cb][b + 3] * inv);
        *reinterpret_cast<i32x2*>(ostage + row * kORow + (32 * cb + 8 * rq + 4 * hi) * 2) = pk;
    });
    // (rows are read back by the wave that wrote them: in-wave LDS ordering suffices)
#pragma unroll 4
    for (int r = 0; r < 32; ++r) {
        const int t = 2 * wave + (r >> 4), h = h0 + (r & 15);
        if (t >= nq || h >= H) continue;  // wave-uniform
        const i32x4 v = *reinterpret_cast<const i32x4*>(ostage + r * kORow + lane * 16);
        *reinterpret_cast<i32x4*>(out + ((int64_t)(s0 + p0 + t) * H + h) * kC + lane * 8) = v;
    }
}

}  // namespace chitu

// The contract of chitu_hip_mla_prefill on mla_prefill_flash_kernel: equal to it within the attention bar, not bit for bit.
extern "C" int chitu_hip_mla_prefill_flash(const void* q_bf16, int64_t q_stride_t, int64_t q_stride_h, const void* kv_bf16,
                                           int64_t kv_stride_t, const int32_t* cu_seqlens, int32_t n_seq, int32_t max_seqlen,
                                           float softmax_scale, void* out_bf16, int32_t heads, int32_t kv_lora_rank,
                                           int32_t rope_dim, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(q_bf16 && kv_bf16 && cu_seqlens && out_bf16 && n_seq >= 0 && max_seqlen >= 0 && heads >= 1);
    if (kv_lora_rank != pff::kC || rope_dim != pff::kR) return CHITU_ERR_UNSUPPORTED;
    CHITU_REQUIRE(q_stride_t % 8 == 0 && q_stride_h % 8 == 0 && kv_stride_t % 8 == 0);
    CHITU_REQUIRE(((uintptr_t)q_bf16 | (uintptr_t)kv_bf16 | (uintptr_t)out_bf16) % 16 == 0);
    if (n_seq == 0 || max_seqlen == 0) return CHITU_OK;
    const size_t lds = 2 * (size_t)pff::kBuf;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)mla_prefill_flash_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const dim3 grid((unsigned)((max_seqlen + pff::kBQ - 1) / pff::kBQ), (unsigned)n_seq, (unsigned)((heads + 15) / 16));
    hipLaunchKernelGGL(mla_prefill_flash_kernel, grid, dim3(256), lds, (hipStream_t)stream, (const bf16_t*)q_bf16, q_stride_t,
                       q_stride_h, (const bf16_t*)kv_bf16, kv_stride_t, cu_seqlens, softmax_scale, (bf16_t*)out_bf16,
                       (int)heads);
    CHITU_RETURN_LAUNCH_STATUS();
}
