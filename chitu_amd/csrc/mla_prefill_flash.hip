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
//   * 32-key blocks arrive by LDS-DMA (global_load_lds_dwordx4, no staging VGPRs) into a 4-slot ring, two blocks ahead of the
//     MFMAs, one workgroup barrier per block; the QK^T MFMAs of block b + 1 carry the softmax of block b between them.
// LDS image of a block (conflict-free for both readers): latent part [32][1088 B] (1024 + 64 pad: four consecutive keys
// sit in four different 64-B bank quarters for the transposed reads), 16-B chunk c of key r stored at c ^ ((r >> 2) & 3)
// (sixteen keys x one chunk column = sixteen different bank groups for the ds_read_b128 K fragments); rope part [32][128 B],
// chunk c of key r at c ^ ((r >> 1) & 7).  The DMA writes lane-linear, so the XOR is applied to the per-lane SOURCE address.
// Causal work per workgroup grows with the block index: blocks are issued heaviest first.
// Not bit-equal to the decode kernel (other summation order, deferred max): the attention bar, 1e-2 of the peak.
#include <type_traits>

#include "common.h"
#include "lds_dma.h"

namespace chitu {

namespace pff {
constexpr int kC = 512, kR = 64;
constexpr int kBQ = 8;                   // query tokens per workgroup (x 16 heads = 128 Q rows, 32 per wave)
constexpr int kRowA = 1088;              // latent part: LDS row stride in bytes
constexpr int kRowB = 128;               // rope part
constexpr int kORow = 1040;              // epilogue: O rows staged as bf16 [128][1040 B] over the ring
constexpr float kDefer = 8.0f;           // deferred-rescale threshold (exp2 domain): P <= 2^8 before the next rescale
}  // namespace pff

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_ff;

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
// the same, pinned in program order (volatile): the pipelined kernel places softmax work of the previous key block between them
__device__ __forceinline__ void s_mfma0_v(f32x16& s, const s16x8& a, const s16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(s) : "v"(a), "v"(b));
}
__device__ __forceinline__ void s_mfma_v(f32x16& s, const s16x8& a, const s16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s) : "v"(a), "v"(b));
}
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// ---- the 32 x 512 fp32 accumulator lives in a[0:255], named literally in the asm text, so the compiler does not allocate (or
// move) it at all.  hipcc's own allocation of a 400-register live set does not work here: with the MFMA builtins it shuttled the
// accumulator between the two register files every tile (1040 v_accvgpr moves, 184 scratch accesses per tile); with "+a" asm
// operands it held it still inside one loop body but shuffled the 16 tiles through scratch at the seams of this kernel's four
// inlined iteration bodies (1240 spilled registers).  So: every PV MFMA, the zero fill, the rescale and the read-out are asm
// statements on fixed AGPRs; fa_reserve()'s clobber list makes the kernel descriptor allocate the AGPR file; the compiler is
// left with a 256-VGPR problem (Q 144, two S tiles 32, fragments, addresses) that it solves without a spill -- and must never
// touch the AGPR file itself: tests/test_flash_asm_audit.py compiles this file and requires that no v_accvgpr_* / a[...]
// operand appears outside ;;#ASMSTART .. ;;#ASMEND and that there is no scratch access (tools/check_flash_asm.py, literal mode).
// Volatile asm pins memory operations, so the K and V^T fragment reads are pipelined by hand in source order.
// Hazards hipcc does not pad around asm (guide 5.7): VALU-written operand -> MFMA (s_nop 1), MFMA result -> v_accvgpr_read
// (fa_settle) / VALU read (s_settle), v_accvgpr_write -> MFMA SrcC (s_nop 3 after a rescale).
#define A4(n) "a" #n "0", "a" #n "1", "a" #n "2", "a" #n "3", "a" #n "4", "a" #n "5", "a" #n "6", "a" #n "7", "a" #n "8", "a" #n "9"
__device__ __forceinline__ void fa_reserve() {
    asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", A4(1), A4(2), A4(3), A4(4), A4(5), A4(6), A4(7),
                 A4(8), A4(9), A4(10), A4(11), A4(12), A4(13), A4(14), A4(15), A4(16), A4(17), A4(18), A4(19), A4(20), A4(21),
                 A4(22), A4(23), A4(24), "a250", "a251", "a252", "a253", "a254", "a255");
}
#undef A4
template <int I>
__device__ __forceinline__ void fa_zero() {
    asm volatile("v_accvgpr_write_b32 a%c0, 0" ::"i"(I));
}
template <int I>
__device__ __forceinline__ float fa_read() {
    float t;
    asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(t) : "i"(I));
    return t;
}
template <int I>
__device__ __forceinline__ void fa_write(float t) {
    asm volatile("v_accvgpr_write_b32 a%c0, %1" ::"i"(I), "v"(t));
}
__device__ __forceinline__ void fa_settle() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }
template <int CB, bool PAD>  // a[16 CB : 16 CB + 15] += A(vf) x B(pb)
__device__ __forceinline__ void fa_mfma(const s16x8& vf, const s16x8& pb) {
    if (PAD)
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(vf), "v"(pb), "i"(CB * 16), "i"(CB * 16 + 15));
    else
        asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(vf), "v"(pb), "i"(CB * 16), "i"(CB * 16 + 15));
}

// ---------------------------------------------------------------- the kernel: software-pipelined over 32-key blocks
// A wave is alone on its SIMD, so a key block run as QK^T -> softmax -> PV in series idles the matrix pipe during the ~100 VALU
// instructions of the softmax (the first form of this kernel, round 5: PMC at 2048 tokens MFMA busy 24 %, 3 VALU per MFMA).
// Here the unit is the 32-key block in a FOUR-slot ring, and the 36 QK^T MFMAs of block b + 1 are issued with the softmax of
// block b between them in program order (two or three VALU per MFMA: inside the 32-cycle shadow of each), then the 32 PV MFMAs
// of block b with their V^T reads.  One barrier per block; block b + 3's DMA is issued two blocks before its K fragments are
// needed (a counted s_waitcnt leaves the younger block in flight).  Measured against the serial form (same box): 2048 tokens
// 125.9 -> 120.9 us, 8192 tokens 1086 -> 1148 TFLOP/s (0.46 of the MFMA peak) with five fragments in flight.
namespace pfp {
constexpr int kKeys = 32;
constexpr int kSlotA = kKeys * pff::kRowA;   // 34816
constexpr int kSlotB = kKeys * pff::kRowB;   // 4096
constexpr int kSlot = kSlotA + kSlotB;       // 38912
constexpr int kRing = 4;                     // 155648 B
constexpr int kPieces = 9;                   // LDS-DMA pieces per wave and block: 8 latent rows + 8 rope rows
}  // namespace pfp

__global__ __launch_bounds__(256, 1) void mla_prefill_flash_pipe_kernel(
    const bf16_t* __restrict__ q, int64_t q_st, int64_t q_sh, const bf16_t* __restrict__ kv, int64_t kv_st,
    const int32_t* __restrict__ cu_seqlens, float scale, bf16_t* __restrict__ out, int H) {
    using namespace pff;
    using namespace pfp;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = lane & 31, hi = lane >> 5;
    const int seq = blockIdx.y, h0 = blockIdx.z * 16;
    const int s0 = cu_seqlens[seq], L = cu_seqlens[seq + 1] - s0;
    const int p0 = ((int)gridDim.x - 1 - (int)blockIdx.x) * kBQ;  // heaviest (latest) block first
    if (p0 >= L) return;
    const int nq = min(kBQ, L - p0);
    const int n_keys = p0 + nq;
    const int nb = (n_keys + kKeys - 1) / kKeys;  // key blocks 0 .. nb - 1; only the last one reaches into the block's own tokens
    const int tq = 2 * wave + (row >> 4);
    const int pq = p0 + min(tq, nq - 1);
    const int head = min(h0 + (row & 15), H - 1);
    const bf16_t* kbase = kv + (int64_t)s0 * kv_st;
    const uint32_t lds0 = lds_offset_of(smem);

    // ---- block DMA: wave w brings keys 8 w .. 8 w + 7 of the block (a piece per latent row, one piece of 8 rope rows)
    const int rope_chunk = 64 + ((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7));
    auto issue = [&](int b) {
        const int t0 = b * kKeys;
        const uint32_t dst = lds0 + (uint32_t)((b & (kRing - 1)) * kSlot);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = wave * 8 + i;
            const int grow = min(t0 + r, L - 1);  // rows past the sequence repeat its last key (finite, masked by causality)
            const uint32_t voff = (uint32_t)((lane ^ ((r >> 2) & 3)) << 4);
            glds16_sbase(kbase + (int64_t)grow * kv_st, voff, dst + (uint32_t)(r * kRowA));
        }
        const int r = wave * 8 + (lane >> 3);
        const int grow = min(t0 + r, L - 1);
        glds16_vaddr(kbase + (int64_t)grow * kv_st + rope_chunk * 8, dst + (uint32_t)(kSlotA + wave * 1024));
    };
    issue(0);
    if (nb > 1) issue(1);
    if (nb > 2) issue(2);

    s16x8 qf[36];
    {
        const bf16_t* qp = q + (int64_t)(s0 + pq) * q_st + (int64_t)head * q_sh + hi * 8;
#pragma unroll
        for (int kk = 0; kk < 36; ++kk) qf[kk] = *reinterpret_cast<const s16x8*>(qp + kk * 16);
#pragma unroll
        for (int kk = 0; kk < 36; ++kk) asm volatile("" : "+v"(qf[kk]));  // (hipcc waits here, not in the loop: see the kernel above)
    }
    fa_reserve();  // O^T[column 32 cb + crow(reg, hi)][this lane's Q row] = a[16 cb + reg]
    static_for<0, 256>([&](auto i) { fa_zero<decltype(i)::value>(); });
    float m = -INFINITY, l = 0.f;
    const float c2 = scale * 1.4426950408889634f;

    const int xl = (lane >> 2) & 3, gbl = (lane >> 1) & 7;
    const int k_off = row * kRowA + ((hi ^ (xl & 1)) * 16);
    const int k_sw = xl >> 1;
    const int kb_off = kSlotA + row * kRowB;
    const int v_row = 4 * hi + ((lane & 15) >> 2);
    const int v_cl = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
    int v_off[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) v_off[u] = (v_row + 8 * u) * kRowA + ((v_cl ^ (hi | (2 * u))) * 16) + (lane & 1) * 8;

    // K fragment kk of a block (slot base sb): chunk 2 kk + hi of row `row`, latent part kk < 32, rope part above
    auto kread = [&](const uint8_t* sb, auto kk_) -> s16x8 {
        constexpr int kk = decltype(kk_)::value;
        if constexpr (kk < 32) return *reinterpret_cast<const s16x8*>(sb + k_off + (((kk & 1) ^ k_sw) * 32) + (kk >> 1) * 64);
        else return *reinterpret_cast<const s16x8*>(sb + kb_off + (((2 * (kk - 32) + hi) ^ gbl) * 16));
    };

    // ---- block 0's scores, not overlapped with anything
    if (nb > 2) glds_wait_leaving<2 * kPieces>();
    else if (nb > 1) glds_wait_leaving<kPieces>();
    else glds_wait_all();
    __syncthreads();
    f32x16 sA, sB;
    static_for<0, 36>([&](auto kk_) {
        constexpr int kk = decltype(kk_)::value;
        const s16x8 kf = kread(smem, kk_);
        if constexpr (kk == 0) s_mfma0(sA, kf, qf[0]);
        else s_mfma(sA, kf, qf[kk]);
    });
    s_settle(sA);

    // iteration b: [scores of block b + 1 into sN, with the softmax of block b (scores sC) between the MFMAs] then O^T += V^T P^T of b
    auto iter = [&](auto last_, int b, f32x16& sC, f32x16& sN) {
        constexpr bool LAST = decltype(last_)::value;  // b == nb - 1: no next block, causal mask on this one
        const uint8_t* sbV = smem + (b & (kRing - 1)) * kSlot;
        const uint8_t* sbK = smem + ((b + 1) & (kRing - 1)) * kSlot;
        if constexpr (!LAST) {
            // block b + 1 has landed (this wave's pieces; block b + 2's, issued one iteration ago, may still fly)
            if (b + 2 < nb) glds_wait_leaving<kPieces>();
            else glds_wait_all();
        }
        __syncthreads();  // everyone's pieces of b + 1 are there; nobody reads slot (b - 1) & 3 any more
        if (b + 3 < nb) issue(b + 3);

        float pmax = -INFINITY, psum = 0.f;  // (the scaled scores overwrite sC in place)
        s16x8 pb[2];
        constexpr int kR = 5;  // K fragments in flight
        s16x8 kf[kR];
        if constexpr (!LAST) static_for<0, kR>([&](auto i) { kf[decltype(i)::value] = kread(sbK, i); });
        // softmax of block b in slices: slots 0-7 scale (+ mask) and the lane's maximum, slot 8 the (rare) rescale decision,
        // slots 9-24 exp2 / row sum / bf16 packing, two elements each
        auto soft = [&](auto k_) {
            constexpr int k = decltype(k_)::value;
            if constexpr (k < 8) {
#pragma unroll
                for (int r = 2 * k; r < 2 * k + 2; ++r) {
                    if constexpr (LAST) {
                        const int key = b * kKeys + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        sC[r] = key <= pq ? sC[r] * c2 : -INFINITY;
                    } else {
                        sC[r] = sC[r] * c2;
                    }
                    pmax = __builtin_fmaxf(pmax, sC[r]);
                }
            } else if constexpr (k == 8) {
                if (__builtin_amdgcn_ballot_w64(pmax > m + kDefer) != 0) {  // m = -inf (first block): every lane votes
                    const float mx = __builtin_fmaxf(pmax, __shfl_xor(pmax, 32, 64));
                    const float m_new = __builtin_fmaxf(m, mx);
                    const float alpha = __builtin_amdgcn_exp2f(m - m_new);
                    m = m_new;
                    l *= alpha;
                    fa_settle();  // (the PV MFMAs of the previous block finished 36 S MFMAs ago at the earliest; belt and braces)
                    static_for<0, 32>([&](auto g) {
                        constexpr int base = decltype(g)::value * 8;
                        float t[8];
                        static_for<0, 8>([&](auto i) { t[decltype(i)::value] = fa_read<base + decltype(i)::value>(); });
                        static_for<0, 8>([&](auto i) { fa_write<base + decltype(i)::value>(t[decltype(i)::value] * alpha); });
                    });
                    asm volatile("s_nop 3" ::: "memory");
                }
            } else if constexpr (k >= 9 && k < 17) {
                constexpr int r = 2 * (k - 9);
                const float a = __builtin_amdgcn_exp2f(sC[r] - m);  // masked keys: exp2(-inf) = 0
                const float c = __builtin_amdgcn_exp2f(sC[r + 1] - m);
                psum += a + c;
                const uint32_t pk = f32x2_to_bf16x2(a, c);
                pb[r >> 3][r & 7] = (short)(pk & 0xffffu);
                pb[r >> 3][(r & 7) + 1] = (short)(pk >> 16);
            }
        };
        if constexpr (!LAST) {
            static_for<0, 36>([&](auto kk_) {
                constexpr int kk = decltype(kk_)::value;
                if constexpr (kk == 0) s_mfma0_v(sN, kf[0], qf[0]);
                else s_mfma_v(sN, kf[kk % kR], qf[kk]);
                if constexpr (kk + kR < 36) kf[kk % kR] = kread(sbK, std::integral_constant<int, kk + kR>{});
                soft(kk_);
            });
        } else {
            static_for<0, 17>(soft);
        }
        l += psum;
        {
            constexpr int kAhead = 5;
            s16x8 vf[kAhead];
            auto vread = [&](auto n_) {
                constexpr int n = decltype(n_)::value;
                const uint8_t* p = sbV + (n >> 4) * 16 * kRowA + (n & 15) * 64;
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
                fa_mfma<(n & 15), (n & 15) == 0>(vf[n % kAhead], pb[n >> 4]);
                if constexpr (n + kAhead < 32) vread(std::integral_constant<int, n + kAhead>{});
            });
        }
    };
    int b = 0;
    for (; b + 2 < nb; b += 2) {
        iter(std::false_type{}, b, sA, sB);
        iter(std::false_type{}, b + 1, sB, sA);
    }
    if (b + 1 < nb) {
        iter(std::false_type{}, b, sA, sB);
        iter(std::true_type{}, b + 1, sB, sA);
    } else {
        iter(std::true_type{}, b, sA, sB);
    }

    // ---- epilogue (as above)
    __syncthreads();
    const float inv = 1.0f / (l + __shfl_xor(l, 32, 64));
    uint8_t* ostage = smem + wave * 32 * kORow;
    fa_settle();
    static_for<0, 64>([&](auto g) {
        constexpr int cb = decltype(g)::value >> 2, rq = decltype(g)::value & 3, bb = cb * 16 + rq * 4;
        i32x2 pk;
        pk[0] = (int)f32x2_to_bf16x2(fa_read<bb>() * inv, fa_read<bb + 1>() * inv);
        pk[1] = (int)f32x2_to_bf16x2(fa_read<bb + 2>() * inv, fa_read<bb + 3>() * inv);
        *reinterpret_cast<i32x2*>(ostage + row * kORow + (32 * cb + 8 * rq + 4 * hi) * 2) = pk;
    });
#pragma unroll 4
    for (int r = 0; r < 32; ++r) {
        const int t = 2 * wave + (r >> 4), h = h0 + (r & 15);
        if (t >= nq || h >= H) continue;  // wave-uniform
        const i32x4 v = *reinterpret_cast<const i32x4*>(ostage + r * kORow + lane * 16);
        *reinterpret_cast<i32x4*>(out + ((int64_t)(s0 + p0 + t) * H + h) * kC + lane * 8) = v;
    }
}

}  // namespace chitu

// The contract of chitu_hip_mla_prefill on mla_prefill_flash_pipe_kernel: equal to it within the attention bar, not bit for bit.
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
    const size_t lds = (size_t)pfp::kRing * pfp::kSlot;
    // (set on every call: the opt-in is per device, a process-wide "done" flag would skip the second GPU of a multi-device process)
    (void)hipFuncSetAttribute((const void*)mla_prefill_flash_pipe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const dim3 grid((unsigned)((max_seqlen + pff::kBQ - 1) / pff::kBQ), (unsigned)n_seq, (unsigned)((heads + 15) / 16));
    hipLaunchKernelGGL(mla_prefill_flash_pipe_kernel, grid, dim3(256), lds, (hipStream_t)stream, (const bf16_t*)q_bf16, q_stride_t,
                       q_stride_h, (const bf16_t*)kv_bf16, kv_stride_t, cu_seqlens, softmax_scale, (bf16_t*)out_bf16,
                       (int)heads);
    CHITU_RETURN_LAUNCH_STATUS();
}
