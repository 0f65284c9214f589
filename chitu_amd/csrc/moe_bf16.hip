// Fused MoE with bf16 ACTIVATIONS: the two modes of the reference's fused_moe_kernel that do not quantise the
// activations -- bf16 expert weights (use_fp8_w8a8=False) and fp8 weights decoded to bf16 on the fly (soft_fp8=True).
//
// Replaces (reference, read-only):
//   chitu/fused_moe.py:62-307     fused_moe_kernel, branches `else: accumulator += tl.dot(a, b)` (:298) and
//                                 `if soft_fp8:` (:232-276: bit-placement decode, x (b_scale * 2^120), -> bf16, tl.dot)
//   chitu/fused_moe.py:1130-1307  fused_experts_impl driven with use_fp8_w8a8=False / soft_fp8=True
//   callers: chitu/models/model_deepseek_v3.py:950-956 (scale-less experts), :968-974 (soft_fp8 on NVIDIA),
//            :975-993 (soft-fp8 dequantised to bf16 first, then the bf16 kernel -- every other vendor)
//
// Design: the same weight-streaming grouped GEMM as moe.hip (16 sorted slots = one MFMA tile, a wave owns 16 weight
// rows, the workgroup's WK waves split K, every wave-load takes whole 128-B lines of 8 weight rows and the even / odd
// half-row products are combined in-lane, gemm_common.h), on v_mfma_f32_16x16x32_bf16.  bf16 weights: a 128-B line is 64
// k values; fp8 weights: 128 k values, decoded in registers to the bf16 the reference multiplies with
// (soft_decode8: one f32 multiply by scale * 2^120, one rounding to bf16 -- bit for bit its arithmetic, NaN codes
// included).  fp32 accumulation over the whole K, x routed weight on the fp32 sum (GEMM2), one rounding to bf16 --
// the reference's rounding points (fused_moe.py:298-306).  SILU: a wave owns the gate tile and the matching up tile and
// writes h = bf16(bf16(silu(bf16(g))) * bf16(u)) (SiluAndMul on bf16 tensors, fused_moe.py:24-39) -- one launch less.
#include "common.h"
#include "gemm_common.h"

namespace chitu {

constexpr int kWBf16 = 0, kWSoftFp8 = 1;

// One K block of one 16-row weight tile, as the bf16 MFMA A fragments of the full-line layout, plus the matching
// activation fragments.  bf16 weights: K block = 64, fragments w[half][0] only.  fp8: K block = 128, w[half][0..1].
template <int WKIND>
struct MoeBfStage {
    i32x4 w[2];        // raw 16 B of rows (n0 + j/2) and (n0 + 8 + j/2)
    s16x8 x[WKIND == kWSoftFp8 ? 4 : 2];
    float ws;
};

template <int WKIND>
struct MoeBfAcc {
    f32x4 e0, o0, e1, o1;
    __device__ __forceinline__ void zero() { e0 = o0 = e1 = o1 = f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ f32x4 fold() const { return f32x4{e0[0] + o0[1], e0[2] + o0[3], e1[0] + o1[1], e1[2] + o1[3]}; }
};

template <int WKIND>
__device__ __forceinline__ void moe_bf_mma(MoeBfAcc<WKIND>& a, const i32x4 (&w)[2], const s16x8* x, float ws) {
    if (WKIND == kWBf16) {
        const s16x8 w0 = __builtin_bit_cast(s16x8, w[0]), w1 = __builtin_bit_cast(s16x8, w[1]);
        a.e0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, x[0], a.e0, 0, 0, 0);
        a.o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, x[1], a.o0, 0, 0, 0);
        a.e1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, x[0], a.e1, 0, 0, 0);
        a.o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, x[1], a.o1, 0, 0, 0);
    } else {
        const float s2 = ws * __uint_as_float(0x7B800000u);  // b_scale * 2^120 (fused_moe.py:259-268)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const s16x8 wa = soft_decode8((uint32_t)w[h][0], (uint32_t)w[h][1], s2);
            const s16x8 wb = soft_decode8((uint32_t)w[h][2], (uint32_t)w[h][3], s2);
            f32x4& e = h == 0 ? a.e0 : a.e1;
            f32x4& o = h == 0 ? a.o0 : a.o1;
            e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, x[0], e, 0, 0, 0);
            e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, x[1], e, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, x[2], o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, x[3], o, 0, 0, 0);
        }
    }
}

// grid (n_tiles, max_mblocks); block 64 * WK.  X: bf16 [rows, K], row of a slot = slot / a_div (GEMM1: topk, GEMM2: 1).
// W: [E, Nw, K] (bf16 or fp8), Ws: [E, ceil(Nw/128), K/128] f32 (fp8 only).  SILU: Nw = 2*N_out (gate rows | up rows).
// out: bf16 [numel, N_out].
template <int WKIND, int WK, bool SILU>
__global__ __launch_bounds__(64 * WK) void moe_gemm_bf16_kernel(
    const bf16_t* __restrict__ X, const void* __restrict__ Wv, const float* __restrict__ Ws,
    const int32_t* __restrict__ sorted_ids, const int32_t* __restrict__ expert_ids,
    const int32_t* __restrict__ num_post_pad, const void* __restrict__ topk_w, int w_dt, int mul_routed,
    bf16_t* __restrict__ out, int numel, int a_div, int N_out, int K) {
    constexpr int kTiles = SILU ? 2 : 1;
    __shared__ float red[WK > 1 ? WK * kTiles * 256 : 1];
    const int mb = blockIdx.y;
    if (mb * 16 >= *num_post_pad) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: SGPR index math
    const int j = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int Nw = SILU ? 2 * N_out : N_out;
    constexpr int kBlk = WKIND == kWSoftFp8 ? 128 : 64;  // k values per 128-B weight line
    constexpr int kEl = WKIND == kWSoftFp8 ? 1 : 2;      // bytes per weight element
    const int KB = K / kBlk;
    const int slot = sorted_ids[mb * 16 + j];
    const bool valid = slot < numel;
    const int e = expert_ids[mb];
    MoeBfAcc<WKIND> acc[kTiles];
#pragma unroll
    for (int t = 0; t < kTiles; ++t) acc[t].zero();
    if (e >= 0) {
        const int kb0 = KB * wave / WK, kb1 = KB * (wave + 1) / WK;
        // padded slots read the tile's first row (always valid): the wave's activation loads touch only real rows
        const int slot0 = __builtin_amdgcn_readfirstlane(slot);
        const int row = (valid ? slot : min(slot0, numel - 1)) / a_div;
        const bf16_t* xp = X + (size_t)row * K + g * (WKIND == kWSoftFp8 ? 16 : 8);
        const uint8_t* Wb = (const uint8_t*)Wv + (size_t)e * Nw * K * kEl;
        const int loff = ((j & 1) * 4 + g) * 16;  // byte offset of the lane's 16 B inside the 128-B line
        const uint8_t* wp[kTiles][2];
        const float* wsp[kTiles];
#pragma unroll
        for (int t = 0; t < kTiles; ++t) {
            const int base = t * N_out + n0, lim = t * N_out + N_out - 1;
            wp[t][0] = Wb + (size_t)min(base + (j >> 1), lim) * K * kEl + loff;
            wp[t][1] = Wb + (size_t)min(base + 8 + (j >> 1), lim) * K * kEl + loff;
            wsp[t] = WKIND == kWSoftFp8 ? Ws + ((size_t)e * ((Nw + 127) >> 7) + (base >> 7)) * KB : nullptr;
        }
        constexpr int D = SILU ? 2 : 3;
        MoeBfStage<WKIND> ring[D][kTiles];
        auto load = [&](MoeBfStage<WKIND> (&st)[kTiles], int kb) {
            const size_t off = (size_t)kb * 128;
#pragma unroll
            for (int t = 0; t < kTiles; ++t) {
                st[t].w[0] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(wp[t][0] + off));
                st[t].w[1] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(wp[t][1] + off));
                st[t].ws = WKIND == kWSoftFp8 ? wsp[t][kb] : 1.0f;
            }
            const bf16_t* xk = xp + (size_t)kb * kBlk;
            if (WKIND == kWSoftFp8) {
                // even A rows hold k = g*16 + [0,16) of the block, odd rows 64 + g*16 + [0,16): two fragments each
                st[0].x[0] = *reinterpret_cast<const s16x8*>(xk);
                st[0].x[1] = *reinterpret_cast<const s16x8*>(xk + 8);
                st[0].x[2] = *reinterpret_cast<const s16x8*>(xk + 64);
                st[0].x[3] = *reinterpret_cast<const s16x8*>(xk + 72);
            } else {
                st[0].x[0] = *reinterpret_cast<const s16x8*>(xk);
                st[0].x[1] = *reinterpret_cast<const s16x8*>(xk + 32);
            }
        };
        auto compute = [&](const MoeBfStage<WKIND> (&st)[kTiles]) {
#pragma unroll
            for (int t = 0; t < kTiles; ++t) moe_bf_mma<WKIND>(acc[t], st[t].w, st[0].x, st[t].ws);
        };
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (kb0 + d < kb1) load(ring[d], kb0 + d);
        for (int kb = kb0; kb < kb1; kb += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (kb + d < kb1) {
                    compute(ring[d]);
                    if (kb + d + D < kb1) load(ring[d], kb + d + D);
                }
            }
        }
    }
    f32x4 r[kTiles];
#pragma unroll
    for (int t = 0; t < kTiles; ++t) r[t] = acc[t].fold();
    if (WK > 1) {
#pragma unroll
        for (int t = 0; t < kTiles; ++t) *reinterpret_cast<f32x4*>(&red[((wave * kTiles + t) * 64 + lane) * 4]) = r[t];
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int t = 0; t < kTiles; ++t) {
            r[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < WK; ++w) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(&red[((w * kTiles + t) * 64 + lane) * 4]);
#pragma unroll
                for (int i = 0; i < 4; ++i) r[t][i] += v[i];
            }
        }
    }
    if (!valid) return;
    f32x4 res = r[0];
    float rw = 1.0f;
    if (SILU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float gv = round_bf16(r[0][i]), uv = round_bf16(r[1][i]);  // GEMM1's bf16 output
            const float s = round_bf16(gv / (1.0f + __expf(-gv)));           // F.silu on a bf16 tensor
            res[i] = s * uv;                                                 // bf16 * bf16, rounded by the store
        }
    } else if (mul_routed) {
        rw = w_dt == 0 ? bf16_to_f32(((const bf16_t*)topk_w)[slot]) : w_dt == 1 ? f16_to_f32(((const uint16_t*)topk_w)[slot])
                                                                                : ((const float*)topk_w)[slot];
    }
    bf16_t* orow = out + (size_t)slot * N_out;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int n = n0 + h * 8 + 2 * g;
        const uint16_t a = f32_to_bf16(res[2 * h] * rw), b = f32_to_bf16(res[2 * h + 1] * rw);
        if (n + 1 < N_out && (N_out & 1) == 0) *reinterpret_cast<uint32_t*>(orow + n) = (uint32_t)a | ((uint32_t)b << 16);
        else {
            if (n < N_out) orow[n] = a;
            if (n + 1 < N_out) orow[n + 1] = b;
        }
    }
}

}  // namespace chitu

using namespace chitu;

// weight_kind 0 = bf16 [E, Nw, K]; 1 = fp8 e4m3fn [E, Nw, K] + scales [E, ceil(Nw/128), K/128] (soft-fp8 decode).
// silu 1: Nw = 2 * n_out, out = SiluAndMul(x W^T) [numel, n_out]; silu 0: Nw = n_out.
extern "C" int chitu_hip_moe_gemm_bf16(const void* a_bf16, int32_t a_div, const void* w, const float* w_scale,
                                       int32_t weight_kind, const int32_t* sorted_token_ids, const int32_t* expert_ids,
                                       const int32_t* num_tokens_post_pad, const void* topk_weights,
                                       int32_t weights_dtype, int32_t mul_routed_weight, int32_t silu, void* out_bf16,
                                       int64_t numel, int64_t n_out, int64_t K, int64_t max_mblocks, void* stream) {
    CHITU_REQUIRE(a_bf16 && w && sorted_token_ids && expert_ids && num_tokens_post_pad && out_bf16);
    CHITU_REQUIRE(a_div >= 1 && numel >= 0 && n_out >= 1 && K >= 1 && max_mblocks >= 0 && max_mblocks <= 65535);
    CHITU_REQUIRE(weight_kind == 0 || weight_kind == 1);
    CHITU_REQUIRE(weight_kind == 0 || w_scale);
    CHITU_REQUIRE(!mul_routed_weight || (topk_weights && weights_dtype >= 0 && weights_dtype <= 2));
    CHITU_REQUIRE(!(silu && mul_routed_weight));
    if (K % 128 != 0 || K >= (1ll << 30) || n_out >= (1ll << 30) || numel >= (1ll << 30)) return CHITU_ERR_UNSUPPORTED;
    if (weight_kind == 1 && silu && n_out % 128 != 0) return CHITU_ERR_UNSUPPORTED;  // gate / up scale rows must not straddle
    if (numel == 0 || max_mblocks == 0) return CHITU_OK;
    hipStream_t st = (hipStream_t)stream;
    const int tiles = (int)((n_out + 15) / 16);
    const int KB = (int)(K / (weight_kind == 1 ? 128 : 64));
    // K split over the waves of a workgroup while the grid alone leaves the chip short of waves
    int WK = 1;
    while (WK < 8 && (int64_t)tiles * max_mblocks * WK < 2048 && KB / (WK * 2) >= 2) WK *= 2;
    const dim3 grid((unsigned)tiles, (unsigned)max_mblocks);
#define LAUNCH(KIND, WKV, S)                                                                                            \
    hipLaunchKernelGGL((moe_gemm_bf16_kernel<KIND, WKV, S>), grid, dim3(64 * WKV), 0, st, (const bf16_t*)a_bf16, w, w_scale, \
                       sorted_token_ids, expert_ids, num_tokens_post_pad, topk_weights, (int)weights_dtype,              \
                       (int)mul_routed_weight, (bf16_t*)out_bf16, (int)numel, (int)a_div, (int)n_out, (int)K)
#define LAUNCH_WK(KIND, S)               \
    switch (WK) {                        \
        case 8: LAUNCH(KIND, 8, S); break; \
        case 4: LAUNCH(KIND, 4, S); break; \
        case 2: LAUNCH(KIND, 2, S); break; \
        default: LAUNCH(KIND, 1, S); break; \
    }
    if (weight_kind == 0) {
        if (silu) LAUNCH_WK(kWBf16, true)
        else LAUNCH_WK(kWBf16, false)
    } else {
        if (silu) LAUNCH_WK(kWSoftFp8, true)
        else LAUNCH_WK(kWSoftFp8, false)
    }
#undef LAUNCH_WK
#undef LAUNCH
    CHITU_RETURN_LAUNCH_STATUS();
}
