// Batch-1 decode of a block-FP8 model: [top-k sum +] residual add + RMSNorm + act_quant as the PROLOGUE of the W8A8 GEMM
// that consumes the quantised row -- the wqkv_a projection behind attn_norm -- so a decoder layer loses the stand-alone
// norm launch in front of its attention.
//
// Replaces (reference, read-only):
//   chitu/models/model_deepseek_v3.py:1107-1113  TransformerBlockDeepSeekV3.forward: x = x + ffn(...); attn(attn_norm(x))
//   chitu/models/model.py:29-78                  RMSNorm.forward
//   chitu/models/model_deepseek_v3.py:98-100     linear_deepseek_v3: act_quant_deepseek_v3 + fp8_gemm_deepseek_v3
//   chitu/fused_moe.py:1299-1305                 the experts' top-k sum (folded in like chitu_hip_rmsnorm(add_terms > 1))
// and, in this library, chitu_hip_rmsnorm(add = ..., quant_mode = 1) followed by chitu_hip_fp8_gemm_blockscale(_tm).
//
// Why: at batch 1 the norm launch is ~6 us of a ~81 us layer for 0.14 MB of data (profiles/r03_step_breakdown_bs1_final.txt:
// rmsnorm_add_kernel<1, 16> 5.9 us x 57).  Here every workgroup of the GEMM (132 for wqkv_a) redoes the sum + add + norm +
// quantisation of the row while its weight tiles are on their way: 1024 "virtual threads" (64 * WK real ones, each taking
// 1024 / (64 * WK) 8-element chunks) run exactly rmsnorm_add_kernel's arithmetic (norm_common.h: fp32 sum of the terms in
// term order with one rounding, bf16 residual add, per-chunk sequential square sum, wave butterfly, the 16 wave sums in
// order, act_quant per 16 lanes), the fp8 codes and scales go to LDS, and the K loop takes its activation fragments from
// there.  Workgroup 0 also writes the new residual stream.  Same K split, same per-block fold (dot * a_s) * b_s and the same
// wave-order reduce as fp8_gemm_kernel (fp8_gemm.hip): the output is BIT-IDENTICAL to the two launches.
#include "common.h"
#include "gemm_common.h"
#include "norm_common.h"

namespace chitu {

constexpr int kFp8NormMaxRows = 2;

// WK: waves per workgroup = K split (fp8_gemm.hip's plan for the shape); MAXT: term registers (>= terms); MR: rows held.
template <int WK, int MAXT, int MR>
__global__ __launch_bounds__(64 * WK) void fp8_gemm_add_norm_kernel(
    const bf16_t* x, int64_t x_stride, const bf16_t* add, int64_t add_stride, int terms, int64_t term_stride, bf16_t* sum_out,
    int64_t sum_stride, const bf16_t* __restrict__ nw, float eps, const fp8_t* __restrict__ W, const float* __restrict__ WS,
    void* __restrict__ out, int out_dt, int M, int N, int K) {
    constexpr int T = 64 * WK, NCH = kNormWideThreads / T;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    __shared__ float red[WK * 256];
    __shared__ float nred[kFp8NormMaxRows * 16];
    const int KB = K >> 7;
    fp8_t* qbuf = dyn_lds;                                              // [M][K] e4m3 codes
    float* sbuf = reinterpret_cast<float*>(dyn_lds + (size_t)M * K);     // [M][K / 128]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int kb0 = (int)((long)KB * wave / WK), kb1 = (int)((long)KB * (wave + 1) / WK);
    const int n_chunks = K >> 3;
    // ---- every load of the prologue first (loads return in order: the norm then never waits for a weight tile)
    i32x4 xr[MR][NCH], wr[NCH], tr[MR][NCH][MAXT];
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        const int row = min(m, M - 1);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = min(tid + i * T, n_chunks - 1);
            xr[m][i] = *reinterpret_cast<const i32x4*>(x + (int64_t)row * x_stride + c * 8);
#pragma unroll
            for (int k = 0; k < MAXT; ++k)
                tr[m][i][k] = *reinterpret_cast<const i32x4*>(add + (int64_t)row * add_stride + (int64_t)min(k, terms - 1) * term_stride + c * 8);
        }
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) wr[i] = *reinterpret_cast<const i32x4*>(nw + min(tid + i * T, n_chunks - 1) * 8);
    // ---- the wave's weight blocks (WK = 8: its whole K range when that is <= 8 blocks, one round trip)
    const fp8_t *wp0, *wp1;
    w8_lane_ptrs(W, n0, N, K, j, g, wp0, wp1);
    const float* wsp = WS + (size_t)(n0 >> 7) * KB;
    constexpr int D = WK == 8 ? 8 : 4;
    W8Frag ring[D];
    float wsr[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        if (kb0 + d < kb1) {
            ring[d].w[0] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(wp0 + ((kb0 + d) << 7)));
            ring[d].w[1] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(wp1 + ((kb0 + d) << 7)));
            wsr[d] = wsp[kb0 + d];
        }
    }
    // ---- [terms sum ->] residual add -> mean square: rmsnorm_add_kernel + rmsnorm_wide_finish, virtual thread vt
    const bool owner = blockIdx.x == 0;
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        if (m < M) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int vt = tid + i * T;
                const bool act = vt < n_chunks;
                const i32x4 a = MAXT == 1 ? tr[m][i][0] : sum_terms_bf16x8<MAXT>(tr[m][i], terms);
                float v[8];
                i32x4 sraw;
                add_bf16x8(xr[m][i], a, v, sraw);
                xr[m][i] = sraw;
                if (owner && act) *reinterpret_cast<i32x4*>(sum_out + (int64_t)m * sum_stride + vt * 8) = sraw;
                float ss = 0.f;
                if (act) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) ss += v[k] * v[k];
                }
                ss = wave_reduce_sum(ss);
                if (lane == 0) nred[m * 16 + wave + i * WK] = ss;  // virtual wave vt / 64
            }
        }
    }
    __syncthreads();
    // ---- y = bf16((x_new * rr) * w), act_quant of the rounded y (16 lanes = one 128-wide group) -> LDS
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        if (m < M) {
            float ss = 0.f;
#pragma unroll
            for (int k = 0; k < kNormWideThreads / 64; ++k) ss += nred[m * 16 + k];
            const float rr = rsqrtf(ss / (float)K + eps);
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int vt = tid + i * T;
                const bool act = vt < n_chunks;
                float v[8], o[8];
                unpack_bf16x8(xr[m][i], v);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t u = (uint32_t)wr[i][k];
                    const uint32_t h2 = f32x2_to_bf16x2((v[2 * k] * rr) * __uint_as_float(u << 16),
                                                        (v[2 * k + 1] * rr) * __uint_as_float(u & 0xffff0000u));
                    o[2 * k] = act ? __uint_as_float(h2 << 16) : 0.f;
                    o[2 * k + 1] = act ? __uint_as_float(h2 & 0xffff0000u) : 0.f;
                }
                float amax = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) amax = __builtin_fmaxf(amax, __builtin_fabsf(o[k]));
                amax = row16_reduce_max(amax);  // K % 128 == 0: a 16-lane group is all inside the row or all outside
                const float sc = amax / 448.0f;
                const i32x2 packed = quant8_fp8<false>(o, act ? sc : 1.0f);
                if (act) {
                    *reinterpret_cast<i32x2*>(qbuf + (size_t)m * K + vt * 8) = packed;
                    if ((tid & 15) == 0) sbuf[m * KB + (vt >> 4)] = sc;
                }
            }
        }
    }
    __syncthreads();
    // ---- K loop of fp8_gemm_kernel<1, WK>: activation fragments and scales from LDS (all 16 MFMA columns of a token read
    // one address: a broadcast)
    const int row = min(j, M - 1);
    const fp8_t* xq = qbuf + (size_t)row * K + g * 16;
    const float* xs = sbuf + row * KB;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kb = kb0; kb < kb1; kb += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (kb + d < kb1) {
                const i32x4 x0 = *reinterpret_cast<const i32x4*>(xq + ((kb + d) << 7));
                const i32x4 x1 = *reinterpret_cast<const i32x4*>(xq + ((kb + d) << 7) + 64);
                const float a_s = xs[kb + d];
                const f32x4 blk = w8a8_block_dot(ring[d], x0, x1);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] += (blk[r] * a_s) * wsr[d];
                if (kb + d + D < kb1) {
                    const int off = (kb + d + D) << 7;
                    ring[d].w[0] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(wp0 + off));
                    ring[d].w[1] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(wp1 + off));
                    wsr[d] = wsp[kb + d + D];
                }
            }
        }
    }
    f32x4 accv[1] = {acc};
    gemm_epilogue_v2<1, WK>(accv, red, out, out_dt, nullptr, M, N, 1, 0, n0);
}

// the K split chitu_hip_fp8_gemm_blockscale picks for this shape (fp8_gemm.hip::plan_split; kept in step so that the fused
// and the unfused launches accumulate in the same order); 0 = a cross-workgroup split or < 4 waves: not served here
static int fp8_norm_gemm_wk(int64_t N, int64_t K) {
    const int tiles = (int)((N + 15) / 16), KB = (int)(K / 128);
    int T = (1536 + tiles - 1) / tiles;
    if (T > KB) T = KB;
    if (T < 1) T = 1;
    int WK = 1;
    while (WK * 2 <= T && WK < 8) WK *= 2;
    if (tiles * WK < 256 && N * K >= (int64_t)(24 << 20)) return 0;  // plan_split would cut K across workgroups
    while (WK > 1 && WK > KB) WK >>= 1;
    return WK >= 4 ? WK : 0;
}

}  // namespace chitu

extern "C" int chitu_hip_fp8_gemm_add_norm(const void* x_bf16, int64_t x_row_stride, const void* add_bf16,
                                           int64_t add_row_stride, int32_t add_terms, int64_t add_term_stride,
                                           void* sum_out_bf16, int64_t sum_row_stride, const void* norm_weight_bf16, float eps,
                                           const void* w_fp8, const float* w_scale, void* out, int32_t out_dtype, int64_t M,
                                           int64_t N, int64_t K, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(x_bf16 && add_bf16 && sum_out_bf16 && norm_weight_bf16 && w_fp8 && w_scale && out);
    CHITU_REQUIRE(M >= 0 && N >= 1 && N < (1 << 30) && K >= 128 && K < (1 << 30) && out_dtype >= 0 && out_dtype <= 2);
    CHITU_REQUIRE(x_row_stride % 8 == 0 && add_row_stride % 8 == 0 && sum_row_stride % 8 == 0);
    CHITU_REQUIRE(add_terms >= 1 && (add_terms == 1 || add_term_stride % 8 == 0));
    if (M == 0) return CHITU_OK;
    if (K % 128 != 0 || K > kNormWideThreads * 8 || add_terms > kNormMaxTerms) return CHITU_ERR_UNSUPPORTED;
    if (M > kFp8NormMaxRows || (M == 2 && add_terms != 1)) return CHITU_ERR_UNSUPPORTED;
    const int WK = fp8_norm_gemm_wk(N, K);
    if (WK == 0) return CHITU_ERR_UNSUPPORTED;
    const int tiles = (int)((N + 15) / 16);
    const size_t lds = (size_t)M * K + (size_t)M * (K / 128) * 4;
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(WKV, MT, MRV)                                                                                                  \
    hipLaunchKernelGGL((fp8_gemm_add_norm_kernel<WKV, MT, MRV>), dim3((unsigned)tiles), dim3(64 * WKV), lds, st,               \
                       (const bf16_t*)x_bf16, x_row_stride, (const bf16_t*)add_bf16, add_row_stride, (int)add_terms,           \
                       add_term_stride, (bf16_t*)sum_out_bf16, sum_row_stride, (const bf16_t*)norm_weight_bf16, eps,           \
                       (const fp8_t*)w_fp8, w_scale, out, (int)out_dtype, (int)M, (int)N, (int)K)
#define LAUNCH_T(WKV)                                   \
    do {                                                \
        if (M == 2) LAUNCH(WKV, 1, 2);                  \
        else if (add_terms == 1) LAUNCH(WKV, 1, 1);     \
        else if (add_terms <= 10) LAUNCH(WKV, 10, 1);   \
        else LAUNCH(WKV, 16, 1);                        \
    } while (0)
    if (WK == 8) LAUNCH_T(8);
    else LAUNCH_T(4);
#undef LAUNCH_T
#undef LAUNCH
    CHITU_RETURN_LAUNCH_STATUS();
}
