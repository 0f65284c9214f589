// Weight-streaming FP8 GEMMs for decode (M = batch <= 64 per pass), gfx950.
//
// Replaces (reference, read-only):
//   chitu/triton_kernels.py:302-365  fp8_gemm_deepseek_v3_kernel       (W8A8, fp32 block scales)
//   chitu/triton_kernels.py:388-508  soft_fp8_gemm_deepseek_v3_kernel  (fp8 weight -> bf16, bf16 dot)
//   chitu/ops.py:453-511             their launchers
//
// Design (not a translation of the Triton tiling): at M <= 64 the op is a stream of the
// [N,K] e4m3 weight matrix, read exactly once, so the kernel is built around the weight
// load.  One wave owns 16 weight rows x a contiguous range of 128-wide K blocks.  Each
// wave-load takes a whole 128-B line from each of 8 rows (gemm_common.h "full-line" layout:
// 6.5+ TB/s vs 5.1 for 64 B from each of 16 rows); the lane's 16 B are two MFMA A-fragments of
// v_mfma_f32_16x16x32_fp8_fp8 (weights = A), the matching activation fragments come from the
// L2-resident [M,K] fp8 matrix with the same k order.  Per K block the MFMAs accumulate into
// fresh registers and the result is folded in as (dot * a_s) * b_s in fp32 -- the reference's order
// (triton_kernels.py:357).  K is split over the waves of a workgroup (LDS reduce) and, when
// N is too small to fill 256 CUs, over S workgroups (fp32 partials + a tiny ordered reduce
// kernel): deterministic, no atomics.
#include "common.h"
#include "gemm_common.h"
// Profiling aid, compile-time only: a probe build (hipcc -DCHITU_GEMM_PHASE_MASK=<bits>) drops parts of fp8_gemm_kernel
// (1: activation loads, 2: MFMAs, 4: activation-scale loads, 8: weight loads) to price them; such a build computes
// garbage and is never shipped.  16: address the activations as if they were laid out tile-major ([K/16][16 tokens][16 B],
// scales [K/128][16 tokens]) -- timing only.
#ifndef CHITU_GEMM_PHASE_MASK
#define CHITU_GEMM_PHASE_MASK 0
#endif

namespace chitu {

typedef long mfma_ab_t;  // 8 fp8

__device__ __forceinline__ mfma_ab_t pack_lo(const i32x4& v) {
    return (long)(((unsigned long long)(uint32_t)v[1] << 32) | (uint32_t)v[0]);
}
__device__ __forceinline__ mfma_ab_t pack_hi(const i32x4& v) {
    return (long)(((unsigned long long)(uint32_t)v[3] << 32) | (uint32_t)v[2]);
}

template <int MT>
struct GemmStage {
    W8Frag w;
    i32x4 x[MT][2];
    float xs[MT];
    float ws;
};

// ---------------------------------------------------------------- W8A8 block-scaled
// TM: the activations are TILE-MAJOR -- X[tile = m / 16][K / 16][m % 16][16 B], XS[tile][K / 128][m % 16] -- the layout the
// fused step's quantising kernels write for batches up to 64 (chitu_hip_rmsnorm quant_mode + 4, ..._uv_quant_fp8_tm).  A
// 16-lane group (one k chunk, 16 tokens) then reads 256 CONTIGUOUS bytes instead of 16 B from each of 16 rows, and a
// block's 16 scales one 64-B segment instead of 16 scattered dwords: the activation loads of this kernel were 2.9 of its
// 9.9 us at bs 16 (profiles/r03_phase_masks.txt); same arithmetic, bit-identical results.
template <int MT, int WK, bool DEEP = false, bool TM = false>
__global__ __launch_bounds__(64 * WK) void fp8_gemm_kernel(
    const fp8_t* __restrict__ X, const float* __restrict__ XS, const fp8_t* __restrict__ W,
    const float* __restrict__ WS, void* __restrict__ out, int out_dt, float* __restrict__ partial,
    int M, int N, int K, int S, int m_base) {
    __shared__ float red[WK > 1 ? WK * MT * 256 : 1];
    if (DEEP) CHITU_PROBE_MARK(0);
    const int lane = threadIdx.x & 63;
    // wave-uniform by construction; saying so (readfirstlane) lets the compiler keep the K-range arithmetic in SGPRs
    // and fetch the weight block scales with scalar loads instead of one vector load per K block
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int KB = K >> 7;
    const int T = S * WK;
    const int t = blockIdx.y * WK + wave;
    const int kb0 = (int)((long)KB * t / T), kb1 = (int)((long)KB * (t + 1) / T);

    const fp8_t *wp0, *wp1;
    w8_lane_ptrs(W, n0, N, K, j, g, wp0, wp1);
    const float* wsp = WS + (size_t)(n0 >> 7) * KB;
    const fp8_t* xp[MT];
    const float* xsp[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = min(m_base + mt * 16 + j, M - 1);
        xp[mt] = X + (size_t)m * K + g * 16;
        xsp[mt] = XS + (size_t)m * KB;
        if (TM) {
            xp[mt] = X + (size_t)(m >> 4) * 16 * K + (size_t)(g * 16 + (m & 15)) * 16;
            xsp[mt] = XS + (size_t)(m >> 4) * KB * 16 + (m & 15);
        } else if (CHITU_GEMM_PHASE_MASK & 16) {
            xp[mt] = X + (size_t)(g * 16 + (m & 15)) * 16;
            xsp[mt] = XS + (m & 15);
        }
    }

    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int dbg = CHITU_GEMM_PHASE_MASK;  // 0 in every shipped build (see the macro)
    auto load = [&](GemmStage<MT>& st, int kb) {
        const int off = kb << 7;
        if (!(dbg & 8)) {
            st.w.w[0] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(wp0 + off));
            st.w.w[1] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(wp1 + off));
        } else {
            st.w.w[0] = st.w.w[1] = i32x4{kb, lane, 3, 4};
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (TM || (dbg & 16)) {  // chunk c = kb*8 + g (+4): 16 tokens x 16 B = 256 B per chunk
                st.x[mt][0] = *reinterpret_cast<const i32x4*>(xp[mt] + (size_t)kb * 2048);
                st.x[mt][1] = *reinterpret_cast<const i32x4*>(xp[mt] + (size_t)kb * 2048 + 1024);
            } else if (!(dbg & 1)) {
                st.x[mt][0] = *reinterpret_cast<const i32x4*>(xp[mt] + off);
                st.x[mt][1] = *reinterpret_cast<const i32x4*>(xp[mt] + off + 64);
            } else {
                st.x[mt][0] = st.x[mt][1] = i32x4{kb, lane, 1, 2};
            }
            st.xs[mt] = (dbg & 4) ? 1.0f : (TM || (dbg & 16)) ? xsp[mt][kb * 16] : xsp[mt][kb];
        }
        st.ws = wsp[kb];
    };
    auto compute = [&](const GemmStage<MT>& st) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x4 blk;
            if (!(dbg & 2)) blk = w8a8_block_dot(st.w, st.x[mt][0], st.x[mt][1]);
            else
                blk = f32x4{__int_as_float(st.w.w[0][0] ^ st.x[mt][0][0]), __int_as_float(st.w.w[0][1] ^ st.x[mt][1][1]),
                            __int_as_float(st.w.w[1][2] ^ st.x[mt][0][2]), __int_as_float(st.w.w[1][3] ^ st.x[mt][1][3])};
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mt][r] += (blk[r] * st.xs[mt]) * st.ws;
        }
    };

    // D-deep register ring: D K-blocks (2 KB of weights each) in flight per wave while one is consumed.
    // DEEP (single token tile, 5-8 K blocks per wave): the wave's whole K range is requested at once,
    // one HBM round trip instead of two -- these launches are latency-, not bandwidth-bound.
    constexpr int D = DEEP ? 8 : MT >= 4 ? 2 : 4;
    GemmStage<MT> ring[D];
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (kb0 + d < kb1) load(ring[d], kb0 + d);
    if (DEEP) CHITU_PROBE_MARK(1);  // all loads issued
    for (int kb = kb0; kb < kb1; kb += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (kb + d < kb1) {
                compute(ring[d]);
                if (DEEP && kb == kb0 && d == 0) {
                    if (acc[0][0] == 1.2345e30f) CHITU_PROBE_MARK(9);
                    CHITU_PROBE_MARK(2);  // first K block multiplied (its loads have arrived)
                }
                if (kb + d + D < kb1) load(ring[d], kb + d + D);
            }
        }
    }
    if (DEEP) {
        if (acc[0][0] == 1.2345e30f) CHITU_PROBE_MARK(9);
        CHITU_PROBE_MARK(3);  // K loop done
    }

    gemm_epilogue_v2<MT, WK>(acc, red, out, out_dt, partial, M, N, S, m_base, n0);
    if (DEEP) CHITU_PROBE_MARK(4);
}

// out[m][n] = sum_s partial[s][m][n] in s order, then cast.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial,
                                                            void* __restrict__ out, int out_dt,
                                                            int S, int64_t MN) {
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < MN;
         i += (int64_t)gridDim.x * blockDim.x * 4) {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const int cnt = (int)min((int64_t)4, MN - i);
        for (int s = 0; s < S; ++s)
            for (int r = 0; r < cnt; ++r) v[r] += partial[(size_t)s * MN + i + r];
        for (int r = 0; r < cnt; ++r) {
            if (out_dt == 2) ((float*)out)[i + r] = v[r];
            else ((uint16_t*)out)[i + r] = out_dt == 0 ? f32_to_bf16(v[r]) : f32_to_f16(v[r]);
        }
    }
}

// ---------------------------------------------------------------- soft-fp8 (w8 a16)
// b = bf16( bits(((w&0x80)<<24)|((w&0x7f)<<20)) * (b_s * 2^120) ), acc += dot_bf16(a, b)
// (triton_kernels.py:453-488).  Weights = A operand of v_mfma_f32_16x16x32_bf16; a lane's
// 16-B weight load feeds two MFMAs, the bf16 activation fragment uses the same k order.
template <int MT, int WK>
__global__ __launch_bounds__(64 * WK) void soft_fp8_gemm_kernel(
    const bf16_t* __restrict__ X, const fp8_t* __restrict__ W, const float* __restrict__ WS,
    void* __restrict__ out, int out_dt, float* __restrict__ partial, int M, int N, int K, int S,
    int m_base) {
    __shared__ float red[WK > 1 ? WK * MT * 256 : 1];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: SGPR index math
    const int j = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int KB = (K + 127) >> 7;
    const int T = S * WK;
    const int t = blockIdx.y * WK + wave;
    const int kb0 = (int)((long)KB * t / T), kb1 = (int)((long)KB * (t + 1) / T);
    const int nrow = min(n0 + j, N - 1);
    const fp8_t* wp = W + (size_t)nrow * K + g * 16;
    const float* wsp = WS + (size_t)(n0 >> 7) * KB;

    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kb = kb0; kb < kb1; ++kb) {
        const float s2 = wsp[kb] * __uint_as_float(0x7B800000u);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int off = (kb << 7) + c * 64;
            const i32x4 w = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(wp + off));
            const s16x8 wa = soft_decode8((uint32_t)w[0], (uint32_t)w[1], s2);
            const s16x8 wb = soft_decode8((uint32_t)w[2], (uint32_t)w[3], s2);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = min(m_base + mt * 16 + j, M - 1);
                const bf16_t* xq = X + (size_t)m * K + off + g * 16;
                const s16x8 xa = *reinterpret_cast<const s16x8*>(xq);
                const s16x8 xb = *reinterpret_cast<const s16x8*>(xq + 8);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xa, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, xb, acc[mt], 0, 0, 0);
            }
        }
    }

    gemm_epilogue<MT, WK>(acc, red, out, out_dt, partial, M, N, S, m_base, n0);
}

void launch_splitk_reduce(const float* partial, void* out, int out_dt, int S, int64_t MN, hipStream_t st) {
    int blocks = (int)((MN / 4 + 255) / 256);
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, partial, out, out_dt, S, MN);
}

// from this many token rows on, the W8A8 GEMM is tiled for compute (fp8_gemm_tiled.hip) instead of streamed per 64 rows
constexpr int kTiledMinRows = 128;

struct SplitPlan {
    int WK, S;
};

static SplitPlan plan_split(int N, int K) {
    const int tiles = (N + 15) / 16;
    const int KB = (K + 127) / 128;
    int T = (1536 + tiles - 1) / tiles;
    if (T > KB) T = KB;
    if (T < 1) T = 1;
    int WK = 1;
    while (WK * 2 <= T && WK < 8) WK *= 2;
    // A cross-workgroup split costs a second (reduce) launch: measured ~5 us inside the decode
    // graph, more than the few us it saves on <= 16 MB of weights.  Only use it when N alone
    // leaves most of the chip idle AND the matrix is big enough to matter.
    int S = 1;
    if (tiles * WK < 256 && (int64_t)N * K >= (int64_t)(24 << 20)) {
        S = (T + WK - 1) / WK;
        if (S > 8) S = 8;
        if (S * WK > KB) S = KB / WK > 0 ? KB / WK : 1;
    }
    return {WK, S};
}

}  // namespace chitu

#define DISPATCH_WK(KERNEL, MT, ...)                                                          \
    switch (plan.WK) {                                                                        \
        case 1: hipLaunchKernelGGL((KERNEL<MT, 1>), grid, dim3(64), 0, st, __VA_ARGS__); break;   \
        case 2: hipLaunchKernelGGL((KERNEL<MT, 2>), grid, dim3(128), 0, st, __VA_ARGS__); break;  \
        case 4: hipLaunchKernelGGL((KERNEL<MT, 4>), grid, dim3(256), 0, st, __VA_ARGS__); break;  \
        default: hipLaunchKernelGGL((KERNEL<MT, 8>), grid, dim3(512), 0, st, __VA_ARGS__); break; \
    }
// the same for tile-major activations (fp8_gemm_kernel<MT, WK, DEEP, TM = true>)
#define DISPATCH_WK_TM(MT, ...)                                                                                   \
    switch (plan.WK) {                                                                                            \
        case 1: hipLaunchKernelGGL((fp8_gemm_kernel<MT, 1, false, true>), grid, dim3(64), 0, st, __VA_ARGS__); break;   \
        case 2: hipLaunchKernelGGL((fp8_gemm_kernel<MT, 2, false, true>), grid, dim3(128), 0, st, __VA_ARGS__); break;  \
        case 4: hipLaunchKernelGGL((fp8_gemm_kernel<MT, 4, false, true>), grid, dim3(256), 0, st, __VA_ARGS__); break;  \
        default: hipLaunchKernelGGL((fp8_gemm_kernel<MT, 8, false, true>), grid, dim3(512), 0, st, __VA_ARGS__); break; \
    }

static int fp8_gemm_blockscale_launch(const void* a_fp8, const float* a_scale, const void* b_fp8, const float* b_scale,
                                      void* out, int out_dtype, int64_t M, int64_t N, int64_t K, void* workspace,
                                      int64_t workspace_bytes, bool tile_major, void* stream);

extern "C" int chitu_hip_fp8_gemm_blockscale(const void* a_fp8, const float* a_scale,
                                             const void* b_fp8, const float* b_scale, void* out,
                                             int out_dtype, int64_t M, int64_t N, int64_t K,
                                             void* workspace, int64_t workspace_bytes,
                                             void* stream) {
    return fp8_gemm_blockscale_launch(a_fp8, a_scale, b_fp8, b_scale, out, out_dtype, M, N, K, workspace, workspace_bytes,
                                      false, stream);
}

// The same GEMM on TILE-MAJOR activations (decode-sized M only): a_fp8 [ceil(M/16)][K/16][16][16 B], a_scale
// [ceil(M/16)][K/128][16] -- the layout chitu_hip_rmsnorm(quant_mode + 4) and ..._uv_quant_fp8_tm write.  Bit-identical output.
extern "C" int chitu_hip_fp8_gemm_blockscale_tm(const void* a_fp8, const float* a_scale,
                                                const void* b_fp8, const float* b_scale, void* out,
                                                int out_dtype, int64_t M, int64_t N, int64_t K,
                                                void* workspace, int64_t workspace_bytes,
                                                void* stream) {
    if (M >= chitu::kTiledMinRows) return CHITU_ERR_UNSUPPORTED;
    return fp8_gemm_blockscale_launch(a_fp8, a_scale, b_fp8, b_scale, out, out_dtype, M, N, K, workspace, workspace_bytes,
                                      true, stream);
}

static int fp8_gemm_blockscale_launch(const void* a_fp8, const float* a_scale, const void* b_fp8, const float* b_scale,
                                      void* out, int out_dtype, int64_t M, int64_t N, int64_t K, void* workspace,
                                      int64_t workspace_bytes, bool tile_major, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(a_fp8 && a_scale && b_fp8 && b_scale && out);
    CHITU_REQUIRE(M >= 0 && N >= 1 && K >= 128 && N < (1 << 30) && K < (1 << 30));
    CHITU_REQUIRE(out_dtype >= 0 && out_dtype <= 2);
    if (K % 128 != 0) return CHITU_ERR_UNSUPPORTED;  // act_quant's contract, ops.py:345-348
    if (M == 0) return CHITU_OK;
    hipStream_t st = (hipStream_t)stream;
    if (!tile_major && M >= kTiledMinRows && K < (1 << 24) && M * (K / 128) < (1ll << 29) &&
        debug_option(kOptFp8GemmTiled) != 0) {  // (32-bit byte offsets inside a 128-row tile, and of a row's scales: 4 M K/128 < 2^31)
        // prefill-sized M: a GEMM, not a weight stream (fp8_gemm_tiled.hip)
        launch_fp8_gemm_tiled((const fp8_t*)a_fp8, a_scale, (const fp8_t*)b_fp8, b_scale, out, out_dtype, M, N, K, st);
        CHITU_RETURN_LAUNCH_STATUS();
    }
    SplitPlan plan = plan_split((int)N, (int)K);
    if (plan.S > 1 && (!workspace || workspace_bytes < (int64_t)plan.S * M * N * 4)) plan.S = 1;
    debug_override(kOptFp8GemmWK, plan.WK);
    while (plan.WK > 1 && plan.WK * plan.S > (int)(K / 128)) plan.WK >>= 1;
    const dim3 grid((unsigned)((N + 15) / 16), (unsigned)plan.S);
    float* partial = (float*)workspace;
    for (int64_t mb = 0; mb < M; mb += 64) {
        const int rem = (int)(M - mb);
        const int mbase = (int)mb;
        if (tile_major) {
            if (rem <= 16) {
                const int per_wave = (int)(K / 128) / (plan.WK * plan.S);
                if (plan.WK == 8 && per_wave > 4 && per_wave <= 8 && debug_option(kOptFp8GemmDeep) != 0)
                    hipLaunchKernelGGL((fp8_gemm_kernel<1, 8, true, true>), grid, dim3(512), 0, st, (const fp8_t*)a_fp8, a_scale,
                                       (const fp8_t*)b_fp8, b_scale, out, out_dtype, partial, (int)M, (int)N, (int)K,
                                       plan.S, mbase);
                else
                    DISPATCH_WK_TM(1, (const fp8_t*)a_fp8, a_scale, (const fp8_t*)b_fp8, b_scale, out, out_dtype, partial,
                                   (int)M, (int)N, (int)K, plan.S, mbase)
            } else if (rem <= 32) {
                DISPATCH_WK_TM(2, (const fp8_t*)a_fp8, a_scale, (const fp8_t*)b_fp8, b_scale, out, out_dtype, partial,
                               (int)M, (int)N, (int)K, plan.S, mbase)
            } else {
                DISPATCH_WK_TM(4, (const fp8_t*)a_fp8, a_scale, (const fp8_t*)b_fp8, b_scale, out, out_dtype, partial,
                               (int)M, (int)N, (int)K, plan.S, mbase)
            }
            continue;
        }
        if (rem <= 16) {
            const int per_wave = (int)(K / 128) / (plan.WK * plan.S);
            if (plan.WK == 8 && per_wave > 4 && per_wave <= 8 && debug_option(kOptFp8GemmDeep) != 0)
                hipLaunchKernelGGL((fp8_gemm_kernel<1, 8, true>), grid, dim3(512), 0, st, (const fp8_t*)a_fp8, a_scale,
                                   (const fp8_t*)b_fp8, b_scale, out, out_dtype, partial, (int)M, (int)N, (int)K,
                                   plan.S, mbase);
            else
                DISPATCH_WK(fp8_gemm_kernel, 1, (const fp8_t*)a_fp8, a_scale, (const fp8_t*)b_fp8, b_scale,
                            out, out_dtype, partial, (int)M, (int)N, (int)K, plan.S, mbase)
        } else if (rem <= 32) {
            DISPATCH_WK(fp8_gemm_kernel, 2, (const fp8_t*)a_fp8, a_scale, (const fp8_t*)b_fp8, b_scale,
                        out, out_dtype, partial, (int)M, (int)N, (int)K, plan.S, mbase)
        } else {
            DISPATCH_WK(fp8_gemm_kernel, 4, (const fp8_t*)a_fp8, a_scale, (const fp8_t*)b_fp8, b_scale,
                        out, out_dtype, partial, (int)M, (int)N, (int)K, plan.S, mbase)
        }
    }
    if (plan.S > 1) {
        const int64_t MN = M * N;
        int blocks = (int)((MN / 4 + 255) / 256);
        if (blocks < 1) blocks = 1;
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, partial, out,
                           out_dtype, plan.S, MN);
    }
    CHITU_RETURN_LAUNCH_STATUS();
}

// Split-K form for the few dense GEMMs whose N alone cannot fill the chip (wqkv_a: 132 tiles on 256 CUs): the K
// range is cut over `num_splits` workgroups per tile and the fp32 partial planes [num_splits, M, N] are the
// OUTPUT -- no reduce launch; the consumer (chitu_hip_mla_qkv_post with num_partials) sums the planes in order
// as it loads them.
extern "C" int chitu_hip_fp8_gemm_blockscale_partials(const void* a_fp8, const float* a_scale, const void* b_fp8,
                                                      const float* b_scale, float* partials, int64_t M, int64_t N,
                                                      int64_t K, int32_t num_splits, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(a_fp8 && a_scale && b_fp8 && b_scale && partials);
    CHITU_REQUIRE(M >= 0 && N >= 1 && K >= 128 && N < (1 << 30) && K < (1 << 30));
    if (K % 128 != 0) return CHITU_ERR_UNSUPPORTED;
    const int KB = (int)(K / 128);
    CHITU_REQUIRE(num_splits >= 2 && num_splits <= 16 && num_splits <= KB);
    if (M == 0) return CHITU_OK;
    hipStream_t st = (hipStream_t)stream;
    SplitPlan plan{1, (int)num_splits};
    while (plan.WK < 8 && plan.WK * 2 * plan.S <= KB) plan.WK *= 2;
    const dim3 grid((unsigned)((N + 15) / 16), (unsigned)plan.S);
    for (int64_t mb = 0; mb < M; mb += 64) {
        const int rem = (int)(M - mb);
        const int mbase = (int)mb;
        if (rem <= 16) {
            DISPATCH_WK(fp8_gemm_kernel, 1, (const fp8_t*)a_fp8, a_scale, (const fp8_t*)b_fp8, b_scale, nullptr, 2,
                        partials, (int)M, (int)N, (int)K, plan.S, mbase)
        } else if (rem <= 32) {
            DISPATCH_WK(fp8_gemm_kernel, 2, (const fp8_t*)a_fp8, a_scale, (const fp8_t*)b_fp8, b_scale, nullptr, 2,
                        partials, (int)M, (int)N, (int)K, plan.S, mbase)
        } else {
            DISPATCH_WK(fp8_gemm_kernel, 4, (const fp8_t*)a_fp8, a_scale, (const fp8_t*)b_fp8, b_scale, nullptr, 2,
                        partials, (int)M, (int)N, (int)K, plan.S, mbase)
        }
    }
    CHITU_RETURN_LAUNCH_STATUS();
}

extern "C" int chitu_hip_soft_fp8_gemm(const void* a_bf16, const void* b_fp8, const float* b_scale,
                                       void* out, int out_dtype, int64_t M, int64_t N, int64_t K,
                                       void* workspace, int64_t workspace_bytes, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(a_bf16 && b_fp8 && b_scale && out);
    CHITU_REQUIRE(M >= 0 && N >= 1 && K >= 128 && N < (1 << 30) && K < (1 << 30));
    CHITU_REQUIRE(out_dtype >= 0 && out_dtype <= 2);
    if (K % 128 != 0) return CHITU_ERR_UNSUPPORTED;
    if (M == 0) return CHITU_OK;
    hipStream_t st = (hipStream_t)stream;
    SplitPlan plan = plan_split((int)N, (int)K);
    if (plan.S > 1 && (!workspace || workspace_bytes < (int64_t)plan.S * M * N * 4)) plan.S = 1;
    const dim3 grid((unsigned)((N + 15) / 16), (unsigned)plan.S);
    float* partial = (float*)workspace;
    for (int64_t mb = 0; mb < M; mb += 32) {
        const int rem = (int)(M - mb);
        const int mbase = (int)mb;
        if (rem <= 16) {
            DISPATCH_WK(soft_fp8_gemm_kernel, 1, (const bf16_t*)a_bf16, (const fp8_t*)b_fp8, b_scale, out,
                        out_dtype, partial, (int)M, (int)N, (int)K, plan.S, mbase)
        } else {
            DISPATCH_WK(soft_fp8_gemm_kernel, 2, (const bf16_t*)a_bf16, (const fp8_t*)b_fp8, b_scale, out,
                        out_dtype, partial, (int)M, (int)N, (int)K, plan.S, mbase)
        }
    }
    if (plan.S > 1) {
        const int64_t MN = M * N;
        int blocks = (int)((MN / 4 + 255) / 256);
        if (blocks < 1) blocks = 1;
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, partial, out,
                           out_dtype, plan.S, MN);
    }
    CHITU_RETURN_LAUNCH_STATUS();
}

CHITU_PROBE_READER(fp8_gemm)
