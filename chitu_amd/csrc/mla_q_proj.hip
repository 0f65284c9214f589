// MLA decode, the q projection in ONE launch: q_norm + act_quant of wqkv_a's q_a slice as the PROLOGUE of the wq_b
// GEMM, and the token's [kv_norm(kv_c) | RoPE(k_pe)] page append on a few extra workgroups of the same grid.
//
// Replaces (reference, read-only):
//   chitu/models/model_deepseek_v3.py:488      q = wq_b(q_norm(q_a))                       (_run_linear)
//   chitu/models/model_deepseek_v3.py:493-496  RoPE(k_pe);  :684-686  kv_norm(kv), the row appended to the page
//   chitu/triton_kernels.py:194-216            act_quant_deepseek_v3_kernel  (the fp8 input of wq_b)
//   chitu/triton_kernels.py:302-365            fp8_gemm_deepseek_v3_kernel
// and, in this library, the pair chitu_hip_mla_qkv_post + chitu_hip_fp8_gemm_blockscale (kept: prefill, batches
// above 32, q_lora_rank above 2048, the split-K planes of wqkv_a; the model takes this launch for one token tile,
// batch <= 16: with two tiles a workgroup redoes the norm + quant of 32 rows and the pair was the faster form).
//
// Why: at decode batch sizes both launches are a few microseconds of dependent memory round trips behind a launch
// (4.8 us each for 0.1 MB and 4.7 MB) and the norm + quant of a [<= 32, 1536] matrix is
// cheap enough to redo in every workgroup of the GEMM: a wave already loads exactly the K blocks of the activation
// rows it multiplies, so it loads them as bf16 instead of fp8, the row's mean square is the sum of the 8 waves'
// partial sums (one LDS exchange, which the K split needs anyway), and norm, rounding, per-128 amax and the fp8
// conversion happen in the registers that become the MFMA operand.  The arithmetic per element is that of
// rmsnorm_row<1> (norm_common.h): y = bf16((x * rr) * w), scale = amax(|y|) / 448, q = fp8(y / scale); only the
// order in which the mean square is summed differs (per-lane partials, then wave-major), i.e. the last bit of rr.
#include "common.h"
#include "gemm_common.h"
#include "mla_kv_row.h"

namespace chitu {

#ifdef CHITU_PROBE  // marks of the first GEMM workgroup (blockIdx.x == kv_blocks) / the first KV workgroup
#define QPROJ_MARK(i, blk) do { if (threadIdx.x == 0 && (int)blockIdx.x == (blk)) g_probe_marks[i] = wall_clock64(); } while (0)
#else
#define QPROJ_MARK(i, blk) do {} while (0)
#endif

constexpr int kQProjWaves = 8;   // K split over the waves of a workgroup
constexpr int kQProjMaxB = 2;    // K blocks per wave => q_lora_rank <= 8 * 2 * 128 = 2048

// max over the 4 lanes {j, j+16, j+32, j+48} on the VALU (gfx950 permlane swaps; __shfl_xor would be two LDS round trips)
__device__ __forceinline__ float col4_reduce_max(float v) {
    const uint32_t u = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float m = __builtin_fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const uint32_t um = __float_as_uint(m);
    const auto b = __builtin_amdgcn_permlane32_swap(um, um, false, false);
    return __builtin_fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

template <int MT>
__global__ __launch_bounds__(64 * kQProjWaves) void mla_q_proj_kernel(
    const bf16_t* __restrict__ qkv, int64_t row_stride, int K, const bf16_t* __restrict__ q_norm_w, float q_eps,
    const fp8_t* __restrict__ W, const float* __restrict__ WS, void* __restrict__ out, int out_dt, int M, int N,
    int kv_blocks, const bf16_t* __restrict__ kv_norm_w, float kv_eps, const float* __restrict__ cos,
    const float* __restrict__ sin, bf16_t* __restrict__ cache, int64_t num_pages, int page_size,
    const int32_t* __restrict__ table, int pages_per_seq, const int32_t* __restrict__ old_lens) {
    constexpr int WK = kQProjWaves, MAXB = kQProjMaxB;
    __shared__ float red[WK * MT * 256];
    // rows padded to 36 floats: row j starts at bank (36 j) % 64 -- 16 distinct multiples of 4, so the 16-byte reads of the
    // 16 rows (and the scalar writes) spread over all 64 banks; unpadded (32 floats) they met on two bank groups (PMC:
    // 85 % of this kernel's LDS cycles were conflicts, profiles/r03_pmc_step.json)
#ifndef CHITU_QPROJ_SSQ_PAD
#define CHITU_QPROJ_SSQ_PAD 4
#endif
    __shared__ __attribute__((aligned(16))) float ssq[MT][16][WK * 4 + CHITU_QPROJ_SSQ_PAD];
    if ((int)blockIdx.x < kv_blocks) {  // the first workgroups: one token's KV row each (waves 0 and 1)
        const int b = blockIdx.x;
        QPROJ_MARK(10, 0);
        mla_kv_row(b, qkv + (int64_t)b * row_stride + K, kv_norm_w, kv_eps, cos, sin, cache, num_pages, page_size, table,
                   pages_per_seq, old_lens);
        QPROJ_MARK(11, 0);
        return;
    }
    QPROJ_MARK(0, kv_blocks);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int n0 = ((int)blockIdx.x - kv_blocks) * 16;
    const int KB = K >> 7;
    // K ranges: 0..MAXB blocks per wave.  Waves w and w + 4 share a SIMD and the prologue below is VALU-bound, so
    // the ranges are dealt out so that each SIMD gets the same number of blocks (q_lora_rank 1536: 12 blocks over 8
    // waves = sizes 1,2,1,2,...; w -> range (w % 4) * 2 + w / 4 pairs every 1 with a 2)
    const int t = (wave & 3) * 2 + (wave >> 2);
    const int kb0 = KB * t / WK, kb1 = KB * (t + 1) / WK;

    // every load of the wave issued up front, addresses clamped instead of branched on: one memory round trip
    const fp8_t *wp0, *wp1;
    w8_lane_ptrs(W, n0, N, K, j, g, wp0, wp1);
    const float* wsp = WS + (size_t)(n0 >> 7) * KB;
    W8Frag wf[MAXB];
    float ws[MAXB];
    i32x4 xr[MT][MAXB][4], wn[MAXB][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const bf16_t* xp = qkv + (int64_t)min(mt * 16 + j, M - 1) * row_stride + g * 16;
#pragma unroll
        for (int i = 0; i < MAXB; ++i) {
            const bf16_t* p = xp + (min(kb0 + i, KB - 1) << 7);
            xr[mt][i][0] = *reinterpret_cast<const i32x4*>(p);
            xr[mt][i][1] = *reinterpret_cast<const i32x4*>(p + 8);
            xr[mt][i][2] = *reinterpret_cast<const i32x4*>(p + 64);
            xr[mt][i][3] = *reinterpret_cast<const i32x4*>(p + 72);
        }
    }
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        const bf16_t* p = q_norm_w + (min(kb0 + i, KB - 1) << 7) + g * 16;
        wn[i][0] = *reinterpret_cast<const i32x4*>(p);
        wn[i][1] = *reinterpret_cast<const i32x4*>(p + 8);
        wn[i][2] = *reinterpret_cast<const i32x4*>(p + 64);
        wn[i][3] = *reinterpret_cast<const i32x4*>(p + 72);
        ws[i] = wsp[min(kb0 + i, KB - 1)];
    }
    // the weights LAST: loads return in order, so the prologue below waits for the activation rows (L2 / memory-side
    // cache) only and runs while the weight tile is still on its way from HBM
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        const int off = min(kb0 + i, KB - 1) << 7;
        wf[i].w[0] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(wp0 + off));
        wf[i].w[1] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(wp1 + off));
    }
    QPROJ_MARK(1, kv_blocks);

    // mean square of each row: per-lane partial over this wave's K blocks -> LDS -> every lane sums the 32 partials
    // of its row j (8 waves x 4 lane groups) in one fixed order
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        float s4[4] = {0.f, 0.f, 0.f, 0.f};  // one chain per 16-byte chunk: 4 short dependent chains, not one of 64
#pragma unroll
        for (int i = 0; i < MAXB; ++i) {
            if (kb0 + i < kb1) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const uint32_t u = (uint32_t)xr[mt][i][c][k];
                        const float lo = __uint_as_float(u << 16), hi = __uint_as_float(u & 0xffff0000u);
                        s4[c] += lo * lo;
                        s4[c] += hi * hi;
                    }
            }
        }
        ssq[mt][j][wave * 4 + g] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    }
    QPROJ_MARK(2, kv_blocks);
    __syncthreads();
    float rr[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        float tot = 0.f;
#pragma unroll
        for (int c = 0; c < WK; ++c) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&ssq[mt][j][c * 4]);
            tot += (v[0] + v[1]) + (v[2] + v[3]);
        }
        rr[mt] = rsqrtf(tot / (float)K + q_eps);
    }
    QPROJ_MARK(3, kv_blocks);

    // y = bf16((x * rr) * w); act_quant over each 128-block (4 lanes x 32 values); straight into the MFMA operands
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MAXB; ++i) {
            if (kb0 + i < kb1) {  // wave-uniform
                float o[4][8];
                float am[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t u = (uint32_t)xr[mt][i][c][k], w = (uint32_t)wn[i][c][k];
                        const uint32_t h2 =
                            f32x2_to_bf16x2((__uint_as_float(u << 16) * rr[mt]) * __uint_as_float(w << 16),
                                            (__uint_as_float(u & 0xffff0000u) * rr[mt]) * __uint_as_float(w & 0xffff0000u));
                        o[c][2 * k] = __uint_as_float(h2 << 16);
                        o[c][2 * k + 1] = __uint_as_float(h2 & 0xffff0000u);
                        am[c] = __builtin_fmaxf(am[c], __builtin_fmaxf(__builtin_fabsf(o[c][2 * k]), __builtin_fabsf(o[c][2 * k + 1])));
                    }
                const float amax = col4_reduce_max(__builtin_fmaxf(__builtin_fmaxf(am[0], am[1]), __builtin_fmaxf(am[2], am[3])));
                const float sc = amax / 448.0f;
                const i32x2 qa = quant8_fp8<false>(o[0], sc), qb = quant8_fp8<false>(o[1], sc);
                const i32x2 qc = quant8_fp8<false>(o[2], sc), qd = quant8_fp8<false>(o[3], sc);
                const f32x4 blk = w8a8_block_dot(wf[i], i32x4{qa[0], qa[1], qb[0], qb[1]}, i32x4{qc[0], qc[1], qd[0], qd[1]});
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mt][r] += (blk[r] * sc) * ws[i];
                if (mt == 0 && i == 0) {
#ifdef CHITU_PROBE
                    if (acc[0][0] == 1.2345e30f) QPROJ_MARK(9, kv_blocks);
#endif
                    QPROJ_MARK(4, kv_blocks);
                }
            }
        }
    }
#ifdef CHITU_PROBE
    if (acc[0][0] == 1.2345e30f) QPROJ_MARK(9, kv_blocks);
#endif
    QPROJ_MARK(5, kv_blocks);
    gemm_epilogue_v2<MT, WK>(acc, red, out, out_dt, nullptr, M, N, 1, 0, n0);
    QPROJ_MARK(6, kv_blocks);
}

}  // namespace chitu

extern "C" int chitu_hip_mla_q_proj(const void* qkv_a_bf16, int64_t row_stride, int32_t q_lora_rank,
                                    const void* q_norm_weight_bf16, float q_eps, const void* wq_b_fp8,
                                    const float* wq_b_scale, void* out, int32_t out_dtype, int64_t N,
                                    const void* kv_norm_weight_bf16, float kv_eps, const float* cos, const float* sin,
                                    void* kv_cache, int64_t num_pages, int32_t page_size, const int32_t* page_table,
                                    int32_t pages_per_seq, const int32_t* old_seq_lens, int32_t batch,
                                    int32_t kv_lora_rank, int32_t rope_dim, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(qkv_a_bf16 && q_norm_weight_bf16 && wq_b_fp8 && wq_b_scale && out && kv_norm_weight_bf16 && cos && sin);
    CHITU_REQUIRE(kv_cache && page_table && old_seq_lens);
    CHITU_REQUIRE(batch >= 0 && N >= 1 && N < (1 << 30) && num_pages >= 1 && page_size >= 1 && pages_per_seq >= 1);
    CHITU_REQUIRE(q_lora_rank >= 128 && out_dtype >= 0 && out_dtype <= 2);
    if (kv_lora_rank != 512 || rope_dim != 64) return CHITU_ERR_UNSUPPORTED;
    if (q_lora_rank % 128 != 0 || q_lora_rank > kQProjWaves * kQProjMaxB * 128 || batch > 32) return CHITU_ERR_UNSUPPORTED;
    CHITU_REQUIRE(row_stride % 8 == 0 && row_stride >= q_lora_rank + 576);
    if (batch == 0) return CHITU_OK;
    const dim3 grid((unsigned)(batch + (N + 15) / 16)), block(64 * kQProjWaves);
#define LAUNCH(MT)                                                                                                   \
    hipLaunchKernelGGL((mla_q_proj_kernel<MT>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)qkv_a_bf16,       \
                       row_stride, (int)q_lora_rank, (const bf16_t*)q_norm_weight_bf16, q_eps, (const fp8_t*)wq_b_fp8, \
                       wq_b_scale, out, (int)out_dtype, (int)batch, (int)N, (int)batch,                               \
                       (const bf16_t*)kv_norm_weight_bf16, kv_eps, cos, sin, (bf16_t*)kv_cache, num_pages,            \
                       (int)page_size, page_table, (int)pages_per_seq, old_seq_lens)
    if (batch <= 16) LAUNCH(1);
    else LAUNCH(2);  // two token tiles keep the bf16 rows of both in registers; more would spill
#undef LAUNCH
    CHITU_RETURN_LAUNCH_STATUS();
}

CHITU_PROBE_READER(mla_q_proj)
