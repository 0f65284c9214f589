// Fused MoE with INT8 W8A8 experts (per-token activation scales, per-output-channel weight scales)
// for decode on gfx950 -- BASELINE config 4 (Mixtral-8x7B W8A8).
//
// The reference has no int8-W8A8 fused path (SURVEY gap G2): Mixtral loops over its experts
// (chitu/models/model_hf_mixtral.py:76-94) and `simple_w8a8` turns every expert linear into a W8A8Linear
// (chitu/quantize/w8a8.py:38-164, quantizer.py:117-145), i.e. per expert and token
//   h13 = fp16( (q(x) . q(w13)^T) * s_x * s_w13 );  a = silu(h13[:I]) * h13[I:];
//   y   = fp16( (q(a) . q(w2)^T) * s_a * s_w2 ) * routing_weight
// with q() = quant_act (per-row absmax / 127, round half even).  Here the same arithmetic runs grouped
// over moe_align's sorted slots, on the weight-streaming shape of moe.hip: 16-slot m-tiles, a wave owns
// the gate tile and the matching up tile, full-line weight loads (a lane's 16 B = one
// v_mfma_i32_16x16x64_i8 fragment), exact int32 accumulation over the whole K range, K split over the
// workgroup's waves (integer reduce through LDS: order-free, bit-reproducible).
// Activations are bf16 here (the reference's W8A8Linear emits fp16; the rounding points are the same).
#include "common.h"
#include "gemm_common.h"

namespace chitu {

typedef int i32x4m __attribute__((ext_vector_type(4)));

struct I8Stage2 {
    i32x4m wg[2], wu[2], x[2];
};
struct I8Stage1 {
    i32x4m w[2], x[2];
};

__device__ __forceinline__ i32x4m i8_combine(const i32x4m& e0, const i32x4m& o0, const i32x4m& e1, const i32x4m& o1) {
    return i32x4m{e0[0] + o0[1], e0[2] + o0[3], e1[0] + o1[1], e1[2] + o1[3]};
}

// write 4 values of one lane (tile columns 2g, 2g+1, 8+2g, 8+2g+1 of slot row) as bf16
__device__ __forceinline__ void i8_store_tile(bf16_t* row, int n0, int g, int N, const float (&v)[4]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int n = n0 + h * 8 + 2 * g;
        if (n + 1 < N && (N & 1) == 0) *reinterpret_cast<uint32_t*>(row + n) = f32x2_to_bf16x2(v[2 * h], v[2 * h + 1]);
        else {
            if (n < N) row[n] = f32_to_bf16(v[2 * h]);
            if (n + 1 < N) row[n + 1] = f32_to_bf16(v[2 * h + 1]);
        }
    }
}

// ---- GEMM1 + SiluAndMul: grid (I/16, max_mblocks); block 64*WK.  out: a bf16 [numel, I].
template <int WK>
__global__ __launch_bounds__(64 * WK) void moe_i8_gemm1_silu_kernel(
    const int8_t* __restrict__ Xq, const float* __restrict__ Xs, const int8_t* __restrict__ W,
    const float* __restrict__ Ws, const int32_t* __restrict__ sorted_ids, const int32_t* __restrict__ expert_ids,
    const int32_t* __restrict__ num_post_pad, bf16_t* __restrict__ out, int numel, int topk, int I, int K) {
    __shared__ int red[WK > 1 ? WK * 512 : 1];
    const int mb = blockIdx.y;
    if (mb * 16 >= *num_post_pad) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: SGPR index math
    const int j = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int N = 2 * I;
    const int KB = K >> 7;
    const int slot = sorted_ids[mb * 16 + j];
    const bool valid = slot < numel;
    const int e = expert_ids[mb];
    const int slot0 = __builtin_amdgcn_readfirstlane(slot);
    const int token = (valid ? slot : min(slot0, numel - 1)) / topk;
    i32x4m ge0 = {0, 0, 0, 0}, go0 = ge0, ge1 = ge0, go1 = ge0, ue0 = ge0, uo0 = ge0, ue1 = ge0, uo1 = ge0;
    if (e >= 0) {
        const int kb0 = KB * wave / WK, kb1 = KB * (wave + 1) / WK;
        const int8_t* xp = Xq + (size_t)token * K + g * 16;
        const fp8_t* Wb = reinterpret_cast<const fp8_t*>(W) + (size_t)e * N * K;
        const fp8_t *gp0, *gp1, *up0, *up1;
        w8_lane_ptrs(Wb, n0, N, K, j, g, gp0, gp1);
        w8_lane_ptrs(Wb, I + n0, N, K, j, g, up0, up1);
        auto load = [&](I8Stage2& st, int kb) {
            const int off = kb << 7;
            st.wg[0] = __builtin_nontemporal_load(reinterpret_cast<const i32x4m*>(gp0 + off));
            st.wg[1] = __builtin_nontemporal_load(reinterpret_cast<const i32x4m*>(gp1 + off));
            st.wu[0] = __builtin_nontemporal_load(reinterpret_cast<const i32x4m*>(up0 + off));
            st.wu[1] = __builtin_nontemporal_load(reinterpret_cast<const i32x4m*>(up1 + off));
            st.x[0] = *reinterpret_cast<const i32x4m*>(xp + off);
            st.x[1] = *reinterpret_cast<const i32x4m*>(xp + off + 64);
        };
        auto compute = [&](const I8Stage2& st) {
            ge0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(st.wg[0], st.x[0], ge0, 0, 0, 0);
            go0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(st.wg[0], st.x[1], go0, 0, 0, 0);
            ge1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(st.wg[1], st.x[0], ge1, 0, 0, 0);
            go1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(st.wg[1], st.x[1], go1, 0, 0, 0);
            ue0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(st.wu[0], st.x[0], ue0, 0, 0, 0);
            uo0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(st.wu[0], st.x[1], uo0, 0, 0, 0);
            ue1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(st.wu[1], st.x[0], ue1, 0, 0, 0);
            uo1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(st.wu[1], st.x[1], uo1, 0, 0, 0);
        };
        constexpr int D = 3;
        I8Stage2 ring[D];
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
    i32x4m ag = i8_combine(ge0, go0, ge1, go1), au = i8_combine(ue0, uo0, ue1, uo1);
    if (WK > 1) {
        *reinterpret_cast<i32x4m*>(&red[(wave * 128 + lane) * 4]) = ag;
        *reinterpret_cast<i32x4m*>(&red[(wave * 128 + 64 + lane) * 4]) = au;
        __syncthreads();
        if (wave != 0) return;
        ag = au = i32x4m{0, 0, 0, 0};
#pragma unroll
        for (int w = 0; w < WK; ++w) {
            const i32x4m vg = *reinterpret_cast<const i32x4m*>(&red[(w * 128 + lane) * 4]);
            const i32x4m vu = *reinterpret_cast<const i32x4m*>(&red[(w * 128 + 64 + lane) * 4]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ag[r] += vg[r];
                au[r] += vu[r];
            }
        }
    }
    if (!valid) return;
    const float sx = Xs[token];
    const float* wse = Ws + (size_t)max(e, 0) * N;
    float h[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = w8_out_col(n0, g, r);
        const int nc = min(n, I - 1);
        const float gv = e >= 0 ? round_bf16(((float)ag[r] * sx) * wse[nc]) : 0.f;  // W8A8Linear output, rounded
        const float uv = e >= 0 ? round_bf16(((float)au[r] * sx) * wse[I + nc]) : 0.f;
        h[r] = round_bf16(gv / (1.0f + expf(-gv))) * uv;
    }
    i8_store_tile(out + (size_t)slot * I, n0, g, I, h);
}

// ---- GEMM2: grid (N/16, max_mblocks); block 64*WK.  a int8 [numel, I] + per-slot scales.
// out[slot, n] = bf16( bf16((acc * s_a[slot]) * s_w2[e][n]) * routing_weight[slot] )
template <int WK>
__global__ __launch_bounds__(64 * WK) void moe_i8_gemm2_kernel(
    const int8_t* __restrict__ Aq, const float* __restrict__ As, const int8_t* __restrict__ W,
    const float* __restrict__ Ws, const int32_t* __restrict__ sorted_ids, const int32_t* __restrict__ expert_ids,
    const int32_t* __restrict__ num_post_pad, const void* __restrict__ topk_w, int w_dt, bf16_t* __restrict__ out,
    int numel, int N, int I, int mul_weight) {
    __shared__ int red[WK > 1 ? WK * 256 : 1];
    const int mb = blockIdx.y;
    if (mb * 16 >= *num_post_pad) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: SGPR index math
    const int j = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int KB = I >> 7;
    const int slot = sorted_ids[mb * 16 + j];
    const bool valid = slot < numel;
    const int e = expert_ids[mb];
    const int row = valid ? slot : 0;
    i32x4m e0 = {0, 0, 0, 0}, o0 = e0, e1 = e0, o1 = e0;
    if (e >= 0) {
        const int kb0 = KB * wave / WK, kb1 = KB * (wave + 1) / WK;
        const int8_t* xp = Aq + (size_t)row * I + g * 16;
        const fp8_t *wp0, *wp1;
        w8_lane_ptrs(reinterpret_cast<const fp8_t*>(W) + (size_t)e * N * I, n0, N, I, j, g, wp0, wp1);
        auto load = [&](I8Stage1& st, int kb) {
            const int off = kb << 7;
            st.w[0] = __builtin_nontemporal_load(reinterpret_cast<const i32x4m*>(wp0 + off));
            st.w[1] = __builtin_nontemporal_load(reinterpret_cast<const i32x4m*>(wp1 + off));
            st.x[0] = *reinterpret_cast<const i32x4m*>(xp + off);
            st.x[1] = *reinterpret_cast<const i32x4m*>(xp + off + 64);
        };
        auto compute = [&](const I8Stage1& st) {
            e0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(st.w[0], st.x[0], e0, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(st.w[0], st.x[1], o0, 0, 0, 0);
            e1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(st.w[1], st.x[0], e1, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(st.w[1], st.x[1], o1, 0, 0, 0);
        };
        constexpr int D = 4;
        I8Stage1 ring[D];
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
    i32x4m acc = i8_combine(e0, o0, e1, o1);
    if (WK > 1) {
        *reinterpret_cast<i32x4m*>(&red[(wave * 64 + lane) * 4]) = acc;
        __syncthreads();
        if (wave != 0) return;
        acc = i32x4m{0, 0, 0, 0};
#pragma unroll
        for (int w = 0; w < WK; ++w) {
            const i32x4m v = *reinterpret_cast<const i32x4m*>(&red[(w * 64 + lane) * 4]);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += v[r];
        }
    }
    if (!valid) return;
    float rw = 1.0f;
    if (mul_weight)
        rw = w_dt == 0 ? bf16_to_f32(((const bf16_t*)topk_w)[slot]) : w_dt == 1 ? f16_to_f32(((const uint16_t*)topk_w)[slot]) : ((const float*)topk_w)[slot];
    const float sa = As[row];
    const float* wse = Ws + (size_t)max(e, 0) * N;
    float y[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = min(w8_out_col(n0, g, r), N - 1);
        y[r] = e >= 0 ? round_bf16(((float)acc[r] * sa) * wse[n]) * rw : 0.f;
    }
    i8_store_tile(out + (size_t)slot * N, n0, g, N, y);
}

static int pick_wk(int64_t wgs, int KB) {
    // K split over the waves of a workgroup.  Round 6 (same-box sweep on the Mixtral step, profiles/r06_mixtral_int8.txt): a one-wave
    // workgroup per 16-row tile (the old choice above 3200 workgroups) streams at 5.0 TB/s, four waves per tile at 5.6-5.8 -- with K
    // blocks to spare (Mixtral: 32 and 112) the split costs an LDS reduce and buys four loads in flight per tile: never below 4 there.
    int WK = wgs <= 640 ? 8 : 4;
    if (KB < 16) WK = wgs <= 640 ? 8 : wgs <= 1536 ? 4 : wgs <= 3200 ? 2 : 1;
    debug_override(kOptMoeI8WK, WK);
    while (WK > 1 && WK > KB) WK >>= 1;
    return WK;
}

}  // namespace chitu

extern "C" int chitu_hip_moe_i8_gemm1_silu(const void* a_int8, const float* a_scale, const void* w1_int8,
                                           const float* w1_scale, const int32_t* sorted_token_ids,
                                           const int32_t* expert_ids, const int32_t* num_tokens_post_pad,
                                           void* h_bf16, int64_t numel, int32_t topk, int64_t inter_size, int64_t K,
                                           int64_t max_mblocks, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(a_int8 && a_scale && w1_int8 && w1_scale && sorted_token_ids && expert_ids && num_tokens_post_pad && h_bf16);
    CHITU_REQUIRE(numel >= 0 && topk >= 1 && inter_size >= 16 && K >= 128 && max_mblocks >= 0);
    if (K % 128 != 0 || inter_size % 16 != 0) return CHITU_ERR_UNSUPPORTED;
    if (numel == 0 || max_mblocks == 0) return CHITU_OK;
    const int n_tiles = (int)(inter_size / 16);
    const int WK = pick_wk(2 * (int64_t)n_tiles * (numel < max_mblocks ? numel : max_mblocks), (int)(K / 128));
    const dim3 grid((unsigned)n_tiles, (unsigned)max_mblocks);
    hipStream_t st = (hipStream_t)stream;
#define L1(WKV)                                                                                                  \
    hipLaunchKernelGGL((moe_i8_gemm1_silu_kernel<WKV>), grid, dim3(64 * WKV), 0, st, (const int8_t*)a_int8, a_scale, \
                       (const int8_t*)w1_int8, w1_scale, sorted_token_ids, expert_ids, num_tokens_post_pad,       \
                       (bf16_t*)h_bf16, (int)numel, (int)topk, (int)inter_size, (int)K)
    if (WK == 8) L1(8); else if (WK == 4) L1(4); else if (WK == 2) L1(2); else L1(1);
#undef L1
    CHITU_RETURN_LAUNCH_STATUS();
}

extern "C" int chitu_hip_moe_i8_gemm2(const void* a_int8, const float* a_scale, const void* w2_int8,
                                      const float* w2_scale, const int32_t* sorted_token_ids,
                                      const int32_t* expert_ids, const int32_t* num_tokens_post_pad,
                                      const void* topk_weights, int weights_dtype, int32_t mul_routed_weight,
                                      void* out_bf16, int64_t numel, int64_t N, int64_t inter_size,
                                      int64_t max_mblocks, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(a_int8 && a_scale && w2_int8 && w2_scale && sorted_token_ids && expert_ids && num_tokens_post_pad && out_bf16);
    CHITU_REQUIRE((topk_weights || !mul_routed_weight) && weights_dtype >= 0 && weights_dtype <= 2);
    CHITU_REQUIRE(numel >= 0 && N >= 1 && inter_size >= 128 && max_mblocks >= 0);
    if (inter_size % 128 != 0) return CHITU_ERR_UNSUPPORTED;
    if (numel == 0 || max_mblocks == 0) return CHITU_OK;
    const int n_tiles = (int)((N + 15) / 16);
    const int WK = pick_wk((int64_t)n_tiles * (numel < max_mblocks ? numel : max_mblocks), (int)(inter_size / 128));
    const dim3 grid((unsigned)n_tiles, (unsigned)max_mblocks);
    hipStream_t st = (hipStream_t)stream;
#define L2(WKV)                                                                                                   \
    hipLaunchKernelGGL((moe_i8_gemm2_kernel<WKV>), grid, dim3(64 * WKV), 0, st, (const int8_t*)a_int8, a_scale,    \
                       (const int8_t*)w2_int8, w2_scale, sorted_token_ids, expert_ids, num_tokens_post_pad,        \
                       topk_weights, weights_dtype, (bf16_t*)out_bf16, (int)numel, (int)N, (int)inter_size,        \
                       (int)mul_routed_weight)
    if (WK == 8) L2(8); else if (WK == 4) L2(4); else if (WK == 2) L2(2); else L2(1);
#undef L2
    CHITU_RETURN_LAUNCH_STATUS();
}
