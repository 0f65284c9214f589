// Fused MoE for decode on gfx950: grouped FP8 (W8A8, 128x128 block scales) GEMMs over
// expert-sorted token slots, SiLU-and-mul + re-quantisation, routed-weight epilogue, top-k sum.
//
// Replaces (reference, read-only):
//   chitu/fused_moe.py:62-307     fused_moe_kernel (Triton grouped GEMM, BLOCK_M=64 tiles)
//   chitu/fused_moe.py:796-891    invoke_fused_moe_kernel
//   chitu/fused_moe.py:24-39      SiluAndMul
//   chitu/fused_moe.py:670-710    _per_token_group_quant_fp8 (between the two GEMMs)
//   chitu/fused_moe.py:1299-1305  moe_sum
//
// Design.  In decode each routed expert sees 1-3 tokens, so the reference's 64-row tiles are
// >90% padding and its cost is the expert weights, streamed once.  Here the m-tile is 16 sorted
// slots (one MFMA tile; moe_align is run with block 16) and everything else mirrors the dense
// weight-streaming GEMM of fp8_gemm.hip: a wave owns 16 weight rows, every wave-load takes whole
// 128-B lines (gemm_common.h full-line layout; non-temporal: each weight byte is used once), the
// same k order is applied to the gathered activation rows, per-128 block scales are folded in fp32
// in the reference's order
// (fused_moe.py:281).  GEMM1 splits K over the waves of a workgroup; GEMM2 (K = moe_inter/tp,
// two K blocks at TP=8) gives every wave its own 16-row tiles and keeps the 16x256 activation
// fragment in registers.  Rounding points are the reference's: GEMM outputs -> bf16, silu in
// fp32 -> bf16, product -> bf16, routed weight multiplied on the fp32 accumulator, top-k sum in
// fp32 over bf16 values.  No atomics anywhere: results are run-to-run identical.
#include "common.h"
#include "gemm_common.h"
// Profiling aid, compile-time only (probe builds, never shipped): 1 = moe_gemm1_silu_kernel addresses the activations as
// if they were laid out tile-major ([K/16][16 tokens][16 B], scales [K/128][16 tokens]; batches <= 16) -- timing only.
#ifndef CHITU_MOE_PROBE_MASK
#define CHITU_MOE_PROBE_MASK 0
#endif

namespace chitu {

struct MoeStage {
    W8Frag w;
    i32x4 x[2];
    float xs, ws;
};

// write one lane's 4 results (tile columns 2g,2g+1,8+2g,8+2g+1 of token slot) as bf16, scaled by rw
__device__ __forceinline__ void moe_store_tile(bf16_t* out_row, int n0, int g, int N, const f32x4& acc, float rw) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int n = n0 + h * 8 + 2 * g;
        const uint16_t a = f32_to_bf16(acc[2 * h] * rw), b = f32_to_bf16(acc[2 * h + 1] * rw);
        if (n + 1 < N && (N & 1) == 0) *reinterpret_cast<uint32_t*>(out_row + n) = (uint32_t)a | ((uint32_t)b << 16);
        else {
            if (n < N) out_row[n] = a;
            if (n + 1 < N) out_row[n + 1] = b;
        }
    }
}

__device__ __forceinline__ float moe_routed_weight(const void* topk_w, int w_dt, int slot) {
    if (w_dt == 0) return bf16_to_f32(((const bf16_t*)topk_w)[slot]);
    if (w_dt == 1) return f16_to_f32(((const uint16_t*)topk_w)[slot]);
    return ((const float*)topk_w)[slot];
}

// ---------------------------------------------------------------- GEMM1: x[token] . W1[e]^T
// grid (n_tiles / NW, max_mblocks); block 64*WK*NW: NW waves own neighbouring 16-row tiles of the same
// m-block (their activation fragments are the same addresses at about the same time -> L1 hits),
// WK waves split K.  out: bf16 [numel, N] (row = sorted slot id).
template <int WK, int NW, int D>
__global__ __launch_bounds__(64 * WK * NW) void moe_gemm1_kernel(
    const fp8_t* __restrict__ Xq, const float* __restrict__ Xs, const fp8_t* __restrict__ W,
    const float* __restrict__ Ws, const int32_t* __restrict__ sorted_ids,
    const int32_t* __restrict__ expert_ids, const int32_t* __restrict__ num_post_pad,
    bf16_t* __restrict__ out, int numel, int topk, int N, int K) {
    static_assert(WK == 1 || NW == 1, "either split K or tile N inside a workgroup");
    __shared__ float red[WK > 1 ? WK * 256 : 1];
    const int mb = blockIdx.y;
    if (mb * 16 >= *num_post_pad) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: SGPR index math
    const int j = lane & 15, g = lane >> 4;
    const int n0 = (NW > 1 ? blockIdx.x * NW + wave : blockIdx.x) * 16;
    if (NW > 1 && n0 >= N) return;
    const int KB = K >> 7;
    const int slot = sorted_ids[mb * 16 + j];
    const bool valid = slot < numel;
    const int e = expert_ids[mb];
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (e >= 0) {
        const int kw = NW > 1 ? 0 : wave;
        const int kb0 = KB * kw / WK, kb1 = KB * (kw + 1) / WK;
        // padded slots read the tile's first token (always valid) instead of token 0: the wave's
        // activation loads then touch only the rows of tokens this expert really has
        const int slot0 = __builtin_amdgcn_readfirstlane(slot);
        const int token = (valid ? slot : min(slot0, numel - 1)) / topk;
        const fp8_t* xp = Xq + (size_t)token * K + g * 16;
        const float* xsp = Xs + (size_t)token * KB;
        const fp8_t *wp0, *wp1;
        w8_lane_ptrs(W + (size_t)e * N * K, n0, N, K, j, g, wp0, wp1);
        const float* wsp = Ws + ((size_t)e * ((N + 127) >> 7) + (n0 >> 7)) * KB;
        auto load = [&](MoeStage& st, int kb) {
            const int off = kb << 7;
            st.w.w[0] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(wp0 + off));
            st.w.w[1] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(wp1 + off));
            st.x[0] = *reinterpret_cast<const i32x4*>(xp + off);
            st.x[1] = *reinterpret_cast<const i32x4*>(xp + off + 64);
            st.xs = xsp[kb];
            st.ws = wsp[kb];
        };
        auto compute = [&](const MoeStage& st) {
            const f32x4 blk = w8a8_block_dot(st.w, st.x[0], st.x[1]);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += (blk[r] * st.xs) * st.ws;
        };
        // D-deep register ring (see fp8_gemm.hip): D x 2 KB of expert weights in flight per wave.
        MoeStage ring[D];
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
    if (WK > 1) {
        *reinterpret_cast<f32x4*>(&red[(wave * 64 + lane) * 4]) = acc;
        __syncthreads();
        if (wave != 0) return;
        acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < WK; ++w) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&red[(w * 64 + lane) * 4]);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += v[r];
        }
    }
    if (!valid) return;
    moe_store_tile(out + (size_t)slot * N, n0, g, N, acc, 1.0f);
}

// ---------------------------------------------------------------- GEMM1 + SiLU-and-mul
// Same loop as moe_gemm1_kernel, but a wave owns the gate tile [n0, n0+16) AND the up tile
// [I+n0, I+n0+16) of W1 (N = 2I): both products share the activation fragments (half the
// activation loads per weight byte) and the wave ends up with g and u of the same 16 x 16 outputs,
// so h = bf16(bf16(silu(bf16(g))) * bf16(u)) -- SiluAndMul with the reference's rounding points,
// fused_moe.py:24-39 -- is formed in the epilogue and written as bf16 [numel, I].  The fp8
// re-quantisation needs a 128-wide group maximum (8 tiles): it is done by GEMM2's prologue
// (moe_gemm2_q_kernel).  grid (I/16, max_mblocks); block 64*WK (WK waves split K).
struct MoeStage2 {
    W8Frag wg, wu;
    i32x4 x[2];
    float xs, wsg, wsu;
};

// Q (needs WK == 1): the workgroup is 8 waves owning the 8 tiles of ONE 128-wide group of h, and the epilogue is
// per_token_group_quant_fp8 of that group (moe_silu_quant_kernel's arithmetic: the group maximum goes through LDS) --
// h leaves as e4m3 codes [numel, I] + scales [numel, I/128], the generic GEMM2's input, and no quantisation launch
// or bf16 h round trip remains.  grid (I/128, max_mblocks); block 512.
template <int WK, int D, bool Q = false>
__global__ __launch_bounds__(Q ? 512 : 64 * WK) void moe_gemm1_silu_kernel(
    const fp8_t* __restrict__ Xq, const float* __restrict__ Xs, const fp8_t* __restrict__ W,
    const float* __restrict__ Ws, const int32_t* __restrict__ sorted_ids,
    const int32_t* __restrict__ expert_ids, const int32_t* __restrict__ num_post_pad,
    bf16_t* __restrict__ out, int numel, int topk, int I, int K, fp8_t* __restrict__ hq = nullptr,
    float* __restrict__ hs = nullptr, float eps = 0.f) {
    static_assert(!Q || WK == 1, "the quantising epilogue owns whole K");
    __shared__ float red[WK > 1 ? WK * 512 : 1];
    __shared__ float qmax[Q ? 8 * 16 : 1];
    const int mb = blockIdx.y;
    if (mb * 16 >= *num_post_pad) return;
    const int lane = threadIdx.x & 63;
    const int wave_id = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: SGPR index math
    const int wave = Q ? 0 : wave_id;                                       // position in the K split
    const int j = lane & 15, g = lane >> 4;
    const int n0 = (Q ? blockIdx.x * 8 + wave_id : blockIdx.x) * 16;
    const int N = 2 * I;
    const int KB = K >> 7;
    const int slot = sorted_ids[mb * 16 + j];
    const bool valid = slot < numel;
    const int e = expert_ids[mb];
    f32x4 ag = f32x4{0.f, 0.f, 0.f, 0.f}, au = f32x4{0.f, 0.f, 0.f, 0.f};
    if (e >= 0) {
        const int kb0 = KB * wave / WK, kb1 = KB * (wave + 1) / WK;
        const int slot0 = __builtin_amdgcn_readfirstlane(slot);
        const int token = (valid ? slot : min(slot0, numel - 1)) / topk;
        const fp8_t* xp = Xq + (size_t)token * K + g * 16;
        const float* xsp = Xs + (size_t)token * KB;
        if (CHITU_MOE_PROBE_MASK & 1) {
            xp = Xq + (size_t)(g * 16 + (token & 15)) * 16;
            xsp = Xs + (token & 15);
        }
        const fp8_t* Wb = W + (size_t)e * N * K;
        const fp8_t *gp0, *gp1, *up0, *up1;
        w8_lane_ptrs(Wb, n0, N, K, j, g, gp0, gp1);
        w8_lane_ptrs(Wb, I + n0, N, K, j, g, up0, up1);
        const float* wsb = Ws + (size_t)e * ((N + 127) >> 7) * KB;
        const float* wsgp = wsb + (size_t)(n0 >> 7) * KB;
        const float* wsup = wsb + (size_t)((I + n0) >> 7) * KB;
        auto load = [&](MoeStage2& st, int kb) {
            const int off = kb << 7;
            st.wg.w[0] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(gp0 + off));
            st.wg.w[1] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(gp1 + off));
            st.wu.w[0] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(up0 + off));
            st.wu.w[1] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(up1 + off));
            if (CHITU_MOE_PROBE_MASK & 1) {
                st.x[0] = *reinterpret_cast<const i32x4*>(xp + (size_t)kb * 2048);
                st.x[1] = *reinterpret_cast<const i32x4*>(xp + (size_t)kb * 2048 + 1024);
                st.xs = xsp[kb * 16];
            } else {
                st.x[0] = *reinterpret_cast<const i32x4*>(xp + off);
                st.x[1] = *reinterpret_cast<const i32x4*>(xp + off + 64);
                st.xs = xsp[kb];
            }
            st.wsg = wsgp[kb];
            st.wsu = wsup[kb];
        };
        auto compute = [&](const MoeStage2& st) {
            const f32x4 bg = w8a8_block_dot(st.wg, st.x[0], st.x[1]);
            const f32x4 bu = w8a8_block_dot(st.wu, st.x[0], st.x[1]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ag[r] += (bg[r] * st.xs) * st.wsg;
                au[r] += (bu[r] * st.xs) * st.wsu;
            }
        };
        MoeStage2 ring[D];
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
    if (WK > 1) {
        *reinterpret_cast<f32x4*>(&red[(wave * 128 + lane) * 4]) = ag;
        *reinterpret_cast<f32x4*>(&red[(wave * 128 + 64 + lane) * 4]) = au;
        __syncthreads();
        if (wave != 0) return;
        ag = f32x4{0.f, 0.f, 0.f, 0.f};
        au = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < WK; ++w) {
            const f32x4 vg = *reinterpret_cast<const f32x4*>(&red[(w * 128 + lane) * 4]);
            const f32x4 vu = *reinterpret_cast<const f32x4*>(&red[(w * 128 + 64 + lane) * 4]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ag[r] += vg[r];
                au[r] += vu[r];
            }
        }
    }
    if (!Q && !valid) return;
    f32x4 h;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float gv = round_bf16(ag[r]), uv = round_bf16(au[r]);  // GEMM1's bf16 output (c1)
        const float sl = round_bf16(gv / (1.0f + expf(-gv)));
        h[r] = round_bf16(sl * uv);
    }
    if (!Q) {
        moe_store_tile(out + (size_t)slot * I, n0, g, I, h, 1.0f);
        return;
    }
    // the lane's 4 values are columns n0 + {2g, 2g+1, 8+2g, 8+2g+1} of row j: the row's maximum over the tile, then
    // over the group's 8 tiles
    float amax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(h[0]), __builtin_fabsf(h[1])),
                                 __builtin_fmaxf(__builtin_fabsf(h[2]), __builtin_fabsf(h[3])));
    amax = __builtin_fmaxf(amax, __shfl_xor(amax, 16, 64));
    amax = __builtin_fmaxf(amax, __shfl_xor(amax, 32, 64));
    if (g == 0) qmax[wave_id * 16 + j] = amax;
    __syncthreads();
    amax = qmax[j];
#pragma unroll
    for (int w = 1; w < 8; ++w) amax = __builtin_fmaxf(amax, qmax[w * 16 + j]);
    const float sc = __builtin_fmaxf(amax, eps) / 448.0f;
    const float v[8] = {h[0], h[1], h[2], h[3], 0.f, 0.f, 0.f, 0.f};
    const uint32_t codes = (uint32_t)quant8_fp8<true>(v, sc)[0];
    if (!valid) return;
    fp8_t* qrow = hq + (size_t)slot * I + n0 + 2 * g;
    *reinterpret_cast<uint16_t*>(qrow) = (uint16_t)(codes & 0xffffu);
    *reinterpret_cast<uint16_t*>(qrow + 8) = (uint16_t)(codes >> 16);
    if (wave_id == 0 && g == 0) hs[(size_t)slot * (I >> 7) + blockIdx.x] = sc;
}

// ---------------------------------------------------------------- SiLU-and-mul + fp8 requant
// c1 [rows, 2I] bf16 -> h = bf16(bf16(silu(gate)) * up) -> q [rows, I] e4m3, s [rows, I/128].
// One 128-wide group per 16 lanes (same shape as act_quant_kernel MODE 1).
__global__ __launch_bounds__(256) void moe_silu_quant_kernel(const bf16_t* __restrict__ c1,
                                                             fp8_t* __restrict__ q,
                                                             float* __restrict__ s, int64_t rows,
                                                             int I, float eps, int act_rule) {
    const int lane16 = threadIdx.x & 15;
    const int gpr = I >> 7;  // groups per row
    const int64_t n_groups = rows * gpr;
    int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 4;
    for (; group < n_groups; group += stride) {
        const int64_t row = group / gpr;
        const int col = (int)(group % gpr) * 128 + lane16 * 8;
        const i32x4 graw = *reinterpret_cast<const i32x4*>(c1 + row * 2 * I + col);
        const i32x4 uraw = *reinterpret_cast<const i32x4*>(c1 + row * 2 * I + I + col);
        float h[8];
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t gu = (uint32_t)graw[i >> 1], uu = (uint32_t)uraw[i >> 1];
            const float gv = (i & 1) ? __uint_as_float(gu & 0xffff0000u) : __uint_as_float(gu << 16);
            const float uv = (i & 1) ? __uint_as_float(uu & 0xffff0000u) : __uint_as_float(uu << 16);
            const float sl = round_bf16(gv / (1.0f + expf(-gv)));
            h[i] = round_bf16(sl * uv);
            amax = __builtin_fmaxf(amax, __builtin_fabsf(h[i]));
        }
        amax = row16_reduce_max(amax);
        // act_rule: act_quant_deepseek_v3 (no eps, no clamp; triton_kernels.py:210-212) for the
        // dense / shared-expert MLP; otherwise per_token_group_quant_fp8 (fused_moe.py:701-703).
        if (!act_rule) amax = __builtin_fmaxf(amax, eps);
        const float sc = amax / 448.0f;
        const i32x2 o = act_rule ? quant8_fp8<false>(h, sc) : quant8_fp8<true>(h, sc);
        *reinterpret_cast<i32x2*>(q + row * I + col) = o;
        if (lane16 == 0) s[group] = sc;
    }
}

// ---------------------------------------------------------------- GEMM2: h[slot] . W2[e]^T * w
// grid (ceil(n_tiles / (4*NT)), max_mblocks); block 256 (4 waves, NT row tiles per wave, no K
// split).  KB = I/128 is small (2 at TP=8): the whole 16 x I activation fragment stays in
// registers (KBMAX blocks), only weights stream.
template <int KBMAX, int NT>
__global__ __launch_bounds__(256) void moe_gemm2_kernel(
    const fp8_t* __restrict__ Hq, const float* __restrict__ Hs, const fp8_t* __restrict__ W,
    const float* __restrict__ Ws, const int32_t* __restrict__ sorted_ids,
    const int32_t* __restrict__ expert_ids, const int32_t* __restrict__ num_post_pad,
    const void* __restrict__ topk_w, int w_dt, bf16_t* __restrict__ out, int numel, int N, int I,
    int mul_weight) {
    const int mb = blockIdx.y;
    if (mb * 16 >= *num_post_pad) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: SGPR index math
    const int j = lane & 15, g = lane >> 4;
    const int KB = I >> 7;
    const int slot = sorted_ids[mb * 16 + j];
    const bool valid = slot < numel;
    const int e = expert_ids[mb];
    const int row = valid ? slot : 0;
    const float rw = (mul_weight && valid) ? moe_routed_weight(topk_w, w_dt, slot) : 1.0f;
    i32x4 x[KBMAX][2];
    float xs[KBMAX];
    if (e >= 0) {
        const fp8_t* xp = Hq + (size_t)row * I + g * 16;
#pragma unroll
        for (int kb = 0; kb < KBMAX; ++kb) {
            if (kb < KB) {
                x[kb][0] = *reinterpret_cast<const i32x4*>(xp + (kb << 7));
                x[kb][1] = *reinterpret_cast<const i32x4*>(xp + (kb << 7) + 64);
                xs[kb] = Hs[(size_t)row * KB + kb];
            }
        }
    }
    const int tile0 = (blockIdx.x * 4 + wave) * NT;
    // all NT tiles' weights are requested before the first is consumed
    W8Frag wf[NT][KBMAX];
    if (e >= 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n0 = (tile0 + t) * 16;
            if (n0 < N) {
                const fp8_t *wp0, *wp1;
                w8_lane_ptrs(W + (size_t)e * N * I, n0, N, I, j, g, wp0, wp1);
#pragma unroll
                for (int kb = 0; kb < KBMAX; ++kb) {
                    if (kb < KB) {
                        wf[t][kb].w[0] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(wp0 + (kb << 7)));
                        wf[t][kb].w[1] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(wp1 + (kb << 7)));
                    }
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n0 = (tile0 + t) * 16;
        if (n0 >= N) break;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        if (e >= 0) {
            const float* wsp = Ws + ((size_t)e * ((N + 127) >> 7) + (n0 >> 7)) * KB;
#pragma unroll
            for (int kb = 0; kb < KBMAX; ++kb) {
                if (kb < KB) {
                    const f32x4 blk = w8a8_block_dot(wf[t][kb], x[kb][0], x[kb][1]);
                    const float ws = wsp[kb];
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] += (blk[r] * xs[kb]) * ws;
                }
            }
        }
        if (valid) moe_store_tile(out + (size_t)slot * N, n0, g, N, acc, rw);
    }
}

// GEMM2 with the fp8 re-quantisation of its input folded into the prologue: h [numel, I] bf16 is
// moe_gemm1_silu_kernel's output.  The 16 x I activation tile of the m-block is quantised once per
// workgroup (per_token_group_quant_fp8 rule, fused_moe.py:701-703; same arithmetic as
// moe_silu_quant_kernel): wave w takes the 64-column half-blocks hb = w, w+4, ..., the two halves
// of a 128-group meet through LDS for the maximum, the bytes are parked in LDS in MFMA-fragment
// order and every wave picks up its B operand there.  The h loads are issued BEFORE the weight
// loads and everything up to the MFMAs is straight-line code, so the prologue's s_waitcnt counts
// past the weight loads queued behind it (vector loads return in order).
template <int KB, int NT, int ROUNDS>
__global__ __launch_bounds__(256) void moe_gemm2_q_kernel(
    const bf16_t* __restrict__ Hb, const fp8_t* __restrict__ W, const float* __restrict__ Ws,
    const int32_t* __restrict__ sorted_ids, const int32_t* __restrict__ expert_ids,
    const int32_t* __restrict__ num_post_pad, const void* __restrict__ topk_w, int w_dt,
    bf16_t* __restrict__ out, int numel, int N, int mul_weight, float eps) {
    constexpr int I = KB * 128;
    constexpr int HPW = (2 * KB + 3) / 4;  // half-blocks per wave
    __shared__ float amax_lds[2 * KB][16];
    __shared__ i32x4 xq_lds[2 * KB][64];
    const int mb = blockIdx.y;
    if (mb * 16 >= *num_post_pad) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: SGPR index math
    const int j = lane & 15, g = lane >> 4;
    const int slot = sorted_ids[mb * 16 + j];
    const bool valid = slot < numel;
    const int e = expert_ids[mb];
    // round r of this wave covers tiles [tile_of(r), tile_of(r) + NT)
    auto tile_of = [&](int r) { return ((blockIdx.x * ROUNDS + r) * 4 + wave) * NT; };
    if (e < 0) {  // expert not on this rank (expert_map): the slot's contribution is zero
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int n0 = (tile_of(r) + t) * 16;
                if (n0 < N && valid) moe_store_tile(out + (size_t)slot * N, n0, g, N, f32x4{0.f, 0.f, 0.f, 0.f}, 1.0f);
            }
        return;
    }
    const int row = valid ? slot : 0;
    const bf16_t* hrow = Hb + (size_t)row * I;
    i32x4 hraw[HPW][2];
#pragma unroll
    for (int q = 0; q < HPW; ++q) {
        const int hb = min(wave + 4 * q, 2 * KB - 1);
        hraw[q][0] = *reinterpret_cast<const i32x4*>(hrow + hb * 64 + g * 16);
        hraw[q][1] = *reinterpret_cast<const i32x4*>(hrow + hb * 64 + g * 16 + 8);
    }
    const int last_tile = (N - 1) >> 4;
    const fp8_t* We = W + (size_t)e * N * I;
    // weights: one round (NT tiles) requested ahead of the one being multiplied, so the workgroup
    // streams ROUNDS x NT x 4 tiles behind ONE slot/expert lookup and ONE quantisation prologue
    W8Frag wf[ROUNDS > 1 ? 2 : 1][NT][KB];
    auto load_round = [&](W8Frag (&dst)[NT][KB], int r) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n0 = min(tile_of(r) + t, last_tile) * 16;  // tail tiles re-read the last one, never stored
            const fp8_t *wp0, *wp1;
            w8_lane_ptrs(We, n0, N, I, j, g, wp0, wp1);
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                dst[t][kb].w[0] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(wp0 + (kb << 7)));
                dst[t][kb].w[1] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(wp1 + (kb << 7)));
            }
        }
    };
    load_round(wf[0], 0);
    const float rw = (mul_weight && valid) ? moe_routed_weight(topk_w, w_dt, slot) : 1.0f;
    float h[HPW][16];
#pragma unroll
    for (int q = 0; q < HPW; ++q) {
        const int hb = wave + 4 * q;
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t u = (uint32_t)hraw[q][i >> 3][(i >> 1) & 3];
            h[q][i] = (i & 1) ? __uint_as_float(u & 0xffff0000u) : __uint_as_float(u << 16);
            amax = __builtin_fmaxf(amax, __builtin_fabsf(h[q][i]));
        }
        amax = __builtin_fmaxf(amax, __shfl_xor(amax, 16, 64));
        amax = __builtin_fmaxf(amax, __shfl_xor(amax, 32, 64));
        if (g == 0 && hb < 2 * KB) amax_lds[hb][j] = amax;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < HPW; ++q) {
        const int hb = wave + 4 * q;
        if (hb < 2 * KB) {
            const float sc = __builtin_fmaxf(__builtin_fmaxf(amax_lds[hb & ~1][j], amax_lds[hb | 1][j]), eps) / 448.0f;
            float lo[8], hi[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                lo[i] = h[q][i];
                hi[i] = h[q][8 + i];
            }
            const i32x2 a = quant8_fp8<true>(lo, sc), b = quant8_fp8<true>(hi, sc);
            xq_lds[hb][lane] = i32x4{a[0], a[1], b[0], b[1]};
        }
    }
    __syncthreads();
    i32x4 x[KB][2];
    float xs[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        x[kb][0] = xq_lds[2 * kb][lane];
        x[kb][1] = xq_lds[2 * kb + 1][lane];
        xs[kb] = __builtin_fmaxf(__builtin_fmaxf(amax_lds[2 * kb][j], amax_lds[2 * kb + 1][j]), eps) / 448.0f;
    }
    const float* wse = Ws + (size_t)e * ((N + 127) >> 7) * KB;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        if (r + 1 < ROUNDS) load_round(wf[(r + 1) & 1], r + 1);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n0 = (tile_of(r) + t) * 16;
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* wsp = wse + (size_t)(min(n0, N - 1) >> 7) * KB;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const f32x4 blk = w8a8_block_dot(wf[r & 1][t][kb], x[kb][0], x[kb][1]);
                const float ws = wsp[kb];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) acc[rr] += (blk[rr] * xs[kb]) * ws;
            }
            if (valid && n0 < N) moe_store_tile(out + (size_t)slot * N, n0, g, N, acc, rw);
        }
    }
}

// (Experts wider than 512 had a variant of this launch with the quantised bytes in LDS, moe_gemm2_qw_kernel, rounds 3-4:
// measured slower than silu_mul_quant + the generic GEMM2 at DeepSeek-V2-Lite's 1408-wide experts -- 39.0 us against
// 4.95 + 27.2, profiles/r03_v2lite_wide_experts.txt -- never the default, removed in round 5: wide experts take the
// three-launch form.)
// Generic-K GEMM2 (any I): reuses the GEMM1 kernel shape with slot-indexed activations.
template <int WK>
__global__ __launch_bounds__(64 * WK) void moe_gemm2_generic_kernel(
    const fp8_t* __restrict__ Hq, const float* __restrict__ Hs, const fp8_t* __restrict__ W,
    const float* __restrict__ Ws, const int32_t* __restrict__ sorted_ids,
    const int32_t* __restrict__ expert_ids, const int32_t* __restrict__ num_post_pad,
    const void* __restrict__ topk_w, int w_dt, bf16_t* __restrict__ out, int numel, int N, int I,
    int mul_weight) {
    __shared__ float red[WK > 1 ? WK * 256 : 1];
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
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (e >= 0) {
        const int kb0 = KB * wave / WK, kb1 = KB * (wave + 1) / WK;
        const fp8_t* xp = Hq + (size_t)row * I + g * 16;
        const float* xsp = Hs + (size_t)row * KB;
        const fp8_t *wp0, *wp1;
        w8_lane_ptrs(W + (size_t)e * N * I, n0, N, I, j, g, wp0, wp1);
        const float* wsp = Ws + ((size_t)e * ((N + 127) >> 7) + (n0 >> 7)) * KB;
        for (int kb = kb0; kb < kb1; ++kb) {
            const int off = kb << 7;
            W8Frag f;
            f.w[0] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(wp0 + off));
            f.w[1] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(wp1 + off));
            const i32x4 x0 = *reinterpret_cast<const i32x4*>(xp + off);
            const i32x4 x1 = *reinterpret_cast<const i32x4*>(xp + off + 64);
            const f32x4 blk = w8a8_block_dot(f, x0, x1);
            const float xs = xsp[kb], ws = wsp[kb];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += (blk[r] * xs) * ws;
        }
    }
    if (WK > 1) {
        *reinterpret_cast<f32x4*>(&red[(wave * 64 + lane) * 4]) = acc;
        __syncthreads();
        if (wave != 0) return;
        acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < WK; ++w) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&red[(w * 64 + lane) * 4]);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += v[r];
        }
    }
    if (!valid) return;
    const float rw = mul_weight ? moe_routed_weight(topk_w, w_dt, slot) : 1.0f;
    moe_store_tile(out + (size_t)slot * N, n0, g, N, acc, rw);
}

// ---------------------------------------------------------------- top-k sum
// out[t][n] = bf16( sum_k float(c3[t][k][n]) ), k ascending (fused_moe.py:1299-1305).
__global__ __launch_bounds__(256) void moe_sum_kernel(const bf16_t* __restrict__ c3,
                                                      bf16_t* __restrict__ out, int64_t M, int topk,
                                                      int64_t N) {
    const int64_t n8 = N >> 3;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < M * n8;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = idx / n8, c = (idx % n8) * 8;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < topk; ++k) {
            const i32x4 raw = *reinterpret_cast<const i32x4*>(c3 + (t * topk + k) * N + c);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t u = (uint32_t)raw[i];
                acc[2 * i] += __uint_as_float(u << 16);
                acc[2 * i + 1] += __uint_as_float(u & 0xffff0000u);
            }
        }
        i32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            o[i] = (int)((uint32_t)f32_to_bf16(acc[2 * i]) | ((uint32_t)f32_to_bf16(acc[2 * i + 1]) << 16));
        *reinterpret_cast<i32x4*>(out + t * N + c) = o;
    }
}

}  // namespace chitu

extern "C" int chitu_hip_moe_gemm1_fp8(const void* a_fp8, const float* a_scale, const void* w1_fp8,
                                       const float* w1_scale, const int32_t* sorted_token_ids,
                                       const int32_t* expert_ids,
                                       const int32_t* num_tokens_post_pad, void* out_bf16,
                                       int64_t numel, int32_t topk, int64_t N, int64_t K,
                                       int64_t max_mblocks, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(a_fp8 && a_scale && w1_fp8 && w1_scale && sorted_token_ids && expert_ids);
    CHITU_REQUIRE(num_tokens_post_pad && out_bf16);
    CHITU_REQUIRE(numel >= 0 && topk >= 1 && N >= 1 && K >= 128 && max_mblocks >= 0);
    if (K % 128 != 0) return CHITU_ERR_UNSUPPORTED;
    if (numel == 0 || max_mblocks == 0) return CHITU_OK;
    const int n_tiles = (int)((N + 15) / 16);
    const dim3 grid((unsigned)n_tiles, (unsigned)max_mblocks);
    const int64_t wgs = (int64_t)n_tiles * (numel < max_mblocks ? numel : max_mblocks);
    const int KB = (int)(K / 128);
    // measured (tools/bench_kernels.py, full-line loads): one wave per tile is fastest once the
    // grid alone fills the chip (>= 2048 waves); below that K is split over the workgroup's waves
    int WK = wgs <= 512 ? 8 : wgs <= 1024 ? 4 : wgs <= 2048 ? 2 : 1;
    int NW = 1, D = 3;  // sweep on MI355X: D=3 5.7 TB/s, D=2 5.4, D=4 5.2; NW>1 (shared L1 activations) loses 5-20%
    debug_override(kOptMoeGemm1WK, WK);  // variant overrides (tests, tools/bench_kernels.py)
    debug_override(kOptMoeGemm1NW, NW);
    debug_override(kOptMoeGemm1D, D);
    while (WK > 1 && WK > KB) WK >>= 1;
    if (WK > 1) NW = 1;
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(WKV, NWV, DV)                                                                          \
    hipLaunchKernelGGL((moe_gemm1_kernel<WKV, NWV, DV>), dim3((unsigned)((n_tiles + NWV - 1) / NWV), (unsigned)max_mblocks), \
                       dim3(64 * WKV * NWV), 0, st, (const fp8_t*)a_fp8, a_scale, (const fp8_t*)w1_fp8, w1_scale, \
                       sorted_token_ids, expert_ids, num_tokens_post_pad, (bf16_t*)out_bf16, (int)numel,  \
                       (int)topk, (int)N, (int)K)
    if (WK == 8) LAUNCH(8, 1, 4);
    else if (WK == 4) LAUNCH(4, 1, 4);
    else if (WK == 2) LAUNCH(2, 1, 4);
    else if (NW == 4 && D == 2) LAUNCH(1, 4, 2);
    else if (NW == 4) LAUNCH(1, 4, 4);
    else if (NW == 2 && D == 2) LAUNCH(1, 2, 2);
    else if (NW == 2) LAUNCH(1, 2, 4);
    else if (D == 2) LAUNCH(1, 1, 2);
    else if (D == 4) LAUNCH(1, 1, 4);
    else LAUNCH(1, 1, 3);
#undef LAUNCH
    CHITU_RETURN_LAUNCH_STATUS();
}

extern "C" int chitu_hip_moe_silu_mul_quant_fp8(const void* c1_bf16, int64_t rows,
                                                int64_t inter_size, int32_t quant_mode, float eps,
                                                void* q_fp8, float* scales, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(c1_bf16 && q_fp8 && scales && rows >= 0 && inter_size >= 128);
    CHITU_REQUIRE(quant_mode == 0 || quant_mode == 1);
    if (inter_size % 128 != 0) return CHITU_ERR_UNSUPPORTED;
    if (rows == 0) return CHITU_OK;
    const int64_t lanes = rows * (inter_size / 128) * 16;
    int64_t blocks = (lanes + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(moe_silu_quant_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       (hipStream_t)stream, (const bf16_t*)c1_bf16, (fp8_t*)q_fp8, scales, rows,
                       (int)inter_size, eps, quant_mode == 0 ? 1 : 0);
    CHITU_RETURN_LAUNCH_STATUS();
}

extern "C" int chitu_hip_moe_gemm2_fp8(const void* h_fp8, const float* h_scale, const void* w2_fp8,
                                       const float* w2_scale, const int32_t* sorted_token_ids,
                                       const int32_t* expert_ids,
                                       const int32_t* num_tokens_post_pad, const void* topk_weights,
                                       int weights_dtype, int32_t mul_routed_weight,
                                       void* out_bf16, int64_t numel, int64_t N, int64_t inter_size,
                                       int64_t max_mblocks, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(h_fp8 && h_scale && w2_fp8 && w2_scale && sorted_token_ids && expert_ids);
    CHITU_REQUIRE(num_tokens_post_pad && out_bf16 && (topk_weights || !mul_routed_weight));
    CHITU_REQUIRE(numel >= 0 && N >= 1 && inter_size >= 128 && max_mblocks >= 0);
    CHITU_REQUIRE(weights_dtype >= 0 && weights_dtype <= 2);
    if (inter_size % 128 != 0) return CHITU_ERR_UNSUPPORTED;
    if (numel == 0 || max_mblocks == 0) return CHITU_OK;
    hipStream_t st = (hipStream_t)stream;
    const int n_tiles = (int)((N + 15) / 16);
    const int KB = (int)(inter_size / 128);
    if (KB <= 4) {
        const int64_t mbs = numel < max_mblocks ? numel : max_mblocks;
#define LAUNCH2(KBM, NTV)                                                                        \
    hipLaunchKernelGGL((moe_gemm2_kernel<KBM, NTV>),                                             \
                       dim3((unsigned)((n_tiles + 4 * NTV - 1) / (4 * NTV)), (unsigned)max_mblocks), \
                       dim3(256), 0, st, (const fp8_t*)h_fp8, h_scale, (const fp8_t*)w2_fp8,     \
                       w2_scale, sorted_token_ids, expert_ids, num_tokens_post_pad, topk_weights, \
                       weights_dtype, (bf16_t*)out_bf16, (int)numel, (int)N, (int)inter_size,    \
                       (int)mul_routed_weight)
        const bool many = (int64_t)n_tiles * mbs > 8192;
        if (KB <= 2) {
            if (many) LAUNCH2(2, 4); else LAUNCH2(2, 2);
        } else {
            if (many) LAUNCH2(4, 4); else LAUNCH2(4, 2);
        }
#undef LAUNCH2
    } else {
        const dim3 grid((unsigned)n_tiles, (unsigned)max_mblocks);
        const int64_t wgs = (int64_t)n_tiles * (numel < max_mblocks ? numel : max_mblocks);
        int WK = wgs <= 512 ? 8 : wgs <= 1024 ? 4 : wgs <= 4096 ? 2 : 1;
        while (WK > 1 && WK > KB) WK >>= 1;
#define LAUNCHG(WKV)                                                                              \
    hipLaunchKernelGGL(moe_gemm2_generic_kernel<WKV>, grid, dim3(64 * WKV), 0, st,                \
                       (const fp8_t*)h_fp8, h_scale, (const fp8_t*)w2_fp8, w2_scale,              \
                       sorted_token_ids, expert_ids, num_tokens_post_pad, topk_weights,           \
                       weights_dtype, (bf16_t*)out_bf16, (int)numel, (int)N, (int)inter_size,     \
                       (int)mul_routed_weight)
        switch (WK) {
            case 8: LAUNCHG(8); break;
            case 4: LAUNCHG(4); break;
            case 2: LAUNCHG(2); break;
            default: LAUNCHG(1); break;
        }
#undef LAUNCHG
    }
    CHITU_RETURN_LAUNCH_STATUS();
}

extern "C" int chitu_hip_moe_gemm1_silu_fp8(const void* a_fp8, const float* a_scale, const void* w1_fp8,
                                            const float* w1_scale, const int32_t* sorted_token_ids,
                                            const int32_t* expert_ids, const int32_t* num_tokens_post_pad,
                                            void* h_bf16, int64_t numel, int32_t topk, int64_t inter_size,
                                            int64_t K, int64_t max_mblocks, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(a_fp8 && a_scale && w1_fp8 && w1_scale && sorted_token_ids && expert_ids);
    CHITU_REQUIRE(num_tokens_post_pad && h_bf16);
    CHITU_REQUIRE(numel >= 0 && topk >= 1 && inter_size >= 16 && K >= 128 && max_mblocks >= 0);
    if (K % 128 != 0 || inter_size % 16 != 0) return CHITU_ERR_UNSUPPORTED;
    if (numel == 0 || max_mblocks == 0) return CHITU_OK;
    const int n_tiles = (int)(inter_size / 16);
    const int64_t wgs = 2 * (int64_t)n_tiles * (numel < max_mblocks ? numel : max_mblocks);
    const int KB = (int)(K / 128);
    // Sweep on MI355X (tools/bench_kernels.py, R1 TP=8 shapes, us per launch; wgs is the worst-case grid,
    // ~80% of it is live): bs 1/2 -> WK 8, D 2 (10.7 / 13.2); bs 4 -> WK 4, D 2 (27.5); bs 8 -> WK 2, D 3
    // (37.5; WK 1: 50.9); bs 16+ -> WK 1, D 3 (61.9; WK 2: 69.0).  Few workgroups: split K over more waves
    // and keep the ring shallow so every workgroup is resident at once; many: one long stream per wave.
    int WK = wgs <= 640 ? 8 : wgs <= 1536 ? 4 : wgs <= 3200 ? 2 : 1;
    debug_override(kOptMoeGemm1WK, WK);
    while (WK > 1 && WK > KB) WK >>= 1;
    int D = WK >= 4 ? 2 : 3;
    debug_override(kOptMoeGemm1D, D);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)n_tiles, (unsigned)max_mblocks);
#define LAUNCH1S(WKV, DV)                                                                                     \
    hipLaunchKernelGGL((moe_gemm1_silu_kernel<WKV, DV>), grid, dim3(64 * WKV), 0, st, (const fp8_t*)a_fp8, a_scale, \
                       (const fp8_t*)w1_fp8, w1_scale, sorted_token_ids, expert_ids, num_tokens_post_pad,      \
                       (bf16_t*)h_bf16, (int)numel, (int)topk, (int)inter_size, (int)K)
    if (WK == 8) { if (D == 2) LAUNCH1S(8, 2); else if (D == 3) LAUNCH1S(8, 3); else LAUNCH1S(8, 4); }
    else if (WK == 4) { if (D == 2) LAUNCH1S(4, 2); else if (D == 3) LAUNCH1S(4, 3); else LAUNCH1S(4, 4); }
    else if (WK == 2) { if (D == 4) LAUNCH1S(2, 4); else LAUNCH1S(2, 3); }
    else { if (D == 4) LAUNCH1S(1, 4); else LAUNCH1S(1, 3); }
#undef LAUNCH1S
    CHITU_RETURN_LAUNCH_STATUS();
}

extern "C" int chitu_hip_moe_gemm2_quant_fp8(const void* h_bf16, const void* w2_fp8, const float* w2_scale,
                                             const int32_t* sorted_token_ids, const int32_t* expert_ids,
                                             const int32_t* num_tokens_post_pad, const void* topk_weights,
                                             int weights_dtype, int32_t mul_routed_weight, void* out_bf16,
                                             int64_t numel, int64_t N, int64_t inter_size, int64_t max_mblocks,
                                             float eps, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(h_bf16 && w2_fp8 && w2_scale && sorted_token_ids && expert_ids);
    CHITU_REQUIRE(num_tokens_post_pad && out_bf16 && (topk_weights || !mul_routed_weight));
    CHITU_REQUIRE(numel >= 0 && N >= 1 && inter_size >= 128 && max_mblocks >= 0);
    CHITU_REQUIRE(weights_dtype >= 0 && weights_dtype <= 2);
    if (inter_size % 128 != 0 || inter_size > 512) return CHITU_ERR_UNSUPPORTED;  // wider experts: the three-launch form
    if (numel == 0 || max_mblocks == 0) return CHITU_OK;
    hipStream_t st = (hipStream_t)stream;
    const int n_tiles = (int)((N + 15) / 16);
    const int KB = (int)(inter_size / 128);
    const int64_t mbs = numel < max_mblocks ? numel : max_mblocks;
#define LAUNCH2Q(KBV, NTV, RV)                                                                       \
    hipLaunchKernelGGL((moe_gemm2_q_kernel<KBV, NTV, RV>),                                           \
                       dim3((unsigned)((n_tiles + 4 * NTV * RV - 1) / (4 * NTV * RV)), (unsigned)max_mblocks), \
                       dim3(256), 0, st, (const bf16_t*)h_bf16, (const fp8_t*)w2_fp8, w2_scale,      \
                       sorted_token_ids, expert_ids, num_tokens_post_pad, topk_weights, weights_dtype, \
                       (bf16_t*)out_bf16, (int)numel, (int)N, (int)mul_routed_weight, eps)
    // sweep on MI355X (us; NT,ROUNDS): bs 16, 106 experts: 2,4 35.6 | 2,8 35.8 | 4,1 37.6 | 2,2 37.8 | 4,2 39.9 | 4,4 40.3;
    // bs 8: 2,1 22.2 | 2,2 22.9 | 4,1 23.4 | 2,4 25.0; bs 1: 2,1 6.7 | 2,4 10.0 -- the multi-round form pays once
    // its 4x fewer workgroups still fill the chip several times over
    const bool many = (int64_t)n_tiles * mbs > 50000;
    int cfg = many ? 24 : 21;  // NT*10 + ROUNDS
    debug_override(kOptMoeGemm2Cfg, cfg);
    switch (KB) {
        case 1: LAUNCH2Q(1, 2, 1); break;
        case 2:
            if (cfg == 42) LAUNCH2Q(2, 4, 2);
            else if (cfg == 44) LAUNCH2Q(2, 4, 4);
            else if (cfg == 22) LAUNCH2Q(2, 2, 2);
            else if (cfg == 24) LAUNCH2Q(2, 2, 4);
            else if (cfg == 28) LAUNCH2Q(2, 2, 8);
            else if (cfg == 21) LAUNCH2Q(2, 2, 1);
            else LAUNCH2Q(2, 4, 1);
            break;
        case 3: LAUNCH2Q(3, 2, 1); break;
        default: LAUNCH2Q(4, 2, 1); break;
    }
#undef LAUNCH2Q
    CHITU_RETURN_LAUNCH_STATUS();
}

extern "C" int chitu_hip_moe_sum(const void* c3_bf16, void* out_bf16, int64_t tokens, int32_t topk,
                                 int64_t N, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(c3_bf16 && out_bf16 && tokens >= 0 && topk >= 1 && N >= 8);
    if (N % 8 != 0) return CHITU_ERR_UNSUPPORTED;
    if (tokens == 0) return CHITU_OK;
    int64_t blocks = (tokens * (N / 8) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(moe_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)c3_bf16, (bf16_t*)out_bf16, tokens, (int)topk, N);
    CHITU_RETURN_LAUNCH_STATUS();
}
