// MoE router for decode on gfx950: skinny bf16 GEMM (scores) + one fused routing kernel.
//
// Replaces (reference, read-only) GateDeepSeekV3.forward, chitu/models/model_deepseek_v3.py:810-842,
// which runs as ~17 small torch launches per layer (F.linear, sigmoid, + bias, view, topk(2), sum,
// topk(groups), zeros_like, scatter_, mul, flatten, topk(k), gather, sum, div, mul, type_as):
//   scores = sigmoid(x W^T) (bf16) | softmax(x W^T) (fp32);  s' = scores + bias
//   group score = sum of the group's top-2 s' (bias present) or its max (no bias)
//   keep the top `topk_groups` groups, zero the rest; pick the top-k experts of s'
//   weights = scores[idx]  (/ their sum for sigmoid)  * route_scale  -> x.dtype
// Rounding points mirror torch's: every bf16 tensor op rounds once (sigmoid, +bias, top-2 sum,
// weight sum, division, scaling).  Ties are broken towards the LOWER index (torch.topk leaves tie
// order unspecified), which makes routing deterministic.
//
// The score GEMM is the same weight-streaming shape as fp8_gemm.hip with bf16 weights as the MFMA A
// operand; N = n_experts is tiny (16 row tiles), so K is split across workgroups and the fp32
// partials are summed, in order, by the routing kernel itself (no separate reduce launch).  The same
// GEMM entry point serves the LM head (N = vocab/tp, no split).
#include "common.h"
#include "gemm_common.h"
#include "moe_align_device.h"

namespace chitu {

// ---------------------------------------------------------------- skinny bf16 GEMM
// out[m][n] = sum_k x[m][k] * w[n][k]; k-block = 64 elements = one 128-B line of a weight row.
// Full-line layout (gemm_common.h): lane (j, g) loads 16 B of weight row n0 + 8*half + j/2 at element
// ((j%2)*4 + g)*8 of the block, so a wave-load takes whole lines from 8 rows; MFMA A-row j then holds
// K elements [0,32) (even j) or [32,64) (odd j) of weight row j/2, the block product is taken with the
// two activation fragments x[0..31], x[32..63] and the halves are added in-lane at the end.  bf16 has no
// per-block scales, so the four accumulators run across the whole K range.
template <int MT>
struct Bf16Stage {
    s16x8 w[2];
    s16x8 x[MT][2];
};

template <int MT, int WK, bool DEEP = false>
__global__ __launch_bounds__(64 * WK) void bf16_gemm_kernel(const bf16_t* __restrict__ X,
                                                            const bf16_t* __restrict__ W,
                                                            void* __restrict__ out, int out_dt,
                                                            float* __restrict__ partial, int M, int N,
                                                            int K, int S, int m_base) {
    __shared__ float red[WK > 1 ? WK * MT * 256 : 1];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: SGPR index math
    const int j = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int KB = K >> 6;
    const int T = S * WK;
    const int t = blockIdx.y * WK + wave;
    const int kb0 = (int)((long)KB * t / T), kb1 = (int)((long)KB * (t + 1) / T);
    const int eoff = ((j & 1) * 4 + g) * 8;
    const bf16_t* wp0 = W + (size_t)min(n0 + (j >> 1), N - 1) * K + eoff;
    const bf16_t* wp1 = W + (size_t)min(n0 + 8 + (j >> 1), N - 1) * K + eoff;
    const bf16_t* xp[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xp[mt] = X + (size_t)min(m_base + mt * 16 + j, M - 1) * K + g * 8;
    f32x4 e0[MT], o0[MT], e1[MT], o1[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) e0[mt] = o0[mt] = e1[mt] = o1[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto load = [&](Bf16Stage<MT>& st, int kb) {
        const int off = kb << 6;
        st.w[0] = __builtin_nontemporal_load(reinterpret_cast<const s16x8*>(wp0 + off));
        st.w[1] = __builtin_nontemporal_load(reinterpret_cast<const s16x8*>(wp1 + off));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            st.x[mt][0] = *reinterpret_cast<const s16x8*>(xp[mt] + off);
            st.x[mt][1] = *reinterpret_cast<const s16x8*>(xp[mt] + off + 32);
        }
    };
    auto compute = [&](const Bf16Stage<MT>& st) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            e0[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.w[0], st.x[mt][0], e0[mt], 0, 0, 0);
            o0[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.w[0], st.x[mt][1], o0[mt], 0, 0, 0);
            e1[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.w[1], st.x[mt][0], e1[mt], 0, 0, 0);
            o1[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.w[1], st.x[mt][1], o1[mt], 0, 0, 0);
        }
    };
    // DEEP (single token tile, <= 8 K blocks per wave): the wave's whole K range in one round trip
    // (a ring of 8 for the long-K form was measured: slower, Llama-3-8B bs 1 3.46 -> 3.50 ms/step)
    constexpr int D = DEEP ? 8 : MT >= 4 ? 2 : 4;
    Bf16Stage<MT> ring[D];
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
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        acc[mt] = f32x4{e0[mt][0] + o0[mt][1], e0[mt][2] + o0[mt][3], e1[mt][0] + o1[mt][1], e1[mt][2] + o1[mt][3]};
    gemm_epilogue_v2<MT, WK>(acc, red, out, out_dt, partial, M, N, S, m_base, n0);
}

// Gate/up projection of an unquantised SwiGLU MLP with SiluAndMul in the epilogue: W = [w1 | w3] rows
// (2*inter x K); a wave owns gate tile [n0, n0+16) and up tile [inter+n0, inter+n0+16), shares the
// activation fragments between them and writes h = bf16(bf16(silu(bf16(g))) * bf16(u)) -- FeedForward
// (models/model.py:212-214: F.silu(w1 x) * w3 x, every torch op rounding once) in one launch.
template <int MT>
struct Bf16Stage2 {
    s16x8 wg[2], wu[2];
    s16x8 x[MT][2];
};

template <int MT, int WK>
__global__ __launch_bounds__(64 * WK) void bf16_gemm_silu_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                                 bf16_t* __restrict__ out, int M, int inter, int K,
                                                                 int m_base) {
    __shared__ float red[WK > 1 ? WK * MT * 512 : 1];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: SGPR index math
    const int j = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int KB = K >> 6;
    const int kb0 = KB * wave / WK, kb1 = KB * (wave + 1) / WK;
    const int eoff = ((j & 1) * 4 + g) * 8;
    const int r0 = min(n0 + (j >> 1), inter - 1), r1 = min(n0 + 8 + (j >> 1), inter - 1);
    const bf16_t* gp0 = W + (size_t)r0 * K + eoff;
    const bf16_t* gp1 = W + (size_t)r1 * K + eoff;
    const bf16_t* up0 = W + (size_t)(inter + r0) * K + eoff;
    const bf16_t* up1 = W + (size_t)(inter + r1) * K + eoff;
    const bf16_t* xp[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xp[mt] = X + (size_t)min(m_base + mt * 16 + j, M - 1) * K + g * 8;
    f32x4 ge0[MT], go0[MT], ge1[MT], go1[MT], ue0[MT], uo0[MT], ue1[MT], uo1[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        ge0[mt] = go0[mt] = ge1[mt] = go1[mt] = ue0[mt] = uo0[mt] = ue1[mt] = uo1[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto load = [&](Bf16Stage2<MT>& st, int kb) {
        const int off = kb << 6;
        st.wg[0] = __builtin_nontemporal_load(reinterpret_cast<const s16x8*>(gp0 + off));
        st.wg[1] = __builtin_nontemporal_load(reinterpret_cast<const s16x8*>(gp1 + off));
        st.wu[0] = __builtin_nontemporal_load(reinterpret_cast<const s16x8*>(up0 + off));
        st.wu[1] = __builtin_nontemporal_load(reinterpret_cast<const s16x8*>(up1 + off));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            st.x[mt][0] = *reinterpret_cast<const s16x8*>(xp[mt] + off);
            st.x[mt][1] = *reinterpret_cast<const s16x8*>(xp[mt] + off + 32);
        }
    };
    auto compute = [&](const Bf16Stage2<MT>& st) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            ge0[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.wg[0], st.x[mt][0], ge0[mt], 0, 0, 0);
            go0[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.wg[0], st.x[mt][1], go0[mt], 0, 0, 0);
            ge1[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.wg[1], st.x[mt][0], ge1[mt], 0, 0, 0);
            go1[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.wg[1], st.x[mt][1], go1[mt], 0, 0, 0);
            ue0[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.wu[0], st.x[mt][0], ue0[mt], 0, 0, 0);
            uo0[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.wu[0], st.x[mt][1], uo0[mt], 0, 0, 0);
            ue1[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.wu[1], st.x[mt][0], ue1[mt], 0, 0, 0);
            uo1[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.wu[1], st.x[mt][1], uo1[mt], 0, 0, 0);
        }
    };
    constexpr int D = MT >= 2 ? 2 : 3;
    Bf16Stage2<MT> ring[D];
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
    f32x4 ag[MT], au[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        ag[mt] = f32x4{ge0[mt][0] + go0[mt][1], ge0[mt][2] + go0[mt][3], ge1[mt][0] + go1[mt][1], ge1[mt][2] + go1[mt][3]};
        au[mt] = f32x4{ue0[mt][0] + uo0[mt][1], ue0[mt][2] + uo0[mt][3], ue1[mt][0] + uo1[mt][1], ue1[mt][2] + uo1[mt][3]};
    }
    if (WK > 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            *reinterpret_cast<f32x4*>(&red[((wave * MT + mt) * 128 + lane) * 4]) = ag[mt];
            *reinterpret_cast<f32x4*>(&red[((wave * MT + mt) * 128 + 64 + lane) * 4]) = au[mt];
        }
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            ag[mt] = au[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < WK; ++w) {
                const f32x4 vg = *reinterpret_cast<const f32x4*>(&red[((w * MT + mt) * 128 + lane) * 4]);
                const f32x4 vu = *reinterpret_cast<const f32x4*>(&red[((w * MT + mt) * 128 + 64 + lane) * 4]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ag[mt][r] += vg[r];
                    au[mt][r] += vu[r];
                }
            }
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = m_base + mt * 16 + j;
        if (m >= M) continue;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int n = n0 + h2 * 8 + 2 * g;
            float hv[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float gv = round_bf16(ag[mt][2 * h2 + q]), uv = round_bf16(au[mt][2 * h2 + q]);
                hv[q] = round_bf16(gv / (1.0f + expf(-gv))) * uv;
            }
            bf16_t* dst = out + (size_t)m * inter + n;
            if (n + 1 < inter && (inter & 1) == 0) *reinterpret_cast<uint32_t*>(dst) = f32x2_to_bf16x2(hv[0], hv[1]);
            else {
                if (n < inter) dst[0] = f32_to_bf16(hv[0]);
                if (n + 1 < inter) dst[1] = f32_to_bf16(hv[1]);
            }
        }
    }
}

// ---------------------------------------------------------------- route + align in one launch
// moe_align_block_size needs every token's ids, and the routing launch has one workgroup per token:
// each workgroup publishes its ids (fence), takes a ticket, and the LAST one to arrive runs the sort
// for the whole batch (moe_align_workgroup) before the launch ends -- the pattern of a single-pass
// reduction, one launch less per MoE layer.  `ticket` is a zero-initialised device word that the
// last workgroup resets, so hipGraph replays and later launches need no memset (a memset NODE ahead of
// the launch was tried and faulted on replay after other graphs of the process had been destroyed --
// ROCm 7.2; the self-resetting word has no such dependency); launches that share a ticket must not
// overlap (one stream per device drives the step).
struct RouteAlign {
    int num_experts;          // experts the sort ranges over (routed + always-on slots); 0 = no align tail
    int block_size;
    int32_t* sorted_ids;
    int64_t sorted_cap;
    int32_t* expert_ids;
    int64_t expert_cap;
    int32_t* num_post_pad;
    int32_t* cumsum;
    const int32_t* expert_map;
    unsigned int* ticket;
    int small;                // 1: the one-workgroup launch sorts with moe_align_small_* (set by the launcher, never by callers)
};

// Called by EVERY thread of every routing workgroup after its ids are written.  The hand-off is the
// agent-scope release / ticket / acquire sequence of the CDNA4 guide (Guideline 16, counter form):
// every wave drains its stores, ONE lane releases at agent scope (the XCD's L2 write-back) and THEN
// takes the ticket; the last arriver's one lane acquires (drops its CU's stale L1 lines) before the
// workgroup reads the other workgroups' ids with plain loads.  Correct for any placement of the
// workgroups over CUs / XCDs.
__device__ __forceinline__ void route_align_tail(const RouteAlign& a, const int64_t* ids, int64_t numel, int* lds) {
    __shared__ int last_flag;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's id / weight stores have left the CU
    __syncthreads();
    if (gridDim.x == 1) {  // a single token (bs 1): this workgroup is the last one by construction
        moe_align_workgroup<int64_t>(ids, numel, a.num_experts, a.block_size, a.sorted_ids, a.sorted_cap, a.expert_ids,
                                     a.expert_cap, a.num_post_pad, a.cumsum, 1, a.expert_map, lds, (int)blockDim.x);
        return;
    }
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the write-back is complete before the ticket
        const unsigned int t = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == gridDim.x - 1) ? 1 : 0;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch starts from 0
        }
        last_flag = last;
    }
    __syncthreads();
    if (!last_flag) return;
    moe_align_workgroup<int64_t>(ids, numel, a.num_experts, a.block_size, a.sorted_ids, a.sorted_cap, a.expert_ids,
                                 a.expert_cap, a.num_post_pad, a.cumsum, 1, a.expert_map, lds, (int)blockDim.x);
}

// ---------------------------------------------------------------- fused routing
__device__ __forceinline__ float bf16r(float v) { return round_bf16(v); }

// One workgroup per token; thread e owns expert e (blockDim = E rounded up to 64, E <= 1024).
// logits: bf16 [M, E] (S == 0) or fp32 partials [S, M, E] to be summed here.
// SIGMOID=1: bf16 score pipeline (DeepSeek-V3); 0: fp32 softmax pipeline (DeepSeek-V2).
template <int SIGMOID>
__global__ __launch_bounds__(1024) void gate_route_kernel(
    const void* __restrict__ logits, int S, int M, int E, const bf16_t* __restrict__ bias, int n_groups,
    int topk_groups, int topk, float route_scale, bf16_t* __restrict__ out_w, int64_t* __restrict__ out_ids,
    int out_stride, int extra_id, float extra_w, int extra_n, int renorm, RouteAlign al) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sc = lds;             // [E] selection score s'
    float* red = lds + E;        // [32] wave partials
    float* gsc = lds + E + 32;   // [n_groups] group scores
    float* wsel = lds + E + 32 + 64;  // [topk] selected original scores
    const int t = blockIdx.x, e = threadIdx.x, lane = e & 63, wave = e >> 6;
    const int nw = blockDim.x >> 6;
    const bool act = e < E;

    float logit = -INFINITY;
    if (act) {
        if (S == 0) {
            logit = bf16_to_f32(((const bf16_t*)logits)[(int64_t)t * E + e]);
        } else {
            // 8 partial loads in flight at a time (a plain `for s < S` loop waits for each load)
            float a = 0.f;
            for (int s0 = 0; s0 < S; s0 += 8) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    v[i] = (s0 + i < S) ? ((const float*)logits)[((int64_t)(s0 + i) * M + t) * E + e] : 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) a += v[i];
            }
            logit = bf16r(a);  // F.linear output in bf16 (model_deepseek_v3.py:820)
        }
    }
    float orig;  // original_scores
    if (SIGMOID) {
        orig = act ? bf16r(1.0f / (1.0f + expf(-logit))) : 0.f;
    } else {
        float m = wave_reduce_max(logit);
        if (lane == 0) red[wave] = m;
        __syncthreads();
        m = red[0];
        for (int w = 1; w < nw; ++w) m = __builtin_fmaxf(m, red[w]);
        __syncthreads();
        const float ex = act ? expf(logit - m) : 0.f;
        float sum = wave_reduce_sum(ex);
        if (lane == 0) red[wave] = sum;
        __syncthreads();
        sum = 0.f;
        for (int w = 0; w < nw; ++w) sum += red[w];
        __syncthreads();
        orig = ex / sum;  // softmax(dim=-1, dtype=float32)
    }
    float sel = orig;
    if (bias && act) sel = SIGMOID ? bf16r(orig + bf16_to_f32(bias[e])) : orig + bf16_to_f32(bias[e]);
    if (act) sc[e] = sel;
    __syncthreads();

    if (n_groups > 1) {
        const int gs = E / n_groups;
        if (gs == 32 || gs == 64) {
            // group = half a wave (or a wave): top-2 by shuffles instead of a serial LDS scan
            const float v = act ? sel : -INFINITY;
            float m1 = v;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1)
                if (off < gs) m1 = __builtin_fmaxf(m1, __shfl_xor(m1, off, 64));
            const unsigned long long gmask = gs == 64 ? ~0ull : (0xffffffffull << (lane & 32));
            const int n_max = __popcll(__ballot(v == m1) & gmask);
            float m2 = v < m1 ? v : -INFINITY;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1)
                if (off < gs) m2 = __builtin_fmaxf(m2, __shfl_xor(m2, off, 64));
            if (n_max >= 2) m2 = m1;
            if (act && (e % gs) == 0) gsc[e / gs] = bias ? (SIGMOID ? bf16r(m1 + m2) : m1 + m2) : m1;
        } else if (e < n_groups) {
            float m1 = -INFINITY, m2 = -INFINITY;
            for (int i = 0; i < gs; ++i) {
                const float v = sc[e * gs + i];
                if (v > m1) { m2 = m1; m1 = v; }
                else if (v > m2) m2 = v;
            }
            // bias: topk(2).sum (bf16 add); no bias: amax  (model_deepseek_v3.py:827-831)
            gsc[e] = bias ? (SIGMOID ? bf16r(m1 + m2) : m1 + m2) : m1;
        }
        __syncthreads();
        if (act) {
            const int grp = e / gs;
            const float mine = gsc[grp];
            int rank = 0;
            for (int g2 = 0; g2 < n_groups; ++g2) {
                const float o = gsc[g2];
                rank += (o > mine) || (o == mine && g2 < grp);
            }
            if (rank >= topk_groups) sel = 0.f;  // scores * mask
        }
        __syncthreads();
        if (act) sc[e] = sel;
        __syncthreads();
    }
    int rank = 0;
    if (act) {
        // rank among all experts: wide broadcast LDS reads, 16 in flight (a scalar loop costs one
        // LDS round trip per expert: ~10 us for 256 experts)
        const int e4 = E & ~3;
#pragma unroll 16
        for (int i = 0; i < e4; i += 4) {
            const f32x4 o = *reinterpret_cast<const f32x4*>(&sc[i]);
            rank += (o[0] > sel) || (o[0] == sel && i < e);
            rank += (o[1] > sel) || (o[1] == sel && i + 1 < e);
            rank += (o[2] > sel) || (o[2] == sel && i + 2 < e);
            rank += (o[3] > sel) || (o[3] == sel && i + 3 < e);
        }
        for (int i = e4; i < E; ++i) {
            const float o = sc[i];
            rank += (o > sel) || (o == sel && i < e);
        }
        if (rank < topk) {
            out_ids[(int64_t)t * out_stride + rank] = e;
            wsel[rank] = orig;
        }
    }
    __syncthreads();
    if (e < topk) {
        float w = wsel[e];
        if (SIGMOID) {
            float sum = 0.f;
            for (int i = 0; i < topk; ++i) sum += wsel[i];
            w = bf16r(w / bf16r(sum));          // weights /= weights.sum(-1, keepdim=True)
            w = bf16r(w * route_scale);         // weights *= route_scale
        } else {
            if (renorm) {  // Mixtral: softmax -> top-k -> weights /= weights.sum() in fp32 (model_hf_mixtral.py:58-64)
                float sum = 0.f;
                for (int i = 0; i < topk; ++i) sum += wsel[i];
                w = w / sum;
            }
            w = w * route_scale;                // fp32, then type_as(x)
        }
        out_w[(int64_t)t * out_stride + e] = f32_to_bf16(w);
    }
    if (e < extra_n && extra_id >= 0) {  // always-on (shared) experts appended as slots topk .. topk+extra_n-1
        out_ids[(int64_t)t * out_stride + topk + e] = extra_id + e;
        out_w[(int64_t)t * out_stride + topk + e] = f32_to_bf16(extra_w);
    }
    if (al.num_experts > 0)  // uniform over the launch
        route_align_tail(al, out_ids, (int64_t)M * out_stride, reinterpret_cast<int*>(lds));
}

// ---------------------------------------------------------------- fused routing, fast path
// Same function as gate_route_kernel<1> (sigmoid scores, bf16 pipeline) for the shapes DeepSeek-V3/R1
// use: group size 32 or 64, at most 64 = waves x topk final candidates, <= 16 logit partials.
// Scores are bf16 values, so (score, lower-index-wins) packs into one unique 32-bit key
//   key = ordered16(score) << 16 | (0xffff - e)
// and every selection becomes integer maxima: group top-2 by DPP row reductions, the top-k by
// topk rounds of a wave-wide max per wave followed by one rank pass over the <= 64 wave winners --
// ~300 VALU ops and 2 barriers instead of a 256-way rank loop (~1400 ops) and 6 barriers.  All
// global loads (partials, bias) are issued up front, straight-line.
__device__ __forceinline__ uint32_t score_key(float v, int e) {
    const uint32_t b = __float_as_uint(v) >> 16;
    return ((b ^ ((b & 0x8000u) ? 0xffffu : 0x8000u)) << 16) | (0xffffu - (uint32_t)e);
}
__device__ __forceinline__ float key_score(uint32_t k) {
    const uint32_t o = k >> 16;
    return __uint_as_float((o ^ ((o & 0x8000u) ? 0x8000u : 0xffffu)) << 16);
}
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t row16_max_u32(uint32_t v) {
    v = max(v, dpp_u32<0x128>(v));
    v = max(v, dpp_u32<0x124>(v));
    v = max(v, dpp_u32<0x122>(v));
    v = max(v, dpp_u32<0x121>(v));
    return v;
}
// max over each aligned group of 8 lanes (DPP: xor 1, xor 2 inside the quads, then the half-row mirror brings the other quad)
__device__ __forceinline__ uint32_t group8_max_u32(uint32_t v) {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false));  // row_half_mirror
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32_uniform(uint32_t v) {
    v = row16_max_u32(v);
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return max(max(a, b), max(c, d));
}

template <int GS>  // experts per group: 32 or 64 (or 0 = ungrouped)
__global__ __launch_bounds__(1024) void gate_route_fast_kernel(
    const void* __restrict__ logits, int S, int M, int E, const bf16_t* __restrict__ bias, int n_groups,
    int topk_groups, int topk, float route_scale, bf16_t* __restrict__ out_w, int64_t* __restrict__ out_ids,
    int out_stride, int extra_id, float extra_w, int extra_n, RouteAlign al) {
    extern __shared__ __attribute__((aligned(16))) int align_lds[];  // only the align tail uses dynamic LDS
    __shared__ float orig_lds[1024];
    __shared__ float gsc[32];
    __shared__ __attribute__((aligned(16))) uint32_t cand[64];
    __shared__ float wsel[64];
    const int t = blockIdx.x, e = threadIdx.x, lane = e & 63, wave = e >> 6;
    const int nw = blockDim.x >> 6;
    const bool act = e < E;
    const int ec = act ? e : E - 1;
    // ---- all global loads up front
    float logit;
    if (S == 0) {
        logit = bf16_to_f32(((const bf16_t*)logits)[(int64_t)t * E + ec]);
    } else {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = ((const float*)logits)[((int64_t)min(i, S - 1) * M + t) * E + ec];
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) a += i < S ? v[i] : 0.f;
        logit = bf16r(a);  // F.linear output in bf16 (model_deepseek_v3.py:820)
    }
    const float bv = bias ? bf16_to_f32(bias[ec]) : 0.f;
    const float orig = bf16r(1.0f / (1.0f + expf(-logit)));  // original_scores
    float sel = bias ? bf16r(orig + bv) : orig;
    orig_lds[e] = orig;
    uint32_t key = act ? score_key(sel, e) : 0u;
    if (GS > 0) {
        // group score = sum of the group's top-2 (bias) or its max (no bias), :827-831
        uint32_t k1 = row16_max_u32(key);
        k1 = max(k1, (uint32_t)__shfl_xor((int)k1, 16, 64));
        if (GS == 64) k1 = max(k1, (uint32_t)__shfl_xor((int)k1, 32, 64));
        uint32_t k2 = row16_max_u32(key == k1 ? 0u : key);
        k2 = max(k2, (uint32_t)__shfl_xor((int)k2, 16, 64));
        if (GS == 64) k2 = max(k2, (uint32_t)__shfl_xor((int)k2, 32, 64));
        const float m1 = key_score(k1), m2 = key_score(k2);
        if (act && (e % GS) == 0) gsc[e / GS] = bias ? bf16r(m1 + m2) : m1;
        __syncthreads();
        const int grp = ec / GS;
        const float mine = gsc[grp];
        int rank = 0;
        for (int g2 = 0; g2 < n_groups; ++g2) {
            const float o = gsc[g2];
            rank += (o > mine) || (o == mine && g2 < grp);
        }
        if (rank >= topk_groups) sel = 0.f;  // scores * mask
        key = act ? score_key(sel, e) : 0u;
    }
    // ---- this wave's topk best keys, one per lane (lane r keeps the r-th)
    uint32_t mine_k = 0u;
    for (int r = 0; r < topk; ++r) {
        const uint32_t m = wave_max_u32_uniform(key);
        if (lane == r) mine_k = m;
        if (key == m) key = 0u;
    }
    if (lane < topk) cand[wave * topk + lane] = mine_k;
    __syncthreads();
    if (wave == 0) {
        // ---- rank the nw * topk (<= 64) wave winners; keys are unique, 0 = empty slot
        const int nc = nw * topk;
        const uint32_t ck = lane < nc ? cand[lane] : 0u;
        int rank = 0;
        for (int i = 0; i < nc; i += 4) {
            const i32x4 o = *reinterpret_cast<const i32x4*>(&cand[i]);
            rank += ((uint32_t)o[0] > ck) + ((uint32_t)o[1] > ck) + ((uint32_t)o[2] > ck) + ((uint32_t)o[3] > ck);
        }
        if (lane < nc && ck != 0u && rank < topk) {
            const int we = 0xffff - (int)(ck & 0xffffu);
            out_ids[(int64_t)t * out_stride + rank] = we;
            wsel[rank] = orig_lds[we];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // wave-local LDS hand-off (wsel)
        __builtin_amdgcn_wave_barrier();
        if (lane < topk) {
            float sum = 0.f;
            for (int i = 0; i < topk; ++i) sum += wsel[i];
            float w = bf16r(wsel[lane] / bf16r(sum));  // weights /= weights.sum(-1, keepdim=True)
            w = bf16r(w * route_scale);                // weights *= route_scale
            out_w[(int64_t)t * out_stride + lane] = f32_to_bf16(w);
        }
        if (lane < extra_n && extra_id >= 0) {  // always-on (shared) experts appended as slots topk .. topk+extra_n-1
            out_ids[(int64_t)t * out_stride + topk + lane] = extra_id + lane;
            out_w[(int64_t)t * out_stride + topk + lane] = f32_to_bf16(extra_w);
        }
    }
    if (al.num_experts > 0)  // uniform over the launch
        route_align_tail(al, out_ids, (int64_t)M * out_stride, align_lds);
}

// ---------------------------------------------------------------- routing + align, one workgroup
// Decode batches of <= 16 tokens on the DeepSeek-V3/R1 router shape (256 experts, sigmoid scores, groups
// of 32 or none): ONE workgroup, one wave per token, four experts per lane (expert e = 4*lane + i, so a
// group of 32 is 8 neighbouring lanes).  Everything a token needs stays inside its wave -- no
// workgroup barrier until the ids of all tokens sit in LDS -- and the same workgroup then sorts them
// (moe_align_workgroup reading the ids from LDS): no ticket, no global round trip between routing and
// sort.  Arithmetic, rounding points and tie rule are gate_route_fast_kernel's, so ids, weights and the
// align outputs are bit-identical to the separate launches.
template <int GS>  // experts per group: 32, or 0 = ungrouped
__global__ __launch_bounds__(1024) void gate_route_align_wg_kernel(
    const void* __restrict__ logits, int S, int M, const bf16_t* __restrict__ bias, int n_groups, int topk_groups,
    int topk, float route_scale, bf16_t* __restrict__ out_w, int64_t* __restrict__ out_ids, int out_stride,
    int extra_id, float extra_w, int extra_n, RouteAlign al) {
    constexpr int E = 256;
    extern __shared__ __attribute__((aligned(16))) int dyn_lds[];
    const int nwaves = blockDim.x >> 6;
    float* orig_lds = reinterpret_cast<float*>(dyn_lds);                      // [nwaves][E]
    int64_t* ids_lds = reinterpret_cast<int64_t*>(dyn_lds + nwaves * E);     // [M * out_stride]
    int* align_lds = dyn_lds + nwaves * E + 2 * ((M * out_stride + 1) & ~1);
    const int lane = threadIdx.x & 63, t = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // the wave's token: wave-uniform
    CHITU_PROBE_MARK(0);
    // The sort's order-free half rides in the routing (moe_align_device.h, moe_align_small_*): the token masks are zeroed
    // and the sentinels written NOW, while the logits are on their way; every token wave ORs its bit into the masks of the
    // experts it picks; after the routing barrier only one scan and the scatter are left.  (SMALL: launcher's choice.)
    const bool small_sort = al.small != 0;
    if (small_sort) {
        moe_align_small_init(al.num_experts, (int64_t)M * out_stride, al.sorted_ids, al.sorted_cap, al.expert_ids, al.expert_cap,
                             align_lds, (int)threadIdx.x, (int)blockDim.x);
        __syncthreads();
    }
    if (t < M) {  // wave-uniform
        // ---- logits of experts 4*lane .. 4*lane+3: all loads up front
        float lg[4];
        if (S == 0) {
            const i32x2 raw = *reinterpret_cast<const i32x2*>((const bf16_t*)logits + (int64_t)t * E + lane * 4);
            lg[0] = bf16_to_f32((bf16_t)((uint32_t)raw[0] & 0xffffu));
            lg[1] = bf16_to_f32((bf16_t)((uint32_t)raw[0] >> 16));
            lg[2] = bf16_to_f32((bf16_t)((uint32_t)raw[1] & 0xffffu));
            lg[3] = bf16_to_f32((bf16_t)((uint32_t)raw[1] >> 16));
        } else {
            f32x4 v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i)
                v[i] = *reinterpret_cast<const f32x4*>((const float*)logits + ((int64_t)min(i, S - 1) * M + t) * E + lane * 4);
            f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 16; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) a[c] += i < S ? v[i][c] : 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) lg[c] = bf16r(a[c]);  // F.linear output in bf16 (model_deepseek_v3.py:820)
        }
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
            const i32x2 raw = *reinterpret_cast<const i32x2*>(bias + lane * 4);
            bv[0] = bf16_to_f32((bf16_t)((uint32_t)raw[0] & 0xffffu));
            bv[1] = bf16_to_f32((bf16_t)((uint32_t)raw[0] >> 16));
            bv[2] = bf16_to_f32((bf16_t)((uint32_t)raw[1] & 0xffffu));
            bv[3] = bf16_to_f32((bf16_t)((uint32_t)raw[1] >> 16));
        }
        float orig[4], sel[4];
        uint32_t key[4];
        if (lg[0] == 12345.678f) CHITU_PROBE_MARK(9);  // (probe builds: forces the logits wait here)
        CHITU_PROBE_MARK(1);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            orig[c] = bf16r(1.0f / (1.0f + expf(-lg[c])));  // original_scores
            sel[c] = bias ? bf16r(orig[c] + bv[c]) : orig[c];
            key[c] = score_key(sel[c], lane * 4 + c);
        }
        *reinterpret_cast<f32x4*>(&orig_lds[t * E + lane * 4]) = f32x4{orig[0], orig[1], orig[2], orig[3]};
        if (key[0] == 0xdeadbeefu) CHITU_PROBE_MARK(9);
        CHITU_PROBE_MARK(5);
        if (GS == 32) {
            // group score = sum of the group's top-2 (bias) or its max (no bias), :827-831; group = 8 lanes
            // maxima over the group's 8 lanes on the VALU (DPP quad permutes + half-row mirror), not through LDS
            const uint32_t k1 = group8_max_u32(max(max(key[0], key[1]), max(key[2], key[3])));
            uint32_t k2 = 0u;
#pragma unroll
            for (int c = 0; c < 4; ++c) k2 = max(k2, key[c] == k1 ? 0u : key[c]);
            k2 = group8_max_u32(k2);
            const float m1 = key_score(k1), m2 = key_score(k2);
            const float mine = bias ? bf16r(m1 + m2) : m1;
            const int grp = lane >> 3;
            int rank = 0;
#pragma unroll
            for (int g2 = 0; g2 < 8; ++g2) {  // E = 256, GS = 32: exactly 8 groups; lane reads, not LDS permutes
                const float o = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), g2 * 8));
                rank += (g2 < n_groups) && ((o > mine) || (o == mine && g2 < grp));
            }
            if (rank >= topk_groups) {  // scores * mask
#pragma unroll
                for (int c = 0; c < 4; ++c) key[c] = score_key(0.f, lane * 4 + c);
            }
        }
        if (key[0] == 0xdeadbeefu) CHITU_PROBE_MARK(9);
        CHITU_PROBE_MARK(6);
        // ---- top-k: topk rounds of a wave-wide maximum; round r's winner is slot r (descending score)
        uint32_t mine_k = 0u;
        for (int r = 0; r < topk; ++r) {
            const uint32_t m = wave_max_u32_uniform(max(max(key[0], key[1]), max(key[2], key[3])));
            if (lane == r) mine_k = m;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (key[c] == m) key[c] = 0u;
        }
        if (mine_k == 0xdeadbeefu) CHITU_PROBE_MARK(9);
        CHITU_PROBE_MARK(7);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // wave-local LDS hand-off (orig_lds)
        __builtin_amdgcn_wave_barrier();
        const int we = 0xffff - (int)(mine_k & 0xffffu);
        const float ws = lane < topk ? orig_lds[t * E + (we & (E - 1))] : 0.f;
        float sum = 0.f;
        if (topk <= 16) {
#pragma unroll
            for (int i = 0; i < 16; ++i)  // same left-to-right sum, by lane reads
                if (i < topk) sum += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ws), i));
        } else {
            for (int i = 0; i < topk; ++i) sum += __shfl(ws, i, 64);
        }
        if (lane < topk) {
            float w = bf16r(ws / bf16r(sum));  // weights /= weights.sum(-1, keepdim=True)
            w = bf16r(w * route_scale);       // weights *= route_scale
            out_w[(int64_t)t * out_stride + lane] = f32_to_bf16(w);
            out_ids[(int64_t)t * out_stride + lane] = we;
            ids_lds[t * out_stride + lane] = we;
            if (small_sort) moe_align_small_mark(align_lds, we, t);
        }
        if (lane < extra_n && extra_id >= 0) {  // always-on (shared) experts appended as slots topk .. topk+extra_n-1
            out_ids[(int64_t)t * out_stride + topk + lane] = extra_id + lane;
            out_w[(int64_t)t * out_stride + topk + lane] = f32_to_bf16(extra_w);
            ids_lds[t * out_stride + topk + lane] = extra_id + lane;
            if (small_sort) moe_align_small_mark(align_lds, extra_id + lane, t);
        }
    }
    CHITU_PROBE_MARK(2);
    __syncthreads();
    CHITU_PROBE_MARK(3);
    // the sort needs a thread per expert, not a wave per token: the other waves leave (a finished wave no longer
    // counts at s_barrier), so the sort's barriers and per-wave histograms span 5 waves instead of up to 16
    const int sort_waves = min((int)(blockDim.x >> 6), max((al.num_experts + 63) >> 6, (M * out_stride + 63) >> 6));
    if (t >= sort_waves) return;
    if (small_sort) {
        moe_align_small_tail(ids_lds, M * out_stride, out_stride, al.num_experts, al.block_size, al.sorted_ids, al.sorted_cap,
                             al.expert_ids, al.expert_cap, al.num_post_pad, al.cumsum, al.expert_map, align_lds, sort_waves * 64);
        CHITU_PROBE_MARK(4);
        return;
    }
    moe_align_workgroup<int64_t>(ids_lds, (int64_t)M * out_stride, al.num_experts, al.block_size, al.sorted_ids,
                                 al.sorted_cap, al.expert_ids, al.expert_cap, al.num_post_pad, al.cumsum, 1,
                                 al.expert_map, align_lds, sort_waves * 64);
    CHITU_PROBE_MARK(4);
}

// ---------------------------------------------------------------- routing + align, one workgroup, softmax routers
// The same one-workgroup form for the softmax routers with at most 64 experts (DeepSeek-V2-Lite: 64 experts, top-6, two
// always-on slots; Mixtral: 8 experts, top-2, renormalised): a wave per token, ONE expert per lane, fp32 scores.
// gate_route_kernel<0>'s arithmetic exactly -- logits summed over the split-K planes in plane order and rounded to bf16,
// softmax in fp32 with the wave's butterfly sum, selection by (score descending, index ascending), weights =
// softmax scores [/ their sum in rank order] * route_scale -- so ids, weights and the align outputs are bit-identical
// to the separate launches (one workgroup per token + ticket + sort by the last one: 12 us at bs 16 on the V2-Lite
// step, most of it the ticket's release / acquire pair and a two-wave sort).
__device__ __forceinline__ uint32_t ordered_u32(float v) {
    const uint32_t b = __float_as_uint(v);
    return b ^ ((b & 0x80000000u) ? 0xffffffffu : 0x80000000u);
}

__global__ __launch_bounds__(1024) void gate_route_align_wg_softmax_kernel(
    const void* __restrict__ logits, int S, int M, int E, const bf16_t* __restrict__ bias, int topk, float route_scale,
    bf16_t* __restrict__ out_w, int64_t* __restrict__ out_ids, int out_stride, int extra_id, float extra_w, int extra_n,
    int renorm, RouteAlign al) {
    extern __shared__ __attribute__((aligned(16))) int dyn_lds[];
    int64_t* ids_lds = reinterpret_cast<int64_t*>(dyn_lds);                  // [M * out_stride]
    int* align_lds = dyn_lds + 2 * ((M * out_stride + 1) & ~1);
    const int lane = threadIdx.x & 63, t = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // the wave's token
    const bool small_sort = al.small != 0;
    if (small_sort) {
        moe_align_small_init(al.num_experts, (int64_t)M * out_stride, al.sorted_ids, al.sorted_cap, al.expert_ids, al.expert_cap,
                             align_lds, (int)threadIdx.x, (int)blockDim.x);
        __syncthreads();
    }
    if (t < M) {  // wave-uniform
        const bool act = lane < E;
        const int ec = act ? lane : E - 1;
        float logit;
        if (S == 0) {
            logit = bf16_to_f32(((const bf16_t*)logits)[(int64_t)t * E + ec]);
        } else {
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = ((const float*)logits)[((int64_t)min(i, S - 1) * M + t) * E + ec];
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) a += i < S ? v[i] : 0.f;  // plane order, as gate_route_kernel
            logit = bf16r(a);
        }
        const float bv = bias ? bf16_to_f32(bias[ec]) : 0.f;
        if (!act) logit = -INFINITY;
        const float m = wave_reduce_max(logit);
        const float ex = act ? expf(logit - m) : 0.f;
        const float sum = wave_reduce_sum(ex);
        const float orig = ex / sum;  // softmax(dim=-1, dtype=float32)
        const float sel = bias ? orig + bv : orig;
        uint32_t key = act ? ordered_u32(sel) : 0u;
        // ---- top-k: round r's winner (greatest score, lowest index among equals) becomes slot r, kept by lane r
        int my_e = 0;
        float my_w = 0.f;
        for (int r = 0; r < topk; ++r) {
            const uint32_t mk = wave_max_u32_uniform(key);
            const unsigned long long hit = __ballot(act && key == mk);
            const int win = __builtin_amdgcn_readfirstlane(__ffsll((long long)hit) - 1);
            const float w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(orig), win));
            if (lane == r) {
                my_e = win;
                my_w = w;
            }
            if (lane == win) key = 0u;  // below every real key: ordered_u32 of a finite or infinite float is never 0
        }
        float w = my_w;
        if (renorm) {  // Mixtral: weights /= weights.sum() in fp32, rank order
            float s2 = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (i < topk) s2 += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), i));
            w = w / s2;
        }
        w = w * route_scale;  // fp32, then type_as(x)
        if (lane < topk) {
            out_w[(int64_t)t * out_stride + lane] = f32_to_bf16(w);
            out_ids[(int64_t)t * out_stride + lane] = my_e;
            ids_lds[t * out_stride + lane] = my_e;
            if (small_sort) moe_align_small_mark(align_lds, my_e, t);
        }
        if (lane < extra_n && extra_id >= 0) {
            out_ids[(int64_t)t * out_stride + topk + lane] = extra_id + lane;
            out_w[(int64_t)t * out_stride + topk + lane] = f32_to_bf16(extra_w);
            ids_lds[t * out_stride + topk + lane] = extra_id + lane;
            if (small_sort) moe_align_small_mark(align_lds, extra_id + lane, t);
        }
    }
    __syncthreads();
    const int sort_waves = min((int)(blockDim.x >> 6), max((al.num_experts + 63) >> 6, (M * out_stride + 63) >> 6));
    if (t >= sort_waves) return;
    if (small_sort) {
        moe_align_small_tail(ids_lds, M * out_stride, out_stride, al.num_experts, al.block_size, al.sorted_ids, al.sorted_cap,
                             al.expert_ids, al.expert_cap, al.num_post_pad, al.cumsum, al.expert_map, align_lds, sort_waves * 64);
        return;
    }
    moe_align_workgroup<int64_t>(ids_lds, (int64_t)M * out_stride, al.num_experts, al.block_size, al.sorted_ids,
                                 al.sorted_cap, al.expert_ids, al.expert_cap, al.num_post_pad, al.cumsum, 1,
                                 al.expert_map, align_lds, sort_waves * 64);
}

}  // namespace chitu

extern "C" int chitu_hip_bf16_gemm(const void* x_bf16, const void* w_bf16, void* out, int out_dtype,
                                   int64_t M, int64_t N, int64_t K, int32_t num_splits,
                                   float* partials, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(x_bf16 && w_bf16 && M >= 0 && N >= 1 && K >= 64 && N < (1 << 30) && K < (1 << 30));
    CHITU_REQUIRE(num_splits >= 1 && num_splits <= 64);
    CHITU_REQUIRE(num_splits > 1 ? partials != nullptr : (out != nullptr && out_dtype >= 0 && out_dtype <= 2));
    if (K % 64 != 0) return CHITU_ERR_UNSUPPORTED;
    if (M == 0) return CHITU_OK;
    hipStream_t st = (hipStream_t)stream;
    const int KB = (int)(K / 64);
    const int tiles = (int)((N + 15) / 16);
    if (num_splits > KB) return CHITU_ERR_BAD_ARG;
    if ((M >= 256 || (M >= 128 && N >= 1024)) && K < (1 << 23) && debug_option(kOptBf16GemmTiled) != 0) {  // (32-bit byte offsets inside a tile)
        // prefill-sized M: a GEMM, not a weight stream (bf16_gemm_tiled.hip); a 128-row prompt against the 256-row router
        // matrix would be two workgroups -- that one stays with the split-K stream.  num_splits > 1: the K range cut over
        // that many workgroups per tile, fp32 planes left in `partials` (the router of a long prompt: few tiles, long K)
        launch_bf16_gemm_tiled((const bf16_t*)x_bf16, (const bf16_t*)w_bf16, out, out_dtype, M, N, K, num_splits, partials, st);
        CHITU_RETURN_LAUNCH_STATUS();
    }
    const int S = num_splits;
    int WK = 8;
    while (WK > 1 && (WK * S > KB || (int64_t)tiles * S * WK > 4096)) WK >>= 1;
    debug_override(kOptBf16GemmWK, WK);
    while (WK > 1 && WK * S > KB) WK >>= 1;
    const dim3 grid((unsigned)tiles, (unsigned)S);
#define LAUNCH(MT, WKV)                                                                            \
    hipLaunchKernelGGL((bf16_gemm_kernel<MT, WKV>), grid, dim3(64 * WKV), 0, st, (const bf16_t*)x_bf16, \
                       (const bf16_t*)w_bf16, out, out_dtype, partials, (int)M, (int)N, (int)K, S, mbase)
#define LAUNCH_WK(MT)                      \
    switch (WK) {                          \
        case 8: LAUNCH(MT, 8); break;      \
        case 4: LAUNCH(MT, 4); break;      \
        case 2: LAUNCH(MT, 2); break;      \
        default: LAUNCH(MT, 1); break;     \
    }
    const int per_wave = KB / (WK * S);
    for (int64_t mb = 0; mb < M; mb += 32) {
        const int mbase = (int)mb;
        if (M - mb <= 16) {
            // the 8-deep ring (the wave's whole K range in one round trip) for grids of at most one workgroup per CU: its
            // register count admits one 8-wave workgroup per CU, so a larger grid would run in rounds (Llama-3-8B's qkv
            // projection, 384 workgroups; see bf16_norm_gemm.hip).  Option 11: 0 never, 1 always, -1 this heuristic.
            const int deep_opt = debug_option(kOptBf16GemmDeep);
            if (WK == 8 && per_wave > 4 && per_wave <= 8 && (deep_opt < 0 ? (int64_t)tiles * S <= 256 : deep_opt != 0))
                hipLaunchKernelGGL((bf16_gemm_kernel<1, 8, true>), grid, dim3(512), 0, st, (const bf16_t*)x_bf16,
                                   (const bf16_t*)w_bf16, out, out_dtype, partials, (int)M, (int)N, (int)K, S, mbase);
            else
                LAUNCH_WK(1)
        } else {
            LAUNCH_WK(2)
        }
    }
#undef LAUNCH_WK
#undef LAUNCH
    // num_splits > 1: partials [S][M][N] are left for the consumer (chitu_hip_gate_route) to sum.
    CHITU_RETURN_LAUNCH_STATUS();
}

static int gate_route_launch(const void* logits, int32_t num_partials, int64_t tokens, int32_t num_experts,
                             const void* bias_bf16, int32_t n_groups, int32_t topk_groups, int32_t topk,
                             int32_t score_func, float route_scale, void* out_weights_bf16, int64_t* out_ids,
                             int32_t out_stride, int32_t extra_expert_id, float extra_weight,
                             int32_t extra_count, const chitu::RouteAlign& al, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(logits && out_weights_bf16 && out_ids && tokens >= 0);
    CHITU_REQUIRE(num_experts >= 1 && num_experts <= 1024 && topk >= 1 && topk <= num_experts && topk <= 64);
    CHITU_REQUIRE(n_groups >= 1 && n_groups <= 64 && num_experts % n_groups == 0);
    CHITU_REQUIRE(topk_groups >= 1 && topk_groups <= n_groups && num_partials >= 0);
    CHITU_REQUIRE(extra_count >= 0 && extra_count <= 32 && out_stride >= topk + (extra_expert_id >= 0 ? extra_count : 0));
    CHITU_REQUIRE(score_func >= 0 && score_func <= 2);  // 0 softmax, 1 sigmoid (V3), 2 softmax + top-k renormalisation (Mixtral)
    if (tokens == 0) return CHITU_OK;
    // the align tail needs one thread per sorted-over expert (routed + always-on slots)
    const int threads = ((max(num_experts, al.num_experts) + 63) / 64) * 64;
    CHITU_REQUIRE(threads <= 1024);
    const size_t align_lds = al.num_experts > 0 ? sizeof(int) * moe_align_lds_ints(al.num_experts, threads) : 0;
    const size_t lds = max(sizeof(float) * (size_t)(num_experts + 32 + 64 + 64), align_lds);
    CHITU_REQUIRE(lds <= 60 * 1024);
    hipStream_t st = (hipStream_t)stream;
    const int gs = n_groups > 1 ? num_experts / n_groups : 0;
    const bool fast = score_func == 1 && (gs == 0 || gs == 32 || gs == 64) && num_partials <= 16 &&
                      (threads / 64) * topk <= 64 && ((threads / 64) * topk) % 4 == 0 && n_groups <= 32 &&
                      debug_option(kOptGateGeneric) <= 0;
    if (fast && al.num_experts > 0 && num_experts == 256 && (gs == 0 || gs == 32) && tokens <= 16 &&
        extra_count <= 32 && debug_option(kOptGateTicket) <= 0) {
        // one workgroup: a wave per token, then the sort (needs a thread per sorted-over expert)
        const int wg_threads = 64 * max((int)tokens, (max(num_experts, al.num_experts) + 63) / 64);
        const size_t ids_ints = 2 * (((size_t)tokens * out_stride + 1) & ~(size_t)1);
        const size_t wg_lds = sizeof(int) * ((size_t)(wg_threads / 64) * 256 + ids_ints +
                                             moe_align_lds_ints(al.num_experts, wg_threads));
        CHITU_REQUIRE(wg_threads <= 1024 && wg_lds <= 64 * 1024);
        // ids distinct within a token (top-k of distinct routed experts + always-on ids past them): the sort's order-free
        // half rides in the routing (moe_align_small_*); option 15 = 0 keeps the general sort (equivalence tests, A/B)
        RouteAlign alw = al;
        // (every id the kernel can emit must index the sort's tables: routed ids < num_experts <= al.num_experts, always-on
        // ids inside [num_experts, al.num_experts) -- otherwise the general sort, which drops out-of-range ids, is used)
        alw.small = (num_experts <= al.num_experts &&
                     (extra_expert_id < 0 || (extra_expert_id >= num_experts && extra_expert_id + extra_count <= al.num_experts)) &&
                     debug_option(kOptGateSmallSort) != 0) ? 1 : 0;
#define LAUNCHW(GSV)                                                                                          \
    hipLaunchKernelGGL(gate_route_align_wg_kernel<GSV>, dim3(1), dim3(wg_threads), wg_lds, st, logits,         \
                       (int)num_partials, (int)tokens, (const bf16_t*)bias_bf16, (int)n_groups, (int)topk_groups, \
                       (int)topk, route_scale, (bf16_t*)out_weights_bf16, out_ids, (int)out_stride,           \
                       (int)extra_expert_id, extra_weight, (int)extra_count, alw)
        if (gs == 32) LAUNCHW(32);
        else LAUNCHW(0);
#undef LAUNCHW
        CHITU_RETURN_LAUNCH_STATUS();
    }
    if (!fast && al.num_experts > 0 && num_experts <= 64 && (score_func == 0 || score_func == 2) && n_groups <= 1 &&
        tokens <= 16 && num_partials <= 16 && topk <= 16 && extra_count <= 32 && debug_option(kOptGateTicket) <= 0 &&
        debug_option(kOptGateGeneric) <= 0) {
        // softmax routers with <= 64 experts (V2-Lite, Mixtral), decode batches: one workgroup, a wave per token, then the sort
        const int wg_threads = 64 * max((int)tokens, (max(num_experts, al.num_experts) + 63) / 64);
        const size_t ids_ints = 2 * (((size_t)tokens * out_stride + 1) & ~(size_t)1);
        const size_t wg_lds = sizeof(int) * (ids_ints + moe_align_lds_ints(al.num_experts, wg_threads));
        CHITU_REQUIRE(wg_threads <= 1024 && wg_lds <= 64 * 1024);
        RouteAlign alw = al;
        // (every id the kernel can emit must index the sort's tables: routed ids < num_experts <= al.num_experts, always-on
        // ids inside [num_experts, al.num_experts) -- otherwise the general sort, which drops out-of-range ids, is used)
        alw.small = (num_experts <= al.num_experts &&
                     (extra_expert_id < 0 || (extra_expert_id >= num_experts && extra_expert_id + extra_count <= al.num_experts)) &&
                     debug_option(kOptGateSmallSort) != 0) ? 1 : 0;
        hipLaunchKernelGGL(gate_route_align_wg_softmax_kernel, dim3(1), dim3(wg_threads), wg_lds, st, logits, (int)num_partials,
                           (int)tokens, (int)num_experts, (const bf16_t*)bias_bf16, (int)topk, route_scale,
                           (bf16_t*)out_weights_bf16, out_ids, (int)out_stride, (int)extra_expert_id, extra_weight,
                           (int)extra_count, score_func == 2 ? 1 : 0, alw);
        CHITU_RETURN_LAUNCH_STATUS();
    }
    if (fast) {
#define LAUNCHF(GSV)                                                                                         \
    hipLaunchKernelGGL(gate_route_fast_kernel<GSV>, dim3((unsigned)tokens), dim3(threads), align_lds, st, logits, \
                       (int)num_partials, (int)tokens, (int)num_experts, (const bf16_t*)bias_bf16, (int)n_groups, \
                       (int)topk_groups, (int)topk, route_scale, (bf16_t*)out_weights_bf16, out_ids,         \
                       (int)out_stride, (int)extra_expert_id, extra_weight, (int)extra_count, al)
        if (gs == 32) LAUNCHF(32);
        else if (gs == 64) LAUNCHF(64);
        else LAUNCHF(0);
#undef LAUNCHF
        CHITU_RETURN_LAUNCH_STATUS();
    }
    if (lds > 48 * 1024) {
        (void)hipFuncSetAttribute((const void*)gate_route_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)gate_route_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (score_func == 1)
        hipLaunchKernelGGL(gate_route_kernel<1>, dim3((unsigned)tokens), dim3(threads), lds, st, logits,
                           (int)num_partials, (int)tokens, (int)num_experts, (const bf16_t*)bias_bf16,
                           (int)n_groups, (int)topk_groups, (int)topk, route_scale, (bf16_t*)out_weights_bf16,
                           out_ids, (int)out_stride, (int)extra_expert_id, extra_weight, (int)extra_count, 0, al);
    else
        hipLaunchKernelGGL(gate_route_kernel<0>, dim3((unsigned)tokens), dim3(threads), lds, st, logits,
                           (int)num_partials, (int)tokens, (int)num_experts, (const bf16_t*)bias_bf16,
                           (int)n_groups, (int)topk_groups, (int)topk, route_scale, (bf16_t*)out_weights_bf16,
                           out_ids, (int)out_stride, (int)extra_expert_id, extra_weight, (int)extra_count,
                           score_func == 2 ? 1 : 0, al);
    CHITU_RETURN_LAUNCH_STATUS();
}

extern "C" int chitu_hip_gate_route(const void* logits, int32_t num_partials, int64_t tokens,
                                    int32_t num_experts, const void* bias_bf16, int32_t n_groups,
                                    int32_t topk_groups, int32_t topk, int32_t score_func,
                                    float route_scale, void* out_weights_bf16, int64_t* out_ids,
                                    int32_t out_stride, int32_t extra_expert_id, float extra_weight,
                                    int32_t extra_count, void* stream) {
    chitu::RouteAlign none{};
    return gate_route_launch(logits, num_partials, tokens, num_experts, bias_bf16, n_groups, topk_groups, topk,
                             score_func, route_scale, out_weights_bf16, out_ids, out_stride, extra_expert_id,
                             extra_weight, extra_count, none, stream);
}

extern "C" int chitu_hip_gate_route_align(const void* logits, int32_t num_partials, int64_t tokens,
                                          int32_t num_experts, const void* bias_bf16, int32_t n_groups,
                                          int32_t topk_groups, int32_t topk, int32_t score_func,
                                          float route_scale, void* out_weights_bf16, int64_t* out_ids,
                                          int32_t out_stride, int32_t extra_expert_id, float extra_weight,
                                          int32_t extra_count, int32_t align_num_experts, int32_t align_block_size,
                                          int32_t* sorted_token_ids, int64_t sorted_cap, int32_t* expert_ids,
                                          int64_t expert_ids_cap, int32_t* num_tokens_post_pad, int32_t* cumsum,
                                          const int32_t* expert_map, uint32_t* ticket, void* stream) {
    CHITU_REQUIRE(sorted_token_ids && expert_ids && num_tokens_post_pad && cumsum && ticket);
    CHITU_REQUIRE(align_num_experts >= 1 && align_num_experts <= 1024 && align_block_size >= 1);
    CHITU_REQUIRE(sorted_cap >= 0 && expert_ids_cap >= 0 && tokens * (int64_t)out_stride < (1ll << 31));
    // the sort runs over out_ids as a dense [tokens * out_stride] array: every column must be written
    CHITU_REQUIRE(out_stride == topk + (extra_expert_id >= 0 ? extra_count : 0));
    chitu::RouteAlign al{align_num_experts, align_block_size, sorted_token_ids, sorted_cap, expert_ids, expert_ids_cap,
                         num_tokens_post_pad, cumsum, expert_map, ticket, 0};
    return gate_route_launch(logits, num_partials, tokens, num_experts, bias_bf16, n_groups, topk_groups, topk,
                             score_func, route_scale, out_weights_bf16, out_ids, out_stride, extra_expert_id,
                             extra_weight, extra_count, al, stream);
}

extern "C" int chitu_hip_bf16_gemm_silu(const void* x_bf16, const void* w13_bf16, void* out_bf16, int64_t M,
                                        int64_t inter, int64_t K, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(x_bf16 && w13_bf16 && out_bf16 && M >= 0 && inter >= 1 && K >= 64 && inter < (1 << 29) && K < (1 << 30));
    if (K % 64 != 0) return CHITU_ERR_UNSUPPORTED;
    if (M == 0) return CHITU_OK;
    hipStream_t st = (hipStream_t)stream;
    const int KB = (int)(K / 64);
    const int tiles = (int)((inter + 15) / 16);
    int WK = 8;
    while (WK > 1 && (WK > KB || (int64_t)tiles * WK > 4096)) WK >>= 1;
    debug_override(kOptBf16SiluWK, WK);
    while (WK > 1 && WK > KB) WK >>= 1;
    const dim3 grid((unsigned)tiles);
#define LAUNCHS(MT, WKV)                                                                                        \
    hipLaunchKernelGGL((bf16_gemm_silu_kernel<MT, WKV>), grid, dim3(64 * WKV), 0, st, (const bf16_t*)x_bf16,    \
                       (const bf16_t*)w13_bf16, (bf16_t*)out_bf16, (int)M, (int)inter, (int)K, mbase)
#define LAUNCHS_WK(MT)                   \
    switch (WK) {                        \
        case 8: LAUNCHS(MT, 8); break;   \
        case 4: LAUNCHS(MT, 4); break;   \
        case 2: LAUNCHS(MT, 2); break;   \
        default: LAUNCHS(MT, 1); break;  \
    }
    for (int64_t mb = 0; mb < M; mb += 32) {
        const int mbase = (int)mb;
        if (M - mb <= 16) { LAUNCHS_WK(1) } else { LAUNCHS_WK(2) }
    }
#undef LAUNCHS_WK
#undef LAUNCHS
    CHITU_RETURN_LAUNCH_STATUS();
}

CHITU_PROBE_READER(gate)
