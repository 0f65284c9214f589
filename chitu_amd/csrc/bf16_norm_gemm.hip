// Small-batch decode of an unquantised (bf16) model: residual add + RMSNorm as the PROLOGUE of the skinny GEMM that
// consumes the normalised row -- the qkv projection behind attn_norm, the gate/up projection (+ SiLU-and-mul)
// behind ffn_norm -- so a decoder layer loses its two stand-alone norm launches.
//
// Replaces (reference, read-only):
//   chitu/models/model.py:29-78     RMSNorm.forward (F.rms_norm on the residual stream)
//   chitu/models/model.py:167-198   Attention.decode_forward_paged's wqkv projection
//   chitu/models/model.py:201-214   FeedForward: F.silu(w1 x) * w3 x
//   chitu/models/model.py:246-251   TransformerBlock: h = x + attention(attention_norm(x)); out = h + ffn(ffn_norm(h))
// and, in this library, chitu_hip_rmsnorm(add=...) followed by chitu_hip_bf16_gemm / chitu_hip_bf16_gemm_silu.
//
// Why: at batch 1-4 a norm launch is ~4.6 us of launch + two cold round trips for 8-32 KB of data (Llama-3-8B bs 1:
// 64 of them = 0.29 ms of a 3.45 ms step).  Here every workgroup of the GEMM redoes the add + norm of the <= 4 rows
// while its first weight tiles are already on their way from HBM: 1024 "virtual threads" (64 * WK real ones, each
// taking 1024 / (64 * WK) 8-element chunks) run exactly the arithmetic of rmsnorm_add_kernel (norm_common.h wide
// form: per-chunk sequential sum, wave butterfly, the 16 wave sums in order), the normalised rows go to LDS as
// bf16, and the K loop takes its activation fragments from there (all 16 MFMA columns of a token read one address:
// a broadcast).  Workgroup 0 also writes the new residual stream.  Same K split, same MFMA order as
// bf16_gemm_kernel / bf16_gemm_silu_kernel (gate.hip): results are BIT-IDENTICAL to the unfused launches.
#include "common.h"
#include "gemm_common.h"
#include "norm_common.h"

namespace chitu {

constexpr int kFusedNormMaxRows = 4;

// MR = rows held in registers (1, 2 or 4: the register budget of the prologue is what limits how many weight tiles
// the K loop can keep in flight)
template <int WK, int MR>
struct AddNormRegs {
    static constexpr int T = 64 * WK, NCH = kNormWideThreads / T;
    i32x4 x[MR][NCH], a[MR][NCH], a2[MR][NCH], w[NCH];  // a2: the second residual term (round 6; loaded only when there is one)
};

// every load of the prologue, issued before the caller's first weight loads (loads return in order: the norm then
// never waits for a weight tile)
template <int WK, int MR>
__device__ __forceinline__ void add_norm_issue(AddNormRegs<WK, MR>& r, const bf16_t* x, int64_t x_stride, const bf16_t* add,
                                               int64_t add_stride, const bf16_t* __restrict__ nw, int M, int n_chunks,
                                               int64_t term2_off = 0) {
    constexpr int T = AddNormRegs<WK, MR>::T, NCH = AddNormRegs<WK, MR>::NCH;
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        const int row = min(m, M - 1);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = min((int)threadIdx.x + i * T, n_chunks - 1);
            r.x[m][i] = *reinterpret_cast<const i32x4*>(x + (int64_t)row * x_stride + c * 8);
            r.a[m][i] = *reinterpret_cast<const i32x4*>(add + (int64_t)row * add_stride + c * 8);
            // two residual terms (the int8 / bf16 MoE's un-summed top-2 outputs, fused_experts(reduce_topk=False)): the second row
            if (term2_off != 0) r.a2[m][i] = *reinterpret_cast<const i32x4*>(add + (int64_t)row * add_stride + term2_off + c * 8);
        }
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = min((int)threadIdx.x + i * T, n_chunks - 1);
        r.w[i] = *reinterpret_cast<const i32x4*>(nw + c * 8);
    }
}

// What else the prologue's owner workgroup (the one that writes the residual stream) can leave behind for launches that are
// NOT this GEMM: the normalised rows (bf16) and / or their fp8 form -- quant_mode 1 = act_quant (no eps), 2 =
// per_token_group_quant (eps, clamp): chitu_hip_rmsnorm's arithmetic on the rounded rows, 16 lanes = one 128-wide group.
struct NormOut {
    bf16_t* y;   // [M, K] or null
    fp8_t* q;    // [M, K] codes or null
    float* qs;   // [M, K / 128]
    float qeps;
    int qmode;   // 0 | 1 | 2
};

// x_new = bf16(x + add) (written to sum_out by the workgroup told to), y = bf16((x_new * rr) * w) -> ybuf [M][K].
// nred: [kFusedNormMaxRows][16] floats.  Ends with a workgroup barrier: ybuf is readable.
template <int WK, int MR, bool POUT = false>
__device__ __forceinline__ void add_norm_finish(AddNormRegs<WK, MR>& r, bf16_t* sum_out, int64_t sum_stride, bool write_sum,
                                                int M, int K, float eps, bf16_t* ybuf, float* nred, const NormOut* po = nullptr,
                                                bool two_terms = false) {
    constexpr int T = AddNormRegs<WK, MR>::T, NCH = AddNormRegs<WK, MR>::NCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_chunks = K >> 3;
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        if (m < M) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int vt = tid + i * T;  // the thread of rmsnorm_add_kernel this chunk belongs to
                const bool act = vt < n_chunks;
                float v[8];
                i32x4 sraw;
                if (two_terms) {  // (workgroup-uniform) add = bf16(float(t0) + float(t1)): chitu_hip_moe_sum's arithmetic for two terms
                    const i32x4 both[2] = {r.a[m][i], r.a2[m][i]};
                    r.a[m][i] = sum_terms_bf16x8<2>(both, 2);
                }
                add_bf16x8(r.x[m][i], r.a[m][i], v, sraw);
                r.x[m][i] = sraw;
                if (write_sum && act) *reinterpret_cast<i32x4*>(sum_out + (int64_t)m * sum_stride + vt * 8) = sraw;
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
                if (vt < n_chunks) {
                    float v[8];
                    unpack_bf16x8(r.x[m][i], v);
                    i32x4 o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t u = (uint32_t)r.w[i][k];
                        o[k] = (int)f32x2_to_bf16x2((v[2 * k] * rr) * __uint_as_float(u << 16),
                                                    (v[2 * k + 1] * rr) * __uint_as_float(u & 0xffff0000u));
                    }
                    *reinterpret_cast<i32x4*>(ybuf + (size_t)m * K + vt * 8) = o;
                    if (POUT && write_sum && po->y) *reinterpret_cast<i32x4*>(po->y + (size_t)m * K + vt * 8) = o;
                }
                if (POUT && write_sum && po->qmode != 0) {  // workgroup-uniform; K % 128 == 0: a 16-lane group is all in or all out
                    const bool act = vt < n_chunks;
                    float ov[8];
                    {
                        float v[8];
                        unpack_bf16x8(r.x[m][i], v);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const uint32_t u = (uint32_t)r.w[i][k];
                            const uint32_t h2 = f32x2_to_bf16x2((v[2 * k] * rr) * __uint_as_float(u << 16),
                                                                (v[2 * k + 1] * rr) * __uint_as_float(u & 0xffff0000u));
                            ov[2 * k] = act ? __uint_as_float(h2 << 16) : 0.f;
                            ov[2 * k + 1] = act ? __uint_as_float(h2 & 0xffff0000u) : 0.f;
                        }
                    }
                    float amax = 0.f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) amax = __builtin_fmaxf(amax, __builtin_fabsf(ov[k]));
                    amax = row16_reduce_max(amax);
                    if (po->qmode == 2) amax = __builtin_fmaxf(amax, po->qeps);
                    const float sc = amax / 448.0f;
                    const i32x2 packed = po->qmode == 2 ? quant8_fp8<true>(ov, act ? sc : 1.0f) : quant8_fp8<false>(ov, act ? sc : 1.0f);
                    if (act) {
                        *reinterpret_cast<i32x2*>(po->q + (size_t)m * K + vt * 8) = packed;
                        if ((tid & 15) == 0) po->qs[(size_t)m * (K >> 7) + (vt >> 4)] = sc;
                    }
                }
            }
        }
    }
    __syncthreads();
}

// QKV epilogue (the merged q|k|v projection of a GQA / MHA layer, interleaved-pair rotary): what
// gqa_qkv_post_kernel (kv.hip) does to the projection's bf16 output, done on the accumulators instead -- q heads
// rotated into `out`, k heads rotated into the token's page row of k_cache, v heads copied into v_cache
// (Attention.decode_forward_paged, models/model.py:167-198).  A 16-column tile lies inside one head (head_dim % 16
// == 0) and holds whole rotary pairs (2i, 2i + 1), so each lane rotates its own two pairs.
struct QkvPostArgs {
    const float* cos;  // [M, head_dim / 2]
    const float* sin;
    bf16_t* k_cache;   // [num_pages, page_size, hkv, head_dim]
    bf16_t* v_cache;
    int64_t num_pages;
    const int32_t* table;  // [M, pages_per_seq]
    const int32_t* old_lens;
    int page_size, pages_per_seq, hq, hkv, d;
};

// ---------------------------------------------------------------- add + norm -> out = y . W^T
// K loop, K split and accumulation order of bf16_gemm_kernel<1, WK> (gate.hip); D = ring depth (8: the wave's whole
// K range in one round trip, the form chitu_hip_bf16_gemm picks for <= 8 blocks per wave).
// POUT: the owner workgroup also leaves the normalised rows / their fp8 form (NormOut).  S > 1 (grid.y): K cut over S
// workgroups per tile, fp32 partial planes [S][M][N] for the consumer to sum in plane order (chitu_hip_bf16_gemm's
// num_splits form: the router scores, summed by the routing launch) -- every workgroup redoes the prologue.
template <int WK, int D, int MR, bool QKV = false, bool POUT = false>
__global__ __launch_bounds__(64 * WK) void bf16_gemm_add_norm_kernel(
    const bf16_t* x, int64_t x_stride, const bf16_t* add, int64_t add_stride, bf16_t* sum_out, int64_t sum_stride,
    const bf16_t* __restrict__ nw, float eps, const bf16_t* __restrict__ W, void* __restrict__ out, int out_dt, int M,
    int N, int K, QkvPostArgs qa, NormOut po = NormOut{}, float* __restrict__ partial = nullptr, int S = 1,
    int64_t term2_off = 0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    __shared__ float red[WK > 1 ? WK * 256 : 1];
    __shared__ float nred[kFusedNormMaxRows * 16];
    bf16_t* ybuf = reinterpret_cast<bf16_t*>(dyn_lds);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int KB = K >> 6;
    const int T = S * WK, tw = (int)blockIdx.y * WK + wave;
    const int kb0 = (int)((long)KB * tw / T), kb1 = (int)((long)KB * (tw + 1) / T);
    // QKV: the token's rotary factors and page row, requested first (their round trips hide behind everything else)
    float rc[2] = {1.f, 1.f}, rs[2] = {0.f, 0.f};
    int old_len = -1;
    const int tok = min(j, M - 1);
    const int head = QKV ? n0 / qa.d : 0, hcol = QKV ? n0 - head * qa.d : 0;  // workgroup-uniform
    if (QKV && wave == 0) {
        if (head < qa.hq + qa.hkv) {
            const int half = qa.d >> 1, i0 = (hcol >> 1) + g;
            rc[0] = qa.cos[(int64_t)tok * half + i0], rs[0] = qa.sin[(int64_t)tok * half + i0];
            rc[1] = qa.cos[(int64_t)tok * half + i0 + 4], rs[1] = qa.sin[(int64_t)tok * half + i0 + 4];
        }
        if (head >= qa.hq) old_len = qa.old_lens[tok];
    }
    AddNormRegs<WK, MR> regs;
    add_norm_issue<WK, MR>(regs, x, x_stride, add, add_stride, nw, M, K >> 3, term2_off);
    const int eoff = ((j & 1) * 4 + g) * 8;
    const bf16_t* wp0 = W + (size_t)min(n0 + (j >> 1), N - 1) * K + eoff;
    const bf16_t* wp1 = W + (size_t)min(n0 + 8 + (j >> 1), N - 1) * K + eoff;
    s16x8 w0[D], w1[D], xa[D], xb[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        if (kb0 + d < kb1) {
            w0[d] = __builtin_nontemporal_load(reinterpret_cast<const s16x8*>(wp0 + ((kb0 + d) << 6)));
            w1[d] = __builtin_nontemporal_load(reinterpret_cast<const s16x8*>(wp1 + ((kb0 + d) << 6)));
        }
    }
    add_norm_finish<WK, MR, POUT>(regs, sum_out, sum_stride, blockIdx.x == 0 && blockIdx.y == 0, M, K, eps, ybuf, nred, &po,
                                  term2_off != 0);
    // QKV: the page row's address -- old_len arrived with the prologue's loads; its dependent table load is issued here
    // so that the round trip runs under the K loop, not after it
    int64_t dst_row = -1;
    if (QKV && wave == 0 && head >= qa.hq) {
        const int pidx = old_len / qa.page_size;
        if (old_len >= 0 && pidx < qa.pages_per_seq) {
            const int64_t page = qa.table[(int64_t)tok * qa.pages_per_seq + pidx];
            if (page >= 0 && page < qa.num_pages) dst_row = (page * qa.page_size + (old_len % qa.page_size)) * (int64_t)qa.hkv * qa.d;
        }
    }
    const bf16_t* yp = ybuf + (size_t)min(j, M - 1) * K + g * 8;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        if (kb0 + d < kb1) {
            xa[d] = *reinterpret_cast<const s16x8*>(yp + ((kb0 + d) << 6));
            xb[d] = *reinterpret_cast<const s16x8*>(yp + ((kb0 + d) << 6) + 32);
        }
    }
    f32x4 e0 = {0.f, 0.f, 0.f, 0.f}, o0 = e0, e1 = e0, o1 = e0;
    for (int kb = kb0; kb < kb1; kb += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (kb + d < kb1) {
                e0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[d], xa[d], e0, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[d], xb[d], o0, 0, 0, 0);
                e1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[d], xa[d], e1, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[d], xb[d], o1, 0, 0, 0);
                if (kb + d + D < kb1) {
                    const int off = (kb + d + D) << 6;
                    w0[d] = __builtin_nontemporal_load(reinterpret_cast<const s16x8*>(wp0 + off));
                    w1[d] = __builtin_nontemporal_load(reinterpret_cast<const s16x8*>(wp1 + off));
                    xa[d] = *reinterpret_cast<const s16x8*>(yp + off);
                    xb[d] = *reinterpret_cast<const s16x8*>(yp + off + 32);
                }
            }
        }
    }
    f32x4 acc[1] = {f32x4{e0[0] + o0[1], e0[2] + o0[3], e1[0] + o1[1], e1[2] + o1[3]}};
    if (!QKV) {
        gemm_epilogue_v2<1, WK>(acc, red, out, out_dt, partial, M, N, S, 0, n0);
        return;
    }
    // ---- QKV: K-split reduce in wave order (gemm_epilogue_v2's), then rotate / scatter
    f32x4 sum = acc[0];
    if (WK > 1) {
        *reinterpret_cast<f32x4*>(&red[(wave * 64 + lane) * 4]) = acc[0];
        __syncthreads();
        if (wave != 0) return;
        sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < WK; ++w) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&red[(w * 64 + lane) * 4]);
#pragma unroll
            for (int r = 0; r < 4; ++r) sum[r] += v[r];
        }
    }
    if (j >= M) return;
    {
#pragma clang fp contract(off)
        uint32_t pk[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // the projection's bf16 output, then RoPE in fp32 with one rounding: gqa_qkv_post_kernel's arithmetic
            const float x0 = round_bf16(sum[2 * h]), x1 = round_bf16(sum[2 * h + 1]);
            if (head < qa.hq + qa.hkv) {
                const float c = rc[h], sn = rs[h];
                pk[h] = (uint32_t)f32_to_bf16(x0 * c - x1 * sn) | ((uint32_t)f32_to_bf16(x1 * c + x0 * sn) << 16);
            } else {
                pk[h] = (uint32_t)f32_to_bf16(x0) | ((uint32_t)f32_to_bf16(x1) << 16);
            }
        }
        bf16_t* dst = nullptr;
        if (head < qa.hq) dst = (bf16_t*)out + (size_t)j * N + n0;
        else if (dst_row >= 0)
            dst = (head < qa.hq + qa.hkv ? qa.k_cache + dst_row + (size_t)(head - qa.hq) * qa.d
                                         : qa.v_cache + dst_row + (size_t)(head - qa.hq - qa.hkv) * qa.d) + hcol;
        if (dst) {
            *reinterpret_cast<uint32_t*>(dst + 2 * g) = pk[0];
            *reinterpret_cast<uint32_t*>(dst + 8 + 2 * g) = pk[1];
        }
    }
}

// ---------------------------------------------------------------- add + norm -> h = silu(y . W1^T) * (y . W3^T)
// K loop, K split, accumulation order and epilogue of bf16_gemm_silu_kernel<1, WK> (gate.hip), ring depth 3.
template <int WK, int MR>
__global__ __launch_bounds__(64 * WK) __attribute__((amdgpu_waves_per_eu(MR == 1 ? 4 : 2, 8))) void bf16_gemm_silu_add_norm_kernel(
    const bf16_t* x, int64_t x_stride, const bf16_t* add, int64_t add_stride, bf16_t* sum_out, int64_t sum_stride,
    const bf16_t* __restrict__ nw, float eps, const bf16_t* __restrict__ W, bf16_t* __restrict__ out, int M, int inter,
    int K) {
    constexpr int D = 3;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    __shared__ float red[WK > 1 ? WK * 512 : 1];
    __shared__ float nred[kFusedNormMaxRows * 16];
    bf16_t* ybuf = reinterpret_cast<bf16_t*>(dyn_lds);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int KB = K >> 6;
    const int kb0 = KB * wave / WK, kb1 = KB * (wave + 1) / WK;
    AddNormRegs<WK, MR> regs;
    add_norm_issue<WK, MR>(regs, x, x_stride, add, add_stride, nw, M, K >> 3);
    const int eoff = ((j & 1) * 4 + g) * 8;
    const int r0 = min(n0 + (j >> 1), inter - 1), r1 = min(n0 + 8 + (j >> 1), inter - 1);
    const bf16_t* gp0 = W + (size_t)r0 * K + eoff;
    const bf16_t* gp1 = W + (size_t)r1 * K + eoff;
    const bf16_t* up0 = W + (size_t)(inter + r0) * K + eoff;
    const bf16_t* up1 = W + (size_t)(inter + r1) * K + eoff;
    s16x8 wg0[D], wg1[D], wu0[D], wu1[D], xa[D], xb[D];
    auto load_w = [&](int d, int kb) {
        const int off = kb << 6;
        wg0[d] = __builtin_nontemporal_load(reinterpret_cast<const s16x8*>(gp0 + off));
        wg1[d] = __builtin_nontemporal_load(reinterpret_cast<const s16x8*>(gp1 + off));
        wu0[d] = __builtin_nontemporal_load(reinterpret_cast<const s16x8*>(up0 + off));
        wu1[d] = __builtin_nontemporal_load(reinterpret_cast<const s16x8*>(up1 + off));
    };
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (kb0 + d < kb1) load_w(d, kb0 + d);
    add_norm_finish<WK, MR>(regs, sum_out, sum_stride, blockIdx.x == 0, M, K, eps, ybuf, nred);
    const bf16_t* yp = ybuf + (size_t)min(j, M - 1) * K + g * 8;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        if (kb0 + d < kb1) {
            xa[d] = *reinterpret_cast<const s16x8*>(yp + ((kb0 + d) << 6));
            xb[d] = *reinterpret_cast<const s16x8*>(yp + ((kb0 + d) << 6) + 32);
        }
    }
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 ge0 = z, go0 = z, ge1 = z, go1 = z, ue0 = z, uo0 = z, ue1 = z, uo1 = z;
    for (int kb = kb0; kb < kb1; kb += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (kb + d < kb1) {
                ge0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wg0[d], xa[d], ge0, 0, 0, 0);
                go0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wg0[d], xb[d], go0, 0, 0, 0);
                ge1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wg1[d], xa[d], ge1, 0, 0, 0);
                go1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wg1[d], xb[d], go1, 0, 0, 0);
                ue0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wu0[d], xa[d], ue0, 0, 0, 0);
                uo0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wu0[d], xb[d], uo0, 0, 0, 0);
                ue1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wu1[d], xa[d], ue1, 0, 0, 0);
                uo1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wu1[d], xb[d], uo1, 0, 0, 0);
                if (kb + d + D < kb1) {
                    load_w(d, kb + d + D);
                    xa[d] = *reinterpret_cast<const s16x8*>(yp + ((kb + d + D) << 6));
                    xb[d] = *reinterpret_cast<const s16x8*>(yp + ((kb + d + D) << 6) + 32);
                }
            }
        }
    }
    f32x4 ag = {ge0[0] + go0[1], ge0[2] + go0[3], ge1[0] + go1[1], ge1[2] + go1[3]};
    f32x4 au = {ue0[0] + uo0[1], ue0[2] + uo0[3], ue1[0] + uo1[1], ue1[2] + uo1[3]};
    if (WK > 1) {
        *reinterpret_cast<f32x4*>(&red[(wave * 128 + lane) * 4]) = ag;
        *reinterpret_cast<f32x4*>(&red[(wave * 128 + 64 + lane) * 4]) = au;
        __syncthreads();
        if (wave != 0) return;
        ag = au = z;
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
    if (j >= M) return;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
        const int n = n0 + h2 * 8 + 2 * g;
        float hv[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float gv = round_bf16(ag[2 * h2 + q]), uv = round_bf16(au[2 * h2 + q]);
            hv[q] = round_bf16(gv / (1.0f + expf(-gv))) * uv;
        }
        bf16_t* dst = out + (size_t)j * inter + n;
        if (n + 1 < inter && (inter & 1) == 0) *reinterpret_cast<uint32_t*>(dst) = f32x2_to_bf16x2(hv[0], hv[1]);
        else {
            if (n < inter) dst[0] = f32_to_bf16(hv[0]);
            if (n + 1 < inter) dst[1] = f32_to_bf16(hv[1]);
        }
    }
}

// the K split chitu_hip_bf16_gemm / chitu_hip_bf16_gemm_silu would pick for this shape (kept in step with gate.hip so
// the fused and the unfused launches accumulate in the same order)
static int gemm_wk(int tiles, int KB) {
    int WK = 8;
    while (WK > 1 && (WK > KB || (int64_t)tiles * WK > 4096)) WK >>= 1;
    return WK;
}

static bool fused_norm_shape_ok(int64_t M, int64_t K) {
    return M >= 1 && M <= kFusedNormMaxRows && K % 64 == 0 && K >= 512 && K <= kNormWideThreads * 8 && M * K * 2 <= 48 * 1024;
}

}  // namespace chitu

extern "C" int chitu_hip_bf16_gemm_add_norm(const void* x_bf16, int64_t x_row_stride, const void* add_bf16,
                                            int64_t add_row_stride, int32_t add_terms, int64_t add_term_stride,
                                            void* sum_out_bf16, int64_t sum_row_stride,
                                            const void* norm_weight_bf16, float eps, const void* w_bf16, void* out,
                                            int32_t out_dtype, int64_t M, int64_t N, int64_t K, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(x_bf16 && add_bf16 && sum_out_bf16 && norm_weight_bf16 && w_bf16 && out);
    CHITU_REQUIRE(add_terms == 1 || (add_terms == 2 && add_term_stride >= 8 && add_term_stride % 8 == 0));
    const int64_t term2_off = add_terms == 2 ? add_term_stride : 0;
    CHITU_REQUIRE(M >= 0 && N >= 1 && N < (1 << 30) && K >= 64 && K < (1 << 30) && out_dtype >= 0 && out_dtype <= 2);
    CHITU_REQUIRE(x_row_stride % 8 == 0 && add_row_stride % 8 == 0 && sum_row_stride % 8 == 0);
    if (M == 0) return CHITU_OK;
    if (!fused_norm_shape_ok(M, K)) return CHITU_ERR_UNSUPPORTED;
    const int KB = (int)(K / 64), tiles = (int)((N + 15) / 16);
    const int WK = gemm_wk(tiles, KB);
    if (WK < 4) return CHITU_ERR_UNSUPPORTED;
    const int per_wave = KB / WK;
    const size_t lds = (size_t)M * K * 2;
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_MR(WKV, DV, MRV)                                                                                          \
    hipLaunchKernelGGL((bf16_gemm_add_norm_kernel<WKV, DV, MRV>), dim3((unsigned)tiles), dim3(64 * WKV), lds, st,        \
                       (const bf16_t*)x_bf16, x_row_stride, (const bf16_t*)add_bf16, add_row_stride, (bf16_t*)sum_out_bf16, \
                       sum_row_stride, (const bf16_t*)norm_weight_bf16, eps, (const bf16_t*)w_bf16, out, (int)out_dtype, \
                       (int)M, (int)N, (int)K, QkvPostArgs{}, NormOut{}, (float*)nullptr, 1, term2_off)
#define LAUNCH(WKV, DV)                          \
    do {                                         \
        if (M == 1) LAUNCH_MR(WKV, DV, 1);       \
        else if (M == 2) LAUNCH_MR(WKV, DV, 2);  \
        else LAUNCH_MR(WKV, DV, 4);              \
    } while (0)
    // Ring depth 8 = the wave's whole K range in one round trip: right for the launches that are latency-bound (<= 256
    // workgroups: one per CU).  With MORE workgroups than CUs the 8-deep ring's ~160 VGPRs allow one 8-wave workgroup per
    // CU only, the grid runs in two rounds and the second is half empty (Llama-3-8B's qkv projection, 384 workgroups:
    // 15.5 us = 3.2 TB/s); the 4-deep ring (~98 VGPRs) keeps two workgroups per CU resident.  Same accumulation order either
    // way.  Option 11 (bf16_gemm_deep): 0 = never 8 deep, 1 = always (A/B), -1 = this heuristic.
    const int deep_opt = debug_option(kOptBf16GemmDeep);
    const bool deep = per_wave <= 8 && (deep_opt < 0 ? tiles <= 256 : deep_opt != 0);
    if (WK == 8) {
        if (deep) LAUNCH(8, 8);
        else LAUNCH(8, 4);
    } else {
        if (deep) LAUNCH(4, 8);
        else LAUNCH(4, 4);
    }
#undef LAUNCH
#undef LAUNCH_MR
    CHITU_RETURN_LAUNCH_STATUS();
}

extern "C" int chitu_hip_bf16_gemm_add_norm_qkv_post(
    const void* x_bf16, int64_t x_row_stride, const void* add_bf16, int64_t add_row_stride, void* sum_out_bf16,
    int64_t sum_row_stride, const void* norm_weight_bf16, float eps, const void* wqkv_bf16, void* qkv_out_bf16, int64_t M,
    int64_t K, int32_t q_heads, int32_t kv_heads, int32_t head_dim, const float* cos, const float* sin, void* k_cache,
    void* v_cache, int64_t num_pages, int32_t page_size, const int32_t* page_table, int32_t pages_per_seq,
    const int32_t* old_seq_lens, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(x_bf16 && add_bf16 && sum_out_bf16 && norm_weight_bf16 && wqkv_bf16 && qkv_out_bf16);
    CHITU_REQUIRE(cos && sin && k_cache && v_cache && page_table && old_seq_lens);
    CHITU_REQUIRE(M >= 0 && K >= 64 && K < (1 << 30) && q_heads >= 1 && kv_heads >= 1 && head_dim >= 16);
    CHITU_REQUIRE(num_pages >= 1 && page_size >= 1 && pages_per_seq >= 1);
    CHITU_REQUIRE(x_row_stride % 8 == 0 && add_row_stride % 8 == 0 && sum_row_stride % 8 == 0);
    if (M == 0) return CHITU_OK;
    const int64_t N = (int64_t)(q_heads + 2 * kv_heads) * head_dim;
    if (!fused_norm_shape_ok(M, K) || head_dim % 16 != 0 || N >= (1 << 30)) return CHITU_ERR_UNSUPPORTED;
    const int KB = (int)(K / 64), tiles = (int)(N / 16);
    const int WK = gemm_wk(tiles, KB);
    if (WK < 4) return CHITU_ERR_UNSUPPORTED;
    const int per_wave = KB / WK;
    const size_t lds = (size_t)M * K * 2;
    hipStream_t st = (hipStream_t)stream;
    const QkvPostArgs qa{cos, sin, (bf16_t*)k_cache, (bf16_t*)v_cache, num_pages, page_table, old_seq_lens,
                         (int)page_size, (int)pages_per_seq, (int)q_heads, (int)kv_heads, (int)head_dim};
#define LAUNCH_MR(WKV, DV, MRV)                                                                                          \
    hipLaunchKernelGGL((bf16_gemm_add_norm_kernel<WKV, DV, MRV, true>), dim3((unsigned)tiles), dim3(64 * WKV), lds, st,  \
                       (const bf16_t*)x_bf16, x_row_stride, (const bf16_t*)add_bf16, add_row_stride, (bf16_t*)sum_out_bf16, \
                       sum_row_stride, (const bf16_t*)norm_weight_bf16, eps, (const bf16_t*)wqkv_bf16, qkv_out_bf16, 0,   \
                       (int)M, (int)N, (int)K, qa)
#define LAUNCH(WKV, DV)                          \
    do {                                         \
        if (M == 1) LAUNCH_MR(WKV, DV, 1);       \
        else if (M == 2) LAUNCH_MR(WKV, DV, 2);  \
        else LAUNCH_MR(WKV, DV, 4);              \
    } while (0)
    // Ring depth 8 = the wave's whole K range in one round trip: right for the launches that are latency-bound (<= 256
    // workgroups: one per CU).  With MORE workgroups than CUs the 8-deep ring's ~160 VGPRs allow one 8-wave workgroup per
    // CU only, the grid runs in two rounds and the second is half empty (Llama-3-8B's qkv projection, 384 workgroups:
    // 15.5 us = 3.2 TB/s); the 4-deep ring (~98 VGPRs) keeps two workgroups per CU resident.  Same accumulation order either
    // way.  Option 11 (bf16_gemm_deep): 0 = never 8 deep, 1 = always (A/B), -1 = this heuristic.
    const int deep_opt = debug_option(kOptBf16GemmDeep);
    const bool deep = per_wave <= 8 && (deep_opt < 0 ? tiles <= 256 : deep_opt != 0);
    if (WK == 8) {
        if (deep) LAUNCH(8, 8);
        else LAUNCH(8, 4);
    } else {
        if (deep) LAUNCH(4, 8);
        else LAUNCH(4, 4);
    }
#undef LAUNCH
#undef LAUNCH_MR
    CHITU_RETURN_LAUNCH_STATUS();
}

extern "C" int chitu_hip_bf16_gemm_silu_add_norm(const void* x_bf16, int64_t x_row_stride, const void* add_bf16,
                                                 int64_t add_row_stride, void* sum_out_bf16, int64_t sum_row_stride,
                                                 const void* norm_weight_bf16, float eps, const void* w13_bf16,
                                                 void* out_bf16, int64_t M, int64_t inter, int64_t K, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(x_bf16 && add_bf16 && sum_out_bf16 && norm_weight_bf16 && w13_bf16 && out_bf16);
    CHITU_REQUIRE(M >= 0 && inter >= 1 && inter < (1 << 29) && K >= 64 && K < (1 << 30));
    CHITU_REQUIRE(x_row_stride % 8 == 0 && add_row_stride % 8 == 0 && sum_row_stride % 8 == 0);
    if (M == 0) return CHITU_OK;
    if (!fused_norm_shape_ok(M, K)) return CHITU_ERR_UNSUPPORTED;
    const int KB = (int)(K / 64), tiles = (int)((inter + 15) / 16);
    const int WK = gemm_wk(tiles, KB);
    if (WK < 4) return CHITU_ERR_UNSUPPORTED;
    const size_t lds = (size_t)M * K * 2;
    hipStream_t st = (hipStream_t)stream;
#define LAUNCHS_MR(WKV, MRV)                                                                                             \
    hipLaunchKernelGGL((bf16_gemm_silu_add_norm_kernel<WKV, MRV>), dim3((unsigned)tiles), dim3(64 * WKV), lds, st,       \
                       (const bf16_t*)x_bf16, x_row_stride, (const bf16_t*)add_bf16, add_row_stride, (bf16_t*)sum_out_bf16, \
                       sum_row_stride, (const bf16_t*)norm_weight_bf16, eps, (const bf16_t*)w13_bf16, (bf16_t*)out_bf16, \
                       (int)M, (int)inter, (int)K)
#define LAUNCHS(WKV)                          \
    do {                                      \
        if (M == 1) LAUNCHS_MR(WKV, 1);       \
        else if (M == 2) LAUNCHS_MR(WKV, 2);  \
        else LAUNCHS_MR(WKV, 4);              \
    } while (0)
    if (WK == 8) LAUNCHS(8);
    else LAUNCHS(4);
#undef LAUNCHS
#undef LAUNCHS_MR
    CHITU_RETURN_LAUNCH_STATUS();
}
