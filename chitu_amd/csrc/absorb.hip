// Per-head projections of MLA "absorb" mode with the FP8 wkv_b de-quantised on the fly.
//
// Replaces (reference, read-only) -- executed every layer of every decode step there:
//   chitu/models/model_deepseek_v3.py:511-531  weight_dequant(wkv_b) -> bf16 tensor in HBM, then
//                                              einsum("shd,hdc->shc", q_nope, wkv_b[:, :128])
//   chitu/models/model_deepseek_v3.py:697      einsum("bshc,hdc->bshd", o, wkv_b[:, -128:])
//   chitu/triton_kernels.py:217-247            weight_dequant_deepseek_v3_kernel (K8)
// i.e. out[b,h,n] = sum_k x[b,h,k] * bf16(float(W[h,n,k]) * s[block]),  fp32 accumulate, bf16 out.
// The dequantised value is rounded to bf16 exactly as the reference's materialised tensor is,
// but it never leaves registers: 2 MB of fp8 is read instead of 2 MB read + 4 MB written + 4 MB
// re-read per layer.  W is [H, N, K] (head stride given) with K contiguous (the contraction index); for the
// W_UK half the host keeps a transposed fp8 copy (layout-only preprocessing, 1 MB per layer).
// One wave per (16 output columns, head, 16 tokens); weights are the MFMA A operand so a lane
// ends with 4 consecutive output columns of one token.
#include "common.h"
#include "mla_kv_row.h"

namespace chitu {

// Optional rider of the W_UK absorb launch (q_lora_rank == 0 models: there is no wq_b launch to carry it): the
// [kv_norm(kv_c) | rope(k_pe)] row of every token of the batch, appended to its page.  kv_in == nullptr: absent.
struct AbsorbKvRow {
    const bf16_t* kv_in;
    int64_t kv_stride;
    const bf16_t* norm_w;
    float eps;
    bf16_t* cache;
    int64_t num_pages;
    int page_size;
    const int32_t* table;
    int pages_per_seq;
    const int32_t* old_lens;
};

// grid (N/16 [+1 [+1]], H, ceil(batch/16)); block 64.  The optional extra block column (rope != null)
// rotates q_pe[b, h, :64] in place for the tile's 16 tokens (the q half of mla_kv_prep_kernel,
// same arithmetic): it needs wq_b's output just like the absorb itself, so it rides in this launch.
// A second extra column (kv.kv_in != null) writes the tile's KV rows (the other half of mla_kv_prep_kernel),
// block h taking tokens h, h + H, ... of the tile.
__global__ __launch_bounds__(64) void absorb_bmm_kernel(
    const bf16_t* __restrict__ x, int64_t x_sb, int64_t x_sh, const fp8_t* __restrict__ W, int64_t w_sh,
    const float* __restrict__ scale, int64_t s_off, int64_t s_sh, int64_t s_sn, int64_t s_sk,
    bf16_t* __restrict__ out, int64_t o_sb, int64_t o_sh, int batch, int N, int K, bf16_t* __restrict__ q_pe,
    int64_t p_sb, int64_t p_sh, const float* __restrict__ cos, const float* __restrict__ sin, AbsorbKvRow kv) {
    const int lane = threadIdx.x, j = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16, h = blockIdx.y, m0 = blockIdx.z * 16;
    if (n0 >= N + 16) {
        for (int b = m0 + h; b < min(m0 + 16, batch); b += gridDim.y)
            mla_kv_row_one_wave(b, kv.kv_in + (int64_t)b * kv.kv_stride, kv.norm_w, kv.eps, cos, sin, kv.cache, kv.num_pages,
                                kv.page_size, kv.table, kv.pages_per_seq, kv.old_lens);
        return;
    }
    if (n0 >= N) {
#pragma clang fp contract(off)
        for (int idx = lane; idx < 16 * 32; idx += 64) {
            const int b = m0 + (idx >> 5), i = idx & 31;
            if (b >= batch) break;
            bf16_t* p = q_pe + b * p_sb + h * p_sh + 2 * i;
            const uint32_t raw = *reinterpret_cast<const uint32_t*>(p);
            const float x0 = __uint_as_float(raw << 16), x1 = __uint_as_float(raw & 0xffff0000u);
            const float c = cos[(int64_t)b * 32 + i], s = sin[(int64_t)b * 32 + i];
            *reinterpret_cast<uint32_t*>(p) = f32x2_to_bf16x2(x0 * c - x1 * s, x1 * c + x0 * s);
        }
        return;
    }
    const int m = min(m0 + j, batch - 1);
    const fp8_t* wp = W + (int64_t)h * w_sh + (int64_t)min(n0 + j, N - 1) * K + g * 16;
    const bf16_t* xp = x + m * x_sb + h * x_sh + g * 16;
    const float* sp = scale + s_off + h * s_sh + (n0 >> 7) * s_sn;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 64) {
        const i32x4 w = *reinterpret_cast<const i32x4*>(wp + k0);
        const float s = sp[(k0 >> 7) * s_sk];
        const s16x8 wa = dequant8_bf16((uint32_t)w[0], (uint32_t)w[1], s);
        const s16x8 wb = dequant8_bf16((uint32_t)w[2], (uint32_t)w[3], s);
        const s16x8 xa = *reinterpret_cast<const s16x8*>(xp + k0);
        const s16x8 xb = *reinterpret_cast<const s16x8*>(xp + k0 + 8);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xa, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, xb, acc, 0, 0, 0);
    }
    if (m0 + j >= batch) return;
    const int n = n0 + g * 4;
    bf16_t* dst = out + (m0 + j) * o_sb + h * o_sh + n;
    if (n + 3 < N) {
        i32x2 o;
        o[0] = (int)((uint32_t)f32_to_bf16(acc[0]) | ((uint32_t)f32_to_bf16(acc[1]) << 16));
        o[1] = (int)((uint32_t)f32_to_bf16(acc[2]) | ((uint32_t)f32_to_bf16(acc[3]) << 16));
        *reinterpret_cast<i32x2*>(dst) = o;
    } else {
        for (int r = 0; r < 4 && n + r < N; ++r) dst[r] = f32_to_bf16(acc[r]);
    }
}

// Same product for N == 128 (the W_UV half: one head = one 128-wide quantisation group of wo's
// input), fused with act_quant_deepseek_v3 of the bf16-rounded result (the launch that precedes the
// wo GEMM in linear_deepseek_v3, model_deepseek_v3.py:98-100).  grid (H, ceil(batch/16)); block 512:
// wave w owns output columns [16w, 16w+16), the per-(token, head) max goes through LDS.
template <int KC>  // K / 64 when known at compile time (8 for kv_lora_rank 512), 0 = runtime loop
__global__ __launch_bounds__(512) void absorb_uv_quant_kernel(
    const bf16_t* __restrict__ x, int64_t x_sb, int64_t x_sh, const fp8_t* __restrict__ W, int64_t w_sh,
    const float* __restrict__ scale, int64_t s_off, int64_t s_sh, int64_t s_sk, fp8_t* __restrict__ q,
    float* __restrict__ qs, int batch, int H, int K) {
    __shared__ float red[8][16];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, m0 = blockIdx.y * 16;
    const int m = min(m0 + j, batch - 1);
    const fp8_t* wp = W + (int64_t)h * w_sh + (int64_t)(wave * 16 + j) * K + g * 16;
    const bf16_t* xp = x + m * x_sb + h * x_sh + g * 16;
    const float* sp = scale + s_off + h * s_sh;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (KC > 0) {
        // all loads of the K loop issued up front (one HBM round trip instead of KC)
        constexpr int KA = KC > 0 ? KC : 1;  // (zero-length arrays are not allowed in device code)
        i32x4 w[KA];
        s16x8 xa[KA], xb[KA];
        float sv[KA];
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            w[c] = *reinterpret_cast<const i32x4*>(wp + c * 64);
            sv[c] = sp[(c >> 1) * s_sk];
            xa[c] = *reinterpret_cast<const s16x8*>(xp + c * 64);
            xb[c] = *reinterpret_cast<const s16x8*>(xp + c * 64 + 8);
        }
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dequant8_bf16((uint32_t)w[c][0], (uint32_t)w[c][1], sv[c]), xa[c], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dequant8_bf16((uint32_t)w[c][2], (uint32_t)w[c][3], sv[c]), xb[c], acc, 0, 0, 0);
        }
    } else {
        for (int k0 = 0; k0 < K; k0 += 64) {
            const i32x4 w = *reinterpret_cast<const i32x4*>(wp + k0);
            const float s = sp[(k0 >> 7) * s_sk];
            const s16x8 wa = dequant8_bf16((uint32_t)w[0], (uint32_t)w[1], s);
            const s16x8 wb = dequant8_bf16((uint32_t)w[2], (uint32_t)w[3], s);
            const s16x8 xa = *reinterpret_cast<const s16x8*>(xp + k0);
            const s16x8 xb = *reinterpret_cast<const s16x8*>(xp + k0 + 8);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xa, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, xb, acc, 0, 0, 0);
        }
    }
    float v[4], amax = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        v[r] = round_bf16(acc[r]);
        amax = __builtin_fmaxf(amax, __builtin_fabsf(v[r]));
    }
    amax = __builtin_fmaxf(amax, __shfl_xor(amax, 16, 64));
    amax = __builtin_fmaxf(amax, __shfl_xor(amax, 32, 64));
    if (g == 0) red[wave][j] = amax;
    __syncthreads();
    amax = red[0][j];
#pragma unroll
    for (int w = 1; w < 8; ++w) amax = __builtin_fmaxf(amax, red[w][j]);
    const float sc = amax / 448.0f;
    if (m0 + j >= batch) return;
    float t[4];
    if (__builtin_amdgcn_ballot_w64(!group_div_fast(sc)) == 0) {
        const float r = group_rcp(sc);
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = group_div(v[k], sc, r);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = v[k] / sc;
    }
    const uint32_t packed = f32x2_to_fp8x2(t[0], t[1]) | (f32x2_to_fp8x2(t[2], t[3]) << 16);
    *reinterpret_cast<uint32_t*>(q + ((int64_t)(m0 + j) * H + h) * 128 + wave * 16 + g * 4) = packed;
    if (wave == 0 && g == 0) qs[(int64_t)(m0 + j) * H + h] = sc;
}

// ---- The two projections over PREFILL-sized row counts (hundreds to thousands of tokens).  Same arithmetic per output as
// absorb_bmm_kernel / absorb_uv_quant_kernel (bit-identical results); what changes is the work per workgroup: run over a
// 2048-token prompt the decode kernels are 67 584 one-wave workgroups of 4 MFMAs each (50 us for 42 MB) and 2048 workgroups
// that each re-read and re-dequantise their head's 64 KB of W_UV (47 us).  Here a workgroup de-quantises its weight
// fragments ONCE into registers and walks kRowsTM token tiles with them.
constexpr int kRowsTM = 8;  // token tiles (of 16) per workgroup
constexpr int kRowsMinBatch = 256;  // below this the decode kernels (more workgroups) are the better shape

// grid (ceil(N/64), H, ceil(batch / (16 kRowsTM))); block 256: wave w owns output columns [64 bx + 16 w, +16), so the four
// waves of a workgroup write whole 128-byte lines of a token row between them; K = 64 KC (KC <= 4: the W_UK half has K = 128).
// The token tiles are walked with two register buffers: tile t + 1's activations are requested before tile t is multiplied.
template <int KC>
__global__ __launch_bounds__(256) void absorb_bmm_rows_kernel(
    const bf16_t* __restrict__ x, int64_t x_sb, int64_t x_sh, const fp8_t* __restrict__ W, int64_t w_sh,
    const float* __restrict__ scale, int64_t s_off, int64_t s_sh, int64_t s_sn, int64_t s_sk,
    bf16_t* __restrict__ out, int64_t o_sb, int64_t o_sh, int batch, int N) {
    constexpr int K = 64 * KC;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = lane & 15, g = lane >> 4;
    const int n0 = (blockIdx.x * 4 + wave) * 16, h = blockIdx.y;
    if (n0 >= N) return;
    const fp8_t* wp = W + (int64_t)h * w_sh + (int64_t)min(n0 + j, N - 1) * K + g * 16;
    const float* sp = scale + s_off + h * s_sh + (n0 >> 7) * s_sn;
    s16x8 wa[KC], wb[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        const i32x4 w = *reinterpret_cast<const i32x4*>(wp + c * 64);
        const float sc = sp[(c >> 1) * s_sk];
        wa[c] = dequant8_bf16((uint32_t)w[0], (uint32_t)w[1], sc);
        wb[c] = dequant8_bf16((uint32_t)w[2], (uint32_t)w[3], sc);
    }
    const int n = n0 + g * 4;
    const int t0 = blockIdx.z * kRowsTM;
    struct XTile {
        s16x8 a[KC], b[KC];
    };
    auto load = [&](XTile& t, int tile) {  // (rows past the batch re-read its last row: never stored)
        const bf16_t* xp = x + (int64_t)min(tile * 16 + j, batch - 1) * x_sb + h * x_sh + g * 16;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            t.a[c] = *reinterpret_cast<const s16x8*>(xp + c * 64);
            t.b[c] = *reinterpret_cast<const s16x8*>(xp + c * 64 + 8);
        }
    };
    auto finish = [&](const XTile& t, int tile) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[c], t.a[c], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[c], t.b[c], acc, 0, 0, 0);
        }
        const int m = tile * 16 + j;
        if (m >= batch) return;
        bf16_t* dst = out + (int64_t)m * o_sb + h * o_sh + n;
        if (n + 3 < N) {
            i32x2 o;
            o[0] = (int)((uint32_t)f32_to_bf16(acc[0]) | ((uint32_t)f32_to_bf16(acc[1]) << 16));
            o[1] = (int)((uint32_t)f32_to_bf16(acc[2]) | ((uint32_t)f32_to_bf16(acc[3]) << 16));
            *reinterpret_cast<i32x2*>(dst) = o;
        } else {
            for (int r = 0; r < 4 && n + r < N; ++r) dst[r] = f32_to_bf16(acc[r]);
        }
    };
    XTile A, B;
    load(A, t0);
#pragma unroll
    for (int tm = 0; tm < kRowsTM; tm += 2) {
        load(B, t0 + tm + 1);
        finish(A, t0 + tm);
        if (tm + 2 < kRowsTM) load(A, t0 + tm + 2);
        finish(B, t0 + tm + 1);
    }
}

// grid (H, ceil(batch / (16 kRowsTM))); block 512; K = 512 (kv_lora_rank), N = 128.  Two activation buffers as above.
__global__ __launch_bounds__(512) void absorb_uv_quant_rows_kernel(
    const bf16_t* __restrict__ x, int64_t x_sb, int64_t x_sh, const fp8_t* __restrict__ W, int64_t w_sh,
    const float* __restrict__ scale, int64_t s_off, int64_t s_sh, int64_t s_sk, fp8_t* __restrict__ q,
    float* __restrict__ qs, int batch, int H) {
    constexpr int KC = 8, K = 512;
    __shared__ float red[2][8][16];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = lane & 15, g = lane >> 4;
    const int h = blockIdx.x;
    const fp8_t* wp = W + (int64_t)h * w_sh + (int64_t)(wave * 16 + j) * K + g * 16;
    const float* sp = scale + s_off + h * s_sh;
    s16x8 wa[KC], wb[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        const i32x4 w = *reinterpret_cast<const i32x4*>(wp + c * 64);
        const float sc = sp[(c >> 1) * s_sk];
        wa[c] = dequant8_bf16((uint32_t)w[0], (uint32_t)w[1], sc);
        wb[c] = dequant8_bf16((uint32_t)w[2], (uint32_t)w[3], sc);
    }
    const int t0 = blockIdx.y * kRowsTM;
    struct XTile {
        s16x8 a[KC], b[KC];
    };
    auto load = [&](XTile& t, int tile) {  // (rows past the batch re-read its last row: never stored)
        const bf16_t* xp = x + (int64_t)min(tile * 16 + j, batch - 1) * x_sb + h * x_sh + g * 16;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            t.a[c] = *reinterpret_cast<const s16x8*>(xp + c * 64);
            t.b[c] = *reinterpret_cast<const s16x8*>(xp + c * 64 + 8);
        }
    };
    auto finish = [&](const XTile& t, int tile, int parity) {  // (every wave of the workgroup runs every tile: one barrier each)
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[c], t.a[c], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[c], t.b[c], acc, 0, 0, 0);
        }
        float v[4], amax = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[r] = round_bf16(acc[r]);
            amax = __builtin_fmaxf(amax, __builtin_fabsf(v[r]));
        }
        amax = __builtin_fmaxf(amax, __shfl_xor(amax, 16, 64));
        amax = __builtin_fmaxf(amax, __shfl_xor(amax, 32, 64));
        if (g == 0) red[parity][wave][j] = amax;
        __syncthreads();  // (the other half of `red` is not rewritten before every wave has read this one)
        amax = red[parity][0][j];
#pragma unroll
        for (int w = 1; w < 8; ++w) amax = __builtin_fmaxf(amax, red[parity][w][j]);
        const float sc = amax / 448.0f;
        float tq[4];
        if (__builtin_amdgcn_ballot_w64(!group_div_fast(sc)) == 0) {
            const float r = group_rcp(sc);
#pragma unroll
            for (int k = 0; k < 4; ++k) tq[k] = group_div(v[k], sc, r);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) tq[k] = v[k] / sc;
        }
        const int m = tile * 16 + j;
        if (m < batch) {
            const uint32_t packed = f32x2_to_fp8x2(tq[0], tq[1]) | (f32x2_to_fp8x2(tq[2], tq[3]) << 16);
            *reinterpret_cast<uint32_t*>(q + ((int64_t)m * H + h) * 128 + wave * 16 + g * 4) = packed;
            if (wave == 0 && g == 0) qs[(int64_t)m * H + h] = sc;
        }
    };
    XTile A, B;
    load(A, t0);
#pragma unroll
    for (int tm = 0; tm < kRowsTM; tm += 2) {
        load(B, t0 + tm + 1);
        finish(A, t0 + tm, 0);
        if (tm + 2 < kRowsTM) load(A, t0 + tm + 2);
        finish(B, t0 + tm + 1, 1);
    }
}

// MLA split-KV merge + W_UV projection + act_quant in one launch (small batches): replaces
// mla_merge_kernel (mla_decode.hip) followed by absorb_uv_quant_kernel, same arithmetic and rounding
// points (merged o -> bf16, projection -> bf16, e4m3 group quantisation), two dependent launches and
// the o round trip less.  grid (H, batch); block 512: thread t merges latent column t of (b, h)
// into LDS while the 64 KB W_UV[h] tile (requested first) is in flight; wave w then owns output
// columns [16w, 16w+16).  The MFMA tile's 16 token columns all carry token b (LDS broadcast).
__global__ __launch_bounds__(512) void mla_merge_uv_quant_kernel(
    const bf16_t* __restrict__ part_o, const float* __restrict__ part_lse, int S, const fp8_t* __restrict__ W,
    int64_t w_sh, const float* __restrict__ scale, int64_t s_off, int64_t s_sh, int64_t s_sk,
    fp8_t* __restrict__ q, float* __restrict__ qs, int H, int tile_major) {
    constexpr int K = 512, KC = 8;
    __shared__ __attribute__((aligned(16))) bf16_t xs[K];
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, b = blockIdx.y;
    CHITU_PROBE_MARK(0);
    const int64_t bh = (int64_t)b * H + h;
    const fp8_t* wp = W + (int64_t)h * w_sh + (int64_t)(wave * 16 + j) * K + g * 16;
    const float* sp = scale + s_off + h * s_sh;
    i32x4 w[KC];
    float sv[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        w[c] = *reinterpret_cast<const i32x4*>(wp + c * 64);
        sv[c] = sp[(c >> 1) * s_sk];
    }
    // ---- merge: out[t] = sum_s w_s * part_o[bh, s, t] / sum_s w_s, w_s = exp(lse_s - max lse)
    // (same operation order as mla_merge_kernel).  Lane s of every wave holds lse_s / w_s; the
    // partial rows are requested 16 at a time, not one round trip per row.
    if (S <= 16) {
        // the usual decode shape: all S partial rows are requested BEFORE the lse values are looked at (their
        // addresses do not depend on them), so the launch pays one memory round trip, not two; same sums, same order
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = bf16_to_f32(part_o[(bh * S + min(i, S - 1)) * K + tid]);
        const float l = lane < S ? part_lse[bh * S + lane] : -INFINITY;
        const float m = wave_reduce_max(l);
        const float wl = l == -INFINITY ? 0.f : __expf(l - m);
        float acc = 0.f, wsum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float ws = i < S ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wl), i)) : 0.f;
            wsum += ws;
            acc = merge_term(acc, ws, v[i]);
        }
        const float inv = wsum > 0.f ? 1.0f / wsum : 0.f;
        xs[tid] = f32_to_bf16(acc * inv);
    } else {
        const float* lse = part_lse + bh * S;
        float m = -INFINITY;
        for (int s0 = 0; s0 < S; s0 += 64) {
            const float l = s0 + lane < S ? lse[s0 + lane] : -INFINITY;
            m = __builtin_fmaxf(m, wave_reduce_max(l));
        }
        float acc = 0.f, wsum = 0.f;
        for (int s0 = 0; s0 < S; s0 += 64) {
            const float l = s0 + lane < S ? lse[s0 + lane] : -INFINITY;
            const float wl = l == -INFINITY ? 0.f : __expf(l - m);
            const int n = min(64, S - s0);
            for (int i0 = 0; i0 < n; i0 += 16) {
                float v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = bf16_to_f32(part_o[(bh * S + s0 + min(i0 + i, n - 1)) * K + tid]);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float ws = i0 + i < n ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wl), (i0 + i) & 63)) : 0.f;
                    wsum += ws;
                    acc = merge_term(acc, ws, v[i]);
                }
            }
        }
        const float inv = wsum > 0.f ? 1.0f / wsum : 0.f;
        xs[tid] = f32_to_bf16(acc * inv);
    }
    CHITU_PROBE_MARK(1);  // merged row in LDS
    __syncthreads();
    CHITU_PROBE_MARK(2);
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        const s16x8 xa = *reinterpret_cast<const s16x8*>(&xs[c * 64 + g * 16]);
        const s16x8 xb = *reinterpret_cast<const s16x8*>(&xs[c * 64 + g * 16 + 8]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dequant8_bf16((uint32_t)w[c][0], (uint32_t)w[c][1], sv[c]), xa, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dequant8_bf16((uint32_t)w[c][2], (uint32_t)w[c][3], sv[c]), xb, acc, 0, 0, 0);
    }
    float v[4], amax = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        v[r] = round_bf16(acc[r]);
        amax = __builtin_fmaxf(amax, __builtin_fabsf(v[r]));
    }
    amax = __builtin_fmaxf(amax, __shfl_xor(amax, 16, 64));
    amax = __builtin_fmaxf(amax, __shfl_xor(amax, 32, 64));
    if (lane == 0) red[wave] = amax;
    CHITU_PROBE_MARK(3);  // projection done (W tile had arrived)
    __syncthreads();
    amax = red[0];
#pragma unroll
    for (int w8 = 1; w8 < 8; ++w8) amax = __builtin_fmaxf(amax, red[w8]);
    const float sc = amax / 448.0f;
    float t[4];
    if (__builtin_amdgcn_ballot_w64(!group_div_fast(sc)) == 0) {
        const float r = group_rcp(sc);
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = group_div(v[k], sc, r);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = v[k] / sc;
    }
    CHITU_PROBE_MARK(4);
    if (j != 0) return;
    const uint32_t packed = f32x2_to_fp8x2(t[0], t[1]) | (f32x2_to_fp8x2(t[2], t[3]) << 16);
    if (tile_major) {  // row b of a [batch, H * 128] matrix, tile-major (gemm_common.h): 16-B chunk index h * 8 + wave
        const int64_t tile = b >> 4;
        const int m = b & 15;
        *reinterpret_cast<uint32_t*>(q + ((tile * (H * 8) + h * 8 + wave) * 16 + m) * 16 + g * 4) = packed;
        if (wave == 0 && g == 0) qs[(tile * H + h) * 16 + m] = sc;
    } else {
        *reinterpret_cast<uint32_t*>(q + bh * 128 + wave * 16 + g * 4) = packed;
        if (wave == 0 && g == 0) qs[bh] = sc;
    }
}

}  // namespace chitu

static int launch_absorb_bmm(const void* x, int64_t x_sb, int64_t x_sh, const void* w, int64_t w_sh, const float* scale,
                             int64_t s_off, int64_t s_sh, int64_t s_sn, int64_t s_sk, void* out, int64_t o_sb,
                             int64_t o_sh, int batch, int heads, int N, int K, void* q_pe, int64_t p_sb, int64_t p_sh,
                             const float* cos, const float* sin, void* stream, const chitu::AbsorbKvRow* kv = nullptr) {
    using namespace chitu;
    if (!q_pe && !kv && batch >= kRowsMinBatch && K == 128) {  // prefill-sized W_UK absorb (qk_nope_head_dim 128), no riders
        const dim3 grid_rows((unsigned)((N + 63) / 64), (unsigned)heads, (unsigned)((batch + 16 * kRowsTM - 1) / (16 * kRowsTM)));
        hipLaunchKernelGGL(absorb_bmm_rows_kernel<2>, grid_rows, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, x_sb, x_sh, (const fp8_t*)w, w_sh,
                           scale, s_off, s_sh, s_sn, s_sk, (bf16_t*)out, o_sb, o_sh, batch, N);
        CHITU_RETURN_LAUNCH_STATUS();
    }
    // the riders' columns are found by n0 >= N (+ 16): N itself a multiple of 16 when they are present
    const dim3 grid((unsigned)((N + 15) / 16 + (q_pe ? 1 : 0) + (kv ? 1 : 0)), (unsigned)heads, (unsigned)((batch + 15) / 16));
    hipLaunchKernelGGL(absorb_bmm_kernel, grid, dim3(64), 0, (hipStream_t)stream, (const bf16_t*)x, x_sb, x_sh,
                       (const fp8_t*)w, w_sh, scale, s_off, s_sh, s_sn, s_sk, (bf16_t*)out, o_sb, o_sh, batch, N, K,
                       (bf16_t*)q_pe, p_sb, p_sh, cos, sin, kv ? *kv : AbsorbKvRow{});
    CHITU_RETURN_LAUNCH_STATUS();
}

extern "C" int chitu_hip_absorb_uv_quant_fp8(const void* x_bf16, int64_t x_stride_b, int64_t x_stride_h,
                                             const void* w_fp8, int64_t w_stride_h, const float* scale,
                                             int64_t scale_offset, int64_t scale_stride_h,
                                             int64_t scale_stride_k, void* q_fp8, float* q_scales,
                                             int32_t batch, int32_t heads, int32_t K, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(x_bf16 && w_fp8 && scale && q_fp8 && q_scales);
    CHITU_REQUIRE(batch >= 0 && heads >= 1 && K >= 64 && w_stride_h % 16 == 0);
    if (K % 64 != 0) return CHITU_ERR_UNSUPPORTED;
    CHITU_REQUIRE(x_stride_b % 8 == 0 && x_stride_h % 8 == 0);
    if (batch == 0) return CHITU_OK;
    if (K == 512 && batch >= chitu::kRowsMinBatch) {  // prefill-sized: weights de-quantised once per kRowsTM token tiles
        const dim3 grid_rows((unsigned)heads, (unsigned)((batch + 16 * kRowsTM - 1) / (16 * kRowsTM)));
        hipLaunchKernelGGL(absorb_uv_quant_rows_kernel, grid_rows, dim3(512), 0, (hipStream_t)stream, (const bf16_t*)x_bf16,
                           x_stride_b, x_stride_h, (const fp8_t*)w_fp8, w_stride_h, scale, scale_offset, scale_stride_h,
                           scale_stride_k, (fp8_t*)q_fp8, q_scales, (int)batch, (int)heads);
        CHITU_RETURN_LAUNCH_STATUS();
    }
    const dim3 grid((unsigned)heads, (unsigned)((batch + 15) / 16));
    if (K == 512)
        hipLaunchKernelGGL(absorb_uv_quant_kernel<8>, grid, dim3(512), 0, (hipStream_t)stream, (const bf16_t*)x_bf16,
                           x_stride_b, x_stride_h, (const fp8_t*)w_fp8, w_stride_h, scale, scale_offset,
                           scale_stride_h, scale_stride_k, (fp8_t*)q_fp8, q_scales, (int)batch, (int)heads, (int)K);
    else
        hipLaunchKernelGGL(absorb_uv_quant_kernel<0>, grid, dim3(512), 0, (hipStream_t)stream, (const bf16_t*)x_bf16,
                           x_stride_b, x_stride_h, (const fp8_t*)w_fp8, w_stride_h, scale, scale_offset,
                           scale_stride_h, scale_stride_k, (fp8_t*)q_fp8, q_scales, (int)batch, (int)heads, (int)K);
    CHITU_RETURN_LAUNCH_STATUS();
}

extern "C" int chitu_hip_absorb_bmm_fp8(const void* x_bf16, int64_t x_stride_b, int64_t x_stride_h,
                                        const void* w_fp8, int64_t w_stride_h, const float* scale,
                                        int64_t scale_offset,
                                        int64_t scale_stride_h, int64_t scale_stride_n,
                                        int64_t scale_stride_k, void* out_bf16, int64_t out_stride_b,
                                        int64_t out_stride_h, int32_t batch, int32_t heads, int32_t N,
                                        int32_t K, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(x_bf16 && w_fp8 && scale && out_bf16);
    CHITU_REQUIRE(batch >= 0 && heads >= 1 && N >= 1 && K >= 64 && w_stride_h % 16 == 0);
    if (K % 64 != 0) return CHITU_ERR_UNSUPPORTED;
    CHITU_REQUIRE(x_stride_b % 8 == 0 && x_stride_h % 8 == 0 && out_stride_b % 4 == 0 && out_stride_h % 4 == 0);
    if (batch == 0) return CHITU_OK;
    return launch_absorb_bmm(x_bf16, x_stride_b, x_stride_h, w_fp8, w_stride_h, scale, scale_offset, scale_stride_h,
                             scale_stride_n, scale_stride_k, out_bf16, out_stride_b, out_stride_h, batch, heads, N, K,
                             nullptr, 0, 0, nullptr, nullptr, stream);
}

extern "C" int chitu_hip_absorb_bmm_rope_fp8(const void* x_bf16, int64_t x_stride_b, int64_t x_stride_h,
                                             const void* w_fp8, int64_t w_stride_h, const float* scale,
                                             int64_t scale_offset, int64_t scale_stride_h,
                                             int64_t scale_stride_n, int64_t scale_stride_k, void* out_bf16,
                                             int64_t out_stride_b, int64_t out_stride_h, int32_t batch,
                                             int32_t heads, int32_t N, int32_t K, void* q_pe_bf16,
                                             int64_t q_pe_stride_b, int64_t q_pe_stride_h, const float* cos,
                                             const float* sin, int32_t rope_dim, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(x_bf16 && w_fp8 && scale && out_bf16 && q_pe_bf16 && cos && sin);
    CHITU_REQUIRE(batch >= 0 && heads >= 1 && N >= 1 && K >= 64 && w_stride_h % 16 == 0);
    if (K % 64 != 0 || rope_dim != 64) return CHITU_ERR_UNSUPPORTED;
    CHITU_REQUIRE(x_stride_b % 8 == 0 && x_stride_h % 8 == 0 && out_stride_b % 4 == 0 && out_stride_h % 4 == 0);
    CHITU_REQUIRE(q_pe_stride_b % 2 == 0 && q_pe_stride_h % 2 == 0);
    if (batch == 0) return CHITU_OK;
    return launch_absorb_bmm(x_bf16, x_stride_b, x_stride_h, w_fp8, w_stride_h, scale, scale_offset, scale_stride_h,
                             scale_stride_n, scale_stride_k, out_bf16, out_stride_b, out_stride_h, batch, heads, N, K,
                             q_pe_bf16, q_pe_stride_b, q_pe_stride_h, cos, sin, stream);
    CHITU_RETURN_LAUNCH_STATUS();
}

extern "C" int chitu_hip_absorb_bmm_rope_kv_fp8(const void* x_bf16, int64_t x_stride_b, int64_t x_stride_h,
                                                const void* w_fp8, int64_t w_stride_h, const float* scale,
                                                int64_t scale_offset, int64_t scale_stride_h,
                                                int64_t scale_stride_n, int64_t scale_stride_k, void* out_bf16,
                                                int64_t out_stride_b, int64_t out_stride_h, int32_t batch,
                                                int32_t heads, int32_t N, int32_t K, void* q_pe_bf16,
                                                int64_t q_pe_stride_b, int64_t q_pe_stride_h, const float* cos,
                                                const float* sin, int32_t rope_dim, const void* kv_in_bf16,
                                                int64_t kv_row_stride, const void* kv_norm_weight_bf16, float eps,
                                                void* kv_cache, int64_t num_pages, int32_t page_size,
                                                const int32_t* page_table, int32_t pages_per_seq,
                                                const int32_t* old_seq_lens, int32_t kv_lora_rank, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(x_bf16 && w_fp8 && scale && out_bf16 && q_pe_bf16 && cos && sin);
    CHITU_REQUIRE(kv_in_bf16 && kv_norm_weight_bf16 && kv_cache && page_table && old_seq_lens);
    CHITU_REQUIRE(batch >= 0 && heads >= 1 && N >= 1 && K >= 64 && w_stride_h % 16 == 0);
    CHITU_REQUIRE(num_pages >= 1 && page_size >= 1 && pages_per_seq >= 1);
    if (K % 64 != 0 || rope_dim != 64 || kv_lora_rank != 512 || N % 16 != 0) return CHITU_ERR_UNSUPPORTED;
    CHITU_REQUIRE(x_stride_b % 8 == 0 && x_stride_h % 8 == 0 && out_stride_b % 4 == 0 && out_stride_h % 4 == 0);
    CHITU_REQUIRE(q_pe_stride_b % 2 == 0 && q_pe_stride_h % 2 == 0 && kv_row_stride % 8 == 0);
    if (batch == 0) return CHITU_OK;
    const AbsorbKvRow kv{(const bf16_t*)kv_in_bf16, kv_row_stride, (const bf16_t*)kv_norm_weight_bf16, eps,
                         (bf16_t*)kv_cache, num_pages, (int)page_size, page_table, (int)pages_per_seq, old_seq_lens};
    return launch_absorb_bmm(x_bf16, x_stride_b, x_stride_h, w_fp8, w_stride_h, scale, scale_offset, scale_stride_h,
                             scale_stride_n, scale_stride_k, out_bf16, out_stride_b, out_stride_h, batch, heads, N, K,
                             q_pe_bf16, q_pe_stride_b, q_pe_stride_h, cos, sin, stream, &kv);
}

static int launch_merge_uv_quant(const void* workspace, int32_t num_splits, const void* w_fp8, int64_t w_stride_h,
                                 const float* scale, int64_t scale_offset, int64_t scale_stride_h, int64_t scale_stride_k,
                                 void* q_fp8, float* q_scales, int32_t batch, int32_t heads, int32_t K, int tile_major,
                                 void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(workspace && w_fp8 && scale && q_fp8 && q_scales);
    CHITU_REQUIRE(batch >= 0 && heads >= 1 && num_splits >= 2 && num_splits <= 256 && w_stride_h % 16 == 0);
    if (K != 512) return CHITU_ERR_UNSUPPORTED;
    if (batch == 0) return CHITU_OK;
    const bf16_t* part_o = (const bf16_t*)workspace;  // chitu_hip_mla_decode's layout: bf16 rows | fp32 LSE
    const float* part_lse = (const float*)(part_o + (int64_t)batch * heads * num_splits * 512);
    hipLaunchKernelGGL(mla_merge_uv_quant_kernel, dim3((unsigned)heads, (unsigned)batch), dim3(512), 0,
                       (hipStream_t)stream, part_o, part_lse, (int)num_splits, (const fp8_t*)w_fp8, w_stride_h,
                       scale, scale_offset, scale_stride_h, scale_stride_k, (fp8_t*)q_fp8, q_scales, (int)heads, tile_major);
    CHITU_RETURN_LAUNCH_STATUS();
}

extern "C" int chitu_hip_mla_merge_absorb_uv_quant_fp8(const void* workspace, int32_t num_splits,
                                                       const void* w_fp8, int64_t w_stride_h,
                                                       const float* scale, int64_t scale_offset,
                                                       int64_t scale_stride_h, int64_t scale_stride_k,
                                                       void* q_fp8, float* q_scales, int32_t batch,
                                                       int32_t heads, int32_t K, void* stream) {
    return launch_merge_uv_quant(workspace, num_splits, w_fp8, w_stride_h, scale, scale_offset, scale_stride_h,
                                 scale_stride_k, q_fp8, q_scales, batch, heads, K, 0, stream);
}

// The same with the fp8 output TILE-MAJOR (see chitu_hip_fp8_gemm_blockscale_tm): q_fp8 [ceil(batch/16)*16, heads*128]
// bytes, q_scales [ceil(batch/16), heads, 16].
extern "C" int chitu_hip_mla_merge_absorb_uv_quant_fp8_tm(const void* workspace, int32_t num_splits,
                                                          const void* w_fp8, int64_t w_stride_h,
                                                          const float* scale, int64_t scale_offset,
                                                          int64_t scale_stride_h, int64_t scale_stride_k,
                                                          void* q_fp8, float* q_scales, int32_t batch,
                                                          int32_t heads, int32_t K, void* stream) {
    return launch_merge_uv_quant(workspace, num_splits, w_fp8, w_stride_h, scale, scale_offset, scale_stride_h,
                                 scale_stride_k, q_fp8, q_scales, batch, heads, K, 1, stream);
}

CHITU_PROBE_READER(absorb)
