// Pieces shared by the weight-streaming GEMM kernels (fp8_gemm.hip, gate.hip).
#pragma once
#include "common.h"

namespace chitu {

// Epilogue shared by both kernels: K-split reduce across the workgroup's waves through LDS
// in fixed wave order, then one token-tile per wave is written (bf16/f16/f32, or the fp32
// partial slab of cross-workgroup split `blockIdx.y`).
template <int MT, int WK>
__device__ __forceinline__ void gemm_epilogue(f32x4 (&acc)[MT], float* red, void* out, int out_dt,
                                              float* partial, int M, int N, int S, int m_base,
                                              int n0) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: SGPR index math
    const int j = lane & 15, g = lane >> 4;
    auto store = [&](int mt, const f32x4& v) {
        const int m = m_base + mt * 16 + j;
        const int n = n0 + g * 4;
        if (m >= M) return;
        if (S > 1) {
            float* dst = partial + ((size_t)blockIdx.y * M + m) * N + n;
            if (n + 3 < N && (N & 3) == 0) *reinterpret_cast<f32x4*>(dst) = v;
            else
                for (int r = 0; r < 4 && n + r < N; ++r) dst[r] = v[r];
        } else if (out_dt == 2) {
            float* dst = (float*)out + (size_t)m * N + n;
            for (int r = 0; r < 4 && n + r < N; ++r) dst[r] = v[r];
        } else {
            uint16_t* dst = (uint16_t*)out + (size_t)m * N + n;
            uint16_t h[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = out_dt == 0 ? f32_to_bf16(v[r]) : f32_to_f16(v[r]);
            if (n + 3 < N && (N & 3) == 0) {
                i32x2 o;
                o[0] = (int)((uint32_t)h[0] | ((uint32_t)h[1] << 16));
                o[1] = (int)((uint32_t)h[2] | ((uint32_t)h[3] << 16));
                *reinterpret_cast<i32x2*>(dst) = o;
            } else {
                for (int r = 0; r < 4 && n + r < N; ++r) dst[r] = h[r];
            }
        }
    };
    if (WK > 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            *reinterpret_cast<f32x4*>(&red[((wave * MT + mt) * 64 + lane) * 4]) = acc[mt];
        __syncthreads();
        for (int mt = wave; mt < MT; mt += WK) {
            f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < WK; ++w) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(&red[((w * MT + mt) * 64 + lane) * 4]);
#pragma unroll
                for (int r = 0; r < 4; ++r) sum[r] += v[r];
            }
            store(mt, sum);
        }
    } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) store(mt, acc[mt]);
    }
}


// ---- full-line ("pattern 2") FP8 weight tile -------------------------------------------------
// Measured on MI355X (tools/probe_stream.hip): a wave-load that takes 64 B from each of 16 rows
// streams at <= 5.1 TB/s, one that takes a whole 128-B line from each of 8 rows at 6.5-6.8 TB/s
// (= a linear stream).  So a lane (j = lane%16, g = lane/16) loads 16 B of weight row
// n0 + 8*half + j/2 at byte ((j%2)*4 + g)*16 of the 128-wide K block: 8 lanes cover one row's line.
// MFMA A-row j then holds only half of weight row j/2 (even j: K bytes 0-63, odd j: 64-127), so the
// block dot is taken with TWO activation fragments -- x[0..63] (valid on even A-rows) and x[64..127]
// (valid on odd A-rows) -- and the halves are added in-lane: C rows 4g+{0,2} come from the "even"
// product, 4g+{1,3} from the "odd" one, i.e. out row 2g + {0,1} = e[0]+o[1], e[2]+o[3].  Twice the
// MFMAs (they have ~10x headroom here), full-line HBM reads, no cross-lane traffic.
struct W8Frag {
    i32x4 w[2];  // rows half 0 (n0 + j/2) and half 1 (n0 + 8 + j/2)
};

__device__ __forceinline__ long frag_lo(const i32x4& v) {
    return (long)(((unsigned long long)(uint32_t)v[1] << 32) | (uint32_t)v[0]);
}
__device__ __forceinline__ long frag_hi(const i32x4& v) {
    return (long)(((unsigned long long)(uint32_t)v[3] << 32) | (uint32_t)v[2]);
}

// One 16 x 16 dot over a whole 128-wide K block of the compute-shaped (prefill) GEMMs: lane (j, g) of each operand holds the two
// 16-byte chunks g and g + 4 of its row.
//   CHITU_FP8_MX = 1 (round 6): ONE v_mfma_scale_f32_16x16x128_f8f6f4 (both formats e4m3, both E8M0 scales 127 = 2^0: the
//     DeepSeek block format's fp32 scales are applied on the fp32 result as before).  The instruction multiplies 128 k values per
//     row pair where the non-scaled fp8 form multiplies 32, at twice the matrix pipe's rate per k (guide: 4.66 PF measured against
//     2.05 PF) and a quarter of the issues; which 32 k values a lane carries is free as long as both operands agree (a dot product
//     does not care), so the LDS fragments are the ones the four 16x16x32 instructions used.  fp32 accumulation inside the block
//     as before; the order of the 128 products' summation inside the instruction differs from four chained 32-wide ones.
//   CHITU_FP8_MX = 0: the four chained v_mfma_f32_16x16x32_fp8_fp8 of rounds 2-5 (A/B builds, tools/build_variant.sh).
#ifndef CHITU_FP8_MX
#define CHITU_FP8_MX 1
#endif
typedef int i32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mfma_fp8_k128(const i32x4& a0, const i32x4& a1, const i32x4& b0, const i32x4& b1) {
    const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
#if CHITU_FP8_MX
    const i32x8 a = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7), b = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, z, 0 /* A: e4m3 */, 0 /* B: e4m3 */, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
#else
    f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(frag_lo(a0), frag_lo(b0), z, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(frag_hi(a0), frag_hi(b0), d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(frag_lo(a1), frag_lo(b1), d, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(frag_hi(a1), frag_hi(b1), d, 0, 0, 0);
#endif
}

// acc += (d * a_s) * b_s for the four results of a tile -- the reference's two roundings in its order (triton_kernels.py:357,
// fused_moe.py:281) -- written on float2 halves so that it compiles to two v_pk_mul_f32 + two v_pk_fma_f32 (4 VALU issues per
// MFMA instead of the 5-8 the compiler's own pairing left; the tiled GEMMs are instruction-issue-bound, rocprof round 6).
__device__ __forceinline__ void fold_scaled(f32x4& acc, const f32x4& d, float a_s, float b_s) {
    const f32x4 t = d * a_s;  // (its result is the fma's multiplicand: contraction cannot merge the two roundings)
    acc = __builtin_elementwise_fma(t, f32x4{b_s, b_s, b_s, b_s}, acc);
}

// lane's weight pointer for K block 0: W + row*K + ((j&1)*4 + g)*16, rows clamped to n_rows-1
__device__ __forceinline__ void w8_lane_ptrs(const fp8_t* Wbase, int n0, int n_rows, int K, int j, int g,
                                             const fp8_t*& p0, const fp8_t*& p1) {
    const int off = ((j & 1) * 4 + g) * 16;
    p0 = Wbase + (size_t)min(n0 + (j >> 1), n_rows - 1) * K + off;
    p1 = Wbase + (size_t)min(n0 + 8 + (j >> 1), n_rows - 1) * K + off;
}

// Block dot of one 16-row x 128-K weight tile with one 16-token activation tile.
// x0 / x1: the lane's 16 B of x[token j][kb*128 + g*16 ..] and x[token j][kb*128 + 64 + g*16 ..].
// Returns {row 2g, row 2g+1, row 8+2g, row 8+2g+1} of (n0 + .) for token j.
__device__ __forceinline__ f32x4 w8a8_block_dot(const W8Frag& f, const i32x4& x0, const i32x4& x1) {
    const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 e0 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(frag_lo(f.w[0]), frag_lo(x0), z, 0, 0, 0);
    f32x4 o0 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(frag_lo(f.w[0]), frag_lo(x1), z, 0, 0, 0);
    f32x4 e1 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(frag_lo(f.w[1]), frag_lo(x0), z, 0, 0, 0);
    f32x4 o1 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(frag_lo(f.w[1]), frag_lo(x1), z, 0, 0, 0);
    e0 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(frag_hi(f.w[0]), frag_hi(x0), e0, 0, 0, 0);
    o0 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(frag_hi(f.w[0]), frag_hi(x1), o0, 0, 0, 0);
    e1 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(frag_hi(f.w[1]), frag_hi(x0), e1, 0, 0, 0);
    o1 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(frag_hi(f.w[1]), frag_hi(x1), o1, 0, 0, 0);
    return f32x4{e0[0] + o0[1], e0[2] + o0[3], e1[0] + o1[1], e1[2] + o1[3]};
}

// output column of accumulator element e (0..3) for lane group g, tile base n0
__device__ __forceinline__ int w8_out_col(int n0, int g, int e) { return n0 + (e >> 1) * 8 + 2 * g + (e & 1); }

// Epilogue for the full-line layout: K-split reduce over the workgroup's waves (LDS, fixed order),
// then per token two 2-column stores (columns 2g,2g+1 and 8+2g,8+2g+1 of the tile).
template <int MT, int WK>
__device__ __forceinline__ void gemm_epilogue_v2(f32x4 (&acc)[MT], float* red, void* out, int out_dt,
                                                 float* partial, int M, int N, int S, int m_base, int n0) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: SGPR index math
    const int j = lane & 15, g = lane >> 4;
    auto store = [&](int mt, const f32x4& v) {
        const int m = m_base + mt * 16 + j;
        if (m >= M) return;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n = n0 + h * 8 + 2 * g;
            const float a = v[2 * h], b = v[2 * h + 1];
            if (S > 1) {
                float* dst = partial + ((size_t)blockIdx.y * M + m) * N + n;
                if (n < N) dst[0] = a;
                if (n + 1 < N) dst[1] = b;
            } else if (out_dt == 2) {
                float* dst = (float*)out + (size_t)m * N + n;
                if (n < N) dst[0] = a;
                if (n + 1 < N) dst[1] = b;
            } else {
                uint16_t* dst = (uint16_t*)out + (size_t)m * N + n;
                const uint16_t ha = out_dt == 0 ? f32_to_bf16(a) : f32_to_f16(a);
                const uint16_t hb = out_dt == 0 ? f32_to_bf16(b) : f32_to_f16(b);
                if (n + 1 < N && (N & 1) == 0) *reinterpret_cast<uint32_t*>(dst) = (uint32_t)ha | ((uint32_t)hb << 16);
                else {
                    if (n < N) dst[0] = ha;
                    if (n + 1 < N) dst[1] = hb;
                }
            }
        }
    };
    if (WK > 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            *reinterpret_cast<f32x4*>(&red[((wave * MT + mt) * 64 + lane) * 4]) = acc[mt];
        __syncthreads();
        for (int mt = wave; mt < MT; mt += WK) {
            f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < WK; ++w) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(&red[((w * MT + mt) * 64 + lane) * 4]);
#pragma unroll
                for (int r = 0; r < 4; ++r) sum[r] += v[r];
            }
            store(mt, sum);
        }
    } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) store(mt, acc[mt]);
    }
}

// soft-fp8 decode of 8 e4m3 bytes to the bf16 values the reference multiplies with (triton_kernels.py:453-488,
// fused_moe.py:232-276): bits -> f32 by bit placement, x (scale * 2^120) in f32, one rounding to bf16.  NaN codes
// come out as +-480 * scale, like the reference's.
__device__ __forceinline__ s16x8 soft_decode8(uint32_t w0, uint32_t w1, float s2) {
    s16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t byte = ((i < 4 ? w0 : w1) >> (8 * (i & 3))) & 0xffu;
        const uint32_t bits = ((byte & 0x80u) << 24) | ((byte & 0x7fu) << 20);
        r[i] = (short)f32_to_bf16(__uint_as_float(bits) * s2);
    }
    return r;
}

// The compute-shaped (128 x 128 tile) form of the W8A8 block-scaled GEMM for prefill-sized M.  Defined in fp8_gemm_tiled.hip.
void launch_fp8_gemm_tiled(const fp8_t* a, const float* a_s, const fp8_t* b, const float* b_s, void* out, int out_dt,
                           int64_t M, int64_t N, int64_t K, hipStream_t st);

// The same for bf16 weights (router scores, unquantised linears) at prefill-sized M.  Defined in bf16_gemm_tiled.hip.
void launch_bf16_gemm_tiled(const bf16_t* x, const bf16_t* w, void* out, int out_dt, int64_t M, int64_t N, int64_t K,
                            int num_splits, float* partials, hipStream_t st);

// out[m][n] = sum_s partial[s][m][n] (s ascending), cast to out_dt.  Defined in fp8_gemm.hip.
void launch_splitk_reduce(const float* partial, void* out, int out_dt, int S, int64_t MN, hipStream_t st);

// ---------------------------------------------------------------- XCD-blocked tile order (prefill-shaped GEMMs)
// Workgroup b of a launch runs on XCD b % 8 (observed dispatch order, MI355X_MICROARCH.md: used for speed only, never for
// correctness), and every XCD has its own 4 MB L2.  A row-major walk over a (tiles_n x tiles_m) grid therefore gives each
// XCD a stripe pattern that touches EVERY weight tile row and EVERY activation tile row: each L2 is filled with the
// whole of both operands.  Here the 8 XCDs are laid out as an xm x xn grid of rectangles (the shape of 8 = xm * xn that
// minimises tile rows per rectangle, Mt + Nt) and XCD k walks only its own rectangle: its L2 sees Mt + Nt operand
// rows instead of tiles_m + tiles_n.  The launch has 8 * Mt * Nt workgroups; those whose tile falls outside the grid exit.
struct XcdTiling {
    int xm, xn, Mt, Nt;
};

__host__ __device__ inline XcdTiling xcd_tiling(int tiles_m, int tiles_n) {
    XcdTiling best{1, 8, tiles_m, (tiles_n + 7) / 8};
    int best_cost = best.Mt + best.Nt;
    for (int xm = 2; xm <= 8; xm *= 2) {
        const int xn = 8 / xm, Mt = (tiles_m + xm - 1) / xm, Nt = (tiles_n + xn - 1) / xn;
        if (Mt + Nt < best_cost) best = XcdTiling{xm, xn, Mt, Nt}, best_cost = Mt + Nt;
    }
    return best;
}

// linear workgroup id -> (tile_m, tile_n); false = no tile (padding of the launch)
__host__ __device__ inline bool xcd_tile_of(int wg, int tiles_m, int tiles_n, int& tile_m, int& tile_n) {
    const XcdTiling t = xcd_tiling(tiles_m, tiles_n);
    const int xcd = wg & 7, slot = wg >> 3;
    tile_m = (xcd / t.xn) * t.Mt + slot / t.Nt;
    tile_n = (xcd % t.xn) * t.Nt + slot % t.Nt;
    return slot < t.Mt * t.Nt && tile_m < tiles_m && tile_n < tiles_n;
}

}  // namespace chitu
