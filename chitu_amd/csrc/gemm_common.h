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
    const int wave = threadIdx.x >> 6;
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


// out[m][n] = sum_s partial[s][m][n] (s ascending), cast to out_dt.  Defined in fp8_gemm.hip.
void launch_splitk_reduce(const float* partial, void* out, int out_dt, int S, int64_t MN, hipStream_t st);

}  // namespace chitu
