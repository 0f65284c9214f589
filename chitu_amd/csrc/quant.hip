// Block-wise FP8 (e4m3fn) activation quantisation and weight de-quantisation, gfx950.
//
// Replaces (reference, read-only):
//   chitu/triton_kernels.py:193-214   act_quant_deepseek_v3_kernel         (no eps, no clamp)
//   chitu/fused_moe.py:670-710        _per_token_group_quant_fp8           (eps + clamp)
//   chitu/triton_kernels.py:217-247   weight_dequant_deepseek_v3_kernel
//   chitu/triton_kernels.py:250-287   weight_dequant_soft_fp8_*_step_1/2   (bit-placement decode)
// All are HBM-bound byte movers: 16-B loads per lane, one 128-element group per 16 lanes,
// group max by 4 xor-shuffles, IEEE division (bit-identical to the reference's fp32 math).
#include "common.h"

namespace chitu {

template <typename T>
__device__ __forceinline__ void load8_as_f32(const T* p, float (&v)[8]);

struct bf16_tag { uint16_t v; };
struct f16_tag { uint16_t v; };

template <>
__device__ __forceinline__ void load8_as_f32<bf16_tag>(const bf16_tag* p, float (&v)[8]) {
    const i32x4 raw = *reinterpret_cast<const i32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t u = (uint32_t)raw[i];
        v[2 * i] = __uint_as_float(u << 16);
        v[2 * i + 1] = __uint_as_float(u & 0xffff0000u);
    }
}
template <>
__device__ __forceinline__ void load8_as_f32<f16_tag>(const f16_tag* p, float (&v)[8]) {
    const i32x4 raw = *reinterpret_cast<const i32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t u = (uint32_t)raw[i];
        v[2 * i] = f16_to_f32((uint16_t)(u & 0xffffu));
        v[2 * i + 1] = f16_to_f32((uint16_t)(u >> 16));
    }
}
template <>
__device__ __forceinline__ void load8_as_f32<float>(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[i] = a[i];
        v[4 + i] = b[i];
    }
}

// One 128-wide group per 16 lanes, 4 groups per wave per step.
// MODE 0: s = max|x| / 448,              y = x / s            (act_quant_deepseek_v3)
// MODE 1: s = max(max|x|, eps) / 448,    y = clamp(x / s)     (per_token_group_quant_fp8)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void act_quant_kernel(const T* __restrict__ x,
                                                        fp8_t* __restrict__ y,
                                                        float* __restrict__ s, int64_t n_groups,
                                                        float eps) {
    const int lane16 = threadIdx.x & 15;
    int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 4;
    for (; group < n_groups; group += stride) {
        float v[8];
        load8_as_f32<T>(x + group * 128 + lane16 * 8, v);
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) amax = __builtin_fmaxf(amax, __builtin_fabsf(v[i]));
        // NaN inputs: fmaxf drops NaN like tl.max's default propagate_nan=NONE.
        amax = row16_reduce_max(amax);
        if (MODE == 1) amax = __builtin_fmaxf(amax, eps);
        const float sc = amax / 448.0f;
        const i32x2 out = quant8_fp8<MODE == 1>(v, sc);
        *reinterpret_cast<i32x2*>(y + group * 128 + lane16 * 8) = out;
        if (lane16 == 0) s[group] = sc;
    }
}

// Weight dequant: y[b][m][n] = dec(x[b][m][n]) * s[b][m/128][n/128], bf16/f16/f32 out.
// SOFT=0: hardware e4m3fn decode (== x.to(float32) in the reference kernel :244).
// SOFT=1: the reference's bit placement ((b&0x80)<<24 | (b&0x7f)<<20) * (s * 2^120)
//         (triton_kernels.py:261,286) -- identical on finite codes, NaN codes give +-480*s.
template <int SOFT, int OUT_DT>
__global__ __launch_bounds__(256) void weight_dequant_kernel(const fp8_t* __restrict__ x,
                                                             const float* __restrict__ s,
                                                             void* __restrict__ y, int64_t B,
                                                             int64_t M, int64_t N) {
    const int64_t n16 = (N + 15) / 16;  // 16-byte chunks per row
    const int64_t total = B * M * n16;
    const int64_t sm = (M + 127) / 128, sn = (N + 127) / 128;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = idx % n16;
        const int64_t row = idx / n16;  // b*M + m
        const int64_t m = row % M, b = row / M;
        const int64_t n0 = c * 16;
        const float sc = s[(b * sm + m / 128) * sn + n0 / 128];
        const fp8_t* src = x + row * N + n0;
        float v[16];
        const bool full = (n0 + 16 <= N) && ((N & 15) == 0);
        uint32_t w[4] = {0, 0, 0, 0};
        if (full) {
            const i32x4 raw = *reinterpret_cast<const i32x4*>(src);
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = (uint32_t)raw[i];
        } else {
            for (int i = 0; i < 16 && n0 + i < N; ++i) w[i >> 2] |= (uint32_t)src[i] << (8 * (i & 3));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (SOFT) {
                const float s2 = sc * __uint_as_float(0x7B800000u);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t byte = (w[i] >> (8 * k)) & 0xffu;
                    const uint32_t bits = ((byte & 0x80u) << 24) | ((byte & 0x7fu) << 20);
                    v[4 * i + k] = __uint_as_float(bits) * s2;
                }
            } else {
                v[4 * i + 0] = fp8_to_f32<0>(w[i]) * sc;
                v[4 * i + 1] = fp8_to_f32<1>(w[i]) * sc;
                v[4 * i + 2] = fp8_to_f32<2>(w[i]) * sc;
                v[4 * i + 3] = fp8_to_f32<3>(w[i]) * sc;
            }
        }
        if (OUT_DT == 2) {
            float* dst = (float*)y + row * N + n0;
            for (int i = 0; i < 16 && n0 + i < N; ++i) dst[i] = v[i];
        } else {
            uint16_t* dst = (uint16_t*)y + row * N + n0;
            uint16_t h[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) h[i] = OUT_DT == 0 ? f32_to_bf16(v[i]) : f32_to_f16(v[i]);
            if (full) {
                i32x4 o0, o1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    o0[i] = (int)((uint32_t)h[2 * i] | ((uint32_t)h[2 * i + 1] << 16));
                    o1[i] = (int)((uint32_t)h[8 + 2 * i] | ((uint32_t)h[8 + 2 * i + 1] << 16));
                }
                *reinterpret_cast<i32x4*>(dst) = o0;
                *reinterpret_cast<i32x4*>(dst + 8) = o1;
            } else {
                for (int i = 0; i < 16 && n0 + i < N; ++i) dst[i] = h[i];
            }
        }
    }
}

static inline int quant_grid(int64_t work_items, int threads) {
    int64_t blocks = (work_items + threads - 1) / threads;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    return (int)blocks;
}

}  // namespace chitu

extern "C" int chitu_hip_act_quant_fp8(const void* x, int act_dtype, int64_t rows, int64_t cols,
                                       int32_t group_size, int32_t mode, float eps, void* y_fp8,
                                       float* scales, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(x && y_fp8 && scales);
    CHITU_REQUIRE(rows >= 0 && cols >= 0);
    if (group_size != 128) return CHITU_ERR_UNSUPPORTED;
    CHITU_REQUIRE(cols % 128 == 0);  // reference asserts divisibility (ops.py:345-348)
    const int64_t groups = rows * (cols / 128);
    if (groups == 0) return CHITU_OK;
    hipStream_t st = (hipStream_t)stream;
    const int grid = quant_grid(groups * 16, 256);
#define LAUNCH(T, MODE)                                                                       \
    hipLaunchKernelGGL((act_quant_kernel<T, MODE>), dim3(grid), dim3(256), 0, st, (const T*)x, \
                       (fp8_t*)y_fp8, scales, groups, eps)
    if (mode == 0) {
        if (act_dtype == 0) LAUNCH(bf16_tag, 0);
        else if (act_dtype == 1) LAUNCH(f16_tag, 0);
        else if (act_dtype == 2) LAUNCH(float, 0);
        else return CHITU_ERR_UNSUPPORTED;
    } else if (mode == 1) {
        if (act_dtype == 0) LAUNCH(bf16_tag, 1);
        else if (act_dtype == 1) LAUNCH(f16_tag, 1);
        else if (act_dtype == 2) LAUNCH(float, 1);
        else return CHITU_ERR_UNSUPPORTED;
    } else {
        return CHITU_ERR_UNSUPPORTED;
    }
#undef LAUNCH
    CHITU_RETURN_LAUNCH_STATUS();
}

extern "C" int chitu_hip_weight_dequant_fp8(const void* w_fp8, const float* scales, int64_t batch,
                                            int64_t rows, int64_t cols, int32_t block_size,
                                            int32_t soft, int out_dtype, void* y, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(w_fp8 && scales && y);
    CHITU_REQUIRE(batch >= 0 && rows >= 0 && cols >= 0);
    if (block_size != 128) return CHITU_ERR_UNSUPPORTED;
    if (batch * rows * cols == 0) return CHITU_OK;
    hipStream_t st = (hipStream_t)stream;
    const int grid = quant_grid(batch * rows * ((cols + 15) / 16), 256);
#define LAUNCH(SOFT, DT)                                                                   \
    hipLaunchKernelGGL((weight_dequant_kernel<SOFT, DT>), dim3(grid), dim3(256), 0, st,    \
                       (const fp8_t*)w_fp8, scales, y, batch, rows, cols)
    if (soft) {
        if (out_dtype == 0) LAUNCH(1, 0);
        else if (out_dtype == 1) LAUNCH(1, 1);
        else if (out_dtype == 2) LAUNCH(1, 2);
        else return CHITU_ERR_UNSUPPORTED;
    } else {
        if (out_dtype == 0) LAUNCH(0, 0);
        else if (out_dtype == 1) LAUNCH(0, 1);
        else if (out_dtype == 2) LAUNCH(0, 2);
        else return CHITU_ERR_UNSUPPORTED;
    }
#undef LAUNCH
    CHITU_RETURN_LAUNCH_STATUS();
}

// ---- arithmetic self-test (tests/test_gpu_fp8.py): the two shortcuts every quantising kernel
// relies on, checked on the device against their slow definitions over caller-supplied operands.
namespace chitu {
__global__ __launch_bounds__(256) void selftest_arith_kernel(const float* __restrict__ num, const float* __restrict__ den,
                                                             int64_t n, unsigned long long* __restrict__ bad) {
    unsigned long long b0 = 0, b1 = 0, b2 = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float x = num[i], d = den[i];
        if (group_div_fast(d)) {
            const float fast = group_div(x, d, group_rcp(d)), slow = x / d;
            b0 += __float_as_uint(fast) != __float_as_uint(slow);
            b1 += f32x2_to_fp8x2_sat(fast, fast) != f32x2_to_fp8x2_sat(slow, slow);
        }
        b2 += f32_to_bf16(x) != f32_to_bf16_sw(x) && x == x;
    }
    if (b0) atomicAdd(bad + 0, b0);
    if (b1) atomicAdd(bad + 1, b1);
    if (b2) atomicAdd(bad + 2, b2);
}
}  // namespace chitu

extern "C" int chitu_hip_selftest_arith(const float* num, const float* den, int64_t n, uint64_t* mismatches,
                                        void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(num && den && mismatches && n >= 0);
    if (n == 0) return CHITU_OK;
    hipLaunchKernelGGL(selftest_arith_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, num, den, n,
                       (unsigned long long*)mismatches);
    CHITU_RETURN_LAUNCH_STATUS();
}

// ---- SiluAndMul for unquantised (bf16) MLPs: out = bf16(bf16(silu(x[:, :d])) * x[:, d:])
// (chitu/fused_moe.py:24-39 / FeedForward.forward, models/model.py:212-214: F.silu(w1 x) * w3 x on bf16
// tensors -- each torch op rounds once).
namespace chitu {
__global__ __launch_bounds__(256) void silu_and_mul_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out,
                                                           int64_t rows, int d) {
    const int chunks = d >> 3;
    const int64_t total = rows * chunks;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / chunks;
        const int c = (int)(i % chunks) * 8;
        const i32x4 g = *reinterpret_cast<const i32x4*>(x + r * 2 * d + c);
        const i32x4 u = *reinterpret_cast<const i32x4*>(x + r * 2 * d + d + c);
        i32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t gu = (uint32_t)g[k], uu = (uint32_t)u[k];
            const float g0 = __uint_as_float(gu << 16), g1 = __uint_as_float(gu & 0xffff0000u);
            const float s0 = round_bf16(g0 / (1.0f + expf(-g0))), s1 = round_bf16(g1 / (1.0f + expf(-g1)));
            o[k] = (int)f32x2_to_bf16x2(s0 * __uint_as_float(uu << 16), s1 * __uint_as_float(uu & 0xffff0000u));
        }
        *reinterpret_cast<i32x4*>(out + r * d + c) = o;
    }
}
}  // namespace chitu

extern "C" int chitu_hip_silu_and_mul(const void* x_bf16, void* out_bf16, int64_t rows, int64_t d, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(x_bf16 && out_bf16 && rows >= 0 && d >= 8 && d < (1ll << 31));
    if (d % 8 != 0) return CHITU_ERR_UNSUPPORTED;
    if (rows == 0) return CHITU_OK;
    int64_t blocks = (rows * (d / 8) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(silu_and_mul_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x_bf16, (bf16_t*)out_bf16, rows, (int)d);
    CHITU_RETURN_LAUNCH_STATUS();
}

