// RMSNorm for the decode path, optionally fused with the FP8 block quantisation of its output.
//
// Replaces (reference, read-only):
//   chitu/models/model.py:29-78   RMSNorm.forward -> F.rms_norm(x.to(compute_dtype), w, eps).to(dtype)
//   + the act_quant_deepseek_v3 launch that follows it in linear_deepseek_v3
//     (chitu/models/model_deepseek_v3.py:98-100, kernel chitu/triton_kernels.py:193-214)
// Optional residual: x <- bf16(x + add) first (also written to sum_out).
// Math: y = (x * rsqrt(mean(x^2) + eps)) * w in fp32, one rounding to the output dtype -- this is
// bit-identical to torch's rms_norm on bf16 input (checked in tests).  The optional second output
// is the e4m3 quantisation of that *rounded* y (what the reference's next kernel would compute),
// so the fused form changes no numerics, it only removes a launch and a round trip through HBM.
// One workgroup per row, 16-B loads, 8 elements per lane per chunk; 16 consecutive lanes own one
// 128-wide quantisation group, exactly as in quant.hip.
#include "common.h"
#include "norm_common.h"

namespace chitu {

template <int QMODE, bool ADD>
__global__ __launch_bounds__(kNormThreads) void rmsnorm_kernel(
    const bf16_t* x, int64_t x_stride, const bf16_t* add, int64_t add_stride, bf16_t* sum_out,
    int64_t sum_stride, const bf16_t* __restrict__ w, bf16_t* y, int64_t y_stride,
    fp8_t* __restrict__ q, float* __restrict__ qs, int dim, float eps, float qeps) {
    rmsnorm_row<QMODE, ADD>(blockIdx.x, x, x_stride, add, add_stride, sum_out, sum_stride, w, y, y_stride, q, qs, dim,
                            eps, qeps);
}

// Residual given as `terms` rows to be summed first -- the top-k sum of the fused MoE folded into
// the norm that consumes it: add[r] = bf16(sum_k float(add[r, k, :])) (= chitu_hip_moe_sum, one
// rounding), then exactly rmsnorm_kernel<QMODE, true>.  1024 threads per row, one 8-element chunk
// per thread (dim <= 8192) so the terms + x + weight loads of a row are one memory round trip.
constexpr int kNormMultiThreads = 1024;
constexpr int kNormMaxTerms = 16;
template <int QMODE>
__global__ __launch_bounds__(kNormMultiThreads) void rmsnorm_multi_kernel(
    const bf16_t* x, int64_t x_stride, const bf16_t* add, int64_t add_stride, int terms, int64_t term_stride,
    bf16_t* sum_out, int64_t sum_stride, const bf16_t* __restrict__ w, bf16_t* y, int64_t y_stride,
    fp8_t* __restrict__ q, float* __restrict__ qs, int dim, float eps, float qeps) {
    __shared__ float red[kNormMultiThreads / 64];
    const int row = blockIdx.x, tid = threadIdx.x;
    const int n_chunks = dim >> 3;
    const bool act = tid < n_chunks;
    const int c = min(tid, n_chunks - 1);
    const i32x4 xraw = *reinterpret_cast<const i32x4*>(x + (int64_t)row * x_stride + c * 8);
    const i32x4 wraw = *reinterpret_cast<const i32x4*>(w + c * 8);
    i32x4 traw[kNormMaxTerms];
#pragma unroll
    for (int k = 0; k < kNormMaxTerms; ++k)
        traw[k] = *reinterpret_cast<const i32x4*>(add + (int64_t)row * add_stride + (int64_t)min(k, terms - 1) * term_stride + c * 8);
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < kNormMaxTerms; ++k) {
        if (k < terms) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t u = (uint32_t)traw[k][i];
                a[2 * i] += __uint_as_float(u << 16);
                a[2 * i + 1] += __uint_as_float(u & 0xffff0000u);
            }
        }
    }
    float v[8], ss = 0.f;
    i32x4 sraw;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t u = (uint32_t)xraw[i];
        const uint32_t t2 = f32x2_to_bf16x2(a[2 * i], a[2 * i + 1]);  // the moe_sum rounding
        const uint32_t s2 = f32x2_to_bf16x2(__uint_as_float(u << 16) + __uint_as_float(t2 << 16),
                                            __uint_as_float(u & 0xffff0000u) + __uint_as_float(t2 & 0xffff0000u));
        v[2 * i] = __uint_as_float(s2 << 16);
        v[2 * i + 1] = __uint_as_float(s2 & 0xffff0000u);
        sraw[i] = (int)s2;
    }
    if (act) {
#pragma unroll
        for (int i = 0; i < 8; ++i) ss += v[i] * v[i];
        if (sum_out) *reinterpret_cast<i32x4*>(sum_out + (int64_t)row * sum_stride + tid * 8) = sraw;
    }
    ss = wave_reduce_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    ss = 0.f;
#pragma unroll
    for (int i = 0; i < kNormMultiThreads / 64; ++i) ss += red[i];
    const float rr = rsqrtf(ss / (float)dim + eps);
    float o[8];
    i32x4 out;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t u = (uint32_t)wraw[i];
        const uint32_t h2 = f32x2_to_bf16x2((v[2 * i] * rr) * __uint_as_float(u << 16),
                                            (v[2 * i + 1] * rr) * __uint_as_float(u & 0xffff0000u));
        out[i] = (int)h2;
        o[2 * i] = act ? __uint_as_float(h2 << 16) : 0.f;
        o[2 * i + 1] = act ? __uint_as_float(h2 & 0xffff0000u) : 0.f;
    }
    if (y && act) *reinterpret_cast<i32x4*>(y + (int64_t)row * y_stride + tid * 8) = out;
    if (QMODE != 0) {
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) amax = __builtin_fmaxf(amax, __builtin_fabsf(o[i]));
        amax = row16_reduce_max(amax);
        if (QMODE == 2) amax = __builtin_fmaxf(amax, qeps);
        const float sc = amax / 448.0f;
        const i32x2 packed = quant8_fp8<QMODE == 2>(o, act ? sc : 1.0f);
        if (act) {
            *reinterpret_cast<i32x2*>(q + (int64_t)row * dim + tid * 8) = packed;
            if ((tid & 15) == 0) qs[(int64_t)row * (dim >> 7) + (tid >> 4)] = sc;
        }
    }
}

}  // namespace chitu

extern "C" int chitu_hip_rmsnorm(const void* x_bf16, int64_t x_row_stride, const void* add_bf16,
                                 int64_t add_row_stride, int32_t add_terms, int64_t add_term_stride,
                                 void* sum_out_bf16, int64_t sum_row_stride,
                                 const void* weight_bf16, void* y_bf16, int64_t y_row_stride,
                                 int64_t rows, int32_t dim, float eps, void* q_fp8, float* q_scales,
                                 int32_t quant_mode, float quant_eps, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(x_bf16 && weight_bf16 && rows >= 0 && dim >= 8);
    CHITU_REQUIRE(y_bf16 || quant_mode != 0);
    if (dim % 8 != 0 || dim > kNormThreads * 8 * kNormMaxChunks) return CHITU_ERR_UNSUPPORTED;
    CHITU_REQUIRE(x_row_stride % 8 == 0 && (!y_bf16 || y_row_stride % 8 == 0));
    CHITU_REQUIRE((!add_bf16 || add_row_stride % 8 == 0) && (!sum_out_bf16 || (add_bf16 && sum_row_stride % 8 == 0)));
    if (quant_mode != 0) {
        CHITU_REQUIRE(q_fp8 && q_scales);
        if (dim % 128 != 0) return CHITU_ERR_UNSUPPORTED;
        CHITU_REQUIRE(quant_mode == 1 || quant_mode == 2);
    }
    CHITU_REQUIRE(add_terms >= 1 && (add_terms == 1 || (add_bf16 && add_term_stride % 8 == 0)));
    if (add_terms > kNormMaxTerms) return CHITU_ERR_UNSUPPORTED;
    if (rows == 0) return CHITU_OK;
    hipStream_t st = (hipStream_t)stream;
    if (add_terms > 1) {
#define LAUNCHM(QM)                                                                                      \
    hipLaunchKernelGGL(rmsnorm_multi_kernel<QM>, dim3((unsigned)rows), dim3(kNormMultiThreads), 0, st,   \
                       (const bf16_t*)x_bf16, x_row_stride, (const bf16_t*)add_bf16, add_row_stride,     \
                       (int)add_terms, add_term_stride, (bf16_t*)sum_out_bf16, sum_row_stride,           \
                       (const bf16_t*)weight_bf16, (bf16_t*)y_bf16, y_row_stride, (fp8_t*)q_fp8, q_scales, \
                       (int)dim, eps, quant_eps)
        if (quant_mode == 0) LAUNCHM(0);
        else if (quant_mode == 1) LAUNCHM(1);
        else LAUNCHM(2);
#undef LAUNCHM
        CHITU_RETURN_LAUNCH_STATUS();
    }
#define LAUNCH(QM, AD)                                                                           \
    hipLaunchKernelGGL((rmsnorm_kernel<QM, AD>), dim3((unsigned)rows), dim3(kNormThreads), 0, st, \
                       (const bf16_t*)x_bf16, x_row_stride, (const bf16_t*)add_bf16, add_row_stride, \
                       (bf16_t*)sum_out_bf16, sum_row_stride, (const bf16_t*)weight_bf16,         \
                       (bf16_t*)y_bf16, y_row_stride, (fp8_t*)q_fp8, q_scales, (int)dim, eps, quant_eps)
    if (add_bf16) {
        if (quant_mode == 0) LAUNCH(0, true);
        else if (quant_mode == 1) LAUNCH(1, true);
        else LAUNCH(2, true);
    } else {
        if (quant_mode == 0) LAUNCH(0, false);
        else if (quant_mode == 1) LAUNCH(1, false);
        else LAUNCH(2, false);
    }
#undef LAUNCH
    CHITU_RETURN_LAUNCH_STATUS();
}
