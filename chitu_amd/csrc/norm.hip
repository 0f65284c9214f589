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

// Residual add in front of the norm, the WIDE row form (norm_common.h): add[r] = one row (MAXT == 1) or
// bf16(sum_k float(add[r, k, :])) over `terms` rows (MAXT == 16: the fused MoE's top-k sum folded into the
// norm that consumes it, = chitu_hip_moe_sum, one rounding), x_new = bf16(x + add[r]), then the norm.
// 1024 threads per row, one 8-element chunk per thread (dim <= 8192), so the terms + x + weight loads of a
// row are one memory round trip.
template <int QMODE, int MAXT>
__global__ __launch_bounds__(kNormWideThreads) void rmsnorm_add_kernel(
    const bf16_t* x, int64_t x_stride, const bf16_t* add, int64_t add_stride, int terms, int64_t term_stride,
    bf16_t* sum_out, int64_t sum_stride, const bf16_t* __restrict__ w, bf16_t* y, int64_t y_stride,
    fp8_t* __restrict__ q, float* __restrict__ qs, int dim, float eps, float qeps, int tile_major) {
    __shared__ float red[kNormWideThreads / 64];
    __shared__ float red2[QMODE == 3 ? kNormWideThreads / 64 : 1];
    if (QMODE == 2 && MAXT == 1) CHITU_PROBE_MARK(0);
    const int row = blockIdx.x, tid = threadIdx.x;
    const int n_chunks = dim >> 3;
    const bool act = tid < n_chunks;
    const int c = min(tid, n_chunks - 1);
    const i32x4 xraw = *reinterpret_cast<const i32x4*>(x + (int64_t)row * x_stride + c * 8);
    const i32x4 wraw = *reinterpret_cast<const i32x4*>(w + c * 8);
    i32x4 traw[MAXT];
#pragma unroll
    for (int k = 0; k < MAXT; ++k)
        traw[k] = *reinterpret_cast<const i32x4*>(add + (int64_t)row * add_stride + (int64_t)min(k, terms - 1) * term_stride + c * 8);
    const i32x4 a = MAXT == 1 ? traw[0] : sum_terms_bf16x8<MAXT>(traw, terms);
    float v[8];
    i32x4 sraw;
    add_bf16x8(xraw, a, v, sraw);
    if (act && sum_out) *reinterpret_cast<i32x4*>(sum_out + (int64_t)row * sum_stride + tid * 8) = sraw;
    if (QMODE == 2 && MAXT == 1) {
        if (v[0] == 1.2345e30f) CHITU_PROBE_MARK(9);
        CHITU_PROBE_MARK(1);  // inputs arrived, residual added
    }
    rmsnorm_wide_finish<QMODE>(v, act, row, wraw, y, y_stride, q, qs, dim, eps, qeps, red, tile_major, red2);
    if (QMODE == 2 && MAXT == 1) CHITU_PROBE_MARK(2);
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
    if (dim % 8 != 0 || dim > kNormThreads * 8 * kNormMaxChunks || dim > kNormWideThreads * 8) return CHITU_ERR_UNSUPPORTED;
    CHITU_REQUIRE(x_row_stride % 8 == 0 && (!y_bf16 || y_row_stride % 8 == 0));
    CHITU_REQUIRE((!add_bf16 || add_row_stride % 8 == 0) && (!sum_out_bf16 || (add_bf16 && sum_row_stride % 8 == 0)));
    // quant_mode + 4: the same codes and scales written TILE-MAJOR (see the header); only with a residual add (wide form)
    const int tile_major = (quant_mode & 4) ? 1 : 0;
    quant_mode &= 3;
    if (quant_mode != 0) {
        CHITU_REQUIRE(q_fp8 && q_scales);
        if (quant_mode != 3 && dim % 128 != 0) return CHITU_ERR_UNSUPPORTED;
        // 3 = per-token int8 (q: int8 codes, q_scales [rows]): the residual-add form with one term only
        CHITU_REQUIRE(quant_mode != 3 || (add_bf16 && add_terms == 1 && !tile_major));
    }
    CHITU_REQUIRE(!tile_major || (quant_mode != 0 && add_bf16));
    CHITU_REQUIRE(add_terms >= 1 && (add_terms == 1 || (add_bf16 && add_term_stride % 8 == 0)));
    if (add_terms > kNormMaxTerms) return CHITU_ERR_UNSUPPORTED;
    if (rows == 0) return CHITU_OK;
    hipStream_t st = (hipStream_t)stream;
    if (add_bf16) {
#define LAUNCHA(QM, MT)                                                                                    \
    hipLaunchKernelGGL((rmsnorm_add_kernel<QM, MT>), dim3((unsigned)rows), dim3(kNormWideThreads), 0, st,  \
                       (const bf16_t*)x_bf16, x_row_stride, (const bf16_t*)add_bf16, add_row_stride,       \
                       (int)add_terms, add_term_stride, (bf16_t*)sum_out_bf16, sum_row_stride,             \
                       (const bf16_t*)weight_bf16, (bf16_t*)y_bf16, y_row_stride, (fp8_t*)q_fp8, q_scales, \
                       (int)dim, eps, quant_eps, tile_major)
        if (add_terms == 1) {
            if (quant_mode == 0) LAUNCHA(0, 1);
            else if (quant_mode == 1) LAUNCHA(1, 1);
            else if (quant_mode == 3) LAUNCHA(3, 1);
            else LAUNCHA(2, 1);
        } else {
            if (quant_mode == 0) LAUNCHA(0, kNormMaxTerms);
            else if (quant_mode == 1) LAUNCHA(1, kNormMaxTerms);
            else LAUNCHA(2, kNormMaxTerms);
        }
#undef LAUNCHA
        CHITU_RETURN_LAUNCH_STATUS();
    }
#define LAUNCH(QM, AD)                                                                           \
    hipLaunchKernelGGL((rmsnorm_kernel<QM, AD>), dim3((unsigned)rows), dim3(kNormThreads), 0, st, \
                       (const bf16_t*)x_bf16, x_row_stride, (const bf16_t*)add_bf16, add_row_stride, \
                       (bf16_t*)sum_out_bf16, sum_row_stride, (const bf16_t*)weight_bf16,         \
                       (bf16_t*)y_bf16, y_row_stride, (fp8_t*)q_fp8, q_scales, (int)dim, eps, quant_eps)
    if (quant_mode == 0) LAUNCH(0, false);
    else if (quant_mode == 1) LAUNCH(1, false);
    else LAUNCH(2, false);
#undef LAUNCH
    CHITU_RETURN_LAUNCH_STATUS();
}

CHITU_PROBE_READER(norm)
