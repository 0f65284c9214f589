// Per-token / per-channel INT8 W8A8 linear for decode on gfx950.
//
// Replaces (reference, read-only):
//   chitu/quantize/w8a8.py:18-26    quant_act   (s = clamp(max|x_row|, 1e-5) / 127, q = round(x / s))
//   chitu/quantize/w8a8.py:97-132   W8A8Linear.forward -> w8a8gemm.mm / w8a8gemv.mv
// The arithmetic of `w8a8gemm` / `w8a8gemv` lives in a closed, un-vendored package
// (third_party/nv_w8a8_kernels/README.md:1); its contract is pinned only by the call sites and by
// test/pytest/test_w8a8.py:13-48:  out_fp16[m][n] = (sum_k q_x[m][k] * q_w[n][k]) * s_act[m] * s_w[n] (+ bias).
// Here: exact int32 dot on v_mfma_i32_16x16x64_i8, scaled in fp32 as (float(acc) * s_act) * s_w.
//
// Same weight-streaming shape as fp8_gemm.hip (full-line layout, gemm_common.h): a lane's 16 B of a
// weight row ARE one 16x16x64 A fragment (16 int8), so a 128-wide K block costs two loads + four MFMAs
// per 16 rows; no per-block scales, so the int32 accumulators run across the whole K range.
#include "common.h"
#include "gemm_common.h"

namespace chitu {

typedef int i32x4v __attribute__((ext_vector_type(4)));

struct bf16_in { uint16_t v; };
struct f16_in { uint16_t v; };
__device__ __forceinline__ float ldf(const bf16_in* p) { return bf16_to_f32(p->v); }
__device__ __forceinline__ float ldf(const f16_in* p) { return f16_to_f32(p->v); }
__device__ __forceinline__ float ldf(const float* p) { return *p; }

// one workgroup per row: q[row][:] = int8(round(x / s)), s = clamp(max|x|, 1e-5) / 127
template <typename T>
__global__ __launch_bounds__(256) void quant_act_int8_kernel(const T* __restrict__ x, int8_t* __restrict__ q,
                                                             float* __restrict__ s, int K) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const T* xr = x + (int64_t)row * K;
    float amax = 0.f;
    for (int i = tid; i < K; i += 256) amax = __builtin_fmaxf(amax, __builtin_fabsf(ldf(xr + i)));
    amax = wave_reduce_max(amax);
    if ((tid & 63) == 0) red[tid >> 6] = amax;
    __syncthreads();
    amax = __builtin_fmaxf(__builtin_fmaxf(red[0], red[1]), __builtin_fmaxf(red[2], red[3]));
    const float sc = __builtin_fmaxf(amax, 1e-5f) / 127.0f;
    for (int i = tid; i < K; i += 256) {
        float v = rintf(ldf(xr + i) / sc);  // round-half-even like torch.round
        v = __builtin_fminf(__builtin_fmaxf(v, -128.f), 127.f);
        q[(int64_t)row * K + i] = (int8_t)v;
    }
    if (tid == 0) s[row] = sc;
}

// 16-bit inputs, K % 8 == 0, K <= 16384: the row is read ONCE with 16-B loads (all in flight together) and
// quantised from registers -- same arithmetic as quant_act_int8_kernel.
template <bool IS_BF16>
__global__ __launch_bounds__(256) void quant_act_int8_vec_kernel(const uint16_t* __restrict__ x, int8_t* __restrict__ q,
                                                                 float* __restrict__ s, int K) {
    __shared__ float red[4];
    constexpr int CH = 8;
    const int row = blockIdx.x, tid = threadIdx.x;
    const int n_chunks = K >> 3;
    const uint16_t* xr = x + (int64_t)row * K;
    i32x4v raw[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) raw[c] = *reinterpret_cast<const i32x4v*>(xr + (size_t)min(tid + c * 256, n_chunks - 1) * 8);
    float v[CH][8];
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const bool act = tid + c * 256 < n_chunks;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t u = (uint32_t)raw[c][k];
            v[c][2 * k] = IS_BF16 ? __uint_as_float(u << 16) : f16_to_f32((uint16_t)(u & 0xffffu));
            v[c][2 * k + 1] = IS_BF16 ? __uint_as_float(u & 0xffff0000u) : f16_to_f32((uint16_t)(u >> 16));
        }
        if (act) {
#pragma unroll
            for (int k = 0; k < 8; ++k) amax = __builtin_fmaxf(amax, __builtin_fabsf(v[c][k]));
        }
    }
    amax = wave_reduce_max(amax);
    if ((tid & 63) == 0) red[tid >> 6] = amax;
    __syncthreads();
    amax = __builtin_fmaxf(__builtin_fmaxf(red[0], red[1]), __builtin_fmaxf(red[2], red[3]));
    const float sc = __builtin_fmaxf(amax, 1e-5f) / 127.0f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        if (tid + c * 256 >= n_chunks) continue;
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float r = rintf(v[c][k] / sc);
            r = __builtin_fminf(__builtin_fmaxf(r, -128.f), 127.f);
            const uint32_t b = (uint32_t)(uint8_t)(int8_t)r;
            if (k < 4) lo |= b << (8 * k); else hi |= b << (8 * (k - 4));
        }
        i32x2 o;
        o[0] = (int)lo;
        o[1] = (int)hi;
        *reinterpret_cast<i32x2*>(q + (int64_t)row * K + (size_t)(tid + c * 256) * 8) = o;
    }
    if (tid == 0) s[row] = sc;
}

// grid (N/16); block 64*WK.  MT token tiles of 16.
template <int MT, int WK>
__global__ __launch_bounds__(64 * WK) void w8a8_int8_gemm_kernel(
    const int8_t* __restrict__ X, const float* __restrict__ XS, const int8_t* __restrict__ W,
    const float* __restrict__ WS, const void* __restrict__ bias, int bias_dt, void* __restrict__ out,
    int out_dt, int M, int N, int K, int m_base) {
    __shared__ int red[WK > 1 ? WK * MT * 256 : 1];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: SGPR index math
    const int j = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int KB = K >> 7;
    const int kb0 = KB * wave / WK, kb1 = KB * (wave + 1) / WK;
    const fp8_t *wp0, *wp1;
    w8_lane_ptrs(reinterpret_cast<const fp8_t*>(W), n0, N, K, j, g, wp0, wp1);
    const int8_t* xp[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xp[mt] = X + (size_t)min(m_base + mt * 16 + j, M - 1) * K + g * 16;

    i32x4v e0[MT], o0[MT], e1[MT], o1[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) e0[mt] = o0[mt] = e1[mt] = o1[mt] = i32x4v{0, 0, 0, 0};

    for (int kb = kb0; kb < kb1; ++kb) {
        const int off = kb << 7;
        const i32x4v w0 = __builtin_nontemporal_load(reinterpret_cast<const i32x4v*>(wp0 + off));
        const i32x4v w1 = __builtin_nontemporal_load(reinterpret_cast<const i32x4v*>(wp1 + off));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const i32x4v x0 = *reinterpret_cast<const i32x4v*>(xp[mt] + off);
            const i32x4v x1 = *reinterpret_cast<const i32x4v*>(xp[mt] + off + 64);
            e0[mt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w0, x0, e0[mt], 0, 0, 0);
            o0[mt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w0, x1, o0[mt], 0, 0, 0);
            e1[mt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w1, x0, e1[mt], 0, 0, 0);
            o1[mt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w1, x1, o1[mt], 0, 0, 0);
        }
    }
    // in-lane combine of the even/odd half-row products (gemm_common.h), exact in int32
    i32x4v acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        acc[mt] = i32x4v{e0[mt][0] + o0[mt][1], e0[mt][2] + o0[mt][3], e1[mt][0] + o1[mt][1], e1[mt][2] + o1[mt][3]};
    auto store = [&](int mt, const i32x4v& v) {
        const int m = m_base + mt * 16 + j;
        if (m >= M) return;
        const float sa = XS[m];
#pragma unroll
        for (int el = 0; el < 4; ++el) {
            const int n = w8_out_col(n0, g, el);
            if (n >= N) continue;
            float r = ((float)v[el] * sa) * WS[n];
            if (bias) r += bias_dt == 0 ? bf16_to_f32(((const bf16_t*)bias)[n]) : bias_dt == 1 ? f16_to_f32(((const uint16_t*)bias)[n]) : ((const float*)bias)[n];
            if (out_dt == 2) ((float*)out)[(size_t)m * N + n] = r;
            else ((uint16_t*)out)[(size_t)m * N + n] = out_dt == 0 ? f32_to_bf16(r) : f32_to_f16(r);
        }
    };
    if (WK > 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            *reinterpret_cast<i32x4v*>(&red[((wave * MT + mt) * 64 + lane) * 4]) = acc[mt];
        __syncthreads();
        for (int mt = wave; mt < MT; mt += WK) {
            i32x4v sum = i32x4v{0, 0, 0, 0};
#pragma unroll
            for (int w = 0; w < WK; ++w) {
                const i32x4v v = *reinterpret_cast<const i32x4v*>(&red[((w * MT + mt) * 64 + lane) * 4]);
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

}  // namespace chitu

extern "C" int chitu_hip_quant_act_int8(const void* x, int act_dtype, int64_t rows, int64_t cols,
                                        void* q_int8, float* scales, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(x && q_int8 && scales && rows >= 0 && cols >= 1 && cols < (1ll << 31));
    if (rows == 0) return CHITU_OK;
    hipStream_t st = (hipStream_t)stream;
    if (act_dtype <= 1 && cols % 8 == 0 && cols <= 16384) {
        if (act_dtype == 0)
            hipLaunchKernelGGL(quant_act_int8_vec_kernel<true>, dim3((unsigned)rows), dim3(256), 0, st, (const uint16_t*)x, (int8_t*)q_int8, scales, (int)cols);
        else
            hipLaunchKernelGGL(quant_act_int8_vec_kernel<false>, dim3((unsigned)rows), dim3(256), 0, st, (const uint16_t*)x, (int8_t*)q_int8, scales, (int)cols);
        CHITU_RETURN_LAUNCH_STATUS();
    }
    if (act_dtype == 0)
        hipLaunchKernelGGL(quant_act_int8_kernel<bf16_in>, dim3((unsigned)rows), dim3(256), 0, st, (const bf16_in*)x, (int8_t*)q_int8, scales, (int)cols);
    else if (act_dtype == 1)
        hipLaunchKernelGGL(quant_act_int8_kernel<f16_in>, dim3((unsigned)rows), dim3(256), 0, st, (const f16_in*)x, (int8_t*)q_int8, scales, (int)cols);
    else if (act_dtype == 2)
        hipLaunchKernelGGL(quant_act_int8_kernel<float>, dim3((unsigned)rows), dim3(256), 0, st, (const float*)x, (int8_t*)q_int8, scales, (int)cols);
    else return CHITU_ERR_UNSUPPORTED;
    CHITU_RETURN_LAUNCH_STATUS();
}

extern "C" int chitu_hip_w8a8_int8_gemm(const void* a_int8, const float* a_scale, const void* b_int8,
                                        const float* b_scale, const void* bias, int bias_dtype, void* out,
                                        int out_dtype, int64_t M, int64_t N, int64_t K, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(a_int8 && a_scale && b_int8 && b_scale && out);
    CHITU_REQUIRE(M >= 0 && N >= 1 && K >= 128 && N < (1 << 30) && K < (1 << 30));
    CHITU_REQUIRE(out_dtype >= 0 && out_dtype <= 2 && bias_dtype >= 0 && bias_dtype <= 2);
    if (K % 128 != 0) return CHITU_ERR_UNSUPPORTED;
    if (M == 0) return CHITU_OK;
    hipStream_t st = (hipStream_t)stream;
    const int tiles = (int)((N + 15) / 16);
    const int KB = (int)(K / 128);
    int WK = tiles >= 1024 ? 2 : tiles >= 384 ? 4 : 8;
    while (WK > 1 && WK > KB) WK >>= 1;
    const dim3 grid((unsigned)tiles);
#define LAUNCH(MT, WKV)                                                                                   \
    hipLaunchKernelGGL((w8a8_int8_gemm_kernel<MT, WKV>), grid, dim3(64 * WKV), 0, st, (const int8_t*)a_int8, \
                       a_scale, (const int8_t*)b_int8, b_scale, bias, bias_dtype, out, out_dtype, (int)M,    \
                       (int)N, (int)K, mbase)
#define LAUNCH_WK(MT)                  \
    switch (WK) {                      \
        case 8: LAUNCH(MT, 8); break;  \
        case 4: LAUNCH(MT, 4); break;  \
        case 2: LAUNCH(MT, 2); break;  \
        default: LAUNCH(MT, 1); break; \
    }
    for (int64_t mb = 0; mb < M; mb += 32) {
        const int mbase = (int)mb;
        if (M - mb <= 16) { LAUNCH_WK(1) } else { LAUNCH_WK(2) }
    }
#undef LAUNCH_WK
#undef LAUNCH
    CHITU_RETURN_LAUNCH_STATUS();
}
