// Shared device/host helpers for the gfx950 decode kernels.
// Wave width is 64 everywhere (CDNA4); nothing here is portable to 32-wide warps.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define CHITU_OK 0
#define CHITU_ERR_BAD_ARG (-1)
#define CHITU_ERR_UNSUPPORTED (-2)

// Every entry point returns CHITU_OK, a negative CHITU_ERR_* for argument
// errors, or the positive hipError_t of a failed launch (never exit()).
#define CHITU_RETURN_LAUNCH_STATUS()            \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        return e__ == hipSuccess ? CHITU_OK : (int)e__; \
    } while (0)

#define CHITU_REQUIRE(cond)                     \
    do {                                        \
        if (!(cond)) return CHITU_ERR_BAD_ARG;  \
    } while (0)

namespace chitu {

// Launch-variant overrides (chitu_hip_debug_option): every op picks its launch variant (K-split width, ring depth,
// fast / generic kernel ...) by a heuristic; tests and tuning sweeps force one through this table instead of the
// process environment (no getenv on any launch path).  -1 = heuristic.  Defined in options.hip.
enum DebugOption {
    kOptMoeGemm1WK = 0, kOptMoeGemm1NW, kOptMoeGemm1D, kOptMoeGemm2Cfg, kOptMoeI8WK, kOptGateGeneric, kOptGateTicket,
    kOptSampleRadix, kOptFp8GemmWK, kOptFp8GemmDeep, kOptBf16GemmWK, kOptBf16GemmDeep, kOptBf16SiluWK,
    kOptFp8GemmTiled, kOptBf16GemmTiled, kOptGateSmallSort, kOptFp8TiledTM, kOptCount
};
extern int g_debug_options[kOptCount];
inline int debug_option(DebugOption o) { return g_debug_options[o]; }
inline void debug_override(DebugOption o, int& v) { if (g_debug_options[o] >= 0) v = g_debug_options[o]; }

// Phase probes (probe builds only: hipcc -DCHITU_PROBE): CHITU_PROBE_MARK(i) stores the 100 MHz wall clock into slot i
// of a device array from thread 0 of workgroup 0; tools/probe_phases.py reads it back.  Compiles to nothing otherwise.
#ifdef CHITU_PROBE
static __device__ unsigned long long g_probe_marks[32];  // one copy per translation unit (no relocatable device code)
#define CHITU_PROBE_READER(tu)                                                                                     \
    extern "C" int chitu_hip_probe_read_##tu(unsigned long long* out32) {                                          \
        (void)hipDeviceSynchronize();                                                                              \
        return (int)hipMemcpyFromSymbol(out32, HIP_SYMBOL(chitu::g_probe_marks), sizeof(unsigned long long) * 32);        \
    }
#define CHITU_PROBE_MARK(i)                                                                            \
    do {                                                                                               \
        if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) g_probe_marks[i] = wall_clock64(); \
    } while (0)
#else
#define CHITU_PROBE_MARK(i) do {} while (0)
#define CHITU_PROBE_READER(tu)
#endif

constexpr int kWave = 64;

typedef uint16_t bf16_t;  // raw bits
typedef uint8_t fp8_t;    // OCP e4m3fn raw bits

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
    return __uint_as_float(((uint32_t)v) << 16);
}

// Round-to-nearest-even (torch's float->bfloat16) on gfx950's v_cvt_pk_bf16_f32; the integer
// emulation (kept as f32_to_bf16_sw, checked against the instruction by chitu_hip_selftest_arith)
// costs 5 VALU ops per value, which is what the small norm/quant kernels are made of.
typedef __bf16 hwbf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t f32x2_to_bf16x2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hwbf16x2));
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(f32x2_to_bf16x2(f, 0.f) & 0xffffu); }
__device__ __forceinline__ bf16_t f32_to_bf16_sw(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

__device__ __forceinline__ float round_bf16(float f) { return bf16_to_f32(f32_to_bf16(f)); }

__device__ __forceinline__ float f16_to_f32(uint16_t h) {
    return (float)__builtin_bit_cast(_Float16, h);
}
__device__ __forceinline__ uint16_t f32_to_f16(float f) {
    return __builtin_bit_cast(uint16_t, (_Float16)f);
}

// OCP e4m3fn byte -> f32 (gfx950's v_cvt_f32_fp8 decodes the OCP encoding).
template <int BYTE>
__device__ __forceinline__ float fp8_to_f32(uint32_t packed) {
    return __builtin_amdgcn_cvt_f32_fp8((int)packed, BYTE);
}

// 8 OCP e4m3fn bytes x one block scale -> 8 bf16, each rounded as the reference's materialised weight_dequant tensor is
// (triton_kernels.py:217-247): the MFMA operand of the absorb projections (absorb.hip, mla_decode.hip's fused tail).
__device__ __forceinline__ s16x8 dequant8_bf16(uint32_t w0, uint32_t w1, float s) {
    s16x8 r;
    r[0] = (short)f32_to_bf16(fp8_to_f32<0>(w0) * s);
    r[1] = (short)f32_to_bf16(fp8_to_f32<1>(w0) * s);
    r[2] = (short)f32_to_bf16(fp8_to_f32<2>(w0) * s);
    r[3] = (short)f32_to_bf16(fp8_to_f32<3>(w0) * s);
    r[4] = (short)f32_to_bf16(fp8_to_f32<0>(w1) * s);
    r[5] = (short)f32_to_bf16(fp8_to_f32<1>(w1) * s);
    r[6] = (short)f32_to_bf16(fp8_to_f32<2>(w1) * s);
    r[7] = (short)f32_to_bf16(fp8_to_f32<3>(w1) * s);
    return r;
}

// Two f32 -> two OCP e4m3fn bytes, RNE, NaN stays NaN (act_quant_deepseek_v3 has
// no clamp: an all-zero group is 0/0 = NaN in the reference, triton_kernels.py:210-212).
__device__ __forceinline__ uint32_t f32x2_to_fp8x2(float a, float b) {
    return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xffffu;
}
// Saturating form: clamp to +-448 first (the reference's tl.clamp, fused_moe.py:646/703).
__device__ __forceinline__ uint32_t f32x2_to_fp8x2_sat(float a, float b) {
    a = __builtin_fminf(__builtin_fmaxf(a, -448.f), 448.f);
    b = __builtin_fminf(__builtin_fmaxf(b, -448.f), 448.f);
    return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xffffu;
}

// x / sc for many x sharing one divisor (a quantisation group's scale): IEEE-exact quotient from a
// refined reciprocal and one residual correction (3 FMAs per element instead of the ~10-instruction
// v_div_scale/v_div_fmas/v_div_fixup sequence).  Exactness needs sc, the quotient and the residual
// (|x| * 2^-24) in the normal range: callers test group_div_fast(sc) wave-uniformly and use plain division otherwise (zero /
// denormal / NaN scales, e.g. the reference's 0/0 = NaN for an all-zero group).
__device__ __forceinline__ bool group_div_fast(float sc) { return sc >= 0x1p-60f && sc <= 0x1p60f; }
__device__ __forceinline__ float group_rcp(float sc) {
    const float r0 = __builtin_amdgcn_rcpf(sc);
    return __builtin_fmaf(__builtin_fmaf(-sc, r0, 1.0f), r0, r0);
}
__device__ __forceinline__ float group_div(float x, float sc, float r) {
    const float q = x * r;
    return __builtin_fmaf(__builtin_fmaf(-q, sc, x), r, q);
}

// 8 values of one lane -> 8 e4m3 bytes of v / sc (SAT: clamped to +-448 first).
template <bool SAT>
__device__ __forceinline__ i32x2 quant8_fp8(const float (&v)[8], float sc) {
    float t[8];
    if (__builtin_amdgcn_ballot_w64(!group_div_fast(sc)) == 0) {
        const float r = group_rcp(sc);
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = group_div(v[k], sc, r);
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = v[k] / sc;
    }
    i32x2 o;
    if (SAT) {
        o[0] = (int)(f32x2_to_fp8x2_sat(t[0], t[1]) | (f32x2_to_fp8x2_sat(t[2], t[3]) << 16));
        o[1] = (int)(f32x2_to_fp8x2_sat(t[4], t[5]) | (f32x2_to_fp8x2_sat(t[6], t[7]) << 16));
    } else {
        o[0] = (int)(f32x2_to_fp8x2(t[0], t[1]) | (f32x2_to_fp8x2(t[2], t[3]) << 16));
        o[1] = (int)(f32x2_to_fp8x2(t[4], t[5]) | (f32x2_to_fp8x2(t[6], t[7]) << 16));
    }
    return o;
}

// One term of the MLA split merge, acc + w_s * v_s, as ONE fused multiply-add spelt out: the three merge forms (mla_merge_kernel,
// mla_merge_uv_quant_kernel, the fused tail of mla_decode_kernel<true>) are bit-identical to each other by construction, not by
// the compiler happening to contract (or pack) `acc += w * v` the same way in three places.  A split without tokens (w = 0) adds
// exactly 0 whatever its row holds.
__device__ __forceinline__ float merge_term(float acc, float ws, float v) { return __builtin_fmaf(ws, ws != 0.f ? v : 0.f, acc); }

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// Reductions inside each 16-lane DPP row (lanes 16r..16r+15) on the VALU: 4 row-rotate steps,
// every lane ends with the row's result.  (__shfl_xor lowers to ds_bpermute = an LDS round trip.)
template <int CTRL>
__device__ __forceinline__ float dpp_row(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_reduce_max(float v) {
    v = __builtin_fmaxf(v, dpp_row<0x128>(v));  // row_ror:8
    v = __builtin_fmaxf(v, dpp_row<0x124>(v));  // row_ror:4
    v = __builtin_fmaxf(v, dpp_row<0x122>(v));  // row_ror:2
    v = __builtin_fmaxf(v, dpp_row<0x121>(v));  // row_ror:1
    return v;
}
__device__ __forceinline__ float row16_reduce_sum(float v) {
    v += dpp_row<0x128>(v);
    v += dpp_row<0x124>(v);
    v += dpp_row<0x122>(v);
    v += dpp_row<0x121>(v);
    return v;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

static inline int ceil_div_i(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace chitu
