// Probe: what keeps the grouped expert GEMM1 (16-slot tile x 16 weight rows per wave, K = 7168) below
// the 6.5-6.8 TB/s its weight access pattern reaches in isolation (probe_stream.hip)?  Same loop as
// moe_gemm1_kernel<1,1,3> with pieces switched off / moved.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ long lo(const i32x4& v) { return (long)(((unsigned long long)(uint32_t)v[1] << 32) | (uint32_t)v[0]); }
__device__ __forceinline__ long hi(const i32x4& v) { return (long)(((unsigned long long)(uint32_t)v[3] << 32) | (uint32_t)v[2]); }
__device__ __forceinline__ f32x4 dot(const i32x4& w0, const i32x4& w1, const i32x4& x0, const i32x4& x1) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 e0 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(lo(w0), lo(x0), z, 0, 0, 0);
    f32x4 o0 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(lo(w0), lo(x1), z, 0, 0, 0);
    f32x4 e1 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(lo(w1), lo(x0), z, 0, 0, 0);
    f32x4 o1 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(lo(w1), lo(x1), z, 0, 0, 0);
    e0 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(hi(w0), hi(x0), e0, 0, 0, 0);
    o0 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(hi(w0), hi(x1), o0, 0, 0, 0);
    e1 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(hi(w1), hi(x0), e1, 0, 0, 0);
    o1 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(hi(w1), hi(x1), o1, 0, 0, 0);
    return f32x4{e0[0] + o0[1], e0[2] + o0[3], e1[0] + o1[1], e1[2] + o1[3]};
}

// V: 0 product shape; 1 scalar ws; 2 scalar ws + xs by 4 blocks; 3 no x loads; 4 no x, no scales;
// 6 product shape with padded slots exec-masked out of the activation loads; 7 = 6 + xs by 4 blocks
template <int V, int D>
__global__ __launch_bounds__(64) void k(const uint8_t* __restrict__ Xq, const float* __restrict__ Xs, const uint8_t* __restrict__ W,
                                        const float* __restrict__ Ws, const int* __restrict__ slot_tok, const int* __restrict__ eids,
                                        uint16_t* __restrict__ out, int N, int K) {
    const int lane = threadIdx.x, j = lane & 15, g = lane >> 4;
    const int mb = blockIdx.y, n0 = blockIdx.x * 16;
    const int KB = K >> 7;
    int e = eids[mb];
    if (V >= 1) e = __builtin_amdgcn_readfirstlane(e);
    const int tok_raw = slot_tok[mb * 16 + j];
    const bool valid = tok_raw >= 0;
    const int token = valid ? tok_raw : 0;
    const uint8_t* xp = Xq + (size_t)token * K + g * 16;
    const float* xsp = Xs + (size_t)token * KB;
    const int off = ((j & 1) * 4 + g) * 16;
    const uint8_t* wp0 = W + ((size_t)e * N + n0 + (j >> 1)) * K + off;
    const uint8_t* wp1 = W + ((size_t)e * N + n0 + 8 + (j >> 1)) * K + off;
    const float* wsp = Ws + ((size_t)e * (N >> 7) + (n0 >> 7)) * KB;
    struct St { i32x4 w0, w1, x0, x1; float xs, ws; };
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    i32x4 cx = {0x38383838, 0x38383838, 0x38383838, 0x38383838};
    f32x4 xs4 = {1.f, 1.f, 1.f, 1.f};
    auto load = [&](St& s, int kb) {
        const int o = kb << 7;
        s.w0 = __builtin_nontemporal_load((const i32x4*)(wp0 + o));
        s.w1 = __builtin_nontemporal_load((const i32x4*)(wp1 + o));
        if (V < 3) { s.x0 = *(const i32x4*)(xp + o); s.x1 = *(const i32x4*)(xp + o + 64); }
        else if (V >= 6) { s.x0 = cx; s.x1 = cx; if (valid) { s.x0 = *(const i32x4*)(xp + o); s.x1 = *(const i32x4*)(xp + o + 64); } }
        else { s.x0 = cx; s.x1 = cx; }
        if (V == 0 || V == 1 || V == 3 || V == 6) s.xs = xsp[kb];
        if (V == 0 || V == 3 || V == 6 || V == 7) s.ws = wsp[kb];
    };
    St ring[D];
#pragma unroll
    for (int d = 0; d < D; ++d) load(ring[d], d);
    for (int kb = 0; kb < KB; kb += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (kb + d < KB) {
                const int kk = kb + d;
                float xs, ws;
                if (V == 2 || V == 5 || V == 7) { if ((kk & 3) == 0) xs4 = *(const f32x4*)(xsp + kk); xs = xs4[kk & 3]; }
                else if (V == 4) xs = 1.f; else xs = ring[d].xs;
                if (V == 1 || V == 2 || V == 5) ws = wsp[kk]; else if (V == 4) ws = 1.f; else ws = ring[d].ws;
                const f32x4 b = dot(ring[d].w0, ring[d].w1, ring[d].x0, ring[d].x1);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] += (b[r] * xs) * ws;
                if (kk + D < KB) load(ring[d], kk + D);
            }
        }
    }
    uint16_t* o = out + ((size_t)mb * 16 + j) * N + n0 + 2 * g;
    o[0] = (uint16_t)(__float_as_uint(acc[0]) >> 16); o[1] = (uint16_t)(__float_as_uint(acc[1]) >> 16);
    o[8] = (uint16_t)(__float_as_uint(acc[2]) >> 16); o[9] = (uint16_t)(__float_as_uint(acc[3]) >> 16);
}

// V8: the m-block's VALID activation rows (<= NVMAX; more -> the global path of V0) and their scales are staged
// in LDS once per workgroup; the k loop then reads x from LDS (ds_read_b128) and only weights from HBM.
template <int D, int NVMAX>
__global__ __launch_bounds__(64) void k8(const uint8_t* __restrict__ Xq, const float* __restrict__ Xs, const uint8_t* __restrict__ W,
                                         const float* __restrict__ Ws, const int* __restrict__ slot_tok, const int* __restrict__ eids,
                                         uint16_t* __restrict__ out, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x, j = lane & 15, g = lane >> 4;
    const int mb = blockIdx.y, n0 = blockIdx.x * 16;
    const int KB = K >> 7;
    const int e = __builtin_amdgcn_readfirstlane(eids[mb]);
    const int tok_raw = slot_tok[mb * 16 + j];
    const bool valid = tok_raw >= 0;
    const int nv = __builtin_popcountll(__builtin_amdgcn_ballot_w64(valid && g == 0));  // valid slots are the first nv
    const int off = ((j & 1) * 4 + g) * 16;
    const uint8_t* wp0 = W + ((size_t)e * N + n0 + (j >> 1)) * K + off;
    const uint8_t* wp1 = W + ((size_t)e * N + n0 + 8 + (j >> 1)) * K + off;
    const float* wsp = Ws + ((size_t)e * (N >> 7) + (n0 >> 7)) * KB;
    const int RS = K + 64;                       // row stride: rows land 16 banks apart
    float* xs_lds = (float*)(lds + NVMAX * RS);  // [NVMAX][KB]
    const bool staged = nv <= NVMAX;
    const int token = valid ? tok_raw : 0;
    const uint8_t* xp = Xq + (size_t)token * K + g * 16;
    const float* xsp = Xs + (size_t)token * KB;
    struct St { i32x4 w0, w1, x0, x1; float xs, ws; };
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    St ring[D];
    auto loadw = [&](St& s, int kb) {
        const int o = kb << 7;
        s.w0 = __builtin_nontemporal_load((const i32x4*)(wp0 + o));
        s.w1 = __builtin_nontemporal_load((const i32x4*)(wp1 + o));
        s.ws = wsp[kb];
    };
    if (staged) {
        // weights of the first D blocks are requested before the staging round trip
#pragma unroll
        for (int d = 0; d < D; ++d) loadw(ring[d], d);
        for (int r = 0; r < nv; ++r) {
            const int tr = __builtin_amdgcn_readlane(tok_raw, r);
            const uint8_t* src = Xq + (size_t)tr * K;
            for (int b = lane * 16; b < K; b += 1024) *(i32x4*)(lds + r * RS + b) = *(const i32x4*)(src + b);
            if (lane < KB) xs_lds[r * KB + lane] = Xs[(size_t)tr * KB + lane];
        }
        __syncthreads();
        const int row = valid ? j : 0;
        const uint8_t* xl = lds + row * RS + g * 16;
        const float* xsl = xs_lds + row * KB;
        for (int kb = 0; kb < KB; kb += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (kb + d < KB) {
                    const int kk = kb + d;
                    const i32x4 x0 = *(const i32x4*)(xl + (kk << 7)), x1 = *(const i32x4*)(xl + (kk << 7) + 64);
                    const float xs = xsl[kk];
                    const f32x4 b = dot(ring[d].w0, ring[d].w1, x0, x1);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] += (b[r] * xs) * ring[d].ws;
                    if (kk + D < KB) loadw(ring[d], kk + D);
                }
            }
        }
    } else {
        auto load = [&](St& s, int kb) {
            loadw(s, kb);
            const int o = kb << 7;
            s.x0 = *(const i32x4*)(xp + o); s.x1 = *(const i32x4*)(xp + o + 64);
            s.xs = xsp[kb];
        };
#pragma unroll
        for (int d = 0; d < D; ++d) load(ring[d], d);
        for (int kb = 0; kb < KB; kb += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (kb + d < KB) {
                    const int kk = kb + d;
                    const f32x4 b = dot(ring[d].w0, ring[d].w1, ring[d].x0, ring[d].x1);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] += (b[r] * ring[d].xs) * ring[d].ws;
                    if (kk + D < KB) load(ring[d], kk + D);
                }
            }
        }
    }
    uint16_t* o = out + ((size_t)mb * 16 + j) * N + n0 + 2 * g;
    o[0] = (uint16_t)(__float_as_uint(acc[0]) >> 16); o[1] = (uint16_t)(__float_as_uint(acc[1]) >> 16);
    o[8] = (uint16_t)(__float_as_uint(acc[2]) >> 16); o[9] = (uint16_t)(__float_as_uint(acc[3]) >> 16);
}

template <int D, int NVMAX>
int run8(const char* name, uint8_t* Xq, float* Xs, uint8_t** W, float* Ws, int* st, int* eids, uint16_t* out, int MB) {
    const int N = 512, K = 7168;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds = (size_t)NVMAX * (K + 64) + (size_t)NVMAX * (K / 128) * 4;
    CK(hipFuncSetAttribute((const void*)k8<D, NVMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    float best = 1e9, sum = 0;
    for (int it = 0; it < 6; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k8<D, NVMAX>), dim3(N / 16, MB), dim3(64), lds, 0, Xq, Xs, W[it & 1], Ws, st, eids, out, N, K);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0) { sum += ms; if (ms < best) best = ms; }
    }
    const double bytes = (double)MB * N * K;
    printf("%-44s D=%d NV=%d avg %.2f us  min %.2f us  %.0f GB/s (min)\n", name, D, NVMAX, sum / 5 * 1e3, best * 1e3, bytes / best / 1e6);
    return 0;
}

template <int V, int D>
int run(const char* name, uint8_t* Xq, float* Xs, uint8_t** W, float* Ws, int* st, int* eids, uint16_t* out, int MB) {
    const int N = 512, K = 7168;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9, sum = 0;
    for (int it = 0; it < 6; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<V, D>), dim3(N / 16, MB), dim3(64), 0, 0, Xq, Xs, W[it & 1], Ws, st, eids, out, N, K);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0) { sum += ms; if (ms < best) best = ms; }
    }
    const double bytes = (double)MB * N * K;
    printf("%-44s D=%d  avg %.2f us  min %.2f us  %.0f GB/s (min)\n", name, D, sum / 5 * 1e3, best * 1e3, bytes / best / 1e6);
    return 0;
}

int main() {
    const int E = 257, N = 512, K = 7168, MB = 103, bs = 16;
    uint8_t* W[2]; uint8_t* Xq; float *Xs, *Ws; int *st, *eids; uint16_t* out;
    for (int i = 0; i < 2; ++i) { CK(hipMalloc(&W[i], (size_t)E * N * K)); CK(hipMemset(W[i], 0x38 + i, (size_t)E * N * K)); }
    CK(hipMalloc(&Xq, bs * K)); CK(hipMemset(Xq, 0x38, bs * K));
    CK(hipMalloc(&Xs, bs * 56 * 4)); CK(hipMalloc(&Ws, (size_t)E * 4 * 56 * 4));
    std::vector<float> ones((size_t)E * 4 * 56, 0.01f);
    CK(hipMemcpy(Ws, ones.data(), ones.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(Xs, ones.data(), bs * 56 * 4, hipMemcpyHostToDevice));
    std::vector<int> hst(MB * 16), he(MB);
    for (int m = 0; m < MB; ++m) { he[m] = (m * 5) % 256; const int nv = m == 0 ? 16 : 1 + (m % 4 == 0); for (int j = 0; j < 16; ++j) hst[m * 16 + j] = j < nv ? (m + j) % bs : -1; }
    CK(hipMalloc(&st, hst.size() * 4)); CK(hipMalloc(&eids, he.size() * 4)); CK(hipMalloc(&out, (size_t)MB * 16 * N * 2));
    CK(hipMemcpy(st, hst.data(), hst.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(eids, he.data(), he.size() * 4, hipMemcpyHostToDevice));
    run<0, 3>("V0 product shape", Xq, Xs, W, Ws, st, eids, out, MB);
    run<0, 4>("V0", Xq, Xs, W, Ws, st, eids, out, MB);
    run<6, 3>("V6 padded slots masked out of x loads", Xq, Xs, W, Ws, st, eids, out, MB);
    run<7, 3>("V7 = V6 + x scales per 4 blocks", Xq, Xs, W, Ws, st, eids, out, MB);
    run<7, 4>("V7", Xq, Xs, W, Ws, st, eids, out, MB);
    run<1, 3>("V1 scalar weight scale", Xq, Xs, W, Ws, st, eids, out, MB);
    run<2, 3>("V2 scalar ws + x scales per 4 blocks", Xq, Xs, W, Ws, st, eids, out, MB);
    run<2, 4>("V2", Xq, Xs, W, Ws, st, eids, out, MB);
    run<2, 6>("V2", Xq, Xs, W, Ws, st, eids, out, MB);
    run8<3, 3>("V8 valid rows staged in LDS", Xq, Xs, W, Ws, st, eids, out, MB);
    run8<4, 3>("V8", Xq, Xs, W, Ws, st, eids, out, MB);
    run8<3, 2>("V8", Xq, Xs, W, Ws, st, eids, out, MB);
    run8<6, 2>("V8", Xq, Xs, W, Ws, st, eids, out, MB);
    run<3, 3>("V3 no activation loads", Xq, Xs, W, Ws, st, eids, out, MB);
    run<4, 3>("V4 no activation loads, no scales", Xq, Xs, W, Ws, st, eids, out, MB);
    run<4, 6>("V4", Xq, Xs, W, Ws, st, eids, out, MB);
    return 0;
}
