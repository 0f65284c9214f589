// Compute-shaped bf16 GEMM for PREFILL-sized M (router scores, every linear of an unquantised model, the LM head over
// a whole prompt), gfx950.  Same tiling as fp8_gemm_tiled.hip -- 128 weight rows x 128 tokens per workgroup, 4 waves as
// 2 x 2, both operand tiles staged through LDS per 128-byte K block (64 bf16) by LDS-DMA (lds_dma.h), double-buffered -- without block scales:
// the 16 accumulator tiles of a wave run across the whole K range (v_mfma_f32_16x16x32_bf16, weights = A operand).
//
// Replaces (reference, read-only), for M >= 128: torch.nn.functional.linear on bf16 weights as used by
//   chitu/models/model_deepseek_v3.py:810-812  GateDeepSeekV3.forward  (scores = linear(x, weight))
//   chitu/models/model.py:104-132, 201-214     Attention / FeedForward projections in prefill
// The skinny kernel (gate.hip: bf16_gemm_kernel) streams the weight matrix once per 32 token rows: 64 launches of
// 4.7 us for the router of a 2048-token prompt (profiles/r02_prefill_*), one launch here.
#include "common.h"
#include "gemm_common.h"
#include "lds_dma.h"

namespace chitu {

constexpr int kBTile = 128;
constexpr int kBTileBytes = kBTile * 128;  // one operand tile of one 64-element K block in LDS: [128 rows][128 B], staged by LDS-DMA
// with the 16-byte chunks XOR-permuted on the source side (lds_dma.h; fp8_gemm_tiled.hip has the measurements)

// TM = token rows per workgroup: 128, or 64 for grids that would leave CUs with fewer than two workgroups (launcher; round 6,
// as fp8_gemm_tiled.hip: the same arithmetic per output element in the same order, twice the workgroups).
template <int TM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void bf16_gemm_tiled_kernel(
    const bf16_t* __restrict__ X, const bf16_t* __restrict__ W, void* __restrict__ out, int out_dt, int M, int N, int K,
    float* __restrict__ partials) {
    static_assert(TM == 128 || TM == 64, "token tile");
    constexpr int MT = TM / 32;  // 16-token MFMA tiles per wave (waves 2 x 2: each 64 weight rows x TM / 2 tokens)
    __shared__ __attribute__((aligned(16))) uint8_t sW[2][kBTileBytes];
    __shared__ __attribute__((aligned(16))) uint8_t sX[2][TM * 128];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int wn = wave & 1, wm = wave >> 1;
    const int n0 = blockIdx.x * kBTile, m0 = blockIdx.y * TM;
    // split-K (gridDim.z > 1): this workgroup's share of the 64-element K blocks; its fp32 tile goes to plane blockIdx.z of
    // `partials` [S][M][N] for the consumer to sum in plane order (chitu_hip_gate_route(num_partials)).  The router's score
    // GEMM of a 2048-token prompt is 32 tiles of 112 K blocks: one workgroup per tile leaves 7/8 of the CUs idle and
    // is a ~1 us-per-block dependent chain (107 us, profiles/r05_*); 8 planes = 256 workgroups of 14 blocks.
    const int KB_all = K >> 6;  // blocks of 64 elements = 128 bytes
    const int S = gridDim.z, per = (KB_all + S - 1) / S;
    const int kb0 = blockIdx.z * per, KB = min(per, KB_all - kb0);
    if (KB <= 0 && S > 1) {
        // (an empty share still owns its plane: zero it)
        for (int idx = threadIdx.x; idx < TM * kBTile; idx += 256) {
            const int m = m0 + idx / kBTile, n = n0 + idx % kBTile;
            if (m < M && n < N) partials[((size_t)blockIdx.z * M + m) * N + n] = 0.f;
        }
        return;
    }

    // staging role: wave w brings rows 32 w .. 32 w + 31 of both tiles, four 8-row LDS-DMA pieces each; byte offsets from the
    // tiles' first rows (32-bit: the launcher bounds 128 K), rows past the matrix re-read its last row (never stored)
    // (TM = 64: the token tile is 8 pieces, two per wave)
    uint32_t woff[4], xoff[MT];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = wave * 4 + i, r = n * 8 + (lane >> 3), c = kblock_src_chunk(lane, n);
        woff[i] = (uint32_t)(min(r, N - 1 - n0) * K * 2 + c * 16);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int n = wave * MT + i, r = n * 8 + (lane >> 3), c = kblock_src_chunk(lane, n);
        xoff[i] = (uint32_t)(min(r, M - 1 - m0) * K * 2 + c * 16);
    }
    const bf16_t* wbase = W + (size_t)n0 * K + (size_t)kb0 * 64;
    const bf16_t* xbase = X + (size_t)m0 * K + (size_t)kb0 * 64;
    const uint32_t ldsW = lds_offset_of(&sW[0][0]), ldsX = lds_offset_of(&sX[0][0]);
    auto issue = [&](int kb) {
        const uint32_t dw = (uint32_t)((kb & 1) * kBTileBytes + wave * 4096), dx = (uint32_t)((kb & 1) * (TM * 128) + wave * (MT * 1024));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            glds16_sbase(wbase + (size_t)kb * 64, woff[i], ldsW + dw + i * 1024);
            if (i < MT) glds16_sbase(xbase + (size_t)kb * 64, xoff[i], ldsX + dx + i * 1024);
        }
    };
    const int foff = kblock_frag_off(j, g);  // this lane's fragment inside a 16-row tile (second half: ^ 64)

    f32x4 acc[4][MT];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // double-buffered, ONE barrier per step: it publishes block kb and retires the buffer block kb - 1 was read from, which
    // the step's DMA then refills
    issue(0);
    for (int kb = 0; kb < KB; ++kb) {
        const int buf = kb & 1;
        glds_wait_all();  // block kb has landed (this wave's pieces) ...
        __syncthreads();  // ... and everyone's; the buffer of block kb - 1 is free
        if (kb + 1 < KB) issue(kb + 1);
        // lane (j, g): elements [8g, 8g+8) of each 32-element half of row j -- one MFMA operand per half
        s16x8 wa[4][2];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint8_t* wr = &sW[buf][(wn * 64 + t * 16) * 128];
            wa[t][0] = *reinterpret_cast<const s16x8*>(wr + foff);
            wa[t][1] = *reinterpret_cast<const s16x8*>(wr + (foff ^ 64));
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const uint8_t* xr = &sX[buf][(wm * (TM / 2) + mt * 16) * 128];
            const s16x8 xb0 = *reinterpret_cast<const s16x8*>(xr + foff), xb1 = *reinterpret_cast<const s16x8*>(xr + (foff ^ 64));
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[nt][0], xb0, acc[nt][mt], 0, 0, 0);
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[nt][1], xb1, acc[nt][mt], 0, 0, 0);
            }
        }
    }

    // C tile (nt, mt): lane holds weight rows n = 4g .. 4g+3 of token column j -> 4 consecutive outputs of one token
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = m0 + wm * (TM / 2) + mt * 16 + j;
        if (m >= M) continue;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = n0 + wn * 64 + nt * 16 + 4 * g;
            if (n >= N) continue;
            const f32x4 v = acc[nt][mt];
            if (S > 1 || out_dt == 2) {
                float* dst = (S > 1 ? partials + (size_t)blockIdx.z * M * N : (float*)out) + (size_t)m * N + n;
                if (n + 3 < N && (N & 3) == 0) *reinterpret_cast<f32x4*>(dst) = v;
                else
                    for (int r = 0; r < 4 && n + r < N; ++r) dst[r] = v[r];
            } else {
                uint16_t h[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) h[r] = out_dt == 0 ? f32_to_bf16(v[r]) : f32_to_f16(v[r]);
                uint16_t* dst = (uint16_t*)out + (size_t)m * N + n;
                if (n + 3 < N && (N & 3) == 0) {
                    i32x2 o;
                    o[0] = (int)((uint32_t)h[0] | ((uint32_t)h[1] << 16));
                    o[1] = (int)((uint32_t)h[2] | ((uint32_t)h[3] << 16));
                    *reinterpret_cast<i32x2*>(dst) = o;
                } else {
                    for (int r = 0; r < 4 && n + r < N; ++r) dst[r] = h[r];
                }
            }
        }
    }
}

// chitu_hip_bf16_gemm's large-M form (declared in gemm_common.h, called from gate.hip)
void launch_bf16_gemm_tiled(const bf16_t* x, const bf16_t* w, void* out, int out_dt, int64_t M, int64_t N, int64_t K,
                            int num_splits, float* partials, hipStream_t st) {
    // 64-token tiles while 128-token ones would leave CUs with fewer than two workgroups (fp8_gemm_tiled.hip has the sweep);
    // option kOptFp8TiledTM forces either
    const int64_t wgs128 = ((N + kBTile - 1) / kBTile) * ((M + 127) / 128) * num_splits;
    int tm = wgs128 < 256 ? 64 : 128;
    debug_override(kOptFp8TiledTM, tm);
    if (tm != 64) tm = 128;
    const dim3 grid((unsigned)((N + kBTile - 1) / kBTile), (unsigned)((M + tm - 1) / tm), (unsigned)num_splits);
    if (tm == 64)
        hipLaunchKernelGGL(bf16_gemm_tiled_kernel<64>, grid, dim3(256), 0, st, x, w, out, out_dt, (int)M, (int)N, (int)K, partials);
    else
        hipLaunchKernelGGL(bf16_gemm_tiled_kernel<128>, grid, dim3(256), 0, st, x, w, out, out_dt, (int)M, (int)N, (int)K, partials);
}

}  // namespace chitu
