// Compute-shaped W8A8 block-scaled GEMM for PREFILL (hundreds to thousands of token rows), gfx950.
//
// Replaces (reference, read-only), for M >= 128:
//   chitu/triton_kernels.py:302-365  fp8_gemm_deepseek_v3_kernel   (W8A8, fp32 128x128 / 1x128 block scales)
//   chitu/ops.py:453-483             its launcher
// The decode kernel (fp8_gemm.hip) streams the weight matrix once per 64 token rows: right when the op is
// a weight stream (M <= 64), 32 passes over the same 15 MB at M = 2048 (profiles/r02_prefill_*: the dense
// projections were 54 % of a prefill layer, at ~1 % of the MFMA peak).  Here the op is tiled like a GEMM:
//
//   workgroup = 128 weight rows x 128 tokens, 4 waves as 2 x 2, each wave 64 x 64 = 4 x 4 MFMA tiles
//   (round 6: ONE v_mfma_scale_f32_16x16x128_f8f6f4 per tile and K block with unit E8M0 scales -- gemm_common.h::mfma_fp8_k128 --
//   where rounds 2-5 chained four v_mfma_f32_16x16x32_fp8_fp8; weights = A operand, tokens = B operand, as in the decode kernels);
//   K advances in the quantisation's own 128-wide blocks: both operand tiles (16 KB each) go global -> LDS by LDS-DMA
//   (global_load_lds_dwordx4, lds_dma.h: no staging registers, no ds_write pass -- rounds 2-4 staged through registers and
//   the ds_write_b128 stream of two resident workgroups alone took ~830 LDS cycles per K block against 1024 MFMA cycles),
//   rows unpadded with the 16-byte chunks XOR-permuted on the source side (conflict-free under ds_read_b128's lane groups;
//   the 144-B padded rows of rounds 2-4 measured SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.33), double-buffered: block
//   kb + 1 lands while block kb is multiplied.  Per block every 16 x 16 tile is a
//   fresh 4-MFMA dot that is folded in as (dot * a_s[token]) * b_s -- the reference's order (triton_kernels.py:357);
//   b_s is one scalar per workgroup and block (128-row tiles are scale-block aligned), a_s one value per lane and
//   token tile.
// Bound: MFMA on paper (the K = 128 form runs at the fp8 rate, ~5 PFLOP/s dense: 16 MFMAs = 512 matrix-pipe cycles per wave and K
// block, against 16 ds_read_b128 and ~90 VALU instructions of scale folding); measured 0.17-0.26 of that peak at 2048 tokens --
// what the waves wait for is each other (one barrier per K block) and the next tile.  Round 6, late: 64-token tiles for small
// grids (below 256 tiles of 128 tokens a CU holds at most one workgroup and nothing fills its barrier and request gaps; the
// 64-token form doubles the workgroups at the same arithmetic per output element: wqkv_a at 256-1024 tokens 48 -> 34-40 us, bit-
// identical), and the step reordered to wait -> barrier -> request -> multiply.  Shader-clock stamps (tools/probe_tiled_steps.py,
// profiles/r06_ab_tiled_dma_spread.txt) put a 128-token step at ~540 cycles of DMA issue (a wave stands ~64 cycles on a 1 KB
// piece; the CU's L2 -> LDS path moves ~54 B per clock, tools/probe_lds_fill.hip), ~480 of LDS reads + MFMAs + folds, ~320 of
// landing wait and ~130 of barrier.  Spreading the pieces between the MFMAs (pinned by unread asm operands) measured 3-8 %
// slower; a 256 x 128 tile on 8 waves, a loader / multiplier split of the waves (three-stage ring) and reads-before-requests were
// built bit-identical and did not beat two 4-wave workgroups per CU either (profiles/r06_ab_fp8_tiled_forms.txt); the scale fold was
// halved (tiled_fold below).  Two more things were measured and NOT kept in
// round 6 (profiles/r06_ab_fp8_mx_mfma.txt): 8 waves per workgroup with a four-stage ring (three K blocks in flight, one
// workgroup per CU): 25-30 % SLOWER at every shape, as the 4-wave rings of round 5 were -- two independent workgroups per CU
// drift out of phase and fill each other's barrier and request gaps, one workgroup of 8 waves moves in lockstep.
// An L2 warm-up (every wave touching one line of each of its 64 staging rows 2-8 K blocks ahead with a 4-byte LDS-DMA into a dump
// area) was measured too: 4-30 % slower, monotonically with the distance (profiles/r06_ab_tiled_l2_warmup.txt) -- a step does not
// wait for HBM, and the touches are 64 more line requests per wave and step through the same L1.
#include <type_traits>

#include "common.h"
#include "gemm_common.h"
#include "lds_dma.h"

namespace chitu {

// probe builds (-DCHITU_PROBE, tools/probe_tiled_steps.py): shader-clock stamps of workgroup 0's thread 0 at five points of K steps
// 8 .. 13 -- top, own DMA pieces landed, barrier passed, next stage requested, block multiplied
#ifdef CHITU_PROBE
#define TILED_MARK(kb, n)                                                                                     \
    do {                                                                                                      \
        if ((kb) >= 8 && (kb) < 14 && threadIdx.x == 0 && blockIdx.x == 0) g_probe_marks[((kb) - 8) * 5 + (n)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define TILED_MARK(kb, n) do {} while (0)
#endif

// acc += d * (a_s * b_s) for the four results of a tile: ONE VALU pass over the tile per MFMA (two v_pk_fma_f32) where
// gemm_common.h::fold_scaled -- (d * a_s) * b_s, the reference's order, triton_kernels.py:357 -- takes two.  Two roundings either
// way (here: the scale product and the fma).  The tiled form is instruction-issue-bound: -6..-10 % from 1024 tokens on
// (profiles/r06_ab_fp8_tiled_forms.txt, section 4); CHITU_TILED_PREMUL=0 builds the two-pass fold.
#ifndef CHITU_TILED_PREMUL
#define CHITU_TILED_PREMUL 1
#endif
__device__ __forceinline__ void tiled_fold(f32x4& acc, const f32x4& d, float a_s, float b_s) {
#if CHITU_TILED_PREMUL
    const float s = a_s * b_s;
    acc = __builtin_elementwise_fma(d, f32x4{s, s, s, s}, acc);
#else
    fold_scaled(acc, d, a_s, b_s);
#endif
}
#ifndef CHITU_TILED_XCD
#define CHITU_TILED_XCD 1  // 0: row-major tile order (A/B builds, tools/build_variant.sh)
#endif
constexpr int kTileN = 128, kTileK = 128;
constexpr int kTiledSmallGrid = 256;  // 128-token tiles below which the 64-token form is launched (sweep: profiles/r06_ab_fp8_tiled_tm64.txt)
constexpr int kTileBytes = 128 * kTileK;  // one operand tile of one K block in LDS

// TM = token rows per workgroup: 128, or 64 for grids that would not put two workgroups on every CU (launcher): the same
// arithmetic per output element in the same order, half the MFMA work per K step and workgroup, twice the workgroups.
template <int TM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void fp8_gemm_tiled_kernel(
    const fp8_t* __restrict__ X, const float* __restrict__ XS, const fp8_t* __restrict__ W, const float* __restrict__ WS,
    void* __restrict__ out, int out_dt, int M, int N, int K) {
    static_assert(TM == 128 || TM == 64, "token tile");
    constexpr int MT = TM / 32;  // 16-token MFMA tiles per wave (waves 2 x 2: each 64 weight rows x TM / 2 tokens)
    __shared__ __attribute__((aligned(16))) uint8_t sW[2][kTileBytes];
    __shared__ __attribute__((aligned(16))) uint8_t sX[2][TM * kTileK];
    __shared__ __attribute__((aligned(16))) float sS[2][TM];  // the K block's activation scale of every token row of the tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int wn = wave & 1, wm = wave >> 1;
#if CHITU_TILED_XCD
    int tile_m, tile_n;  // XCD-blocked order (gemm_common.h): workgroup-uniform, so is the early exit
    if (!xcd_tile_of((int)blockIdx.x, (M + TM - 1) / TM, (N + kTileN - 1) / kTileN, tile_m, tile_n)) return;
    const int n0 = tile_n * kTileN, m0 = tile_m * TM;
#else
    const int n0 = blockIdx.x * kTileN, m0 = blockIdx.y * TM;
#endif
    const int KB = K >> 7;

    // staging role: wave w brings rows 32 w .. 32 w + 31 of both tiles, four 8-row pieces each (lds_dma.h: lane i -> row
    // 8 n + (i >> 3), source chunk kblock_src_chunk); byte offsets from the tiles' first rows, rows past the matrix re-read
    // its last row (never stored)
    // (TM = 64: the token tile is 8 pieces, two per wave)
    uint32_t woff[4], xoff[MT];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = wave * 4 + i, r = n * 8 + (lane >> 3), c = kblock_src_chunk(lane, n);
        woff[i] = (uint32_t)(min(r, N - 1 - n0) * K + c * 16);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int n = wave * MT + i, r = n * 8 + (lane >> 3), c = kblock_src_chunk(lane, n);
        xoff[i] = (uint32_t)(min(r, M - 1 - m0) * K + c * 16);
    }
    const fp8_t* wbase = W + (size_t)n0 * K;
    const fp8_t* xbase = X + (size_t)m0 * K;
    const uint32_t ldsW = lds_offset_of(&sW[0][0]), ldsX = lds_offset_of(&sX[0][0]), ldsS = lds_offset_of(&sS[0][0]);
    // the activation scales of the block ride with it: waves 0 and 1 bring 64 tokens' values each (a 4-byte DMA piece).  As
    // plain loads one step ahead (rounds 2-5a) they put an `s_waitcnt vmcnt(0)` of the compiler's into the middle of the MFMA
    // stream -- for `cur = nxt` -- which also waited for the tile just requested: no overlap left within a wave.
    const uint32_t soff = (uint32_t)(min(m0 + 64 * (wave & (TM / 64 - 1)) + lane, M - 1) * KB * 4);
    auto issue = [&](int kb, int buf) {
        const uint32_t dst = (uint32_t)(buf * kTileBytes + wave * 4096);
        glds16x4_sbase(wbase + (size_t)kb * 128, woff[0], woff[1], woff[2], woff[3], ldsW + dst);
        if constexpr (TM == 128) {
            glds16x4_sbase(xbase + (size_t)kb * 128, xoff[0], xoff[1], xoff[2], xoff[3], ldsX + dst);
        } else {
            const uint32_t dx = (uint32_t)(buf * (TM * kTileK) + wave * 2048);
            glds16_sbase(xbase + (size_t)kb * 128, xoff[0], ldsX + dx);
            glds16_sbase(xbase + (size_t)kb * 128, xoff[1], ldsX + dx + 1024);
        }
        if (wave < TM / 64)
            glds4_sbase(XS + kb, soff, ldsS + (uint32_t)(buf * (TM * 4) + (wave & (TM / 64 - 1)) * 256));
    };
    const float* wsp = WS + (size_t)(n0 >> 7) * KB;  // b_s: one scalar per workgroup and block (scalar loads)
    const int foff = kblock_frag_off(j, g);  // this lane's fragment inside a 16-row tile (second half: ^ 64)

    f32x4 acc[4][MT];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // double-buffered: block kb + 1 lands while block kb is multiplied; ONE barrier per step (it publishes block kb and retires the
    // buffer block kb - 1 was read from, which the step's DMA then refills).  A third stage for the 64-token form (two workgroups
    // per CU still fit) measured equal or slower (profiles/r06_ab_fp8_tiled_tm64.txt).
    issue(0, 0);
    float ws_cur = wsp[0];
    // (the buffer index is a compile-time constant of the step: the loop runs two steps per trip, so every fragment read is its
    // lane's base address + an immediate offset -- with a run-time buffer index each step recomputed ~10 addresses on the VALU)
    auto step = [&](int kb, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        TILED_MARK(kb, 0);
        glds_wait_all();  // block kb has landed (this wave's pieces) ...
        TILED_MARK(kb, 1);
        __syncthreads();  // ... and everyone's; the buffer of block kb - 1 is free
        TILED_MARK(kb, 2);
        const float ws_nxt = wsp[min(kb + 1, KB - 1)];
        if (kb + 1 < KB) issue(kb + 1, buf ^ 1);
        TILED_MARK(kb, 3);
        // fragments: lane (j, g) takes chunks g and g + 4 of row j of each tile -- the same k subset for the weight rows and
        // the token rows, which is all the dot product needs
        i32x4 wa[4][2], xb[MT][2];
        float sc[MT];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint8_t* wr = &sW[buf][(wn * 64 + t * 16) * 128];
            wa[t][0] = *reinterpret_cast<const i32x4*>(wr + foff);
            wa[t][1] = *reinterpret_cast<const i32x4*>(wr + (foff ^ 64));
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const uint8_t* xr = &sX[buf][(wm * (TM / 2) + mt * 16) * 128];
            xb[mt][0] = *reinterpret_cast<const i32x4*>(xr + foff);
            xb[mt][1] = *reinterpret_cast<const i32x4*>(xr + (foff ^ 64));
            sc[mt] = sS[buf][wm * (TM / 2) + mt * 16 + j];
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const f32x4 d = mfma_fp8_k128(wa[nt][0], wa[nt][1], xb[mt][0], xb[mt][1]);
                tiled_fold(acc[nt][mt], d, sc[mt], ws_cur);
            }
#ifdef CHITU_PROBE
        if (acc[3][MT - 1][3] == 12345.678f) g_probe_marks[31] = 1;  // (the stamp below waits for the step's last fold)
#endif
        TILED_MARK(kb, 4);
        ws_cur = ws_nxt;
    };
    for (int kb = 0; kb < KB; kb += 2) {
        step(kb, std::integral_constant<int, 0>{});
        if (kb + 1 < KB) step(kb + 1, std::integral_constant<int, 1>{});
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
            if (out_dt == 2) {
                float* dst = (float*)out + (size_t)m * N + n;
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

// chitu_hip_fp8_gemm_blockscale's large-M form (declared in gemm_common.h, called from fp8_gemm.hip)
void launch_fp8_gemm_tiled(const fp8_t* a, const float* a_s, const fp8_t* b, const float* b_s, void* out, int out_dt,
                           int64_t M, int64_t N, int64_t K, hipStream_t st) {
    // 64-token tiles while 128-token ones would leave CUs with fewer than two workgroups (two per CU fit, and two are what
    // hides a workgroup's barrier and request gaps): R1's wqkv_a at 2048 tokens is 272 tiles of 128.  Option kOptFp8TiledTM forces.
    const int64_t tiles128 = ((M + 127) / 128) * ((N + kTileN - 1) / kTileN);
    int tm = tiles128 < kTiledSmallGrid ? 64 : 128;
    debug_override(kOptFp8TiledTM, tm);
    if (tm != 64) tm = 128;
#if CHITU_TILED_XCD
    const XcdTiling t = xcd_tiling((int)((M + tm - 1) / tm), (int)((N + kTileN - 1) / kTileN));
    const dim3 grid((unsigned)(8 * t.Mt * t.Nt));
#else
    const dim3 grid((unsigned)((N + kTileN - 1) / kTileN), (unsigned)((M + tm - 1) / tm));
#endif
    if (tm == 64)
        hipLaunchKernelGGL(fp8_gemm_tiled_kernel<64>, grid, dim3(256), 0, st, a, a_s, b, b_s, out, out_dt, (int)M, (int)N, (int)K);
    else
        hipLaunchKernelGGL(fp8_gemm_tiled_kernel<128>, grid, dim3(256), 0, st, a, a_s, b, b_s, out, out_dt, (int)M, (int)N, (int)K);
}

}  // namespace chitu

// Host-only: the tile order fp8_gemm_tiled_kernel walks, for a CPU test of the map (every tile exactly once, each XCD's
// workgroups inside one rectangle).  tile_of_wg: 2 x grid ints (tile_m, tile_n; -1, -1 for a padding workgroup).
extern "C" int chitu_hip_selftest_xcd_tile_order(int32_t tiles_m, int32_t tiles_n, int32_t* grid_out, int32_t* tile_of_wg,
                                                 int64_t capacity) {
    using namespace chitu;
    CHITU_REQUIRE(tiles_m >= 1 && tiles_n >= 1 && grid_out);
    const XcdTiling t = xcd_tiling(tiles_m, tiles_n);
    const int grid = 8 * t.Mt * t.Nt;
    *grid_out = grid;
    if (!tile_of_wg) return CHITU_OK;
    CHITU_REQUIRE(capacity >= grid);
    for (int wg = 0; wg < grid; ++wg) {
        int tm = -1, tn = -1;
        if (!xcd_tile_of(wg, tiles_m, tiles_n, tm, tn)) tm = tn = -1;
        tile_of_wg[2 * wg] = tm;
        tile_of_wg[2 * wg + 1] = tn;
    }
    return CHITU_OK;
}

CHITU_PROBE_READER(fp8_gemm_tiled)
