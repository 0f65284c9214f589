// Fused MoE for PREFILL-sized batches: the two grouped FP8 GEMMs tiled for compute (64 sorted slots x 128 weight rows
// per workgroup), gfx950.
//
// Replaces (reference, read-only), for hundreds of tokens and more:
//   chitu/fused_moe.py:62-307     fused_moe_kernel (Triton grouped GEMM, BLOCK_SIZE_M = 64 tiles -- the tile height
//                                 this file uses too: moe_align_block_size is run with block 64 here)
//   chitu/fused_moe.py:24-39      SiluAndMul (fused into GEMM1's epilogue)
// The decode kernels (moe.hip) give every 16-slot tile its own stream of the expert's weights: right when an expert
// sees 1-3 tokens, 4.2 GB of L2 -> CU traffic per layer at 2048 prompt tokens (64 tokens per expert:
// profiles/r02_prefill_*: 1.3 ms of a 2.0 ms layer).  Here a workgroup stages a 64-slot activation tile (rows gathered
// through sorted_token_ids) and a 128-row weight tile of the tile's expert in LDS per 128-wide K block, double-buffered
// like fp8_gemm_tiled.hip, and every wave multiplies 2 weight-row tiles x 4 slot tiles per block.
//   GEMM1 + SiLU:  the 128 weight rows are 64 gate rows [n0, n0+64) and the 64 up rows [I+n0, I+n0+64) of W1 [E, 2I, K];
//                  wave w owns gate rows 16w.. and the matching up rows, so g and u of one output meet in one lane:
//                  h = bf16(bf16(silu(bf16(g))) * bf16(u)) -> bf16 [numel, I]   (moe_gemm1_silu_kernel's rounding points)
//   GEMM2:         128 rows of W2 [E, N, I]; out[slot, n] = bf16(acc * routed_weight[slot]) (moe_gemm2_kernel's)
// Scales: (dot * a_s[slot row]) * w_s per K block, the reference's order (fused_moe.py:281).
#include "common.h"
#include "gemm_common.h"
#include "lds_dma.h"

namespace chitu {

// Slots per tile = the moe_align block size of this path: a template parameter TM, 64 or 128 (the launcher is told which).
//   64 (rounds 2-5): a 2048-token prompt puts ~72 slots on every expert of a 257-expert layer, i.e. two 64-slot blocks on most of
//   them, and the second block streams the expert's weights again -- 1.41 GB fetched per GEMM1 launch against 1.08 GB algorithmic.
//   128 (round 6): one block for nearly every expert, its weights streamed once; the block's all-padding 16-slot sub-tiles are
//   SKIPPED (wave-uniform count of the sub-tiles that hold a token), so the padded half costs loads of a repeated row and no MFMA /
//   scale fold -- round 5 built 128-slot tiles without the skip (and under a compiler-placed wait in the K loop) and measured
//   them slower; two stages instead of three keep two workgroups on a CU (66 KB each).  profiles/r05_ab_moe_tiled.txt, r06_ab_moe_tiled128.txt
// Both tiles of a K block arrive by LDS-DMA (lds_dma.h): unpadded [rows][128 B] with the 16-byte chunks XOR-permuted on the
// source side -- no staging registers, no ds_write pass, conflict-free fragment reads.

__device__ __forceinline__ float moe_tiled_routed_weight(const void* topk_w, int w_dt, int slot) {
    if (w_dt == 0) return bf16_to_f32(((const bf16_t*)topk_w)[slot]);
    if (w_dt == 1) return f16_to_f32(((const uint16_t*)topk_w)[slot]);
    return ((const float*)topk_w)[slot];
}

// probe builds (-DCHITU_PROBE, tools/probe_tiled_steps.py moe): shader-clock stamps of workgroup (0, 0)'s thread 0 at five points of
// steps 8 .. 13 -- top, own DMA pieces landed, barrier passed, next stage requested, block multiplied
#ifdef CHITU_PROBE
#define MOE_TILED_MARK(t, n)                                                                                                  \
    do {                                                                                                                      \
        if ((t) >= 8 && (t) < 14 && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) g_probe_marks[((t) - 8) * 5 + (n)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define MOE_TILED_MARK(t, n) do {} while (0)
#endif
#ifndef CHITU_MOE_TILED_NREP
#define CHITU_MOE_TILED_NREP 4  // 1: a workgroup per tile always (A/B builds, tools/build_variant.sh)
#endif

constexpr int moe_ring_of(int tm) { return tm == 128 ? 2 : 3; }  // stages of (W tile + X tile + scales): 25 KB each at 64 slots, 33 KB at 128
struct MoeTileScales {  // the weight tile's block scales of a K block (workgroup-uniform: scalar loads)
    float ws0, ws1;
};

// grid: GEMM1 form 1-D (XCD-ordered (m-block, n-tile) pairs); GEMM2 form (weight-row tile groups, max m-blocks); block 256.
//   SILU: Nw = 2I rows per expert, blockIdx.x covers output columns [64 bx, 64 bx + 64); `out` = h [numel, I].
//   else: Nw = N rows per expert, blockIdx.x covers rows [128 bx, +128); `out` = [numel, Nw] scaled by the routed weight.
// row_div: activation row of slot s = s / row_div (topk for GEMM1: the token; 1 for GEMM2: the slot's own h row).
// NREP (GEMM2 form only): consecutive 128-row weight tiles a workgroup walks with ONE pipeline -- (tile, K block) pairs are
// one sequence of steps, the next step's operands are in flight while this one is multiplied, a tile's C is stored when its
// last K block is done.  With K = 256 (two K blocks, R1 at TP=8) a workgroup per tile is all prologue and epilogue: its
// dependent chain (padded count -> expert id -> slot ids -> rows -> LDS -> MFMA -> store) is paid once per NREP tiles.
template <bool SILU, int NREP = 1, int TM = 64>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void moe_gemm_tiled_kernel(
    const fp8_t* __restrict__ Xq, const float* __restrict__ Xs, const fp8_t* __restrict__ W, const float* __restrict__ Ws,
    const int32_t* __restrict__ sorted_ids, const int32_t* __restrict__ expert_ids,
    const int32_t* __restrict__ num_post_pad, bf16_t* __restrict__ out, const void* __restrict__ topk_w, int w_dt,
    int numel, int row_div, int Nw, int K) {
    // a ring of kMoeRing stages (one K block of both operand tiles + the slots' scales each): 25 KB a stage, two resident
    // workgroups per CU; two stages are in flight while one is multiplied.  (Measured against a ring of two once the K loop held
    // no compiler wait any more: within noise -- at 2048 tokens these launches are HBM-bound, profiles/r05_ab_moe_tiled.txt.
    // Also measured and not kept: the XCD sequence in groups of four m-blocks, n-tile by n-tile, so that the blocks of one
    // expert stream the same weight rows back to back: 275 vs 270-274 us.)
    constexpr int kMoeTileM = TM, kMoeRing = moe_ring_of(TM);
    static_assert(TM == 64 || TM == 128, "slot tiles of 64 or 128");
    __shared__ __attribute__((aligned(16))) uint8_t sW[kMoeRing][128 * 128];
    __shared__ __attribute__((aligned(16))) uint8_t sX[kMoeRing][kMoeTileM * 128];
    __shared__ __attribute__((aligned(16))) float sS[kMoeRing][4 * 64];  // the K block's activation scales: wave w's piece = slots (TM / 4) w ..
    // GEMM1 form: a 1-D grid walked XCD-aware.  Workgroup L of every run of 8 * n_tiles goes to XCD L % 8 (round-robin
    // dispatch); XCD x is given the CONTIGUOUS m-blocks [x C, x C + C) (C = an eighth of the padded blocks), one per run, all
    // n-tiles of it inside the run: the n-tiles of one m-block -- they stage the same gathered activation rows, 459 KB per 64
    // slots at K = 7168 -- share one L2, and so do the m-blocks of one expert (consecutive in the sorted order).
    constexpr int MT = kMoeTileM / 16;  // slot tiles per workgroup (every wave multiplies all of them)
    int mb, ntile;
    if (SILU) {
        const int n_tiles = Nw >> 7;  // (2I / 128) = I / 64 output-column tiles
        const int L = blockIdx.x, run = L / (8 * n_tiles), within = L % (8 * n_tiles);
        const int nb = (*num_post_pad + kMoeTileM - 1) / kMoeTileM, C = (nb + 7) >> 3;
        if (run >= C) return;
        mb = (within & 7) * C + run;
        ntile = within >> 3;
        if (mb >= nb) return;
    } else {
        mb = blockIdx.y;
        ntile = blockIdx.x;
        if (mb * kMoeTileM >= *num_post_pad) return;
    }
    // a block whose first slot is already padding holds no token at all (real slots come first in an expert's segment):
    // nothing to multiply, nothing to store
    if (sorted_ids[mb * kMoeTileM] >= numel) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int e = expert_ids[mb];
    const int I = Nw >> 1;
    static_assert(!SILU || NREP == 1, "the GEMM1 form keeps one tile per workgroup");
    const int n0 = SILU ? ntile * 64 : ntile * 128 * NREP;
    const int KB = K >> 7;

    // this lane's output slots (token column j of each of the 4 slot tiles)
    int slot[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) slot[mt] = sorted_ids[mb * kMoeTileM + mt * 16 + j];
    // 16-slot sub-tiles that hold a token (real slots come first in a block): the others are neither multiplied nor stored
    int mt_valid = 0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        if (__builtin_amdgcn_ballot_w64(slot[mt] < numel) != 0) mt_valid = mt + 1;
    if (e < 0) {  // another rank's expert (expert parallelism): its slots are zero-filled, fused_moe.py:40-59
        const int cols = SILU ? 64 : 128 * NREP, ldo = SILU ? I : Nw;
        for (int idx = tid; idx < kMoeTileM * (cols / 8); idx += 256) {
            const int r = idx / (cols / 8), c = idx % (cols / 8);
            const int s = sorted_ids[mb * kMoeTileM + r];
            if (s < numel && n0 + c * 8 + 7 < ldo) *reinterpret_cast<i32x4*>(out + (size_t)s * ldo + n0 + c * 8) = i32x4{0, 0, 0, 0};
        }
        return;
    }

    // staging roles (LDS-DMA pieces of 8 rows, lds_dma.h): wave w brings weight pieces 4 w .. 4 w + 3 and activation pieces
    // (kMoeTileM / 32) w ..; byte offsets from the expert's first weight row / the activation matrix (32-bit: the launcher
    // bounds both), rows past the matrix re-read its last row, padded slots a valid row (never stored)
    constexpr int XP = kMoeTileM / 32;  // activation pieces per wave
    const fp8_t* We = W + (size_t)e * Nw * K;
    auto w_off = [&](int i, int rep) -> uint32_t {
        const int n = wave * 4 + i, r = n * 8 + (lane >> 3);
        const int row = SILU ? (r < 64 ? n0 + r : I + n0 + (r - 64)) : n0 + rep * 128 + r;
        return (uint32_t)(min(row, Nw - 1) * K + kblock_src_chunk(lane, n) * 16);
    };
    uint32_t woff[4], xoff[XP];
#pragma unroll
    for (int i = 0; i < 4; ++i) woff[i] = w_off(i, 0);
#pragma unroll
    for (int i = 0; i < XP; ++i) {
        const int n = wave * XP + i;
        const int s = sorted_ids[mb * kMoeTileM + n * 8 + (lane >> 3)];
        xoff[i] = (uint32_t)((min(s, numel - 1) / row_div) * K + kblock_src_chunk(lane, n) * 16);
    }
    // the slots' activation scales ride with the tile: every wave brings the block's values of 16 slots (one 4-byte DMA piece,
    // lanes 0-15; the other lanes repeat them into the piece's unused part -- the same number of pieces for every wave keeps
    // the counted wait below one constant).  As plain loads one step ahead they put a vmcnt(0) of the compiler's into the
    // MFMA stream (fp8_gemm_tiled.hip).
    const uint32_t soff = (uint32_t)((min(sorted_ids[mb * kMoeTileM + wave * (TM / 4) + (lane & (TM / 4 - 1))], numel - 1) / row_div) * KB * 4);
    const float* wsb = Ws + (size_t)e * ((Nw + 127) >> 7) * KB;
    const float* wsp0 = wsb + (size_t)(n0 >> 7) * KB;
    const float* wsp1 = SILU ? wsb + (size_t)((I + n0) >> 7) * KB : wsp0;
    const uint32_t ldsW = lds_offset_of(&sW[0][0]), ldsX = lds_offset_of(&sX[0][0]), ldsS = lds_offset_of(&sS[0][0]);

    using Scales = MoeTileScales;
    constexpr int kPieces = 4 + XP + 1;  // DMA pieces per wave and stage
    auto issue = [&](int stage, int kb) {
        const uint32_t b = (uint32_t)stage;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            glds16_sbase(uniform_ptr(We + (size_t)kb * 128), woff[i], ldsW + b * (128 * 128) + (uint32_t)((wave * 4 + i) * 1024));
#pragma unroll
        for (int i = 0; i < XP; ++i)
            glds16_sbase(uniform_ptr(Xq + (size_t)kb * 128), xoff[i], ldsX + b * (kMoeTileM * 128) + (uint32_t)((wave * XP + i) * 1024));
        glds4_sbase(uniform_ptr(Xs + kb), soff, ldsS + b * 1024 + (uint32_t)(wave * 256));
    };
    auto fetch_scales = [&](Scales& r, int kb, int rep) {
        r.ws0 = wsp0[(size_t)rep * KB + kb];  // one 128-row tile = one row of block scales
        r.ws1 = SILU ? wsp1[kb] : r.ws0;
    };
    const int foff = kblock_frag_off(j, g);  // this lane's fragment inside a 16-row tile (second half: ^ 64)

    // the wave's two weight-row tiles inside the staged 128 rows
    const int wrow0 = SILU ? 16 * wave : 32 * wave, wrow1 = SILU ? 64 + 16 * wave : 32 * wave + 16;
    f32x4 acc[2][MT];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // the slots' routed weights (GEMM2 form), fetched once up front: fetched when a tile leaves (round 5a) they were loads inside
    // the step loop, whose wait also drained the next step's DMA
    float rw[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) rw[mt] = 1.f;
    if (!SILU && topk_w) {  // (one uniform branch per element type: the four loads of a lane are in flight together)
        if (w_dt == 2) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) rw[mt] = ((const float*)topk_w)[min(slot[mt], numel - 1)];
        } else {
            uint16_t raw[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) raw[mt] = ((const uint16_t*)topk_w)[min(slot[mt], numel - 1)];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) rw[mt] = w_dt == 0 ? bf16_to_f32(raw[mt]) : f16_to_f32(raw[mt]);
        }
    }
    // C tile (nt, mt): lane holds weight rows 4g .. 4g+3 of the tile for slot column j
    auto store_tile = [&](int nb) {  // nb = first weight row (GEMM2) / output column (GEMM1) of the finished tile
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int s = slot[mt];
            if (s >= numel) continue;
            const float rwm = rw[mt];
            if (SILU) {
                const int n = nb + 16 * wave + 4 * g;  // output column of r = 0
                uint16_t h[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float gv = round_bf16(acc[0][mt][r]), uv = round_bf16(acc[1][mt][r]);  // GEMM1's bf16 output (c1)
                    const float sl = round_bf16(gv / (1.0f + expf(-gv)));
                    h[r] = f32_to_bf16(sl * uv);
                }
                bf16_t* dst = out + (size_t)s * I + n;
                if (n + 3 < I) {
                    i32x2 o;
                    o[0] = (int)((uint32_t)h[0] | ((uint32_t)h[1] << 16));
                    o[1] = (int)((uint32_t)h[2] | ((uint32_t)h[3] << 16));
                    *reinterpret_cast<i32x2*>(dst) = o;
                } else {
                    for (int r = 0; r < 4 && n + r < I; ++r) dst[r] = h[r];
                }
            } else {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int n = nb + 32 * wave + 16 * nt + 4 * g;
                    uint16_t h[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[r] = f32_to_bf16(acc[nt][mt][r] * rwm);
                    bf16_t* dst = out + (size_t)s * Nw + n;
                    if (n + 3 < Nw) {
                        i32x2 o;
                        o[0] = (int)((uint32_t)h[0] | ((uint32_t)h[1] << 16));
                        o[1] = (int)((uint32_t)h[2] | ((uint32_t)h[3] << 16));
                        *reinterpret_cast<i32x2*>(dst) = o;
                    } else {
                        for (int r = 0; r < 4 && n + r < Nw; ++r) dst[r] = h[r];
                    }
                }
            }
        }
    };

    // (tile, K block) steps: the next step's operands land while this one is multiplied
    const int reps = NREP == 1 ? 1 : min(NREP, (Nw - n0 + 127) >> 7);
    const int steps = reps * KB;
    Scales cur, nxt;
    // the issue pointer runs kMoeRing - 1 steps ahead of the multiply pointer
    int ikb = 0, irep = 0, istage = 0;
    auto issue_next = [&]() {
        if (NREP > 1 && ikb == 0 && irep > 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) woff[i] = w_off(i, irep);
        }
        issue(istage, ikb);
        if (++ikb == KB) ikb = 0, ++irep;
        if (++istage == kMoeRing) istage = 0;
    };
    // every load of the prologue is consumed HERE, ahead of the first request: a wait of the compiler's placed behind it would,
    // counting in order, wait for the tiles as well
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(woff[i]));
#pragma unroll
    for (int i = 0; i < XP; ++i) asm volatile("" ::"v"(xoff[i]));
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) asm volatile("" ::"v"(rw[mt]), "v"(slot[mt]));
    asm volatile("" ::"v"(soff));
    issue_next();
    if (kMoeRing == 3 && steps > 1) issue_next();
    fetch_scales(cur, 0, 0);
    int rep = 0, kb = 0, buf = 0;
    for (int t = 0; t < steps; ++t) {
        int nkb = kb + 1, nrep = rep;
        if (nkb == KB) nkb = 0, nrep = rep + 1;
        // stage t has landed (this wave's pieces; stage t + 1, requested a step ago, may still be in flight) ...
        MOE_TILED_MARK(t, 0);
        if (kMoeRing == 3 && t + 1 < steps) glds_wait_leaving<kPieces>();
        else glds_wait_all();
        MOE_TILED_MARK(t, 1);
        __syncthreads();  // ... and everyone's; everyone is done with stage t - 1, whose buffer the request below overwrites
        MOE_TILED_MARK(t, 2);
        if (t + kMoeRing - 1 < steps) issue_next();
        MOE_TILED_MARK(t, 3);
        if (t + 1 < steps) fetch_scales(nxt, nkb, nrep);
        i32x4 wa[2][2];
        {
            const uint8_t* w0 = &sW[buf][wrow0 * 128];
            const uint8_t* w1 = &sW[buf][wrow1 * 128];
            wa[0][0] = *reinterpret_cast<const i32x4*>(w0 + foff);
            wa[0][1] = *reinterpret_cast<const i32x4*>(w0 + (foff ^ 64));
            wa[1][0] = *reinterpret_cast<const i32x4*>(w1 + foff);
            wa[1][1] = *reinterpret_cast<const i32x4*>(w1 + (foff ^ 64));
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (TM > 64 && mt >= mt_valid) continue;  // (wave-uniform) a sub-tile of padding only
            const uint8_t* xr = &sX[buf][mt * 16 * 128];
            const i32x4 xb0 = *reinterpret_cast<const i32x4*>(xr + foff), xb1 = *reinterpret_cast<const i32x4*>(xr + (foff ^ 64));
            // wave w's scale piece holds slots (TM / 4) w ..: sub-tile mt's 16 values sit in piece mt / (TM / 64) at (mt % (TM / 64)) * 16
            const float sc = sS[buf][(mt / (TM / 64)) * 64 + (mt % (TM / 64)) * 16 + j];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const f32x4 d = mfma_fp8_k128(wa[nt][0], wa[nt][1], xb0, xb1);
                const float wsc = nt == 0 ? cur.ws0 : cur.ws1;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[nt][mt][r] += (d[r] * sc) * wsc;
            }
        }
#ifdef CHITU_PROBE
        if (acc[1][0][3] == 12345.678f) g_probe_marks[31] = 1;  // (the stamp below waits for a fold of this step)
#endif
        MOE_TILED_MARK(t, 4);
        if (kb == KB - 1) {  // this tile's last K block: its C leaves now, under the next tile's loads
            store_tile(n0 + rep * 128);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (t + 1 < steps) cur = nxt;
        kb = nkb, rep = nrep;
        if (++buf == kMoeRing) buf = 0;
    }
}

}  // namespace chitu

extern "C" int chitu_hip_moe_gemm1_silu_fp8_tiled(const void* a_fp8, const float* a_scale, const void* w1_fp8,
                                                  const float* w1_scale, const int32_t* sorted_token_ids,
                                                  const int32_t* expert_ids, const int32_t* num_tokens_post_pad,
                                                  void* h_bf16, int64_t numel, int32_t topk, int64_t inter_size,
                                                  int64_t K, int64_t max_mblocks, int32_t block_m, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(a_fp8 && a_scale && w1_fp8 && w1_scale && sorted_token_ids && expert_ids && num_tokens_post_pad && h_bf16);
    CHITU_REQUIRE(numel >= 0 && numel < (1ll << 31) && topk >= 1 && inter_size >= 1 && K >= 128 && max_mblocks >= 0);
    if (K % 128 != 0 || inter_size % 128 != 0 || inter_size >= (1 << 29) || K >= (1 << 30)) return CHITU_ERR_UNSUPPORTED;
    if (2 * inter_size * K >= (1ll << 31) || (numel / topk + 1) * K >= (1ll << 31)) return CHITU_ERR_UNSUPPORTED;  // 32-bit tile offsets
    if (block_m != 64 && block_m != 128) return CHITU_ERR_UNSUPPORTED;  // the moe_align block size the ids were sorted with
    if (numel == 0 || max_mblocks == 0) return CHITU_OK;
    CHITU_REQUIRE(max_mblocks <= 65535);
    const dim3 grid((unsigned)((inter_size / 64) * ((max_mblocks + 7) / 8 * 8)));  // (m-block, n-tile) pairs in XCD order, see the kernel
#define LAUNCH1T(TMV)                                                                                                              \
    hipLaunchKernelGGL((moe_gemm_tiled_kernel<true, 1, TMV>), grid, dim3(256), 0, (hipStream_t)stream, (const fp8_t*)a_fp8, a_scale, \
                       (const fp8_t*)w1_fp8, w1_scale, sorted_token_ids, expert_ids, num_tokens_post_pad, (bf16_t*)h_bf16,           \
                       (const void*)nullptr, 0, (int)numel, (int)topk, (int)(2 * inter_size), (int)K)
    if (block_m == 128) LAUNCH1T(128);
    else LAUNCH1T(64);
#undef LAUNCH1T
    CHITU_RETURN_LAUNCH_STATUS();
}

extern "C" int chitu_hip_moe_gemm2_fp8_tiled(const void* h_fp8, const float* h_scale, const void* w2_fp8,
                                             const float* w2_scale, const int32_t* sorted_token_ids,
                                             const int32_t* expert_ids, const int32_t* num_tokens_post_pad,
                                             const void* topk_weights, int weights_dtype, int32_t mul_routed_weight,
                                             void* out_bf16, int64_t numel, int64_t N, int64_t inter_size,
                                             int64_t max_mblocks, int32_t block_m, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(h_fp8 && h_scale && w2_fp8 && w2_scale && sorted_token_ids && expert_ids && num_tokens_post_pad && out_bf16);
    CHITU_REQUIRE(numel >= 0 && numel < (1ll << 31) && N >= 1 && inter_size >= 128 && max_mblocks >= 0);
    CHITU_REQUIRE(!mul_routed_weight || (topk_weights && weights_dtype >= 0 && weights_dtype <= 2));
    if (inter_size % 128 != 0 || N % 8 != 0 || N >= (1 << 30) || inter_size >= (1 << 30)) return CHITU_ERR_UNSUPPORTED;
    if (N * inter_size >= (1ll << 31) || (numel + 1) * inter_size >= (1ll << 31)) return CHITU_ERR_UNSUPPORTED;  // 32-bit tile offsets
    if (block_m != 64 && block_m != 128) return CHITU_ERR_UNSUPPORTED;  // the moe_align block size the ids were sorted with
    if (numel == 0 || max_mblocks == 0) return CHITU_OK;
    CHITU_REQUIRE(max_mblocks <= 65535);
    const int n_tiles = (int)((N + 127) / 128);
#define LAUNCH2T(NREPV)                                                                                                  \
    if (block_m == 128)                                                                                                  \
    hipLaunchKernelGGL((moe_gemm_tiled_kernel<false, NREPV, 128>), dim3((unsigned)((n_tiles + NREPV - 1) / NREPV), (unsigned)max_mblocks), \
                       dim3(256), 0, (hipStream_t)stream, (const fp8_t*)h_fp8, h_scale, (const fp8_t*)w2_fp8, w2_scale,         \
                       sorted_token_ids, expert_ids, num_tokens_post_pad, (bf16_t*)out_bf16,                                    \
                       mul_routed_weight ? topk_weights : (const void*)nullptr, (int)weights_dtype, (int)numel, 1, (int)N,      \
                       (int)inter_size);                                                                                        \
    else                                                                                                                 \
    hipLaunchKernelGGL((moe_gemm_tiled_kernel<false, NREPV>), dim3((unsigned)((n_tiles + NREPV - 1) / NREPV), (unsigned)max_mblocks), \
                       dim3(256), 0, (hipStream_t)stream, (const fp8_t*)h_fp8, h_scale, (const fp8_t*)w2_fp8, w2_scale,         \
                       sorted_token_ids, expert_ids, num_tokens_post_pad, (bf16_t*)out_bf16,                                    \
                       mul_routed_weight ? topk_weights : (const void*)nullptr, (int)weights_dtype, (int)numel, 1, (int)N,      \
                       (int)inter_size)
    // few K blocks (R1 at TP=8: two): four tiles per workgroup through one pipeline; long K: a tile per workgroup
    if (inter_size <= 512 && n_tiles >= 8 && CHITU_MOE_TILED_NREP > 1) LAUNCH2T(CHITU_MOE_TILED_NREP);
    else LAUNCH2T(1);
#undef LAUNCH2T
    CHITU_RETURN_LAUNCH_STATUS();
}

CHITU_PROBE_READER(moe_tiled)
