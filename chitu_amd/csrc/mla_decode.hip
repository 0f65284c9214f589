// MLA (absorb mode) paged decode attention for gfx950: split-KV flash decoding in latent space.
//
// Replaces (reference, read-only):
//   chitu/triton_decode_attention.py:20-130   _mla_attn_kernel      (stage 1, fixed 4 splits)
//   chitu/triton_decode_attention.py:185-232  _mla_softmax_reducev  (stage 2, LSE merge)
//   chitu/attn_backend.py:707-774             TritonAttnBackend.mla_attn_with_kvcache
//   (and the third-party flash_mla / flashinfer calls at attn_backend.py:561-571, 678-684)
//
//   out[b,h,:] = softmax_t( scale * (q_nope[b,h,:] . c[t,:512] + q_pe[b,h,:] . c[t,512:]) ) . c[t,:512]
// over the first seqlens[b] cached tokens; V *is* the 512-wide latent (the MLA trick), so a KV
// tile is staged in LDS once and used for both products.
//
// Design.  One workgroup (4 waves) = 16 heads x one KV split of one sequence.  Per 64-token
// tile (64 x 576 bf16 rows, brought into LDS by LDS-DMA, double-buffered; layout below): QK^T: each wave owns
// 16 tokens, 18 x v_mfma_f32_16x16x32_bf16 with Q (A operand) in registers;
// online softmax in fp32 with 16-lane shuffles + a 4-wave LDS exchange; P -> bf16 -> LDS;
// PV: each wave owns 128 latent columns, V^T fragments come straight from the row-major tile
// with ds_read_b64_tr_b16 (gfx950 transpose read), 16 MFMAs per wave per tile.  Splits are
// sized on the host from the batch only (graph-static); a split's token range is derived from
// the device-side sequence length, empty splits publish LSE = -inf.  Stage 2 merges splits.
#include "common.h"
#include "lds_dma.h"

namespace chitu {

constexpr int kC = 512;        // kv_lora_rank (latent / V width)
constexpr int kR = 64;         // qk_rope_head_dim
constexpr int kD = kC + kR;    // cached row width (576)
constexpr int kTile = 64;      // KV tokens per tile
constexpr int kPStride = 72;   // P row stride in bf16 elements (64 + 8 pad)
constexpr int kMaxTilesLds = 512;  // page ids cached in LDS per split (32k tokens)

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// grid (num_splits, batch, heads/16); block 256, one workgroup per CU (152 KB LDS).
//
// How bytes move (round 5; the arithmetic and its order are those of rounds 1-4, output bit-identical).  The KV tile is brought
// by LDS-DMA -- no staging registers, no ds_write pass (phase masks priced them at 4.3 of 19.4 us at bs 16 / ctx 1024 and 17
// of 44 us at ctx 8192, profiles/r03_phase_masks.txt) -- into one of TWO 64-key buffers: the next tile lands while this one
// is multiplied, one barrier per tile instead of two.  Q lives in registers (72 VGPRs: the A operand never changes), which is
// what lets two buffers fit; it gets there coalesced, through buffer 1 while that is still free.  The table's first 256
// entries are requested BESIDE seqlens, not behind it (the chain that heads the kernel is seqlens -> KV rows, not seqlens ->
// page id -> KV rows), and the DMA carries the `nt` hint: a KV row is read once per launch, left to displace the L2's
// resident lines it cost 0.55 us per launch in the step (same-box kernel trace, profiles/r05_ab_mla_decode_dma.txt:
// register-staged 13.29 us, DMA 12.07, DMA + nt 11.52; ctx 8192: 36.1 -> 27.1 us).
//
// LDS image of a tile: [64 rows][1152 B] unpadded, the 16-byte chunk c of row r stored at c ^ swz(r),
// swz(r) = 5 * bit3(r) + 2 * bit1(r).  A DMA piece is 1 KiB of the image, lane-linear (lds_dma.h): image chunk q = 64 n + lane
// -> row q / 72, position q % 72, source chunk (q % 72) ^ swz(row).  Readers: K fragments (ds_read_b128, lane groups
// {0-3,12-15,20-27},...: 16 rows with chunk g or g ^ 1) and V^T fragments (ds_read_b64_tr_b16, 32 lanes = 8 rows x 32 B)
// both land on 16 distinct 16-byte slots of the 256-byte bank row (checked exhaustively for every wave / k step;
// the padded 1184-byte rows of rounds 2-4 left the transpose reads 2-way conflicted).
constexpr int kRowU = kD * 2;            // 1152
constexpr int kTileU = kTile * kRowU;    // 73728
constexpr int kDmaPieces = kTileU / 1024 / 4;  // 18 per wave
__device__ __forceinline__ int kv_swz(int r) { return ((r >> 3) & 1) * 5 + ((r >> 1) & 1) * 2; }

// ---- Round 6: the split merge + W_UV projection + act_quant INSIDE the decode launch (FUSED = true).
//
// Two launches (decode, then mla_merge_uv_quant_kernel of absorb.hip) cost the pair two fixed launch costs, a cold W_UV fetch
// behind the boundary and the partials' trip through memory between them: 11.6 + 6.6 us at bs 16 / ctx 1024
// (profiles/r05_step_breakdown_bs16_final.txt).  Fused, every split workgroup of a (sequence, head block) publishes its partial
// rows as before (write-through), counts itself on the group's arrival word, and -- instead of leaving -- waits for the other
// splits and then does the merge + projection + quantisation of ONE head (split s takes head s, s + S, ...): the tail of a
// sequence runs on as many CUs as it has splits, not on one last arriver, and each workgroup's 64 KB of W_UV was requested
// before its first KV tile was multiplied.  Arithmetic and its order are mla_merge_uv_quant_kernel's: outputs bit-identical to
// the two launches (tests/test_gpu_mla.py), which stay as the cross-check.
//
// Hand-off (guide: "sc1 payload -> vmcnt(0) -> sc1 flag" both sides): partial rows and LSE leave as write-through (sc1) stores,
// every wave drains them, the workgroup meets, one lane adds 1 to arrive[group] (agent scope); a waiter's one lane polls that
// word with sc1 loads + s_sleep until it reads S, the workgroup meets, and the partials are read with sc1 loads (L1 bypassed:
// no acquire needed).  The words reset themselves: every waiter counts itself on depart[group] after its poll, the last one
// stores 0 to both (all S arrivals have happened: it saw S).  Forward progress: a waiter needs the other splits of its group to
// be scheduled; they precede or directly follow it in dispatch order, and a launch has at most ~one workgroup per CU
// (attn_backend.choose_num_splits), so a waiter never holds a CU that an unscheduled split of an EARLIER group needs.  The wait is
// bounded all the same (guide: bound every spin): after kFuseTimeoutTicks the waiter sets the sticky error word behind the
// counters and carries on with what is there -- a wrong row and a reported error instead of a hung GPU.
struct MlaFuse {
    const fp8_t* W;        // W_UV [H, 128, 512] e4m3, head stride w_sh
    int64_t w_sh;
    const float* scale;    // block scales of wkv_b: scale[s_off + h * s_sh + kblock * s_sk]
    int64_t s_off, s_sh, s_sk;
    fp8_t* q;              // wo's quantised input rows [batch, H * 128] (or tile-major, gemm_common.h)
    float* qs;
    int tile_major;
    uint32_t* tickets;     // [2 * kFuseMaxGroups + 1]: (arrive, depart) per (sequence, head block), then the error word; zero at creation
};
constexpr int kFuseMaxGroups = 4096;
constexpr uint64_t kFuseTimeoutTicks = 20000000ull;  // 200 ms of the 100 MHz wall clock

__device__ __forceinline__ void store16_sc1(void* dst, i32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
}
__device__ __forceinline__ uint32_t load_sc1(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// This wave's two 16-column tiles of W_UV[h] (wave w owns output columns [32 w, 32 w + 32)) and the head's four K-block scales.
struct MlaUvW {
    i32x4 w[2][8];
    float sv[4];
};
__device__ __forceinline__ void mla_uv_load(MlaUvW& r, const MlaFuse& f, int h, int wave, int j, int g) {
    const float* sp = f.scale + f.s_off + h * f.s_sh;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const fp8_t* wp = f.W + (int64_t)h * f.w_sh + (int64_t)((2 * wave + t) * 16 + j) * kC + g * 16;
#pragma unroll
        for (int c = 0; c < 8; ++c) r.w[t][c] = *reinterpret_cast<const i32x4*>(wp + c * 64);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) r.sv[c] = sp[c * f.s_sk];
}

// The tail of a split workgroup under FUSED: count, wait, then one head per round.  `scratch`: >= 2 KB of LDS nobody else uses
// any more.  `uw` holds the weights of the first head, head h0 + split (requested long ago), whenever there is one to do.
__device__ __forceinline__ void mla_fused_tail(const MlaFuse& f, MlaUvW& uw, const bf16_t* part_o, const float* part_lse, int b,
                                               int H, int h0, int split, int S, int group, uint8_t* scratch) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, g = lane >> 4;
    bf16_t* xs = reinterpret_cast<bf16_t*>(scratch);            // [512] merged row
    float* red = reinterpret_cast<float*>(scratch + 1024);      // [8] tile maxima
    uint32_t* arrive = f.tickets + 2 * group;
    uint32_t* depart = arrive + 1;
    uint32_t* err = f.tickets + 2 * kFuseMaxGroups;
    const int nh = min(16, H - h0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through partial stores have left
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (split >= nh) return;  // more splits than heads: nothing to merge here
    if (tid == 0) {
        if (load_sc1(arrive) < (uint32_t)S) {
            const uint64_t t0 = wall_clock64();
            for (unsigned spins = 0;; ++spins) {
                if (load_sc1(arrive) >= (uint32_t)S) break;
                __builtin_amdgcn_s_sleep(1);
                if ((spins & 255) == 255 && (load_sc1(err) != 0 || wall_clock64() - t0 > kFuseTimeoutTicks)) {
                    __hip_atomic_fetch_or(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
    }
    __syncthreads();
    if (tid == 64) {  // (a lane of another wave than the poller's, behind the barrier: its returning atomic is nobody's critical path)
        const uint32_t waiters = (uint32_t)min(S, nh);
        if (__hip_atomic_fetch_add(depart, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == waiters - 1) {
            __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the next launch starts from 0
            __hip_atomic_store(depart, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    for (int hm = split; hm < nh; hm += S) {
        const int h = h0 + hm;
        const int64_t bh = (int64_t)b * H + h;
        if (hm != split) mla_uv_load(uw, f, h, wave, j, g);  // (fewer splits than heads: the further heads' weights are fetched here)
        // ---- merge (mla_merge_uv_quant_kernel's sums in its order): thread t owns latent columns 2 t, 2 t + 1.  The first 16
        // partial rows are requested BEFORE the lse values are looked at (their addresses do not depend on them): one memory
        // round trip, not two, at the usual S <= 16
        const float* lse = part_lse + bh * S;
        uint32_t pv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) pv[i] = load_sc1(reinterpret_cast<const uint32_t*>(part_o + (bh * S + min(i, S - 1)) * kC + 2 * tid));
        float m = -INFINITY;
        for (int s0 = 0; s0 < S; s0 += 64) {
            const float l = s0 + lane < S ? __uint_as_float(load_sc1(reinterpret_cast<const uint32_t*>(lse + s0 + lane))) : -INFINITY;
            m = __builtin_fmaxf(m, wave_reduce_max(l));
        }
        float a0 = 0.f, a1 = 0.f, wsum = 0.f;
        for (int s0 = 0; s0 < S; s0 += 64) {
            const float l = s0 + lane < S ? __uint_as_float(load_sc1(reinterpret_cast<const uint32_t*>(lse + s0 + lane))) : -INFINITY;
            const float wl = l == -INFINITY ? 0.f : __expf(l - m);
            const int n = min(64, S - s0);
            for (int i0 = 0; i0 < n; i0 += 16) {
                if (s0 + i0 > 0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        pv[i] = load_sc1(reinterpret_cast<const uint32_t*>(part_o + (bh * S + s0 + min(i0 + i, n - 1)) * kC + 2 * tid));
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float ws = i0 + i < n ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wl), (i0 + i) & 63)) : 0.f;
                    wsum += ws;
                    a0 = merge_term(a0, ws, __uint_as_float(pv[i] << 16));
                    a1 = merge_term(a1, ws, __uint_as_float(pv[i] & 0xffff0000u));
                }
            }
        }
        const float inv = wsum > 0.f ? 1.0f / wsum : 0.f;
        *reinterpret_cast<uint32_t*>(xs + 2 * tid) = (uint32_t)f32_to_bf16(a0 * inv) | ((uint32_t)f32_to_bf16(a1 * inv) << 16);
        __syncthreads();
        // ---- o . W_UV[h]^T: the MFMA tile's 16 token columns all carry this token (LDS broadcast)
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const s16x8 xa = *reinterpret_cast<const s16x8*>(&xs[c * 64 + g * 16]);
            const s16x8 xb = *reinterpret_cast<const s16x8*>(&xs[c * 64 + g * 16 + 8]);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dequant8_bf16((uint32_t)uw.w[t][c][0], (uint32_t)uw.w[t][c][1], uw.sv[c >> 1]), xa, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dequant8_bf16((uint32_t)uw.w[t][c][2], (uint32_t)uw.w[t][c][3], uw.sv[c >> 1]), xb, acc[t], 0, 0, 0);
            }
        }
        float v[2][4], amax = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[t][r] = round_bf16(acc[t][r]);
                amax = __builtin_fmaxf(amax, __builtin_fabsf(v[t][r]));
            }
        amax = __builtin_fmaxf(amax, __shfl_xor(amax, 16, 64));
        amax = __builtin_fmaxf(amax, __shfl_xor(amax, 32, 64));
        if (lane == 0) red[wave] = amax;
        __syncthreads();
        amax = __builtin_fmaxf(__builtin_fmaxf(red[0], red[1]), __builtin_fmaxf(red[2], red[3]));
        const float sc = amax / 448.0f;
        const bool fast = __builtin_amdgcn_ballot_w64(!group_div_fast(sc)) == 0;
        const float rc = group_rcp(sc);
        if (j == 0) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float q4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) q4[k] = fast ? group_div(v[t][k], sc, rc) : v[t][k] / sc;
                const uint32_t packed = f32x2_to_fp8x2(q4[0], q4[1]) | (f32x2_to_fp8x2(q4[2], q4[3]) << 16);
                const int T = 2 * wave + t;
                if (f.tile_major) {  // row b of a [batch, H * 128] matrix, tile-major (gemm_common.h): 16-B chunk index h * 8 + T
                    const int64_t tile = b >> 4;
                    *reinterpret_cast<uint32_t*>(f.q + ((tile * (H * 8) + h * 8 + T) * 16 + (b & 15)) * 16 + g * 4) = packed;
                    if (T == 0 && g == 0) f.qs[(tile * H + h) * 16 + (b & 15)] = sc;
                } else {
                    *reinterpret_cast<uint32_t*>(f.q + bh * 128 + T * 16 + g * 4) = packed;
                    if (T == 0 && g == 0) f.qs[bh] = sc;
                }
            }
        }
        // (the next round's xs / red writes come behind its own loads and the barrier above: no third barrier needed for red;
        // xs is rewritten only after every wave has passed the red barrier, i.e. has finished its MFMA reads of xs)
    }
}

template <bool FUSED>
__global__ __launch_bounds__(256, 1) void mla_decode_kernel(
    const bf16_t* __restrict__ q_nope, int64_t qn_sb, int64_t qn_sh, const bf16_t* __restrict__ q_pe,
    int64_t qp_sb, int64_t qp_sh, const bf16_t* __restrict__ cache, int64_t num_pages, int page_size,
    const int32_t* __restrict__ block_table, int table_stride, const int32_t* __restrict__ seqlens,
    float scale, bf16_t* __restrict__ part_o, float* __restrict__ part_lse, bf16_t* __restrict__ out,
    int H, int num_splits, MlaFuse fuse) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* kv_lds = smem;                                               // [2][64][kRowU]
    bf16_t* p_lds = reinterpret_cast<bf16_t*>(smem + 2 * kTileU);         // [16][72]
    float* red_max = reinterpret_cast<float*>(smem + 2 * kTileU + 16 * kPStride * 2);  // [4][16]
    float* red_sum = red_max + 64;                                        // [4][16]
    int* pages_lds = reinterpret_cast<int*>(red_sum + 64);                // [kMaxTilesLds]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int split = blockIdx.x, b = blockIdx.y, hb = blockIdx.z;
    const int h0 = hb * 16;
    const int32_t* tbl = block_table + (int64_t)b * table_stride;
    const int max_page_idx = table_stride - 1;
    // The chain that heads this kernel is seqlens -> KV rows: the table's first 256 entries are requested with seqlens,
    // not behind it (lane l of every wave: entries l, l + 64, ...), and the one this split starts at is picked by readlane.
    // (asm: as plain loads the compiler sinks the table loads behind the selection below, i.e. behind seqlens again, and
    // schedules the seqlens load itself behind Q's address arithmetic)
    CHITU_PROBE_MARK(8);
    int L_raw;
    asm volatile("s_load_dword %0, %1, 0x0" : "=&s"(L_raw) : "s"(seqlens + b) : "memory");
    int spec[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        asm volatile("global_load_dword %0, %1, off" : "=v"(spec[k]) : "v"(tbl + min(lane + 64 * k, max_page_idx)) : "memory");
    // Q (16 heads x 576): 1152 chunks of 16 B, <= 5 per thread, coalesced; it passes through buffer 1 (free until the second
    // tile is requested) on its way to the registers of the four waves, each of which needs all of it as the A operand
    i32x4 qreg[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int c = min(tid + i * 256, 16 * 72 - 1);
        const int row = c / 72, col = c % 72;
        const int h = min(h0 + row, H - 1);
        const bf16_t* src = col < 64 ? q_nope + b * qn_sb + h * qn_sh + col * 8 : q_pe + b * qp_sb + h * qp_sh + (col - 64) * 8;
        qreg[i] = *reinterpret_cast<const i32x4*>(src);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(L_raw)::"memory");
    CHITU_PROBE_MARK(9);
    const int L = max(L_raw, 0);  // a corrupt negative length is an empty sequence
    const int n_tiles = (L + kTile - 1) / kTile;
    const int tile0 = (int)((unsigned)n_tiles * (unsigned)split / (unsigned)num_splits);
    const int tile1 = (int)((unsigned)n_tiles * (unsigned)(split + 1) / (unsigned)num_splits);
    const bool pages_in_lds = (tile1 - tile0) <= kMaxTilesLds;
    const int p_first = (tile0 * kTile) / page_size;
    // the table entries have landed (and Q, requested just behind them: same round trip, nothing lost)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(spec[0]), "+v"(spec[1]), "+v"(spec[2]), "+v"(spec[3])::"memory");
#pragma unroll
    for (int i = 0; i < 5; ++i) asm volatile("" : "+v"(qreg[i]));  // (the compiler's own wait for Q sits here, not behind the DMA below)
    int first_pg;
    if (p_first < 256) {
        const int pick = p_first < 64 ? spec[0] : p_first < 128 ? spec[1] : p_first < 192 ? spec[2] : spec[3];
        first_pg = __builtin_amdgcn_readlane(pick, p_first & 63);
    } else {
        first_pg = tbl[min(p_first, max_page_idx)];
    }
    auto page_src = [&](int64_t page, int t0) -> const bf16_t* {
        if (page < 0 || page >= num_pages) page = 0;  // corrupt table: stay in bounds
        return cache + (page * page_size + (t0 % page_size)) * (int64_t)kD;
    };
    auto tile_src = [&](int tile) -> const bf16_t* {
        const int t0 = tile * kTile;
        return page_src(pages_in_lds ? pages_lds[tile - tile0] : tbl[min(t0 / page_size, max_page_idx)], t0);
    };
    // this wave's 18 pieces of a tile: piece n = wave + 4 i; lane's image chunk 64 n + lane -> (row, source byte offset)
    int prow[kDmaPieces];
    uint32_t pswz[kDmaPieces];
#pragma unroll
    for (int i = 0; i < kDmaPieces; ++i) {
        const int qi = 64 * (wave + 4 * i) + lane;
        prow[i] = qi / 72;
        pswz[i] = (uint32_t)(((qi % 72) ^ kv_swz(prow[i])) << 4);
    }
    const uint32_t lds0 = lds_offset_of(smem);
    // rows past the sequence end re-read the tile's last valid row: finite, and their probabilities are exactly 0
    auto issue = [&](const bf16_t* src, int valid, int buf) {
        const uint64_t a = (uint64_t)src;
        const bf16_t* sb = (const bf16_t*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                                           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a));
#pragma unroll
        for (int i = 0; i < kDmaPieces; ++i)
            glds16_sbase<true>(sb, (uint32_t)(min(prow[i], valid - 1) * kRowU) + pswz[i],
                             lds0 + (uint32_t)(buf * kTileU + (wave + 4 * i) * 1024));
    };
    // FUSED: an empty split still has a head of the tail to do -- it runs the rest of the kernel over zero tiles (zero rows, LSE = -inf)
    const bool empty = FUSED && tile0 >= tile1;
    if (!FUSED && tile0 >= tile1) {  // an empty split publishes LSE = -inf and zero rows (nothing of it is read by the merge)
        if (num_splits > 1) {
            if (tid < 16 && h0 + tid < H) part_lse[((int64_t)b * H + h0 + tid) * num_splits + split] = -INFINITY;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int chunk = tid + i * 256, hr = chunk >> 6, c8 = chunk & 63;
                if (h0 + hr < H)
                    *reinterpret_cast<i32x4*>(part_o + (((int64_t)b * H + h0 + hr) * num_splits + split) * kC + c8 * 8) = i32x4{0, 0, 0, 0};
            }
        } else {
            for (int i = tid; i < 16 * kC / 8; i += 256)
                if (h0 + (i >> 6) < H) *reinterpret_cast<i32x4*>(out + ((int64_t)b * H + h0 + (i >> 6)) * kC + (i & 63) * 8) = i32x4{0, 0, 0, 0};
        }
        return;
    }
    if (!empty) issue(page_src(first_pg, tile0 * kTile), min(kTile, L - tile0 * kTile), 0);
    // FUSED: this workgroup's head of the tail (split s -> head s) -- its 64 KB of W_UV are requested once the LAST KV tile has
    // landed (no DMA is issued behind them: the counted waits stay what they are) and arrive while that tile is multiplied
    MlaUvW uvw;
    const bool uv_mine = FUSED && num_splits > 1 && split < min(16, H - h0);
    if (empty && uv_mine) mla_uv_load(uvw, fuse, h0 + split, wave, j, g);
    {   // Q into buffer 1, in the tile image's own layout (row = head, chunk c at c ^ swz(row)): read back like a K fragment
        uint8_t* q_lds = kv_lds + kTileU;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int c = tid + i * 256;
            if (c < 16 * 72) *reinterpret_cast<i32x4*>(q_lds + (c / 72) * kRowU + (((c % 72) ^ kv_swz(c / 72)) << 4)) = qreg[i];
        }
    }
    // page ids of the following tiles (none at one tile per split, the short-context shape: no table load, and no wait of the
    // compiler's for one -- which, counting in order, would also wait for the tile requested above)
    if (pages_in_lds && tile1 - tile0 > 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (tid + i * 256 < tile1 - tile0)
                pages_lds[tid + i * 256] = tbl[min(((tile0 + tid + i * 256) * kTile) / page_size, max_page_idx)];
    }

    f32x4 o[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[4], l_run[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        m_run[r] = -INFINITY;
        l_run[r] = 0.f;
    }
    // fragment addressing inside a tile image (see the layout note above)
    const int ksw = kv_swz(j);                                    // row wave*16 + j: bits 1 and 3 are j's
    const int koff0 = (wave * 16 + j) * kRowU + ((g ^ ksw) << 4);  // even k steps; odd ones: chunk ^ 4
    const int koff1 = (wave * 16 + j) * kRowU + (((g ^ ksw) ^ 4) << 4);
    const int vsw = (g & 1) * 5 + ((j >> 3) & 1) * 2;             // rows ks*32 + g*8 + (j>>2) (+4): bit 3 = g & 1, bit 1 = j >> 3
    const int vrow_off = (g * 8 + (j >> 2)) * kRowU + wave * 256 + ((((j >> 1) & 1) ^ (vsw & 1)) << 4) + (j & 1) * 8;
    int vx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) vx[k] = ((2 * k) ^ (vsw & 6)) << 4;

    __syncthreads();  // Q and the page list are visible (the first tile may still be in flight)
    s16x8 qf[18];     // lane (j, g): elements [32 kk + 8 g, +8) of head j
    {
        const uint8_t* q_lds = kv_lds + kTileU;
        const int ksw0 = kv_swz(j);
#pragma unroll
        for (int kk = 0; kk < 18; ++kk)
            qf[kk] = *reinterpret_cast<const s16x8*>(q_lds + j * kRowU + (kk >> 1) * 128 + (((g + 4 * (kk & 1)) ^ ksw0) << 4));
    }
    if (tile0 + 1 < tile1) {
        __syncthreads();  // every wave has its copy of Q: buffer 1 may be overwritten
        issue(tile_src(tile0 + 1), min(kTile, L - (tile0 + 1) * kTile), 1);
    }

    for (int tile = tile0; tile < tile1; ++tile) {
        const int buf = (tile - tile0) & 1;
        const int valid = min(kTile, L - tile * kTile);
        if (tile == tile0 && tile0 + 1 < tile1)
            asm volatile("s_waitcnt vmcnt(18)" ::: "memory");  // the first tile's pieces; the second tile's 18 stay in flight
        else
            glds_wait_all();  // this wave's pieces of the tile
        __syncthreads();   // everyone's; the other buffer and the softmax exchange areas of the previous tile are free
        if (tile > tile0 && tile + 1 < tile1) issue(tile_src(tile + 1), min(kTile, L - (tile + 1) * kTile), buf ^ 1);
        if (FUSED && tile + 1 == tile1 && uv_mine) mla_uv_load(uvw, fuse, h0 + split, wave, j, g);
        const uint8_t* kv = kv_lds + buf * kTileU;
        CHITU_PROBE_MARK(10);

        // ---- S = Q K^T for this wave's 16 tokens (two accumulators: no 18-deep dependent chain)
        f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 18; kk += 2) {
            const s16x8 k0 = *reinterpret_cast<const s16x8*>(kv + koff0 + (kk >> 1) * 128);
            const s16x8 k1 = *reinterpret_cast<const s16x8*>(kv + koff1 + (kk >> 1) * 128);
            s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[kk], k0, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[kk + 1], k1, s1, 0, 0, 0);
        }
        // lane holds S[head 4g+r][token wave*16+j]
        CHITU_PROBE_MARK(11);
        const bool tok_ok = (wave * 16 + j) < valid;
        float sv[4], mx[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sv[r] = tok_ok ? (s0[r] + s1[r]) * scale : -INFINITY;
            mx[r] = row16_reduce_max(sv[r]);
        }
        if (j == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) red_max[wave * 16 + g * 4 + r] = mx[r];
        }
        __syncthreads();
        float alpha[4], psum[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int hh = g * 4 + r;
            const float mt = __builtin_fmaxf(__builtin_fmaxf(red_max[hh], red_max[16 + hh]),
                                             __builtin_fmaxf(red_max[32 + hh], red_max[48 + hh]));
            const float m_new = __builtin_fmaxf(m_run[r], mt);  // finite: the tile's first token is valid
            alpha[r] = __expf(m_run[r] - m_new);
            m_run[r] = m_new;
            const float p = __expf(sv[r] - m_new);
            psum[r] = row16_reduce_sum(p);
            p_lds[hh * kPStride + wave * 16 + j] = f32_to_bf16(p);
        }
        if (j == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) red_sum[wave * 16 + g * 4 + r] = psum[r];
        }
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[c][r] *= alpha[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int hh = g * 4 + r;
            l_run[r] = l_run[r] * alpha[r] + (red_sum[hh] + red_sum[16 + hh] + red_sum[32 + hh] + red_sum[48 + hh]);
        }

        // ---- O += P V : this wave owns latent columns [wave*128, wave*128+128)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const s16x8 pfrag = *reinterpret_cast<const s16x8*>(p_lds + j * kPStride + ks * 32 + g * 8);
            const uint8_t* vbase = kv + vrow_off + ks * 32 * kRowU;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const uint8_t* va = vbase + vx[c & 3] + (c >> 2) * 128;
                const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(va));
                const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(va + 4 * kRowU));
                s16x8 vf;
                vf[0] = v0[0]; vf[1] = v0[1]; vf[2] = v0[2]; vf[3] = v0[3];
                vf[4] = v1[0]; vf[5] = v1[1]; vf[6] = v1[2]; vf[7] = v1[3];
                o[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pfrag, vf, o[c], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: lane holds O[head 4g+r][col wave*128 + c*16 + j]
    CHITU_PROBE_MARK(12);
    if (num_splits == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int h = h0 + g * 4 + r;
            if (h >= H) continue;
            const float inv = empty ? 0.f : 1.0f / l_run[r];
            bf16_t* dst = out + ((int64_t)b * H + h) * kC + wave * 128 + j;
#pragma unroll
            for (int c = 0; c < 8; ++c) dst[c * 16] = f32_to_bf16(o[c][r] * inv);
        }
        return;
    }
    // split partials: transposed through LDS (the KV tiles are dead) so every thread stores 16-B pieces of whole rows; they
    // leave as BF16 (the normalised o of a split is an attention output: the merge's convex combination keeps the 2^-9
    // rounding below the final output's own) + the fp32 LSE
    __syncthreads();
    bf16_t* o_lds = reinterpret_cast<bf16_t*>(kv_lds);  // [16][512] bf16 = 16 KB
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float inv = empty ? 0.f : 1.0f / l_run[r];
#pragma unroll
        for (int c = 0; c < 8; ++c) o_lds[(g * 4 + r) * kC + wave * 128 + c * 16 + j] = f32_to_bf16(o[c][r] * inv);
        const int h = h0 + g * 4 + r;
        if (wave == 0 && j == 0 && h < H) {
            float* lp = part_lse + ((int64_t)b * H + h) * num_splits + split;
            const float lv = empty ? -INFINITY : m_run[r] + __logf(l_run[r]);
            if (FUSED) __hip_atomic_store(lp, lv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through, like the rows
            else *lp = lv;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int chunk = tid + i * 256;
        const int hr = chunk >> 6, c8 = chunk & 63;
        if (h0 + hr < H) {
            const i32x4 v = *reinterpret_cast<const i32x4*>(o_lds + hr * kC + c8 * 8);
            bf16_t* dst = part_o + (((int64_t)b * H + h0 + hr) * num_splits + split) * kC + c8 * 8;
            // write-through (sc1): partials left DIRTY in the L2s would be flushed by the end-of-kernel release, in
            // front of the launch that reads them back; streamed out here they overlap the other workgroups
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
        }
    }
    CHITU_PROBE_MARK(13);
    if (FUSED) {
        // scratch: the second tile buffer's first bytes (the partial transpose above used the first buffer's)
        mla_fused_tail(fuse, uvw, part_o, part_lse, b, H, h0, split, num_splits, b * (int)gridDim.z + hb, kv_lds + kTileU);
    }
}

// Stage 2: out[b,h,:] = sum_s w_s * part_o[b,h,s,:] / sum_s w_s, w_s = exp(lse_s - max lse).  part_o bf16, sums fp32.
__global__ __launch_bounds__(128) void mla_merge_kernel(const bf16_t* __restrict__ part_o,
                                                        const float* __restrict__ part_lse,
                                                        bf16_t* __restrict__ out, int num_splits) {
    const int64_t bh = blockIdx.x;
    const float* lse = part_lse + bh * num_splits;
    float m = -INFINITY;
    for (int s = 0; s < num_splits; ++s) m = __builtin_fmaxf(m, lse[s]);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float wsum = 0.f;
    for (int s0 = 0; s0 < num_splits; s0 += 8) {  // 8 partial rows in flight
        i32x2 v[8];
        float w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int s = s0 + i;
            const float l = s < num_splits ? lse[s] : -INFINITY;
            w[i] = l == -INFINITY ? 0.f : __expf(l - m);
            v[i] = i32x2{0, 0};
            if (w[i] != 0.f) v[i] = *reinterpret_cast<const i32x2*>(part_o + (bh * num_splits + s) * kC + threadIdx.x * 4);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            wsum += w[i];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const uint32_t u = (uint32_t)v[i][k];
                acc[2 * k] = merge_term(acc[2 * k], w[i], __uint_as_float(u << 16));
                acc[2 * k + 1] = merge_term(acc[2 * k + 1], w[i], __uint_as_float(u & 0xffff0000u));
            }
        }
    }
    const float inv = wsum > 0.f ? 1.0f / wsum : 0.f;
    i32x2 o2;
    o2[0] = (int)((uint32_t)f32_to_bf16(acc[0] * inv) | ((uint32_t)f32_to_bf16(acc[1] * inv) << 16));
    o2[1] = (int)((uint32_t)f32_to_bf16(acc[2] * inv) | ((uint32_t)f32_to_bf16(acc[3] * inv) << 16));
    *reinterpret_cast<i32x2*>(out + bh * kC + threadIdx.x * 4) = o2;
}

}  // namespace chitu

// Dynamic LDS of a decode workgroup; the opt-in above 64 KB is set on every call (it is per device and cheap; a process-wide
// "done" flag would leave the second GPU of a multi-device process without it).
template <bool FUSED>
static size_t mla_decode_lds_bytes() {
    const size_t lds = 2 * chitu::kTileU + 16 * chitu::kPStride * 2 + 2 * 64 * sizeof(float) + chitu::kMaxTilesLds * sizeof(int);
    (void)hipFuncSetAttribute((const void*)chitu::mla_decode_kernel<FUSED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return lds;
}

extern "C" int chitu_hip_mla_decode_workspace_bytes(int32_t batch, int32_t heads, int32_t num_splits,
                                                    int64_t* bytes) {
    if (!bytes || batch < 0 || heads < 1 || num_splits < 1) return CHITU_ERR_BAD_ARG;
    *bytes = (int64_t)batch * heads * num_splits * (chitu::kC * 2 + 4);  // bf16 partial rows + fp32 LSE
    return CHITU_OK;
}

extern "C" int chitu_hip_mla_decode(const void* q_nope, int64_t qn_stride_b, int64_t qn_stride_h,
                                    const void* q_pe, int64_t qp_stride_b, int64_t qp_stride_h,
                                    const void* kv_cache, int64_t num_pages, int32_t page_size,
                                    const int32_t* block_table, int32_t table_stride,
                                    const int32_t* seqlens, float softmax_scale, void* out_bf16,
                                    int32_t batch, int32_t heads, int32_t kv_lora_rank,
                                    int32_t rope_dim, int32_t num_splits, void* workspace,
                                    int64_t workspace_bytes, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(q_nope && q_pe && kv_cache && block_table && seqlens);
    CHITU_REQUIRE(out_bf16 || num_splits > 1);  // no out: leave the split partials for a fused consumer
    CHITU_REQUIRE(batch >= 0 && heads >= 1 && num_pages >= 1 && table_stride >= 1);
    CHITU_REQUIRE(((uintptr_t)kv_cache & 15) == 0 && ((uintptr_t)q_nope & 15) == 0 && ((uintptr_t)q_pe & 15) == 0);  // 16-byte loads
    if (kv_lora_rank != kC || rope_dim != kR) return CHITU_ERR_UNSUPPORTED;
    if (page_size < kTile || page_size % kTile != 0) return CHITU_ERR_UNSUPPORTED;
    CHITU_REQUIRE(num_splits >= 1 && num_splits <= 256);
    // the kernel's split arithmetic is 32-bit: tiles the table can address x (splits + 1) must stay below 2^31
    CHITU_REQUIRE((int64_t)table_stride * (page_size / kTile) * (num_splits + 1) < (1ll << 31));
    if (batch == 0) return CHITU_OK;
    bf16_t* part_o = nullptr;
    float* part_lse = nullptr;
    if (num_splits > 1) {
        // workspace: bf16 partial rows [batch, heads, splits, 512] | fp32 LSE [batch, heads, splits]
        const int64_t need = (int64_t)batch * heads * num_splits * (kC * 2 + 4);
        CHITU_REQUIRE(workspace && workspace_bytes >= need);
        part_o = (bf16_t*)workspace;
        part_lse = (float*)(part_o + (int64_t)batch * heads * num_splits * kC);
    }
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)num_splits, (unsigned)batch, (unsigned)((heads + 15) / 16));
    hipLaunchKernelGGL(mla_decode_kernel<false>, grid, dim3(256), mla_decode_lds_bytes<false>(), st, (const bf16_t*)q_nope, qn_stride_b,
                       qn_stride_h, (const bf16_t*)q_pe, qp_stride_b, qp_stride_h, (const bf16_t*)kv_cache,
                       num_pages, (int)page_size, block_table, (int)table_stride, seqlens, softmax_scale,
                       part_o, part_lse, (bf16_t*)out_bf16, (int)heads, (int)num_splits, MlaFuse{});
    if (num_splits > 1 && out_bf16)
        hipLaunchKernelGGL(mla_merge_kernel, dim3((unsigned)(batch * heads)), dim3(128), 0, st, part_o,
                           part_lse, (bf16_t*)out_bf16, (int)num_splits);
    CHITU_RETURN_LAUNCH_STATUS();
}

extern "C" int chitu_hip_mla_decode_tickets_bytes(int64_t* bytes) {
    if (!bytes) return CHITU_ERR_BAD_ARG;
    *bytes = (int64_t)(2 * chitu::kFuseMaxGroups + 1) * 4;
    return CHITU_OK;
}

extern "C" int chitu_hip_mla_decode_merge_uv_quant_fp8(
    const void* q_nope, int64_t qn_stride_b, int64_t qn_stride_h, const void* q_pe, int64_t qp_stride_b, int64_t qp_stride_h,
    const void* kv_cache, int64_t num_pages, int32_t page_size, const int32_t* block_table, int32_t table_stride,
    const int32_t* seqlens, float softmax_scale, int32_t batch, int32_t heads, int32_t kv_lora_rank, int32_t rope_dim,
    int32_t num_splits, void* workspace, int64_t workspace_bytes, const void* w_fp8, int64_t w_stride_h, const float* scale,
    int64_t scale_offset, int64_t scale_stride_h, int64_t scale_stride_k, void* q_fp8, float* q_scales, int32_t tile_major,
    uint32_t* tickets, void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(q_nope && q_pe && kv_cache && block_table && seqlens && w_fp8 && scale && q_fp8 && q_scales && tickets);
    CHITU_REQUIRE(batch >= 0 && heads >= 1 && num_pages >= 1 && table_stride >= 1 && w_stride_h % 16 == 0);
    CHITU_REQUIRE(((uintptr_t)kv_cache & 15) == 0 && ((uintptr_t)q_nope & 15) == 0 && ((uintptr_t)q_pe & 15) == 0);  // 16-byte loads
    CHITU_REQUIRE(((uintptr_t)w_fp8 & 15) == 0 && ((uintptr_t)tickets & 3) == 0);
    if (kv_lora_rank != kC || rope_dim != kR) return CHITU_ERR_UNSUPPORTED;
    if (page_size < kTile || page_size % kTile != 0) return CHITU_ERR_UNSUPPORTED;
    CHITU_REQUIRE(num_splits >= 2 && num_splits <= 256);  // one split: chitu_hip_mla_decode + chitu_hip_absorb_uv_quant_fp8
    CHITU_REQUIRE((int64_t)table_stride * (page_size / kTile) * (num_splits + 1) < (1ll << 31));
    const int hb = (heads + 15) / 16;
    if ((int64_t)batch * hb > kFuseMaxGroups) return CHITU_ERR_UNSUPPORTED;
    if (batch == 0) return CHITU_OK;
    const int64_t need = (int64_t)batch * heads * num_splits * (kC * 2 + 4);
    CHITU_REQUIRE(workspace && workspace_bytes >= need);
    bf16_t* part_o = (bf16_t*)workspace;
    float* part_lse = (float*)(part_o + (int64_t)batch * heads * num_splits * kC);
    const MlaFuse fuse{(const fp8_t*)w_fp8, w_stride_h, scale, scale_offset, scale_stride_h, scale_stride_k, (fp8_t*)q_fp8, q_scales,
                       (int)tile_major, tickets};
    const dim3 grid((unsigned)num_splits, (unsigned)batch, (unsigned)hb);
    hipLaunchKernelGGL(mla_decode_kernel<true>, grid, dim3(256), mla_decode_lds_bytes<true>(), (hipStream_t)stream, (const bf16_t*)q_nope,
                       qn_stride_b, qn_stride_h, (const bf16_t*)q_pe, qp_stride_b, qp_stride_h, (const bf16_t*)kv_cache, num_pages,
                       (int)page_size, block_table, (int)table_stride, seqlens, softmax_scale, part_o, part_lse, (bf16_t*)nullptr,
                       (int)heads, (int)num_splits, fuse);
    CHITU_RETURN_LAUNCH_STATUS();
}

CHITU_PROBE_READER(mla_decode)
