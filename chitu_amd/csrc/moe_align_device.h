// moe_align_block_size as a workgroup-level device function (shared by moe_align.hip and gate.hip).
// See moe_align.hip for the reference it replaces and the output contract.
#pragma once
#include "common.h"

namespace chitu {

// The whole sort, run by ONE workgroup of `nthreads` threads (a multiple of 64, >= E: thread t owns
// expert t in the scan) -- the body of moe_align_kernel, and the tail the last routing workgroup runs
// in gate.hip's fused route + align launch.  Every thread of the workgroup must call it.
// lds: moe_align_lds_ints(E, nthreads) ints = counts[E] | seg_cursor[E] | wave_tot[W] | wave_hist[W][E].
__host__ __device__ inline size_t moe_align_lds_ints(int E, int nthreads) {
    const int W = nthreads / 64;
    return (size_t)2 * E + W + (size_t)W * E;
}

template <typename id_t>
__device__ __forceinline__ void moe_align_workgroup(
    const id_t* __restrict__ ids, int64_t numel, int E, int block_size,
    int32_t* __restrict__ sorted_ids, int64_t sorted_cap, int32_t* __restrict__ expert_ids,
    int64_t expert_cap, int32_t* __restrict__ num_post_pad, int32_t* __restrict__ cumsum,
    int fill, const int32_t* __restrict__ expert_map, int* lds, const int nthreads) {
    const int nwaves = nthreads >> 6;
    int* counts = lds;
    int* cursor = lds + E;
    int* wave_tot = lds + 2 * E;
    int* wave_hist = lds + 2 * E + nwaves;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    for (int i = tid; i < nwaves * E; i += nthreads) wave_hist[i] = 0;
    if (fill) {
        // The reference allocator's sentinels (fused_moe.py:493-502), done here so a
        // graph-captured caller needs no extra fill launches.
        for (int64_t i = tid; i < sorted_cap; i += nthreads) sorted_ids[i] = (int32_t)numel;
        for (int64_t i = tid; i < expert_cap; i += nthreads) expert_ids[i] = 0;
    }
    __syncthreads();
    CHITU_PROBE_MARK(20);

    // Every wave owns one CONTIGUOUS run of the flat ids (a multiple of 64 long): the stable order is then "by wave, then by
    // position", each wave's share of an expert's segment is known after ONE cross-wave scan, and the scatter needs no
    // workgroup barrier at all -- three barriers for any numel.  (Rounds 1-4 walked the ids in 1024-token rounds of three
    // barriers each with per-round cross-wave offsets: 54 us for the 18432 slots of a 2048-token prompt, 5 % of its layer.)
    const int64_t per = (((numel + nwaves - 1) / nwaves) + 63) & ~(int64_t)63;
    const int64_t w0 = (int64_t)wave * per, w1 = w0 + per < numel ? w0 + per : numel;
    int* my_hist = wave_hist + wave * E;

    // Pass 1: this wave's per-expert counts (order-free, LDS atomics are exact for integers).  Four loads in flight per
    // lane: a step is one dependent global load otherwise, ~1.5 us each.
    for (int64_t i0 = w0 + lane; i0 < w1; i0 += 256) {
        int64_t e4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) e4[u] = i0 + 64 * u < w1 ? (int64_t)ids[i0 + 64 * u] : -1;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (e4[u] >= 0 && e4[u] < E) atomicAdd(&my_hist[(int)e4[u]], 1);
    }
    __syncthreads();
    CHITU_PROBE_MARK(21);

    // Pass 2: exclusive scan of padded counts -> segment starts; thread t owns expert t.
    int padded = 0;
    if (tid < E) {
        int total = 0;
        for (int w = 0; w < nwaves; ++w) total += wave_hist[w * E + tid];
        counts[tid] = total;
        padded = ((total + block_size - 1) / block_size) * block_size;
    }
    int incl = padded;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int wave_base = 0;
    for (int w = 0; w < wave; ++w) wave_base += wave_tot[w];
    incl += wave_base;
    const int start = incl - padded;
    if (tid < E) {
        const int32_t own_id = expert_map ? expert_map[tid] : (int32_t)tid;  // local id or -1 (expert parallelism)
        // wave w's slots of this expert start where the earlier waves' end: counts -> exclusive cursors, in place
        int run = start;
        for (int w = 0; w < nwaves; ++w) {
            const int c = wave_hist[w * E + tid];
            wave_hist[w * E + tid] = run;
            run += c;
        }
        cursor[tid] = run;  // (one past the expert's last real slot; kept for probes)
        cumsum[tid + 1] = incl;
        for (int i = start; i < incl; i += block_size) {
            const int b = i / block_size;
            if (b < expert_cap) expert_ids[b] = own_id;
        }
        if (tid == E - 1) *num_post_pad = incl;
    }
    if (tid == 0) cumsum[0] = 0;
    if (expert_map) {
        // expert_ids = expert_map[expert_ids] over the WHOLE array (fused_moe.py:516-517): the blocks
        // past num_tokens_post_pad hold the allocator's 0 and therefore map to expert_map[0]
        int total = 0;
        for (int w = 0; w < nwaves; ++w) total += wave_tot[w];
        const int32_t tail_id = expert_map[0];
        for (int64_t b = total / block_size + tid; b < expert_cap; b += nthreads) expert_ids[b] = tail_id;
    }
    __syncthreads();
    CHITU_PROBE_MARK(22);

    // Pass 3: stable scatter, each wave over its own run, 64 ids per step, no workgroup barrier: a step's lanes read the
    // wave's cursor of their expert, then the first lane of every group of equal ids advances it (one wave's LDS
    // operations are served in program order).
    int nbits = 0;
    while ((1 << nbits) < E) ++nbits;
    for (int64_t base4 = w0; base4 < w1; base4 += 256) {
        int64_t e4[4];  // four steps' ids requested together (see pass 1)
#pragma unroll
        for (int u = 0; u < 4; ++u) e4[u] = base4 + 64 * u + lane < w1 ? (int64_t)ids[base4 + 64 * u + lane] : -1;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t base = base4 + 64 * u;
            if (base >= w1) break;  // wave-uniform
            const int64_t i = base + lane;
            const int64_t e64 = e4[u];
            const bool valid = (e64 >= 0 && e64 < E);
            const int e = valid ? (int)e64 : 0;

            // wave64 match-any: lanes holding the same expert id as this lane.
            unsigned long long same = __ballot(valid);
            for (int b = 0; b < nbits; ++b) {
                const bool bit = (e >> b) & 1;
                const unsigned long long bal = __ballot(valid && bit);
                same &= bit ? bal : ~bal;
            }
            const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
            const int rank = __popcll(same & lt);
            int pos = 0;
            if (valid) pos = my_hist[e] + rank;
            __builtin_amdgcn_wave_barrier();  // (compiler: every lane's cursor read stays ahead of the updates below)
            if (valid) {
                if (rank == 0) my_hist[e] = pos + __popcll(same);
                if (pos < sorted_cap) sorted_ids[pos] = (int32_t)i;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---- the sort for ONE decode batch of <= 32 tokens whose ids are DISTINCT WITHIN A TOKEN (a router's top-k + always-on
// slots), split in two so that its order-free half rides inside the routing itself (gate.hip):
//   moe_align_small_mark:  tmask[e] |= 1 << token            (LDS atomic OR: exact and order-free)
//   moe_align_small_tail:  count[e] = popcount(tmask[e]); segment starts by one scan; the slot of (token t, expert e)
//                          is start[e] + popcount(tmask[e] & ((1 << t) - 1)) -- its rank among the tokens that chose e,
//                          i.e. exactly the stable order of moe_align_workgroup (flat index = t * stride + k grows with t).
// Bit-identical to moe_align_workgroup(fill = 1) on such ids, in 2 barriers instead of 7 and without the per-wave
// histograms.  lds: tmask[E] | start[E] | wave_tot[nthreads / 64]; tmask zeroed and the sentinels written by the caller
// (moe_align_small_init) before the marks.  `nthreads` threads (a multiple of 64, >= E and >= numel) call the tail.
__host__ __device__ inline size_t moe_align_small_lds_ints(int E, int nthreads) { return (size_t)2 * E + nthreads / 64; }

__device__ __forceinline__ void moe_align_small_init(int E, int64_t numel, int32_t* __restrict__ sorted_ids, int64_t sorted_cap,
                                                     int32_t* __restrict__ expert_ids, int64_t expert_cap, int* lds,
                                                     int tid, int nthreads) {
    for (int i = tid; i < E; i += nthreads) lds[i] = 0;
    for (int64_t i = tid; i < sorted_cap; i += nthreads) sorted_ids[i] = (int32_t)numel;  // the allocator's sentinels
    for (int64_t i = tid; i < expert_cap; i += nthreads) expert_ids[i] = 0;                // (fused_moe.py:493-502)
}

// (e is inside [0, E) by the launcher's choice of this sort -- gate.hip, `alw.small`; E is not known here)
__device__ __forceinline__ void moe_align_small_mark(int* lds, int e, int token) { atomicOr(&lds[e], 1 << token); }

__device__ __forceinline__ void moe_align_small_tail(
    const int64_t* __restrict__ ids_lds, int numel, int stride, int E, int block_size, int32_t* __restrict__ sorted_ids,
    int64_t sorted_cap, int32_t* __restrict__ expert_ids, int64_t expert_cap, int32_t* __restrict__ num_post_pad,
    int32_t* __restrict__ cumsum, const int32_t* __restrict__ expert_map, int* lds, const int nthreads) {
    const int nwaves = nthreads >> 6;
    int* tmask = lds;
    int* start_l = lds + E;
    int* wave_tot = lds + 2 * E;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int padded = 0;
    if (tid < E) padded = ((__popc((unsigned)tmask[tid]) + block_size - 1) / block_size) * block_size;
    int incl = padded;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int wave_base = 0, total = 0;
    for (int w = 0; w < nwaves; ++w) {
        const int t = wave_tot[w];
        if (w < wave) wave_base += t;
        total += t;
    }
    incl += wave_base;
    const int start = incl - padded;
    if (tid < E) {
        const int32_t own_id = expert_map ? expert_map[tid] : (int32_t)tid;
        start_l[tid] = start;
        cumsum[tid + 1] = incl;
        for (int i = start; i < incl; i += block_size) {
            const int b = i / block_size;
            if (b < expert_cap) expert_ids[b] = own_id;
        }
        if (tid == E - 1) *num_post_pad = incl;
    }
    if (tid == 0) cumsum[0] = 0;
    if (expert_map) {  // expert_ids = expert_map[expert_ids] over the WHOLE array: the tail holds expert_map[0]
        const int32_t tail_id = expert_map[0];
        for (int64_t b = total / block_size + tid; b < expert_cap; b += nthreads) expert_ids[b] = tail_id;
    }
    __syncthreads();
    if (tid < numel) {
        const int e = (int)ids_lds[tid];
        const int t = tid / stride;
        if (e >= 0 && e < E) {  // like the general sort: an id outside the table is dropped, never written through
            const int pos = start_l[e] + __popc((unsigned)tmask[e] & ((1u << t) - 1u));
            if (pos >= 0 && pos < sorted_cap) sorted_ids[pos] = (int32_t)tid;
        }
    }
}

}  // namespace chitu
