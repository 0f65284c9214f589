// moe_align_block_size for gfx950: stable counting sort of flat top-k expert ids,
// each expert segment padded to `block_size`.
//
// Replaces (reference, read-only):
//   csrc/moe_align_kernel.cu:27-120  moe_align_block_size_kernel + launcher
//   chitu/fused_moe.py:314-442       moe_align_block_size_stage1..4 (Triton)
// Output contract = the Triton path's: token order inside an expert segment is
// the original flat order (stable), so results are bit-exact and deterministic.
// The CUDA kernel ranks with shared-memory atomicAdd (order inside a segment is
// arbitrary) and hard-codes 32-wide warps; neither is carried over.  Here a
// wave64 match-any built from ballots gives every token its rank among equal
// ids in its wave, a per-wave LDS histogram gives the cross-wave offset, and a
// running per-expert cursor carries the order across 1024-token chunks.
#include "common.h"

namespace chitu {

constexpr int kAlignThreads = 1024;
constexpr int kAlignWaves = kAlignThreads / kWave;
constexpr int kAlignMaxExperts = 1024;  // one scan slot per thread

// LDS: counts[E] | seg_cursor[E] | wave_tot[16] | wave_hist[16][E]
template <typename id_t>
__global__ __launch_bounds__(kAlignThreads) void moe_align_kernel(
    const id_t* __restrict__ ids, int64_t numel, int E, int block_size,
    int32_t* __restrict__ sorted_ids, int64_t sorted_cap, int32_t* __restrict__ expert_ids,
    int64_t expert_cap, int32_t* __restrict__ num_post_pad, int32_t* __restrict__ cumsum,
    int fill, const int32_t* __restrict__ expert_map) {
    extern __shared__ __attribute__((aligned(16))) int lds[];
    int* counts = lds;
    int* cursor = lds + E;
    int* wave_tot = lds + 2 * E;
    int* wave_hist = lds + 2 * E + kAlignWaves;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    for (int i = tid; i < E; i += kAlignThreads) counts[i] = 0;
    for (int i = tid; i < kAlignWaves * E; i += kAlignThreads) wave_hist[i] = 0;
    if (fill) {
        // The reference allocator's sentinels (fused_moe.py:493-502), done here so a
        // graph-captured caller needs no extra fill launches.
        for (int64_t i = tid; i < sorted_cap; i += kAlignThreads) sorted_ids[i] = (int32_t)numel;
        for (int64_t i = tid; i < expert_cap; i += kAlignThreads) expert_ids[i] = 0;
    }
    __syncthreads();

    // Pass 1: per-expert totals (order-free, LDS atomics are exact for integers).
    for (int64_t i = tid; i < numel; i += kAlignThreads) {
        const int64_t e = (int64_t)ids[i];
        if (e >= 0 && e < E) atomicAdd(&counts[(int)e], 1);
    }
    __syncthreads();

    // Pass 2: exclusive scan of padded counts -> segment starts; thread t owns expert t.
    int padded = 0;
    if (tid < E) padded = ((counts[tid] + block_size - 1) / block_size) * block_size;
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
        cursor[tid] = start;
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
        for (int w = 0; w < kAlignWaves; ++w) total += wave_tot[w];
        const int32_t tail_id = expert_map[0];
        for (int64_t b = total / block_size + tid; b < expert_cap; b += kAlignThreads) expert_ids[b] = tail_id;
    }
    __syncthreads();

    // Pass 3: stable scatter, 1024 tokens per round.
    int nbits = 0;
    while ((1 << nbits) < E) ++nbits;
    for (int64_t base = 0; base < numel; base += kAlignThreads) {
        const int64_t i = base + tid;
        int64_t e64 = -1;
        if (i < numel) e64 = (int64_t)ids[i];
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
        if (valid && rank == 0) wave_hist[wave * E + e] = __popcll(same);
        __syncthreads();

        if (valid) {
            int pos = cursor[e] + rank;
            for (int w = 0; w < wave; ++w) pos += wave_hist[w * E + e];
            if (pos < sorted_cap) sorted_ids[pos] = (int32_t)i;
        }
        __syncthreads();

        if (tid < E) {
            int add = 0;
#pragma unroll
            for (int w = 0; w < kAlignWaves; ++w) {
                add += wave_hist[w * E + tid];
                wave_hist[w * E + tid] = 0;
            }
            cursor[tid] += add;
        }
        __syncthreads();
    }
}

}  // namespace chitu

// ids_dtype follows torch's integral ScalarType numbering, the set the reference
// dispatches over (moe_align_kernel.cu:17-25): 0=u8 1=i8 2=i16 3=i32 4=i64.
extern "C" int chitu_hip_moe_align_block_size_mapped(const void* topk_ids, int ids_dtype, int64_t numel,
                                                     int32_t num_experts, int32_t block_size,
                                                     int32_t* sorted_token_ids, int64_t sorted_cap,
                                                     int32_t* expert_ids, int64_t expert_ids_cap,
                                                     int32_t* num_tokens_post_pad, int32_t* cumsum,
                                                     int32_t fill_sentinels, const int32_t* expert_map,
                                                     void* stream) {
    using namespace chitu;
    CHITU_REQUIRE(sorted_token_ids && expert_ids && num_tokens_post_pad && cumsum);
    CHITU_REQUIRE(numel >= 0 && (numel == 0 || topk_ids));
    CHITU_REQUIRE(numel < (1ll << 31));
    CHITU_REQUIRE(num_experts >= 1 && num_experts <= kAlignMaxExperts);
    CHITU_REQUIRE(block_size >= 1);
    CHITU_REQUIRE(sorted_cap >= 0 && expert_ids_cap >= 0);
    const size_t lds = sizeof(int) * (size_t)(2 * num_experts + kAlignWaves + kAlignWaves * num_experts);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(T)                                                                              \
    if (lds > 48 * 1024)                                                                       \
        (void)hipFuncSetAttribute((const void*)moe_align_kernel<T>,                            \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
    hipLaunchKernelGGL(moe_align_kernel<T>, dim3(1), dim3(kAlignThreads), lds, s,              \
                       (const T*)topk_ids, numel, (int)num_experts, (int)block_size,           \
                       sorted_token_ids, sorted_cap, expert_ids, expert_ids_cap,               \
                       num_tokens_post_pad, cumsum, (int)fill_sentinels, expert_map)
    switch (ids_dtype) {
        case 0: LAUNCH(uint8_t); break;
        case 1: LAUNCH(int8_t); break;
        case 2: LAUNCH(int16_t); break;
        case 3: LAUNCH(int32_t); break;
        case 4: LAUNCH(int64_t); break;
        default: return CHITU_ERR_UNSUPPORTED;
    }
#undef LAUNCH
    CHITU_RETURN_LAUNCH_STATUS();
}

extern "C" int chitu_hip_moe_align_block_size(const void* topk_ids, int ids_dtype, int64_t numel,
                                              int32_t num_experts, int32_t block_size,
                                              int32_t* sorted_token_ids, int64_t sorted_cap,
                                              int32_t* expert_ids, int64_t expert_ids_cap,
                                              int32_t* num_tokens_post_pad, int32_t* cumsum,
                                              int32_t fill_sentinels, void* stream) {
    return chitu_hip_moe_align_block_size_mapped(topk_ids, ids_dtype, numel, num_experts, block_size,
                                                 sorted_token_ids, sorted_cap, expert_ids, expert_ids_cap,
                                                 num_tokens_post_pad, cumsum, fill_sentinels, nullptr, stream);
}
