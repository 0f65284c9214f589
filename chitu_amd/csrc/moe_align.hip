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
#include "moe_align_device.h"

namespace chitu {

constexpr int kAlignThreads = 1024;
constexpr int kAlignMaxExperts = 1024;  // one scan slot per thread

template <typename id_t>
__global__ __launch_bounds__(kAlignThreads) void moe_align_kernel(
    const id_t* __restrict__ ids, int64_t numel, int E, int block_size,
    int32_t* __restrict__ sorted_ids, int64_t sorted_cap, int32_t* __restrict__ expert_ids,
    int64_t expert_cap, int32_t* __restrict__ num_post_pad, int32_t* __restrict__ cumsum,
    int fill, const int32_t* __restrict__ expert_map) {
    extern __shared__ __attribute__((aligned(16))) int lds[];
    moe_align_workgroup<id_t>(ids, numel, E, block_size, sorted_ids, sorted_cap, expert_ids, expert_cap,
                              num_post_pad, cumsum, fill, expert_map, lds, kAlignThreads);
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
    const size_t lds = sizeof(int) * moe_align_lds_ints(num_experts, kAlignThreads);
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
