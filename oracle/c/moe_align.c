/* Plain-C restatement of the reference's stable moe_align (test infrastructure only).
 * Follows chitu/fused_moe.py:314-393 (stage1..4): counts, padded prefix sum, block->expert,
 * then a scatter that walks tokens in flat order so ranks inside a segment are stable.
 * Buffers follow the allocator contract of fused_moe.py:570-588 (caller pre-fills). */
#include <stdint.h>
#include <stdlib.h>

int oracle_moe_align_block_size(const int64_t* ids, int64_t numel, int32_t num_experts,
                                int32_t block_size, int32_t* sorted_ids, int32_t* expert_ids,
                                int32_t* num_post_pad, int32_t* cumsum) {
    int32_t* counts = (int32_t*)calloc((size_t)num_experts, sizeof(int32_t));
    int32_t* cursor = (int32_t*)calloc((size_t)num_experts, sizeof(int32_t));
    if (!counts || !cursor) return -1;
    for (int64_t i = 0; i < numel; ++i) counts[ids[i]]++;
    cumsum[0] = 0;
    for (int32_t e = 0; e < num_experts; ++e) {
        int32_t padded = (counts[e] + block_size - 1) / block_size * block_size;
        cumsum[e + 1] = cumsum[e] + padded;
        cursor[e] = cumsum[e];
        for (int32_t i = cumsum[e]; i < cumsum[e + 1]; i += block_size) expert_ids[i / block_size] = e;
    }
    *num_post_pad = cumsum[num_experts];
    for (int64_t i = 0; i < numel; ++i) sorted_ids[cursor[ids[i]]++] = (int32_t)i;
    free(counts);
    free(cursor);
    return 0;
}
