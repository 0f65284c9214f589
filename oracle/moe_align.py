"""Oracle for moe_align_block_size (numpy; integers, bit-exact).

Restates the reference's *stable* path:
  chitu/fused_moe.py:398-442  moe_align_block_size_triton (stages 1-4, :314-393)
  chitu/fused_moe.py:522-596  moe_align_block_size_native (allocation contract)
The CUDA kernel (csrc/moe_align_kernel.cu:27-96) produces the same segments but an
arbitrary order inside each expert's segment (atomicAdd ranks); the reference's own
test only checks set membership (test/pytest/test_moe_align.py:52-74).
"""

import numpy as np


def moe_align_block_size(topk_ids, block_size, num_experts, expert_map=None):
    """Returns (sorted_ids, expert_ids, num_tokens_post_pad, cumsum), all int32.

    sorted_ids: len numel + E*(block-1), padding slots hold `numel`   (fused_moe.py:574-578)
    expert_ids: len ceil(len(sorted_ids)/block), unused slots are 0     (fused_moe.py:579-584)
    """
    ids = np.asarray(topk_ids).reshape(-1).astype(np.int64)
    numel = ids.size
    E = int(num_experts)
    max_padded = numel + E * (block_size - 1)
    sorted_ids = np.full((max_padded,), numel, dtype=np.int32)
    max_blocks = (max_padded + block_size - 1) // block_size
    expert_ids = np.zeros((max_blocks,), dtype=np.int32)
    cumsum = np.zeros((E + 1,), dtype=np.int32)

    # stage 1-3: per-expert counts, padded prefix sum (fused_moe.py:314-364)
    counts = np.bincount(ids, minlength=E)[:E]
    padded = (counts + block_size - 1) // block_size * block_size
    cumsum[1:] = np.cumsum(padded)
    # stage 4: block -> expert, stable scatter (fused_moe.py:367-393)
    for e in range(E):
        for i in range(int(cumsum[e]), int(cumsum[e + 1]), block_size):
            expert_ids[i // block_size] = e
    order = np.argsort(ids, kind="stable")
    rank = np.zeros(numel, dtype=np.int64)
    start_of = np.cumsum(counts) - counts
    rank[order] = np.arange(numel) - start_of[ids[order]]
    sorted_ids[cumsum[ids] + rank] = np.arange(numel, dtype=np.int32)
    num_post_pad = np.array([cumsum[E]], dtype=np.int32)
    if expert_map is not None:
        expert_ids = np.asarray(expert_map)[expert_ids]
    return sorted_ids, expert_ids, num_post_pad, cumsum
