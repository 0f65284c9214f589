/* chitu_hip.h -- C ABI of libchitu_hip.so: the MI355X (gfx950) decode hot path of
 * thu-pacman/chitu, re-written as hand-authored HIP kernels.
 *
 * Conventions (the contract a binding must follow):
 *   - every pointer is a DEVICE pointer unless stated; the caller owns and allocates all
 *     inputs, outputs and scratch (reference ownership rule: csrc/common.h:46-55,
 *     fused_moe.py:489-506).  Kernels never allocate and never synchronise.
 *   - `stream` is a hipStream_t; launches are enqueued on it and are hipGraph-capturable.
 *   - return value: 0 = enqueued; <0 = argument error (CHITU_ERR_*); >0 = hipError_t.
 *     Nothing here calls exit() (the reference's ASSERTWITH does, csrc/common.h:20-30).
 *   - tensors are dense row-major ("contiguous" in the reference's asserts) unless a
 *     stride argument is given.  fp8 = OCP e4m3fn bytes (torch.float8_e4m3fn), bf16 =
 *     raw uint16 bits.  `act_dtype`: 0 = bf16, 1 = f16, 2 = f32.
 */
#ifndef CHITU_HIP_H
#define CHITU_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CHITU_HIP_ABI_VERSION 1

/* ---- fused MoE: token alignment -------------------------------------------------
 * Replaces chitu_backend.cuda_moe_align_block_size (reference csrc/binding.cpp:11,
 * csrc/moe_kernel.h:7-11, kernel csrc/moe_align_kernel.cu:27-120) and the Triton
 * 4-stage path chitu/fused_moe.py:314-442.  Stable (flat token order inside each expert
 * segment) => bit-identical to the Triton path.
 *   topk_ids            [numel] integers, ids_dtype: 0=u8 1=i8 2=i16 3=i32 4=i64
 *   sorted_token_ids    [sorted_cap]  (cap >= numel + E*(block-1)); padding = numel
 *   expert_ids          [expert_ids_cap] one expert per block of `block_size` slots
 *   num_tokens_post_pad [1]; cumsum [num_experts+1]
 *   fill_sentinels      0: caller pre-filled sorted=numel / expert_ids=0 as the reference
 *                          allocator does (fused_moe.py:493-502); 1: the kernel does it.
 * Limits: 1 <= num_experts <= 1024 (reference CUDA kernel: <= 256). */
int chitu_hip_moe_align_block_size(const void* topk_ids, int ids_dtype, int64_t numel,
                                   int32_t num_experts, int32_t block_size,
                                   int32_t* sorted_token_ids, int64_t sorted_cap,
                                   int32_t* expert_ids, int64_t expert_ids_cap,
                                   int32_t* num_tokens_post_pad, int32_t* cumsum,
                                   int32_t fill_sentinels, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CHITU_HIP_H */
