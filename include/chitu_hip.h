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

#define CHITU_HIP_ABI_VERSION 3  /* 3 (round 6): + chitu_hip_mla_decode_merge_uv_quant_fp8 / _tickets_bytes; INTEGRATION.md lists what 2 -> 3 removed or tightened */

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
/* Same launch with the expert-parallel remap folded in: expert_map [num_experts] i32 on the device
 * (local expert id, or -1 for an expert held by another rank) is applied to EVERY entry of expert_ids,
 * exactly as `expert_ids = expert_map[expert_ids]` does after the sort (chitu/fused_moe.py:516-517):
 * blocks of expert e carry expert_map[e], the unused tail carries expert_map[0] (the allocator's zeros
 * mapped).  expert_map == NULL: identical to chitu_hip_moe_align_block_size. */
int chitu_hip_moe_align_block_size_mapped(const void* topk_ids, int ids_dtype, int64_t numel,
                                          int32_t num_experts, int32_t block_size,
                                          int32_t* sorted_token_ids, int64_t sorted_cap,
                                          int32_t* expert_ids, int64_t expert_ids_cap,
                                          int32_t* num_tokens_post_pad, int32_t* cumsum,
                                          int32_t fill_sentinels, const int32_t* expert_map,
                                          void* stream);

/* ---- FP8 activation quantisation ---------------------------------------------------
 * mode 0 replaces act_quant_deepseek_v3 (chitu/ops.py:330-353, kernel
 *        chitu/triton_kernels.py:193-214): s = max|x|/448, y = x/s, no eps, no clamp.
 * mode 1 replaces per_token_group_quant_fp8 (chitu/fused_moe.py:713-793, kernel :670-710):
 *        s = max(max|x|, eps)/448, y = clamp(x/s, +-448).
 *   x [rows, cols] act_dtype; y_fp8 [rows, cols] e4m3fn; scales [rows, cols/group] f32.
 * group_size must be 128 and divide cols (the reference asserts this, ops.py:345-348). */
int chitu_hip_act_quant_fp8(const void* x, int act_dtype, int64_t rows, int64_t cols,
                            int32_t group_size, int32_t mode, float eps, void* y_fp8,
                            float* scales, void* stream);

/* ---- FP8 weight de-quantisation -------------------------------------------------------
 * Replaces weight_dequant_deepseek_v3 (chitu/ops.py:357-392, kernel triton_kernels.py:217-247)
 * when soft = 0, and weight_dequant_soft_fp8_deepseek_v3 (chitu/ops.py:396-449, kernels
 * triton_kernels.py:250-287: bit placement * (s * 2^120)) when soft = 1.
 *   w_fp8 [batch, rows, cols]; scales [batch, ceil(rows/128), ceil(cols/128)] f32;
 *   y [batch, rows, cols] of out_dtype (0 bf16, 1 f16, 2 f32). */
int chitu_hip_weight_dequant_fp8(const void* w_fp8, const float* scales, int64_t batch,
                                 int64_t rows, int64_t cols, int32_t block_size, int32_t soft,
                                 int out_dtype, void* y, void* stream);

/* ---- FP8 x FP8 block-scaled GEMM (W8A8) --------------------------------------------------
 * Replaces fp8_gemm_deepseek_v3 (chitu/ops.py:453-483, kernel triton_kernels.py:302-365):
 *   out[m][n] = sum_kb dot(a[m, kb], b[n, kb]) * a_scale[m, kb] * b_scale[n/128, kb]
 *   a_fp8 [M, K]; a_scale [M, K/128] f32; b_fp8 [N, K]; b_scale [ceil(N/128), K/128] f32;
 *   out [M, N] of out_dtype.  K % 128 == 0.  Built for decode (M <= 64 per weight pass).
 * workspace: optional device scratch for cross-workgroup split-K partials
 * (needs 8 * M * N * 4 bytes at most); NULL/too small => no cross-workgroup split. */
int chitu_hip_fp8_gemm_blockscale(const void* a_fp8, const float* a_scale, const void* b_fp8,
                                  const float* b_scale, void* out, int out_dtype, int64_t M,
                                  int64_t N, int64_t K, void* workspace,
                                  int64_t workspace_bytes, void* stream);
/* The same GEMM on TILE-MAJOR activations, the layout the fused decode step keeps its fp8 activations in between two of
 * its own launches (never across the drop-in op surface, which stays row-major like the reference's):
 *   a_fp8   [ceil(M/16)][K/16][16 rows][16 B]   -- the 16 tokens of a tile side by side for every 16-byte k chunk
 *   a_scale [ceil(M/16)][K/128][16 rows] f32
 * so that a 16-lane group of the MFMA B-operand load (one k chunk, 16 tokens) reads 256 contiguous bytes instead of 16 B
 * from each of 16 rows, and a block's 16 scales one 64-B segment (the activation loads were 2.9 of the 9.9 us of the wqkv_a
 * launch at bs 16).  Written by chitu_hip_rmsnorm / chitu_hip_comm_allreduce_rmsnorm with quant_mode + 4 and by
 * chitu_hip_mla_merge_absorb_uv_quant_fp8_tm.  M < 128; same arithmetic, bit-identical output. */
int chitu_hip_fp8_gemm_blockscale_tm(const void* a_fp8, const float* a_scale, const void* b_fp8,
                                     const float* b_scale, void* out, int out_dtype, int64_t M,
                                     int64_t N, int64_t K, void* workspace,
                                     int64_t workspace_bytes, void* stream);
/* The same contraction with the K range cut over num_splits (2..16) workgroups per output tile and the fp32 partial
 * planes as the OUTPUT: partials [num_splits, M, N], plane s = the contribution of its K blocks, no reduce launch.
 * For the dense GEMMs whose N cannot fill 256 CUs on its own (wqkv_a: 2112 rows = 132 tiles); the consumer sums
 * the planes in plane order (chitu_hip_mla_qkv_post with num_partials). */
int chitu_hip_fp8_gemm_blockscale_partials(const void* a_fp8, const float* a_scale, const void* b_fp8,
                                           const float* b_scale, float* partials, int64_t M, int64_t N,
                                           int64_t K, int32_t num_splits, void* stream);

/* ---- FP8-weight x bf16-activation GEMM ("soft fp8") ------------------------------------
 * Replaces soft_fp8_gemm_deepseek_v3 (chitu/ops.py:487-511, kernel triton_kernels.py:388-508):
 * weights decoded to bf16 as bits((b&0x80)<<24 | (b&0x7f)<<20) * (b_scale * 2^120), bf16 dot,
 * fp32 accumulate.  a_bf16 [M, K]; b_fp8 [N, K]; b_scale [ceil(N/128), K/128]; out [M, N]. */
int chitu_hip_soft_fp8_gemm(const void* a_bf16, const void* b_fp8, const float* b_scale,
                            void* out, int out_dtype, int64_t M, int64_t N, int64_t K,
                            void* workspace, int64_t workspace_bytes, void* stream);

/* ---- paged KV append ----------------------------------------------------------------------
 * Replaces append_to_paged_kv_cache (chitu/ops.py:51-91, kernel triton_kernels.py:18-48):
 *   cache[page_table[b][L_b / page_size]][L_b % page_size] = this_kv[b],  L_b = old_seq_lens[b]
 *   kv_cache [num_pages, page_size, row_bytes] bytes; page_table [batch, pages_per_seq] i32;
 *   this_kv [batch, row_bytes]; old_seq_lens [batch] i32.  Out-of-table positions are dropped. */
int chitu_hip_append_paged_kv(void* kv_cache, int64_t num_pages, int32_t page_size,
                              int64_t row_bytes, const int32_t* page_table,
                              int32_t pages_per_seq, const void* this_kv,
                              const int32_t* old_seq_lens, int32_t batch, void* stream);

/* ---- rotary embedding ---------------------------------------------------------------------
 * Replaces apply_rotary_pos_emb (chitu/ops.py:311-326; kernels triton_kernels.py:51-190; torch
 * form ops.py:243-272).  layout 0 = "llama" interleaved (re, im) pairs, 1 = "hf-llama" halves.
 *   q [batch, q_heads, head_dim], k [batch, k_heads, head_dim] (element strides given),
 *   cos/sin [batch, head_dim/2] f32; fp32 math, one rounding to act_dtype. */
int chitu_hip_rope(const void* q, const void* k, void* out_q, void* out_k, const float* cos,
                   const float* sin, int act_dtype, int32_t batch, int32_t q_heads,
                   int32_t k_heads, int32_t head_dim, int64_t q_sb, int64_t q_sh, int64_t k_sb,
                   int64_t k_sh, int64_t oq_sb, int64_t oq_sh, int64_t ok_sb, int64_t ok_sh,
                   int32_t layout, void* stream);

/* ---- fused MoE, fp8 W8A8 with 128x128 block scales (decode) -------------------------------
 * Together these replace fused_experts_impl (chitu/fused_moe.py:1130-1307): the two
 * invoke_fused_moe_kernel calls (:796-891, Triton kernel :62-307), SiluAndMul (:24-39), the
 * activation re-quantisation (:829 -> :713-793) and moe_sum (:1299-1305).  All take the outputs
 * of chitu_hip_moe_align_block_size run with block_size = 16 (one MFMA tile of sorted slots).
 *   sorted_token_ids / expert_ids / num_tokens_post_pad: from moe_align (block 16); expert -1
 *   (expert_map, not on this rank) writes zeros like write_zeros_to_output (:40-59).
 *   max_mblocks: grid bound, min(len(expert_ids), numel) is always enough.
 *
 * gemm1:  c1[slot, :] = bf16( sum_kb dot(a[slot/topk, kb], w1[e, :, kb]) * a_s * w1_s )
 *   a_fp8 [tokens, K], a_scale [tokens, K/128]; w1 [E, N, K] fp8, w1_scale [E, ceil(N/128), K/128];
 *   out_bf16 [numel, N] with numel = tokens*topk (row = flat slot id t*topk+k).
 * silu_mul_quant:  h = bf16(bf16(silu(c1[:, :I])) * c1[:, I:]); per-128 group
 *   s = max(max|h|, eps)/448, q = clamp(h/s) -> e4m3 (quant_mode 1, the MoE rule) or s = max|h|/448,
 *   q = h/s (quant_mode 0, act_quant_deepseek_v3: dense / shared-expert MLP, model_deepseek_v3.py:936-949).
 *   c1 [rows, 2I]; q [rows, I]; scales [rows, I/128].
 * gemm2:  c3[slot, :] = bf16( (sum_kb dot(h[slot, kb], w2[e, :, kb]) * h_s * w2_s) * topk_w[slot] )
 *   h_fp8 [numel, I]; w2 [E, N, I]; out [numel, N]; topk_weights [numel] of weights_dtype.
 * moe_sum: out[t, :] = bf16( sum_k float(c3[t, k, :]) ).
 */
int chitu_hip_moe_gemm1_fp8(const void* a_fp8, const float* a_scale, const void* w1_fp8,
                            const float* w1_scale, const int32_t* sorted_token_ids,
                            const int32_t* expert_ids, const int32_t* num_tokens_post_pad,
                            void* out_bf16, int64_t numel, int32_t topk, int64_t N, int64_t K,
                            int64_t max_mblocks, void* stream);
int chitu_hip_moe_silu_mul_quant_fp8(const void* c1_bf16, int64_t rows, int64_t inter_size,
                                     int32_t quant_mode, float eps, void* q_fp8, float* scales,
                                     void* stream);
int chitu_hip_moe_gemm2_fp8(const void* h_fp8, const float* h_scale, const void* w2_fp8,
                            const float* w2_scale, const int32_t* sorted_token_ids,
                            const int32_t* expert_ids, const int32_t* num_tokens_post_pad,
                            const void* topk_weights, int weights_dtype,
                            int32_t mul_routed_weight, void* out_bf16, int64_t numel, int64_t N,
                            int64_t inter_size, int64_t max_mblocks, void* stream);
/* Two-launch form of gemm1 -> silu_mul_quant -> gemm2 (same arithmetic and rounding points):
 * gemm1_silu: a wave owns the gate tile and the up tile of the same 16 columns of W1 [E, 2I, K] and
 *   writes h[slot, :] = bf16(bf16(silu(bf16(gate))) * bf16(up)) as bf16 [numel, I]; inter_size % 16 == 0.
 * gemm2_quant: per_token_group_quant_fp8 (eps rule, 128-wide groups) of h in the prologue, then gemm2.
 *   inter_size % 128 == 0 and <= 512 (CHITU_ERR_UNSUPPORTED otherwise: wider experts take the three-launch form). */
int chitu_hip_moe_gemm1_silu_fp8(const void* a_fp8, const float* a_scale, const void* w1_fp8,
                                 const float* w1_scale, const int32_t* sorted_token_ids,
                                 const int32_t* expert_ids, const int32_t* num_tokens_post_pad,
                                 void* h_bf16, int64_t numel, int32_t topk, int64_t inter_size, int64_t K,
                                 int64_t max_mblocks, void* stream);

/* The same two grouped GEMMs tiled for PREFILL-sized batches (block_m = 64 or 128 sorted slots x 128 weight rows per workgroup
 * through LDS; fused_moe.py:62-307 runs BLOCK_SIZE_M = 64): sorted_token_ids / expert_ids / num_tokens_post_pad must come from
 * chitu_hip_moe_align_block_size with block_size = block_m (128: one block, i.e. one pass over its weights, for an expert with
 * up to 128 slots; all-padding 16-slot sub-tiles are skipped); max_mblocks <= 65535.  Same outputs for either block_m.
 *   gemm1_silu_tiled: h[slot, :] = bf16(bf16(silu(bf16(gate))) * bf16(up)) as bf16 [numel, I]; I % 128 == 0, K % 128 == 0.
 *   gemm2_tiled: out[slot, :] = bf16((h_fp8[slot] . W2[e]^T, block-scaled) * routed_weight[slot]); h quantised by the
 *   caller (chitu_hip_act_quant_fp8 mode 1); I % 128 == 0, N % 8 == 0.
 * Same rounding points as the decode kernels; the order of the sum inside a 128-block differs (<= 1e-5 of the result). */
int chitu_hip_moe_gemm1_silu_fp8_tiled(const void* a_fp8, const float* a_scale, const void* w1_fp8,
                                       const float* w1_scale, const int32_t* sorted_token_ids,
                                       const int32_t* expert_ids, const int32_t* num_tokens_post_pad,
                                       void* h_bf16, int64_t numel, int32_t topk, int64_t inter_size, int64_t K,
                                       int64_t max_mblocks, int32_t block_m, void* stream);
int chitu_hip_moe_gemm2_fp8_tiled(const void* h_fp8, const float* h_scale, const void* w2_fp8,
                                  const float* w2_scale, const int32_t* sorted_token_ids,
                                  const int32_t* expert_ids, const int32_t* num_tokens_post_pad,
                                  const void* topk_weights, int weights_dtype, int32_t mul_routed_weight,
                                  void* out_bf16, int64_t numel, int64_t N, int64_t inter_size,
                                  int64_t max_mblocks, int32_t block_m, void* stream);
int chitu_hip_moe_gemm2_quant_fp8(const void* h_bf16, const void* w2_fp8, const float* w2_scale,
                                  const int32_t* sorted_token_ids, const int32_t* expert_ids,
                                  const int32_t* num_tokens_post_pad, const void* topk_weights,
                                  int weights_dtype, int32_t mul_routed_weight, void* out_bf16,
                                  int64_t numel, int64_t N, int64_t inter_size, int64_t max_mblocks,
                                  float eps, void* stream);
int chitu_hip_moe_sum(const void* c3_bf16, void* out_bf16, int64_t tokens, int32_t topk, int64_t N,
                      void* stream);

/* ---- fused MoE with bf16 ACTIVATIONS: bf16 experts and soft-fp8 experts (csrc/moe_bf16.hip) --------
 * The two modes of fused_moe_kernel (chitu/fused_moe.py:62-307) that do not quantise the activations:
 *   weight_kind 0 -- bf16 weights, `accumulator += tl.dot(a, b)` (:298): fused_experts(use_fp8_w8a8=False), what the
 *     reference's MoEDeepSeekV3.forward runs for scale-less checkpoints (model_deepseek_v3.py:950-956) and, on every
 *     non-NVIDIA device, for soft-fp8 ones after weight_dequant_soft_fp8 (:975-993);
 *   weight_kind 1 -- fp8 e4m3fn weights + [128,128] block scales decoded to bf16 in registers,
 *     b = bf16(bits(((w & 0x80) << 24) | ((w & 0x7f) << 20)) * (scale * 2^120)) (:232-276): fused_experts(
 *     use_fp8_w8a8=True, soft_fp8=True), the NVIDIA branch (:968-974) -- same values as kind 0 on the dequantised weights.
 * One grouped GEMM over moe_align (block 16) output, fp32 accumulation over the whole K, one rounding to bf16:
 *   out[slot, :] = bf16( (a[slot / a_div, :] . w[e, :, :]^T) * (mul_routed_weight ? topk_weights[slot] : 1) )
 *   a bf16 [rows, K] (a_div = topk for the first GEMM, 1 for the second); w [E, Nw, K]; w_scale [E, ceil(Nw/128), K/128]
 *   (kind 1 only); out bf16 [numel, n_out]; expert -1 (expert_map: another rank's) writes zeros.
 *   silu = 1: Nw = 2*n_out (gate rows | up rows) and out = bf16(bf16(silu(bf16(g))) * bf16(u)) -- GEMM1 followed by
 *   SiluAndMul on bf16 tensors (fused_moe.py:24-39, :1262-1265) in one launch; silu = 0: Nw = n_out.
 *   K % 128 == 0; kind 1 with silu: n_out % 128 == 0.  Sum over the top-k with chitu_hip_moe_sum. */
int chitu_hip_moe_gemm_bf16(const void* a_bf16, int32_t a_div, const void* w, const float* w_scale,
                            int32_t weight_kind, const int32_t* sorted_token_ids, const int32_t* expert_ids,
                            const int32_t* num_tokens_post_pad, const void* topk_weights, int32_t weights_dtype,
                            int32_t mul_routed_weight, int32_t silu, void* out_bf16, int64_t numel, int64_t n_out,
                            int64_t K, int64_t max_mblocks, void* stream);

/* ---- GQA / MHA decode: RoPE(q in place, k) + paged append of k and v in one launch ----------------
 * Replaces apply_rotary_pos_emb (ops.py:311-326) + the two in-place appends of attn_with_kvcache
 * (attn_backend.py:108-115) in Attention.decode_forward_paged (models/model.py:167-198), same arithmetic.
 *   qkv [batch, q_heads + 2*kv_heads, head_dim] bf16 rows (row stride in elements): q heads rotated IN
 *   PLACE; k heads rotated into k_cache, v heads copied into v_cache [num_pages, page_size, kv_heads, head_dim]
 *   at row (page_table[b][L/page], L%page), L = old_seq_lens[b].  layout 0 = interleaved pairs, 1 = half-split. */
int chitu_hip_gqa_qkv_post(void* qkv_bf16, int64_t row_stride, int32_t q_heads, int32_t kv_heads,
                           int32_t head_dim, const float* cos, const float* sin, int32_t layout,
                           void* k_cache, void* v_cache, int64_t num_pages, int32_t page_size,
                           const int32_t* page_table, int32_t pages_per_seq, const int32_t* old_seq_lens,
                           int32_t batch, void* stream);

/* Gate/up projection of an unquantised SwiGLU MLP with SiluAndMul in the epilogue (FeedForward,
 * models/model.py:212-214): out[m, n] = bf16( bf16(silu(bf16(x[m] . w13[n]))) * bf16(x[m] . w13[inter + n]) ),
 * x [M, K] bf16, w13 [2*inter, K] bf16 (w1 rows then w3 rows), out [M, inter] bf16; K % 64 == 0. */
int chitu_hip_bf16_gemm_silu(const void* x_bf16, const void* w13_bf16, void* out_bf16, int64_t M, int64_t inter,
                             int64_t K, void* stream);

/* ---- [top-k sum +] residual add + RMSNorm + act_quant as the prologue of the W8A8 GEMM that consumes it (batch 1-2) ----
 * TransformerBlockDeepSeekV3.forward (model_deepseek_v3.py:1107-1113): x = x + ffn(...); attn(attn_norm(x)) with the first
 * linear of the attention (wqkv_a; linear_deepseek_v3 :98-100 = act_quant_deepseek_v3 + fp8_gemm_deepseek_v3) in ONE launch:
 *   a[m]       = add_terms == 1 ? add[m] : bf16(sum_k float(add[m, k, :]))   (the experts' top-k sum, fused_moe.py:1299-1305)
 *   sum_out[m] = bf16(x[m] + a[m])                                          (the new residual stream)
 *   y[m]       = bf16((sum_out[m] * rsqrt(mean(sum_out[m]^2) + eps)) * norm_weight);  (q, s) = act_quant(y, 128)
 *   out        = fp8_gemm(q, s, w_fp8, w_scale)                              [M, N], out_dtype as chitu_hip_fp8_gemm_blockscale
 * Bit-identical to chitu_hip_rmsnorm(add, add_terms, quant_mode = 1) followed by chitu_hip_fp8_gemm_blockscale (same
 * summation orders, same K split).  M == 1 (any add_terms <= 16) or M == 2 with add_terms == 1; K % 128 == 0, K <= 8192; a
 * shape whose GEMM runs as one workgroup per 16 output rows with 4 or 8 waves of K split (wqkv_a of R1 / V2-Lite, w1w3 of a
 * dense layer); anything else: CHITU_ERR_UNSUPPORTED (use the two launches).  add row / term strides in elements. */
int chitu_hip_fp8_gemm_add_norm(const void* x_bf16, int64_t x_row_stride, const void* add_bf16, int64_t add_row_stride,
                                int32_t add_terms, int64_t add_term_stride, void* sum_out_bf16, int64_t sum_row_stride,
                                const void* norm_weight_bf16, float eps, const void* w_fp8, const float* w_scale, void* out,
                                int32_t out_dtype, int64_t M, int64_t N, int64_t K, void* stream);

/* ---- residual add + RMSNorm as the prologue of the bf16 GEMM that consumes it (decode batches of 1-4 rows) -------
 * TransformerBlock (models/model.py:246-251): h = x + attention(attention_norm(x)); out = h + ffn(ffn_norm(h)) --
 * the add and the norm (RMSNorm.forward, models/model.py:29-78) in front of a layer's qkv projection and of its
 * gate/up projection, folded into those launches:
 *   sum_out[m] = bf16(x[m] + add[m])                       (the new residual stream; may alias x or add)
 *   y[m]       = bf16((sum_out[m] * rsqrt(mean(sum_out[m]^2) + eps)) * norm_weight)
 *   out        = y . w^T                                   (chitu_hip_bf16_gemm_add_norm)
 *   out        = silu(y . w1^T) * (y . w3^T)               (chitu_hip_bf16_gemm_silu_add_norm, w13 = [w1; w3])
 * Bit-identical to chitu_hip_rmsnorm(add = ...) followed by chitu_hip_bf16_gemm / chitu_hip_bf16_gemm_silu (same
 * summation orders).  M <= 4, K % 64 == 0, 512 <= K <= 8192, M * K <= 24576 and at least 4 waves of K split for the
 * shape; anything else: CHITU_ERR_UNSUPPORTED (use the two launches).  Row strides in elements, multiples of 8. */
/* (round 6) add_terms 1 | 2, add_term_stride (elements): with two terms the residual is bf16(float(add[r]) + float(add[r] + stride)) --
 * chitu_hip_moe_sum's arithmetic for a top-2 MoE whose un-summed outputs [M, 2, K] are handed over (fused_experts(reduce_topk=False)). */
int chitu_hip_bf16_gemm_add_norm(const void* x_bf16, int64_t x_row_stride, const void* add_bf16,
                                 int64_t add_row_stride, int32_t add_terms, int64_t add_term_stride,
                                 void* sum_out_bf16, int64_t sum_row_stride,
                                 const void* norm_weight_bf16, float eps, const void* w_bf16, void* out,
                                 int32_t out_dtype, int64_t M, int64_t N, int64_t K, void* stream);
/* chitu_hip_bf16_gemm_add_norm for the merged q|k|v projection of a GQA / MHA layer with chitu_hip_gqa_qkv_post
 * (layout 0: interleaved rotary pairs) applied in the epilogue: qkv_out [M, (q_heads + 2*kv_heads) * head_dim] receives
 * the ROTATED q heads only; the rotated k heads and the v heads go straight into the token's page rows of k_cache /
 * v_cache (Attention.decode_forward_paged, models/model.py:167-198).  Bit-identical to the three separate launches.
 * head_dim % 16 == 0; other limits as chitu_hip_bf16_gemm_add_norm. */
int chitu_hip_bf16_gemm_add_norm_qkv_post(const void* x_bf16, int64_t x_row_stride, const void* add_bf16,
                                          int64_t add_row_stride, void* sum_out_bf16, int64_t sum_row_stride,
                                          const void* norm_weight_bf16, float eps, const void* wqkv_bf16,
                                          void* qkv_out_bf16, int64_t M, int64_t K, int32_t q_heads, int32_t kv_heads,
                                          int32_t head_dim, const float* cos, const float* sin, void* k_cache,
                                          void* v_cache, int64_t num_pages, int32_t page_size,
                                          const int32_t* page_table, int32_t pages_per_seq,
                                          const int32_t* old_seq_lens, void* stream);
int chitu_hip_bf16_gemm_silu_add_norm(const void* x_bf16, int64_t x_row_stride, const void* add_bf16,
                                      int64_t add_row_stride, void* sum_out_bf16, int64_t sum_row_stride,
                                      const void* norm_weight_bf16, float eps, const void* w13_bf16,
                                      void* out_bf16, int64_t M, int64_t inter, int64_t K, void* stream);

/* ---- SiluAndMul (unquantised MLPs: Llama FeedForward, models/model.py:212-214; fused_moe.py:24-39)
 *   out[r, :] = bf16( bf16(silu(x[r, :d])) * x[r, d:2d] ),  x [rows, 2d] bf16, out [rows, d] bf16, d % 8 == 0. */
int chitu_hip_silu_and_mul(const void* x_bf16, void* out_bf16, int64_t rows, int64_t d, void* stream);

/* ---- decode-step prologue: embedding lookup + rotary-row gather in one launch -----------------------
 * Replaces VocabParallelEmbedding.forward's mask / lookup / zero-fill (chitu/tensor_parallel.py:199-208; the
 * all-reduce that follows stays the caller's) and prepare_freqs_cis_decode's gather of the rotary row of every
 * sequence's current position (chitu/models/model.py:429-448):
 *   h[b, :] = vocab_start <= tokens[b] < vocab_start + vocab_local ? embed[tokens[b] - vocab_start, :] : 0
 *   cos_out[b, :] = cos_table[positions[b], :], sin_out likewise ([table_rows, half] f32; skipped when cos_out NULL)
 *   tokens [batch] i64, embed [vocab_local, dim] bf16 (dim % 8 == 0), h [batch, dim] bf16, positions [batch] i32. */
int chitu_hip_embed_rope_gather(const int64_t* tokens, const void* embed_bf16, int64_t vocab_start,
                                int64_t vocab_local, int32_t dim, void* h_bf16, const int32_t* positions,
                                const float* cos_table, const float* sin_table, int64_t table_rows,
                                int32_t half, float* cos_out, float* sin_out, int32_t batch, void* stream);

/* ---- launch-variant override (tests, tuning sweeps) -------------------------------------------------
 * Several ops choose between launch variants of IDENTICAL results (K-split width, ring depth, key-based vs
 * generic routing kernel, candidate vs radix sampler ...) by a shape heuristic.  This entry forces one, so the
 * equivalence tests can run every variant and a sweep can time them; value -1 restores the heuristic.  Process-wide,
 * not thread-safe, never needed in production.  option: 0 MoE GEMM1 K-split waves, 1 GEMM1 tiles per workgroup,
 * 2 GEMM1 ring depth, 3 GEMM2 config (NT*10 + ROUNDS), 4 int8 MoE K-split waves, 5 generic routing kernel (1),
 * 6 ticket-based route + align (1), 7 radix-only sampler (1), 8 dense fp8 GEMM K-split waves, 9 its whole-K-in-flight
 * form for <= 8 blocks per wave (0 = off), 10 / 11 the same two for the bf16 GEMM, 12 K-split waves of the bf16 SwiGLU GEMM,
 * 13 the tiled (compute-shaped) form of the dense fp8 GEMM for M >= 128 (0 = keep streaming the weights per 64 rows),
 * 14 the same for the bf16 GEMM (0 = per 32 rows), 15 the in-routing sort of the one-workgroup route + align launch (0 = the
 * general sort after the routing barrier), 16 the token-tile height of the tiled fp8 GEMM (64 | 128; heuristic: 64 for small grids). */
int chitu_hip_debug_option(int32_t option, int32_t value);

/* ---- arithmetic self-test ----------------------------------------------------------------------
 * The quantising kernels divide a group's values by its scale with a refined-reciprocal +
 * residual-correction sequence instead of the IEEE division expansion, and round to bf16 with
 * v_cvt_pk_bf16_f32 instead of integer arithmetic.  This entry checks both on the device, element
 * by element, against the slow definitions: mismatches[0] += #(fast quotient != num/den bitwise),
 * [1] += #(their e4m3 encodings differ), [2] += #(hardware bf16 != software RNE), over the pairs
 * whose divisor is in the fast path's range [2^-60, 2^60] (callers keep |num/den| and |num| * 2^-24
 * inside the normal range, as the kernels' |x| <= 448 * scale does).  mismatches: 3 x u64, caller-zeroed. */
int chitu_hip_selftest_arith(const float* num, const float* den, int64_t n, uint64_t* mismatches,
                             void* stream);

/* Host-only (no launch): the XCD-blocked tile order of the prefill-shaped dense GEMM (csrc/gemm_common.h::xcd_tile_of:
 * workgroup b runs on XCD b % 8, so each XCD is given one rectangle of a tiles_m x tiles_n grid).  *grid_out = workgroups
 * launched (8 * Mt * Nt); tile_of_wg (may be NULL; capacity >= *grid_out pairs) receives (tile_m, tile_n) per workgroup,
 * (-1, -1) for padding workgroups.  Exists so that the map can be tested without a GPU. */
int chitu_hip_selftest_xcd_tile_order(int32_t tiles_m, int32_t tiles_n, int32_t* grid_out, int32_t* tile_of_wg,
                                      int64_t capacity);

/* ---- fused MoE with INT8 W8A8 experts (BASELINE config 4: Mixtral-8x7B W8A8) ----------------------
 * The reference has no int8-W8A8 fused path (Mixtral loops over experts, model_hf_mixtral.py:76-94, each
 * linear a W8A8Linear after simple_w8a8, quantize/w8a8.py:38-164); these run the same per-expert arithmetic
 * grouped over moe_align's sorted slots (block 16):
 *   gemm1_silu: a[slot, :] = bf16( bf16(silu(g)) * u ), g/u = bf16( (q_x[token] . w1[e, n | I+n]) * s_x[token] * s_w1[e, n | I+n] )
 *     a_int8 [tokens, K] + a_scale [tokens] (chitu_hip_quant_act_int8 of the hidden states); w1 [E, 2I, K] int8,
 *     w1_scale [E, 2I] f32; h_bf16 [numel, I].
 *   gemm2: out[slot, :] = bf16( bf16( (q_a[slot] . w2[e, n]) * s_a[slot] * s_w2[e, n] ) * topk_weights[slot] )
 *     a_int8 [numel, I] + a_scale [numel] (chitu_hip_quant_act_int8 of a); w2 [E, N, I] int8, w2_scale [E, N].
 *   Sum over the top-k with chitu_hip_moe_sum.  K % 128 == 0, I % 128 == 0; int32 accumulation is exact. */
int chitu_hip_moe_i8_gemm1_silu(const void* a_int8, const float* a_scale, const void* w1_int8,
                                const float* w1_scale, const int32_t* sorted_token_ids,
                                const int32_t* expert_ids, const int32_t* num_tokens_post_pad, void* h_bf16,
                                int64_t numel, int32_t topk, int64_t inter_size, int64_t K, int64_t max_mblocks,
                                void* stream);
int chitu_hip_moe_i8_gemm2(const void* a_int8, const float* a_scale, const void* w2_int8, const float* w2_scale,
                           const int32_t* sorted_token_ids, const int32_t* expert_ids,
                           const int32_t* num_tokens_post_pad, const void* topk_weights, int weights_dtype,
                           int32_t mul_routed_weight, void* out_bf16, int64_t numel, int64_t N,
                           int64_t inter_size, int64_t max_mblocks, void* stream);

/* ---- MLA absorb-mode paged decode attention ------------------------------------------------
 * Replaces mla_decode (chitu/triton_decode_attention.py:259-290: _mla_attn_kernel :20-130 +
 * _mla_softmax_reducev_kernel :185-232) as called from TritonAttnBackend.mla_attn_with_kvcache
 * (chitu/attn_backend.py:707-774), and the third-party flash_mla / flashinfer MLA calls
 * (attn_backend.py:561-571, 678-684):
 *   out[b,h,:] = softmax_t(scale * (q_nope[b,h,:].c[t,:C] + q_pe[b,h,:].c[t,C:])) . c[t,:C]
 *   q_nope [batch, heads, C=512] bf16 (element strides given, multiples of 8); q_pe [batch, heads, R=64];
 *   kv_cache [num_pages, page_size, C+R] bf16 (one layer), page_size % 64 == 0, base 16-byte aligned (the 64-key tiles
 *   go global -> LDS by LDS-DMA, 16 bytes per lane; the launch needs 152 KB of the CU's 160 KB of LDS);
 *   block_table [batch, table_stride] i32; seqlens [batch] i32 = tokens to attend (incl. the
 *   row appended this step -- append is chitu_hip_append_paged_kv); out [batch, heads, C] bf16.
 *   num_splits: KV splits per sequence (graph-static; any value >= 1 gives the same result up to
 *   fp32 rounding).  workspace: >= chitu_hip_mla_decode_workspace_bytes when num_splits > 1;
 *   it then holds part_o [batch, heads, num_splits, C] bf16 (each split's normalised output) followed by
 *   part_lse [batch, heads, num_splits] f32.  out_bf16 == NULL (num_splits > 1 only): skip the merge pass and leave the
 *   partials for chitu_hip_mla_merge_absorb_uv_quant_fp8. */
int chitu_hip_mla_decode_workspace_bytes(int32_t batch, int32_t heads, int32_t num_splits,
                                         int64_t* bytes);
int chitu_hip_mla_decode(const void* q_nope, int64_t qn_stride_b, int64_t qn_stride_h,
                         const void* q_pe, int64_t qp_stride_b, int64_t qp_stride_h,
                         const void* kv_cache, int64_t num_pages, int32_t page_size,
                         const int32_t* block_table, int32_t table_stride, const int32_t* seqlens,
                         float softmax_scale, void* out_bf16, int32_t batch, int32_t heads,
                         int32_t kv_lora_rank, int32_t rope_dim, int32_t num_splits,
                         void* workspace, int64_t workspace_bytes, void* stream);

/* MLA absorb-mode causal prefill attention (MQA, head dims 576 / 512): the attn_varlen_func call of
 * AttentionDeepSeekV3.prefill_forward (chitu/models/model_deepseek_v3.py:589-599; chitu/attn_backend.py:39-90).
 *   out[t,h,:] = softmax over keys s <= t of the same sequence ( scale * q[t,h,:] . kv[s,:] ) . kv[s,:512]
 *   q [T, heads, 576] bf16 (token / head strides in elements, multiples of 8); kv [T, 576] bf16 (row stride);
 *   cu_seqlens [n_seq + 1] i32 (device); max_seqlen bounds the grid; out [T, heads, 512] bf16 contiguous.
 * Per query token the arithmetic and its order are chitu_hip_mla_decode's with num_splits = 1. */
int chitu_hip_mla_prefill(const void* q_bf16, int64_t q_stride_t, int64_t q_stride_h, const void* kv_bf16,
                          int64_t kv_stride_t, const int32_t* cu_seqlens, int32_t n_seq, int32_t max_seqlen,
                          float softmax_scale, void* out_bf16, int32_t heads, int32_t kv_lora_rank,
                          int32_t rope_dim, void* stream);
/* The same contract (attn_varlen_func for the absorb-mode MQA shape, model_deepseek_v3.py:589-599; attn_backend.py:39-90;
 * the reference runs it on third-party flash_attn) on the flash kernel: 8 query tokens x 16 heads = 128 Q rows per workgroup,
 * S^T = K Q^T with Q in registers, in-lane softmax with deferred rescale, O^T = V^T P^T accumulated in the AGPR file, 32-key
 * blocks by LDS-DMA into a 4-slot ring.  Equal to chitu_hip_mla_prefill within the attention bar (1e-2 of the peak), not bit for bit.
 * All three base pointers 16-byte aligned. */
int chitu_hip_mla_prefill_flash(const void* q_bf16, int64_t q_stride_t, int64_t q_stride_h, const void* kv_bf16,
                          int64_t kv_stride_t, const int32_t* cu_seqlens, int32_t n_seq, int32_t max_seqlen,
                          float softmax_scale, void* out_bf16, int32_t heads, int32_t kv_lora_rank,
                          int32_t rope_dim, void* stream);

/* Causal GQA / MHA prefill attention, head_dim 128: the attn_varlen_func call of Attention.prefill_forward
 * (chitu/models/model.py:104-132; chitu/attn_backend.py:39-90; the reference runs it on third-party flash_attn, :167-206).
 *   out[t,h,:] = softmax over keys s <= t of the same sequence ( scale * q[t,h,:] . k[s,h/G,:] ) . v[s,h/G,:],  G = q_heads / kv_heads
 *   q [T, q_heads, 128], k / v [T, kv_heads, 128] bf16 (token / head strides in elements, multiples of 8; 16-byte aligned
 *   bases); cu_seqlens [n_seq + 1] i32 (device); max_seqlen bounds the grid; out [T, q_heads, 128] bf16 contiguous.
 * G must be a power of two <= 32.  Flash form (csrc/gqa_prefill_flash.hip): KV read once per 128 / G query tokens. */
int chitu_hip_gqa_prefill(const void* q_bf16, int64_t q_stride_t, int64_t q_stride_h, const void* k_bf16,
                          int64_t k_stride_t, int64_t k_stride_h, const void* v_bf16, int64_t v_stride_t,
                          int64_t v_stride_h, const int32_t* cu_seqlens, int32_t n_seq, int32_t max_seqlen,
                          float softmax_scale, void* out_bf16, int32_t q_heads, int32_t kv_heads, int32_t head_dim,
                          void* stream);

/* chitu_hip_mla_decode (num_splits >= 2) AND the merge + W_UV projection + act_quant of the entry below in ONE launch:
 * AttentionDeepSeekV3.decode_forward's mla_attn_with_kvcache -> einsum("bshc,hdc->bshd") -> act_quant of wo's input
 * (chitu/models/model_deepseek_v3.py:672-699, chitu/attn_backend.py:707-774).  Every split workgroup publishes its partial
 * rows, counts itself on its (sequence, head block)'s word in `tickets`, waits for the other splits (bounded: 200 ms, then
 * the word tickets[bytes/4 - 1] is set and stays set) and finishes one head.  Outputs are bit-identical to chitu_hip_mla_decode(
 * out_bf16 = NULL) followed by chitu_hip_mla_merge_absorb_uv_quant_fp8[_tm] (tile_major = 1).
 *   tickets: chitu_hip_mla_decode_tickets_bytes() bytes of device memory, zero when first used and never written by anyone
 *   else (the kernel leaves it zero); launches that share it must not overlap.  batch * ceil(heads/16) <= 4096.
 *   Other arguments: as the two entries it replaces; workspace as chitu_hip_mla_decode_workspace_bytes. */
int chitu_hip_mla_decode_tickets_bytes(int64_t* bytes);
int chitu_hip_mla_decode_merge_uv_quant_fp8(const void* q_nope, int64_t qn_stride_b, int64_t qn_stride_h, const void* q_pe,
                                            int64_t qp_stride_b, int64_t qp_stride_h, const void* kv_cache,
                                            int64_t num_pages, int32_t page_size, const int32_t* block_table,
                                            int32_t table_stride, const int32_t* seqlens, float softmax_scale,
                                            int32_t batch, int32_t heads, int32_t kv_lora_rank, int32_t rope_dim,
                                            int32_t num_splits, void* workspace, int64_t workspace_bytes,
                                            const void* w_fp8, int64_t w_stride_h, const float* scale,
                                            int64_t scale_offset, int64_t scale_stride_h, int64_t scale_stride_k,
                                            void* q_fp8, float* q_scales, int32_t tile_major, uint32_t* tickets,
                                            void* stream);

/* Split-KV merge + W_UV projection (model_deepseek_v3.py:697) + act_quant of wo's input in one
 * launch, for small batches (one workgroup per (head, token)): the same arithmetic and rounding
 * points as chitu_hip_mla_decode's merge pass followed by chitu_hip_absorb_uv_quant_fp8.
 * workspace: as left by chitu_hip_mla_decode(out_bf16 = NULL) with the same batch/heads/num_splits
 * (>= 2); weight/scale arguments as chitu_hip_absorb_uv_quant_fp8; K must be 512. */
/* ..._tm: the same launch writing q / scales TILE-MAJOR (see chitu_hip_fp8_gemm_blockscale_tm): q_fp8 holds
 * ceil(batch/16)*16 rows of heads*128 bytes, q_scales [ceil(batch/16), heads, 16]. */
int chitu_hip_mla_merge_absorb_uv_quant_fp8_tm(const void* workspace, int32_t num_splits,
                                               const void* w_fp8, int64_t w_stride_h, const float* scale,
                                               int64_t scale_offset, int64_t scale_stride_h,
                                               int64_t scale_stride_k, void* q_fp8, float* q_scales,
                                               int32_t batch, int32_t heads, int32_t K, void* stream);
int chitu_hip_mla_merge_absorb_uv_quant_fp8(const void* workspace, int32_t num_splits,
                                            const void* w_fp8, int64_t w_stride_h, const float* scale,
                                            int64_t scale_offset, int64_t scale_stride_h,
                                            int64_t scale_stride_k, void* q_fp8, float* q_scales,
                                            int32_t batch, int32_t heads, int32_t K, void* stream);

/* ---- RMSNorm (+ fused FP8 quantisation of its output) ----------------------------------------
 * Replaces RMSNorm.forward (chitu/models/model.py:29-78: F.rms_norm in fp32, one rounding) and,
 * when quant_mode != 0, also the act-quant launch of the fp8 linear that consumes it
 * (chitu/models/model_deepseek_v3.py:98-100).  quant_mode 1 = act_quant_deepseek_v3 rule,
 * 2 = per_token_group_quant_fp8 rule (eps = quant_eps); the codes are those of the bf16-rounded y.
 * quant_mode + 4 (5, 6; only with add_bf16): the same codes and scales written TILE-MAJOR for
 * chitu_hip_fp8_gemm_blockscale_tm -- q_fp8 ceil(rows/16)*16*dim bytes, q_scales ceil(rows/16)*(dim/128)*16 floats.
 *   add_bf16 (optional): residual branch folded in first, x <- bf16(x + add) -- the reference's
 *   `x = x + attn(...)` / `x = x + ffn(...)` (model_deepseek_v3.py:1107-1113); sum_out receives it.
 *   add_terms > 1 (<= 16): the residual of row r is first formed as bf16(sum_k float(add[r*add_row_stride +
 *   k*add_term_stride + :])) -- chitu_hip_moe_sum's arithmetic, i.e. the fused MoE's top-k sum
 *   (fused_moe.py:1299-1305) folded into the norm that consumes it; add_terms == 1: plain residual.
 *   x [rows, dim] bf16 (row stride given); weight [dim] bf16; y [rows, dim] bf16 or NULL;
 *   q_fp8 [rows, dim], q_scales [rows, dim/128] (dim % 128 == 0 when quantising); dim <= 8192.  * quant_mode 3 (round 6; residual-add form with one term, row-major): the per-token INT8 quantisation of the rounded y
 * instead (quant_act of the reference's W8A8Linear, chitu/quantize/w8a8.py:18-26: scale = max(|y|, 1e-5) / 127 over the row,
 * code = clamp(rint(y / scale), -128, 127)): q_fp8 then holds int8 codes [rows, dim], q_scales one fp32 per row.
 */
int chitu_hip_rmsnorm(const void* x_bf16, int64_t x_row_stride, const void* add_bf16,
                      int64_t add_row_stride, int32_t add_terms, int64_t add_term_stride,
                      void* sum_out_bf16, int64_t sum_row_stride,
                      const void* weight_bf16, void* y_bf16, int64_t y_row_stride, int64_t rows,
                      int32_t dim, float eps, void* q_fp8, float* q_scales, int32_t quant_mode,
                      float quant_eps, void* stream);

/* ---- MLA absorb projections with in-register FP8 dequant ---------------------------------------
 * Replaces weight_dequant(wkv_b) + einsum("shd,hdc->shc") / einsum("bshc,hdc->bshd")
 * (chitu/models/model_deepseek_v3.py:511-531, 697; kernel triton_kernels.py:217-247):
 *   out[b,h,n] = bf16( sum_k x[b,h,k] * bf16(float(w[h,n,k]) * scale[off + h*sh + (n/128)*sn + (k/128)*sk]) )
 *   x [batch, heads, K] bf16 (strides in elements, multiples of 8); w [heads, N, K] fp8 with head
 *   stride w_stride_h elements (so a row-slice of wkv_b can be passed in place); K % 64 == 0;
 *   out [batch, heads, N] bf16 (strides multiples of 4). */
int chitu_hip_absorb_bmm_fp8(const void* x_bf16, int64_t x_stride_b, int64_t x_stride_h,
                             const void* w_fp8, int64_t w_stride_h, const float* scale,
                             int64_t scale_offset, int64_t scale_stride_h, int64_t scale_stride_n, int64_t scale_stride_k,
                             void* out_bf16, int64_t out_stride_b, int64_t out_stride_h,
                             int32_t batch, int32_t heads, int32_t N, int32_t K, void* stream);

/* chitu_hip_absorb_bmm_fp8 plus, in the same launch, RoPE applied IN PLACE to q_pe [batch, heads, 64]
 * (interleaved pairs, cos/sin [batch, 32] f32; the q half of chitu_hip_mla_kv_prep, same arithmetic):
 * both consume wq_b's output, so they share one launch on the decode path. */
int chitu_hip_absorb_bmm_rope_fp8(const void* x_bf16, int64_t x_stride_b, int64_t x_stride_h,
                                  const void* w_fp8, int64_t w_stride_h, const float* scale,
                                  int64_t scale_offset, int64_t scale_stride_h, int64_t scale_stride_n,
                                  int64_t scale_stride_k, void* out_bf16, int64_t out_stride_b,
                                  int64_t out_stride_h, int32_t batch, int32_t heads, int32_t N, int32_t K,
                                  void* q_pe_bf16, int64_t q_pe_stride_b, int64_t q_pe_stride_h,
                                  const float* cos, const float* sin, int32_t rope_dim, void* stream);

/* chitu_hip_absorb_bmm_rope_fp8 plus, still in the same launch, the KV half of chitu_hip_mla_kv_prep: kv_norm(kv_c),
 * RoPE(k_pe) and the page append of every token's kv_in row [kv_c (512) | k_pe (64)] (row stride kv_row_stride
 * elements, a multiple of 8).  The decode path of models WITHOUT a q low-rank projection (DeepSeek-V2-Lite,
 * q_lora_rank == 0; the reference asserts q_lora_rank > 0, model_deepseek_v3.py:477): their q, kv_c and k_pe all come
 * out of one merged GEMM, so everything that consumes its output fits one launch.  N % 16 == 0, kv_lora_rank == 512,
 * rope_dim == 64 (CHITU_ERR_UNSUPPORTED otherwise).  Same arithmetic as the two separate entries, bit for bit. */
int chitu_hip_absorb_bmm_rope_kv_fp8(const void* x_bf16, int64_t x_stride_b, int64_t x_stride_h,
                                     const void* w_fp8, int64_t w_stride_h, const float* scale,
                                     int64_t scale_offset, int64_t scale_stride_h, int64_t scale_stride_n,
                                     int64_t scale_stride_k, void* out_bf16, int64_t out_stride_b,
                                     int64_t out_stride_h, int32_t batch, int32_t heads, int32_t N, int32_t K,
                                     void* q_pe_bf16, int64_t q_pe_stride_b, int64_t q_pe_stride_h,
                                     const float* cos, const float* sin, int32_t rope_dim, const void* kv_in_bf16,
                                     int64_t kv_row_stride, const void* kv_norm_weight_bf16, float eps,
                                     void* kv_cache, int64_t num_pages, int32_t page_size,
                                     const int32_t* page_table, int32_t pages_per_seq,
                                     const int32_t* old_seq_lens, int32_t kv_lora_rank, void* stream);

/* ---- skinny bf16 GEMM (router scores, LM head) ---------------------------------------------
 * Replaces the F.linear calls on bf16 weights on the decode path: gate scores
 * (chitu/models/model_deepseek_v3.py:820) and the LM head (tensor_parallel.py:93, model.py:468-475).
 *   out[m][n] = sum_k x[m][k] * w[n][k];  x [M, K] bf16, w [N, K] bf16, K % 64 == 0.
 *   num_splits == 1: out [M, N] of out_dtype is written.  num_splits > 1 (tiny N): K is split across
 *   workgroups and fp32 partials [num_splits, M, N] are written to `partials` for the consumer
 *   (chitu_hip_gate_route sums them in order); `out` is ignored. */
int chitu_hip_bf16_gemm(const void* x_bf16, const void* w_bf16, void* out, int out_dtype, int64_t M,
                        int64_t N, int64_t K, int32_t num_splits, float* partials, void* stream);

/* ---- fused MoE routing ------------------------------------------------------------------------
 * Replaces everything after the score GEMM in GateDeepSeekV3.forward
 * (chitu/models/model_deepseek_v3.py:821-842; ~16 torch launches): score function, bias, group-limited
 * top-k, weight gather / normalise / scale.  Ties go to the lower index (torch.topk: unspecified).
 *   logits: bf16 [tokens, E] when num_partials == 0, else fp32 [num_partials, tokens, E] summed here
 *   and rounded to bf16 like F.linear's output.  bias_bf16 [E] or NULL.  score_func 1 = sigmoid
 *   (bf16 pipeline, weights normalised), 0 = softmax (fp32 pipeline), 2 = softmax followed by renormalising the
 *   selected weights in fp32 (Mixtral's router, chitu/models/model_hf_mixtral.py:58-64).
 *   out_weights_bf16 / out_ids (int64) [tokens, out_stride]: slots 0..topk-1 sorted by descending
 *   selection score; if extra_expert_id >= 0, slots topk .. topk+extra_count-1 = (extra_expert_id + i,
 *   extra_weight) -- used to
 *   run the always-on shared expert (stacked last, model_deepseek_v3.py:883-919) through the same
 *   grouped GEMMs as the routed ones. */
int chitu_hip_gate_route(const void* logits, int32_t num_partials, int64_t tokens, int32_t num_experts,
                         const void* bias_bf16, int32_t n_groups, int32_t topk_groups, int32_t topk,
                         int32_t score_func, float route_scale, void* out_weights_bf16,
                         int64_t* out_ids, int32_t out_stride, int32_t extra_expert_id,
                         float extra_weight, int32_t extra_count, void* stream);
/* chitu_hip_gate_route and chitu_hip_moe_align_block_size_mapped in ONE launch (decode batches): the routing
 * workgroups (one per token) publish their ids and take a ticket; the last to arrive sorts the whole
 * [tokens * out_stride] id array (sentinels written by the kernel, fill_sentinels = 1 semantics).  Outputs
 * are those of the two separate launches, bit for bit.  align_num_experts covers every id that can appear
 * (routed + always-on slots), <= 1024; out_stride must equal topk + extra slots.  `ticket`: one
 * zero-initialised uint32 in device memory, reset by the kernel; launches sharing it must not overlap. */
int chitu_hip_gate_route_align(const void* logits, int32_t num_partials, int64_t tokens, int32_t num_experts,
                               const void* bias_bf16, int32_t n_groups, int32_t topk_groups, int32_t topk,
                               int32_t score_func, float route_scale, void* out_weights_bf16,
                               int64_t* out_ids, int32_t out_stride, int32_t extra_expert_id,
                               float extra_weight, int32_t extra_count, int32_t align_num_experts,
                               int32_t align_block_size, int32_t* sorted_token_ids, int64_t sorted_cap,
                               int32_t* expert_ids, int64_t expert_ids_cap, int32_t* num_tokens_post_pad,
                               int32_t* cumsum, const int32_t* expert_map, uint32_t* ticket, void* stream);

/* ---- MLA decode KV prep (kv_norm + RoPE + append, fused) ------------------------------------
 * Replaces four launches of AttentionDeepSeekV3.decode_forward_paged: apply_rotary_pos_emb on
 * (q_pe, k_pe) (chitu/models/model_deepseek_v3.py:493-500), kv_norm (:684), torch.cat (:686) and
 * append_to_paged_kv_cache (attn_backend.py:720-722).  Same arithmetic as the separate ops.
 *   kv_in_bf16: row b holds [kv_c (512) | k_pe (64)] (the tail of wqkv_a's output), row stride given;
 *   q_pe_bf16 [batch, heads, 64] rotated IN PLACE (strides in elements); cos/sin [batch, 32] f32;
 *   kv_cache [num_pages, page_size, 576] bf16: row (page_table[b][L/page], L%page), L = old_seq_lens[b],
 *   receives [rmsnorm(kv_c) * w | rope(k_pe)]. */
int chitu_hip_mla_kv_prep(const void* kv_in_bf16, int64_t kv_row_stride, void* q_pe_bf16,
                          int64_t q_stride_b, int64_t q_stride_h, int32_t heads, const float* cos,
                          const float* sin, const void* kv_norm_weight_bf16, float eps, void* kv_cache,
                          int64_t num_pages, int32_t page_size, const int32_t* page_table,
                          int32_t pages_per_seq, const int32_t* old_seq_lens, int32_t batch,
                          int32_t kv_lora_rank, int32_t rope_dim, void* stream);

/* Everything that consumes wqkv_a's output row [q_a (q_lora_rank) | kv_c (512) | k_pe (64)] in one
 * launch: q_norm + act_quant of q_a (= chitu_hip_rmsnorm quant_mode 1 -> q_fp8 [batch, q_lora_rank],
 * q_scales [batch, q_lora_rank/128], the input of the wq_b GEMM; model_deepseek_v3.py:480-487) and the
 * kv half of chitu_hip_mla_kv_prep (kv_norm + RoPE(k_pe) + page append).  q_pe is NOT rotated here
 * (it does not exist yet): use chitu_hip_absorb_bmm_rope_fp8 after wq_b.
 *   qkv_a: num_partials == 0: the GEMM's bf16 output [batch, row_stride]; num_partials >= 1: the fp32 split-K
 *   planes [num_partials, batch, row_stride] of chitu_hip_fp8_gemm_blockscale_partials -- each value is then
 *   bf16(plane 0 + plane 1 + ...), the rounding the GEMM's own epilogue would apply (q_lora_rank <= 2048). */
int chitu_hip_mla_qkv_post(const void* qkv_a, int32_t num_partials, int64_t row_stride, int32_t q_lora_rank,
                           const void* q_norm_weight_bf16, float q_eps, void* q_fp8, float* q_scales,
                           const void* kv_norm_weight_bf16, float kv_eps, const float* cos,
                           const float* sin, void* kv_cache, int64_t num_pages, int32_t page_size,
                           const int32_t* page_table, int32_t pages_per_seq,
                           const int32_t* old_seq_lens, int32_t batch, int32_t kv_lora_rank,
                           int32_t rope_dim, void* stream);

/* chitu_hip_mla_qkv_post AND the wq_b GEMM in one launch (decode, batch <= 32, q_lora_rank <= 2048):
 *   out[batch, N] = fp8_gemm(act_quant(q_norm(qkv_a[:, :q_lora_rank])), wq_b)   (model_deepseek_v3.py:488 with the
 *   linear's act_quant + fp8 GEMM, triton_kernels.py:194-216 / :302-365), and every token's
 *   [kv_norm(kv_c) | RoPE(k_pe)] row appended to its page (model_deepseek_v3.py:493-496, :684-686).
 * q_norm + act_quant run as the GEMM's prologue in every workgroup (per element the arithmetic of
 * chitu_hip_rmsnorm quant_mode 1; the mean square is summed in this kernel's own fixed order, so the rounded norm
 * can differ from chitu_hip_mla_qkv_post's in the last bf16 bit of a few elements).  wq_b [N, q_lora_rank] e4m3 with
 * scales [ceil(N/128), q_lora_rank/128]; out_dtype 0 bf16 / 1 f16 / 2 f32.  Other shapes: CHITU_ERR_UNSUPPORTED
 * (use chitu_hip_mla_qkv_post + chitu_hip_fp8_gemm_blockscale). */
int chitu_hip_mla_q_proj(const void* qkv_a_bf16, int64_t row_stride, int32_t q_lora_rank,
                         const void* q_norm_weight_bf16, float q_eps, const void* wq_b_fp8,
                         const float* wq_b_scale, void* out, int32_t out_dtype, int64_t N,
                         const void* kv_norm_weight_bf16, float kv_eps, const float* cos, const float* sin,
                         void* kv_cache, int64_t num_pages, int32_t page_size, const int32_t* page_table,
                         int32_t pages_per_seq, const int32_t* old_seq_lens, int32_t batch,
                         int32_t kv_lora_rank, int32_t rope_dim, void* stream);

/* ---- W_UV absorb projection fused with the FP8 quantisation of wo's input ---------------------
 * chitu_hip_absorb_bmm_fp8 for N = 128 (einsum "bshc,hdc->bshd", model_deepseek_v3.py:697) followed by
 * act_quant_deepseek_v3 of the bf16-rounded result (model_deepseek_v3.py:98-100 inside wo):
 *   q_fp8 [batch, heads*128] e4m3, q_scales [batch, heads] (one 128-group per head). */
int chitu_hip_absorb_uv_quant_fp8(const void* x_bf16, int64_t x_stride_b, int64_t x_stride_h,
                                  const void* w_fp8, int64_t w_stride_h, const float* scale,
                                  int64_t scale_offset, int64_t scale_stride_h, int64_t scale_stride_k,
                                  void* q_fp8, float* q_scales, int32_t batch, int32_t heads, int32_t K,
                                  void* stream);

/* ---- INT8 W8A8 linear (per-token activations x per-output-channel weights) --------------------
 * Replaces quant_act (chitu/quantize/w8a8.py:18-26) and the closed w8a8gemm.mm / w8a8gemv.mv behind
 * W8A8Linear.forward (chitu/quantize/w8a8.py:97-132; contract pinned by test/pytest/test_w8a8.py:13-48).
 *   quant_act_int8: s[row] = clamp(max|x[row]|, 1e-5)/127, q = int8(round_half_even(x/s)).
 *   w8a8_int8_gemm: out[m][n] = (sum_k a[m][k]*b[n][k] exact in i32) * a_scale[m] * b_scale[n] (+ bias[n]);
 *   a_int8 [M, K], b_int8 [N, K] (K % 128 == 0), scales f32, bias optional (bias_dtype), out of out_dtype. */
int chitu_hip_quant_act_int8(const void* x, int act_dtype, int64_t rows, int64_t cols, void* q_int8,
                             float* scales, void* stream);
int chitu_hip_w8a8_int8_gemm(const void* a_int8, const float* a_scale, const void* b_int8,
                             const float* b_scale, const void* bias, int bias_dtype, void* out,
                             int out_dtype, int64_t M, int64_t N, int64_t K, void* stream);

/* ---- GQA / MHA paged decode attention (head_dim 128) ------------------------------------------
 * Replaces the third-party flash_attn.flash_attn_with_kvcache call behind
 * FlashAttnBackend.attn_with_kvcache (chitu/attn_backend.py:208-243; contract :92-164) on the paged,
 * single-query path used by Attention.decode_forward_paged (chitu/models/model.py:167-198):
 *   out[b,h,:] = softmax_t(scale * q[b,h,:].K[t, h/(Hq/Hkv), :]) . V[t, h/(Hq/Hkv), :],  t < seqlens[b]
 *   q [batch, q_heads, 128] bf16 (element strides, multiples of 8); k_cache / v_cache
 *   [num_pages, page_size, kv_heads, 128] bf16, page_size % 16 == 0; block_table [batch, table_stride] i32;
 *   seqlens [batch] i32 = tokens to attend INCLUDING the row appended this step (append =
 *   chitu_hip_append_paged_kv on each cache); out [batch, q_heads, 128] bf16; q_heads/kv_heads <= 16.
 *   workspace >= batch*q_heads*num_splits*129*4 bytes when num_splits > 1. */
int chitu_hip_gqa_decode(const void* q_bf16, int64_t q_stride_b, int64_t q_stride_h, const void* k_cache,
                         const void* v_cache, int64_t num_pages, int32_t page_size, int32_t kv_heads,
                         const int32_t* block_table, int32_t table_stride, const int32_t* seqlens,
                         float softmax_scale, void* out_bf16, int32_t batch, int32_t q_heads,
                         int32_t head_dim, int32_t num_splits, void* workspace, int64_t workspace_bytes,
                         void* stream);

/* ---- token sampling (the step after the path) -----------------------------------
 * Replaces NormalExecutor.update_response's device work (chitu/executor.py:82-112) and
 * top_k_top_p_min_p_sampling_from_probs_torch (chitu/utils.py:62-81: full sort + cumsum + multinomial).
 *   frequency_penalty: logits[row, t] -= penalties[row] once per occurrence of t in
 *     tokens[offsets[row] .. offsets[row+1]) for rows with penalties[row] > 0 (executor.py:89-102);
 *     logits [rows, row_stride] f32 in place, tokens i32 (ids outside [0, vocab) ignored).
 *   sample: one token per row, out_tokens [rows] i64.
 *     top_ks == NULL: greedy for every row = first index of the row maximum (executor.py:103-104).
 *     else per row: top_k == 1 -> arg-max; otherwise weights e = exp(x/T - max) (probs_mode 0: `logits`
 *     are logits, temperatures [rows]) or p / max p (probs_mode 1: `logits` are probabilities,
 *     temperatures may be NULL); the entry at position pos of the descending order (ties: lower
 *     index first) is kept iff pos < top_k (top_k <= 0: no limit) and the exclusive cumulative
 *     weight before it is <= top_p * sum(weights) (top_p >= 1: no limit) -- utils.py:72-76;
 *     the token is the inverse CDF of uniforms[row] in [0, 1) over the kept weights in index
 *     order.  Weights are accumulated as integers (scaled by 2^40), so results are deterministic.
 *     n_kept_out [rows] i32 / kept_mass_out [rows] f32 (kept weight / total) are optional.
 *   logits of act_dtype (0 bf16, 1 f16, 2 f32), row_stride in elements, any vocab > 0. */
int chitu_hip_frequency_penalty(float* logits, int64_t row_stride, int64_t rows, int32_t vocab,
                                const int32_t* tokens, const int32_t* offsets, const float* penalties,
                                void* stream);
int chitu_hip_sample(const void* logits, int act_dtype, int64_t row_stride, int64_t rows, int32_t vocab,
                     const float* temperatures, const int32_t* top_ks, const float* top_ps,
                     const float* uniforms, int32_t probs_mode, int64_t* out_tokens,
                     int32_t* n_kept_out, float* kept_mass_out, void* stream);

/* ---- in-graph tensor-parallel collectives over xGMI (csrc/comm.hip) --------------------------------
 * Replace the NCCL calls of the reference's decode step -- dist.all_reduce after every RowParallelLinear
 * (chitu/tensor_parallel.py:157-169) and after the MoE (chitu/models/model_deepseek_v3.py:1010-1011), the
 * all_gather_into_tensor of a gather_output ColumnParallelLinear (tensor_parallel.py:94-102) -- with plain
 * kernel launches that a hipGraph captures like any other (the reference captures its NCCL calls into the
 * step's CUDA graph, chitu/models/model.py:554-617).
 * SETUP entries (these DO allocate / synchronise; call them once, outside any capture):
 *   comm_create: one uncached device buffer on the current device for rank `rank` of `world` (<= 8) ranks:
 *     all-reduce of up to max_rows x max_dim bf16 (max_dim <= 8192), all-gather of up to gather_bytes per rank;
 *     every wait inside the kernels gives up after timeout_ms and sets a sticky error word (comm_status).
 *   comm_ipc_handle / comm_open_peer: the 64-byte hipIpcMemHandle_t of this rank's buffer / map a peer's
 *     (ranks in different processes; exchange the handles over any host channel, e.g. the process group's
 *     store).  comm_local_ptr / comm_set_peer: the same for ranks that share one process (raw pointers).
 *     Every rank must have created (and thereby zeroed) its buffer before any peer launches a collective.
 *   comm_status: blocking read of the error word (0 = fine, bit 0 = a wait timed out).
 *   comm_poll_error: NON-blocking read of the pinned host copy of that word (the kernel that times out stores
 *     both); a serving loop calls it between steps -- where the reference would see NCCL's watchdog abort the
 *     process (its all_reduce never returns garbage silently) -- without synchronising the stream.
 * COLLECTIVES (enqueue only; every rank of the group must issue the same sequence of calls):
 *   comm_allreduce_rmsnorm -- one launch for [top-k sum ->] all-reduce -> residual add -> RMSNorm -> fp8 quant:
 *     part_r = terms > 1 ? bf16(sum_k float(part[row*part_row_stride + k*term_stride + :])) : part[row]
 *              (terms <= 16: chitu_hip_moe_sum's arithmetic, fused_moe.py:1299-1305)
 *     a      = bf16(sum over ranks r = 0..world-1, in that order, of float(part_r))   -- identical on every rank
 *     v      = x ? bf16(x + a) : a            -> sum_out [rows, dim]   (optional when a weight is given)
 *     weight: y = rmsnorm(v) * weight -> y_bf16 and / or (q_fp8, q_scales), exactly chitu_hip_rmsnorm's
 *     arithmetic and quant modes; weight == NULL: plain all-reduce (y, q must be NULL, sum_out required).
 *     rows <= max_rows, dim <= max_dim, dim % 8 == 0 (CHITU_ERR_UNSUPPORTED otherwise: use the library path).
 *   comm_all_gather: out[row, r*cols + j] = in_r[row, j] (rank-major concat of the last dimension);
 *     in bf16 [rows, cols] (row stride given, cols % 8 == 0, rows*cols*2 <= gather_bytes); out_dtype 0 = bf16,
 *     2 = f32 (the logits' `.float()`, models/model.py:475, rides along); out [rows, world*cols] dense.
 *   phase (both): 0 = the whole collective; 1 = contribute only (push this rank's data, signal, return);
 *     2 = complete only (wait for the peers, reduce / gather, advance the call counter) -- issue it after a
 *     phase-1 launch with the same arguments.  Split phases let work sit between the push and the wait. */
int chitu_hip_comm_create(int32_t rank, int32_t world, int32_t max_rows, int32_t max_dim,
                          int64_t gather_bytes, int32_t timeout_ms, void** comm_out);
int chitu_hip_comm_ipc_handle(void* comm, void* handle_out_64);
int chitu_hip_comm_open_peer(void* comm, int32_t peer, const void* handle_64);
int chitu_hip_comm_local_ptr(void* comm, void** ptr_out);
int chitu_hip_comm_set_peer(void* comm, int32_t peer, void* ptr);
int chitu_hip_comm_status(void* comm, uint32_t* err_out);
int chitu_hip_comm_poll_error(void* comm, uint32_t* err_out);
/* All-reduces of >= min_bytes per rank take the two-shot form (reduce-scatter + all-gather inside the same launch: 2/world
 * of the bytes per xGMI link, one more flag hop; bit-identical to the one-shot form).  Default 256 KB; same value on every
 * rank; host-side, set before capture.  With it, phase 3 = "wait for hop 1, reduce my slice, send hop 2" of a split launch
 * (1 = hop 1, 3, 2 = wait for hop 2 + finish). */
int chitu_hip_comm_set_two_shot(void* comm, int64_t min_bytes);
int chitu_hip_comm_destroy(void* comm);
int chitu_hip_comm_allreduce_rmsnorm(void* comm, const void* part_bf16, int64_t part_row_stride,
                                     int32_t terms, int64_t term_stride, const void* x_bf16,
                                     int64_t x_row_stride, void* sum_out_bf16, int64_t sum_row_stride,
                                     const void* weight_bf16, void* y_bf16, int64_t y_row_stride,
                                     int64_t rows, int32_t dim, float eps, void* q_fp8, float* q_scales,
                                     int32_t quant_mode, float quant_eps, int32_t phase, void* stream);
int chitu_hip_comm_all_gather(void* comm, const void* in_bf16, int64_t in_row_stride, int64_t rows,
                              int64_t cols, void* out, int32_t out_dtype, int32_t phase, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CHITU_HIP_H */
