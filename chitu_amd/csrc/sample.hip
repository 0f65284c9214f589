// Token sampling for the decode step: frequency penalty, greedy argmax, temperature softmax with
// top-k / top-p filtering and one draw per row -- the step right after the hot path.
//
// Replaces (reference, read-only):
//   chitu/executor.py:82-112   NormalExecutor.update_response: index_add_ of -frequency_penalty per
//                              generated token, argmax if all rows are greedy, else
//                              softmax(logits / temperature) -> top_k_top_p_min_p_sampling_from_probs_torch
//   chitu/utils.py:62-81       full descending sort of [bs, vocab], cumsum, two masks, multinomial
//
// The reference sorts the whole vocabulary to cut a prefix of the sorted order.  Here the prefix is
// found without sorting: the kept set is {elements with weight > tau} plus the first c ties at tau,
// and (tau, c) comes from a 3-level radix descent (10 bits per level) over the 30-bit float pattern of
// the weight e = exp(x - max) in (0, 1].  Each level is one pass over the row (L2-resident, 517 KB at
// vocab 129280) into a 1024-bin LDS histogram of (count, fixed-point sum); sums are integers
// (weights scaled by 2^40), so the result does not depend on the order atomics land in and two runs
// agree bit for bit.  A prefix position `pos` of the descending order is kept iff pos < top_k and
// exclusive_cumsum(pos) <= top_p * Z -- the two masks of utils.py:72-76 (ties in sorted order broken
// by lower index; torch.sort leaves that order unspecified).  The draw is an inverse CDF over the
// kept weights in INDEX order with a caller-supplied uniform u in [0, 1): same distribution as
// torch.multinomial on the masked probabilities, and reproducible given u.
//
// One workgroup (1024 threads) per row; rows are independent.
#include "common.h"

namespace chitu {

constexpr int kSampleThreads = 1024;
constexpr int kSampleWaves = kSampleThreads / 64;
constexpr int kSampleBins = 1024;
constexpr float kSampleFixScale = 1099511627776.0f;  // 2^40: weights in (0, 1] -> integers <= 2^40

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_row_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ uint64_t dpp_row_u64(uint64_t v) {
    return ((uint64_t)dpp_row_u32<CTRL>((uint32_t)(v >> 32)) << 32) | dpp_row_u32<CTRL>((uint32_t)v);
}
// Sum over the 64 lanes, result in every lane (wave-uniform): DPP rotations inside each 16-lane row,
// then the four row results through v_readlane.  Must be called by the whole wave.
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
    v += dpp_row_u64<0x128>(v);  // row_ror:8
    v += dpp_row_u64<0x124>(v);  // row_ror:4
    v += dpp_row_u64<0x122>(v);  // row_ror:2
    v += dpp_row_u64<0x121>(v);  // row_ror:1
    const int lo = (int)(uint32_t)v, hi = (int)(uint32_t)(v >> 32);
    uint64_t s = 0;
#pragma unroll
    for (int l = 0; l < 64; l += 16)
        s += ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(hi, l) << 32) | (uint32_t)__builtin_amdgcn_readlane(lo, l);
    return s;
}
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int off) {
    return ((uint64_t)(uint32_t)__shfl_xor((int)(v >> 32), off, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)v, off, 64);
}
__device__ __forceinline__ uint64_t shfl_up_u64(uint64_t v, int off) {
    return ((uint64_t)(uint32_t)__shfl_up((int)(v >> 32), off, 64) << 32) | (uint32_t)__shfl_up((int)(uint32_t)v, off, 64);
}

// total order on floats as unsigned integers (larger float <-> larger key; NaN sorts above +inf,
// so a NaN logit is the arg-max, as in torch.argmax)
__device__ __forceinline__ uint32_t ordered_key(float v) {
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_key_inv(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// Four consecutive elements i0 .. i0+3 of a row (i0 % 4 == 0, i0 < vocab); returns how many are in
// range.  `vec`: the row start is 16-B (f32) / 8-B (16-bit types) aligned, so one wide load is legal.
template <int DT>
__device__ __forceinline__ int load4(const void* row, int i0, int vocab, bool vec, float (&v)[4]) {
    const int n = min(4, vocab - i0);
    if (DT == 2) {
        const float* p = reinterpret_cast<const float*>(row) + i0;
        if (vec && n == 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = t[k];
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = k < n ? p[k] : 0.f;
        }
    } else {
        const uint16_t* p = reinterpret_cast<const uint16_t*>(row) + i0;
        uint16_t h[4];
        if (vec && n == 4) {
            const i32x2 t = *reinterpret_cast<const i32x2*>(p);
            h[0] = (uint16_t)((uint32_t)t[0] & 0xffffu);
            h[1] = (uint16_t)((uint32_t)t[0] >> 16);
            h[2] = (uint16_t)((uint32_t)t[1] & 0xffffu);
            h[3] = (uint16_t)((uint32_t)t[1] >> 16);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) h[k] = k < n ? p[k] : (uint16_t)0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = DT == 0 ? bf16_to_f32(h[k]) : f16_to_f32(h[k]);
    }
    return n;
}

// One pass over a row: every wave visits its 256-element chunks (wave w: [256 w + 4096 j, + 256)), a lane
// four consecutive elements per chunk.  The loads of U chunks are issued before the first is consumed:
// a pass is otherwise one dependent HBM / L2 round trip per chunk (measured: 25 us per pass over
// 129280 floats with one load in flight per lane).  body(i0, v[4], n) is called by the WHOLE wave for
// every chunk (n = 0 for lanes past the end of the row), so it may use wave-wide operations.
constexpr int kSampleUnroll = 8;
template <int DT, int U, typename F>
__device__ __forceinline__ void for_each_quad(const void* row, int vocab, bool vec, F&& body) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = wave * 256; base < vocab; base += kSampleThreads * 4 * U) {  // wave-uniform trip count
        float v[U][4];
        int n[U];
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int i0 = base + q * kSampleThreads * 4 + lane * 4;
            n[q] = i0 < vocab ? load4<DT>(row, i0, vocab, vec, v[q]) : 0;
        }
#pragma unroll
        for (int q = 0; q < U; ++q) {
            if (base + q * kSampleThreads * 4 < vocab)  // wave-uniform
                body(base + q * kSampleThreads * 4 + lane * 4, v[q], n[q]);
        }
    }
}

// x = logits / temperature (executor.py:106, an IEEE f32 division) in logits mode; the probability
// itself in probs mode (utils.py:62 takes probabilities).
__device__ __forceinline__ float sample_x(float v, float temperature, int probs_mode) {
    // v / 1.0f == v bit for bit: greedy batches and unit temperatures skip the ~10-instruction IEEE
    // division (the passes over a row are VALU-bound: one workgroup per row)
    return (probs_mode || temperature == 1.0f) ? v : v / temperature;
}
// weight in [0, 1] relative to the row maximum m: exp(x - m) (softmax numerator) or p / max p.  The
// same expression is evaluated in every pass, so every pass sees the same bits.
__device__ __forceinline__ float sample_e(float x, float m, int probs_mode) {
#pragma clang fp contract(off)
    const float e = probs_mode ? x / m : __expf(x - m);
    return __builtin_fminf(__builtin_fmaxf(e, 0.f), 1.f);  // NaN -> 0
}
__device__ __forceinline__ uint64_t sample_fix(float e) { return (uint64_t)(e * kSampleFixScale); }

struct SampleShared {
    uint32_t cnt[kSampleBins];
    uint64_t sum[kSampleBins];
    uint64_t wave_u64[kSampleWaves];
    uint32_t wave_u32[kSampleWaves];
    uint64_t seg_strict[kSampleWaves];
    uint32_t seg_ties[kSampleWaves];
    // boundary bin of the current level
    int b_found;
    uint32_t b_bin, b_cexcl, b_cnt;
    uint64_t b_sexcl, b_sum, total;
};

// One histogram level: elements whose key >> (shift + 10) == prefix (level 1: all) are binned by
// (key >> shift) & 1023 with (count, fixed-point sum).  Lanes of a wave that hit the same bin as the
// wave's first active lane are combined in registers first (two rounds): flat distributions put
// most of the vocabulary in a handful of bins and same-address LDS atomics serialise.
template <int DT>
__device__ __forceinline__ void sample_hist_level(SampleShared& sh, const void* row, int vocab, bool vec,
                                                  float temperature, float m, int probs_mode, int shift,
                                                  bool use_prefix, uint32_t prefix) {
    const int tid = threadIdx.x, lane = tid & 63;
    for (int b = tid; b < kSampleBins; b += kSampleThreads) {
        sh.cnt[b] = 0;
        sh.sum[b] = 0;
    }
    __syncthreads();
    for_each_quad<DT, kSampleUnroll>(row, vocab, vec, [&](int i0, const float (&v)[4], int n) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float e = sample_e(sample_x(v[k], temperature, probs_mode), m, probs_mode);
            const uint32_t key = __float_as_uint(e);
            const uint64_t w = sample_fix(e);
            bool act = k < n && (!use_prefix || (key >> (shift + 10)) == prefix);
            const int b = (int)((key >> shift) & (kSampleBins - 1));
            uint64_t rem = __builtin_amdgcn_ballot_w64(act);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (rem == 0) break;  // wave-uniform
                const int leader = __builtin_amdgcn_readfirstlane(__builtin_ctzll(rem));
                const int lb = __builtin_amdgcn_readlane(b, leader);
                const bool same = act && b == lb;
                const uint64_t smask = __builtin_amdgcn_ballot_w64(same);
                const uint64_t ws = wave_sum_u64(same ? w : 0ull);
                if (lane == leader) {
                    atomicAdd(&sh.cnt[lb], (uint32_t)__builtin_popcountll(smask));
                    atomicAdd(reinterpret_cast<unsigned long long*>(&sh.sum[lb]), (unsigned long long)ws);
                }
                act = act && !same;
                rem &= ~smask;
            }
            if (act) {
                atomicAdd(&sh.cnt[b], 1u);
                atomicAdd(reinterpret_cast<unsigned long long*>(&sh.sum[b]), (unsigned long long)w);
            }
        }
    });
    __syncthreads();
}

// Walk the bins from the largest weight down: the boundary bin is the first whose inclusive count
// reaches k_rem or whose inclusive sum exceeds p_rem (everything before it is kept whole).  Thread t
// owns bin 1023 - t.  Leaves {b_found, b_bin, b_cexcl, b_sexcl, b_cnt, b_sum, total} in LDS.
// `p_from_total`: level 1 derives p_rem = floor(top_p * total) once the total is known.
__device__ __forceinline__ void sample_find_boundary(SampleShared& sh, uint32_t k_rem, uint64_t& p_rem,
                                                     bool p_from_total, float top_p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = kSampleBins - 1 - tid;
    const uint32_t c_own = sh.cnt[b];
    const uint64_t s_own = sh.sum[b];
    uint32_t c = c_own;
    uint64_t s = s_own;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t c2 = (uint32_t)__shfl_up((int)c, off, 64);
        const uint64_t s2 = shfl_up_u64(s, off);
        if (lane >= off) {
            c += c2;
            s += s2;
        }
    }
    if (lane == 63) {
        sh.wave_u32[wave] = c;
        sh.wave_u64[wave] = s;
    }
    if (tid == 0) sh.b_found = 0;
    __syncthreads();
    uint32_t co = 0;
    uint64_t so = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kSampleWaves; ++w) {
        if (w < wave) {
            co += sh.wave_u32[w];
            so += sh.wave_u64[w];
        }
        tot += sh.wave_u64[w];
    }
    if (p_from_total) {
        // floor(top_p * Z) in f64: one IEEE multiply of two exactly-converted operands
        // (top_p >= 1 keeps everything: the mathematical meaning of utils.py:72, whose f32 cumsum can
        // overshoot 1.0 by an ulp)
        const double pz = (double)__builtin_fmaxf(top_p, 0.f) * (double)tot;
        p_rem = (top_p >= 1.0f || pz >= 18446744073709549568.0) ? ~0ull : (uint64_t)pz;
    }
    const uint32_t c_incl = c + co, c_excl = c_incl - c_own;
    const uint64_t s_incl = s + so, s_excl = s_incl - s_own;
    const bool cond = c_incl >= k_rem || s_incl > p_rem;
    const bool cond_prev = c_excl >= k_rem || s_excl > p_rem;
    if (cond && !cond_prev) {  // exactly one bin (an empty bin has cond == cond_prev)
        sh.b_found = 1;
        sh.b_bin = (uint32_t)b;
        sh.b_cexcl = c_excl;
        sh.b_sexcl = s_excl;
        sh.b_cnt = c_own;
        sh.b_sum = s_own;
    }
    if (tid == 0) sh.total = tot;
    __syncthreads();
}

// ---------------------------------------------------------------- candidate fast path
// Peaked rows (what a language model emits) keep a few dozen tokens out of 129280, so the prefix can be
// cut from a short candidate list instead of three histogram passes over the row:
//   pass 1  Z = sum of the fixed-point weights, and how many weights reach each of five thresholds
//           2^-4 .. 2^-20 of the maximum (register counters, no atomics);
//   pass 2  the elements above the lowest threshold that admits <= kCandCap of them are appended
//           (key, index) to LDS (one LDS atomic per wave-load);
//   then    a bitonic sort in the specification's order (key descending, index ascending), a scan of
//           the exact integer weights for the top-k / top-p cut, a second sort of the kept entries by
//           index and a scan for the inverse CDF.
// The list provably holds the whole kept prefix iff the cut falls inside it, or it is as long as
// top_k, or its mass already exceeds top_p * Z, or it is the whole row; otherwise (flat rows, huge
// top_k, no limits at all) the radix descent below runs.  Both paths evaluate the same integer
// specification (oracle/sampling.py::sample_fixed_point), so which one ran cannot be told from the
// result.
constexpr int kCandCap = 1024;
constexpr int kCandLevels = 5;
__device__ __forceinline__ uint32_t cand_threshold_key(int j) { return (uint32_t)(127 - 4 * (j + 1)) << 23; }  // bits of 2^-(4j+4)

__device__ __forceinline__ uint64_t block_sum_u64(SampleShared& sh, uint64_t v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_sum_u64(v);
    __syncthreads();  // earlier readers of wave_u64 are done
    if (lane == 0) sh.wave_u64[wave] = v;
    __syncthreads();
    uint64_t s = 0;
#pragma unroll
    for (int w = 0; w < kSampleWaves; ++w) s += sh.wave_u64[w];
    return s;
}

// exclusive prefix sum over the workgroup's threads (thread order); `total` = sum over all threads
__device__ __forceinline__ uint64_t block_excl_scan_u64(SampleShared& sh, uint64_t v, uint64_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint64_t o = shfl_up_u64(incl, off);
        if (lane >= off) incl += o;
    }
    __syncthreads();
    if (lane == 63) sh.wave_u64[wave] = incl;
    __syncthreads();
    uint64_t before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kSampleWaves; ++w) {
        if (w < wave) before += sh.wave_u64[w];
        tot += sh.wave_u64[w];
    }
    total = tot;
    return before + incl - v;
}

// in-place bitonic sort of a[0 .. n) in LDS, n a power of two <= kSampleThreads, one element per thread
template <bool DESCENDING>
__device__ __forceinline__ void bitonic_sort_lds(uint64_t* a, int n) {
    const int tid = threadIdx.x;
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int p = tid ^ j;
            if (tid < n && p > tid) {
                const uint64_t x = a[tid], y = a[p];
                const bool first_half = (tid & k) == 0;
                const bool swap = (DESCENDING == first_half) ? x < y : x > y;
                if (swap) {
                    a[tid] = y;
                    a[p] = x;
                }
            }
            __syncthreads();
        }
    }
}

// Returns true when the token (and the optional statistics) have been written; false = not applicable,
// nothing written, the caller runs the radix descent.  Must be called by the whole workgroup.
template <int DT>
__device__ __forceinline__ bool sample_candidates(SampleShared& sh, const void* row, int vocab, bool vec, float temperature,
                                                  float m, int probs_mode, int top_k, float top_p, float u, int row_id,
                                                  int64_t arg_max, int64_t* out_tokens, int32_t* n_kept_out,
                                                  float* kept_mass_out) {
    const int tid = threadIdx.x, lane = tid & 63;
    uint64_t* arr = sh.sum;  // the histogram's sum array doubles as the sort buffer (kCandCap == kSampleBins)
    // ---- pass 1: Z and the five threshold counts (21 bits each: vocab < 2^21 per the entry point)
    uint64_t z = 0, ca = 0, cb = 0;
    for_each_quad<DT, kSampleUnroll>(row, vocab, vec, [&](int i0, const float (&v)[4], int n) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < n) {
                const float e = sample_e(sample_x(v[k], temperature, probs_mode), m, probs_mode);
                const uint32_t key = __float_as_uint(e);
                z += sample_fix(e);
                ca += (uint64_t)(key >= cand_threshold_key(0)) | ((uint64_t)(key >= cand_threshold_key(1)) << 21) |
                      ((uint64_t)(key >= cand_threshold_key(2)) << 42);
                cb += (uint64_t)(key >= cand_threshold_key(3)) | ((uint64_t)(key >= cand_threshold_key(4)) << 21);
            }
        }
    });
    z = block_sum_u64(sh, z);
    ca = block_sum_u64(sh, ca);
    cb = block_sum_u64(sh, cb);
    const uint32_t counts[kCandLevels] = {(uint32_t)(ca & 0x1fffffu), (uint32_t)((ca >> 21) & 0x1fffffu),
                                          (uint32_t)((ca >> 42) & 0x1fffffu), (uint32_t)(cb & 0x1fffffu),
                                          (uint32_t)((cb >> 21) & 0x1fffffu)};
    if (z == 0 || counts[0] > (uint32_t)kCandCap) return false;
    int level = 0;
#pragma unroll
    for (int j = 1; j < kCandLevels; ++j)
        if (counts[j] <= (uint32_t)kCandCap) level = j;
    const uint32_t tkey = cand_threshold_key(level);
    const int n_cand = (int)counts[level];
    uint64_t p_rem;
    {
        const double pz = (double)__builtin_fmaxf(top_p, 0.f) * (double)z;  // as sample_find_boundary
        p_rem = (top_p >= 1.0f || pz >= 18446744073709549568.0) ? ~0ull : (uint64_t)pz;
    }
    const uint32_t k_eff = top_k <= 0 ? 0xffffffffu : (uint32_t)top_k;
    // ---- pass 2: append (key, index) of every weight >= the threshold; one LDS atomic per wave-load
    if (tid == 0) sh.b_cnt = 0;
    __syncthreads();
    for_each_quad<DT, kSampleUnroll>(row, vocab, vec, [&](int i0, const float (&v)[4], int n) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t key = 0;
            if (k < n) key = __float_as_uint(sample_e(sample_x(v[k], temperature, probs_mode), m, probs_mode));
            const bool take = k < n && key >= tkey;
            const uint64_t mask = __builtin_amdgcn_ballot_w64(take);
            if (mask) {  // wave-uniform
                uint32_t start = 0;
                if (lane == 0) start = atomicAdd(&sh.b_cnt, (uint32_t)__builtin_popcountll(mask));
                start = (uint32_t)__builtin_amdgcn_readfirstlane((int)start);
                const uint64_t below = lane == 0 ? 0ull : (mask & (~0ull >> (64 - lane)));
                if (take)
                    arr[start + (uint32_t)__builtin_popcountll(below)] =
                        ((uint64_t)key << 32) | (uint32_t)(0xffffffffu - (uint32_t)(i0 + k));
            }
        }
    });
    int npow = 64;
    while (npow < n_cand) npow <<= 1;
    __syncthreads();
    if (tid >= n_cand && tid < npow) arr[tid] = 0ull;  // below every real entry (a real key is >= the threshold > 0)
    __syncthreads();
    bitonic_sort_lds<true>(arr, npow);  // key descending, ties: lower index first
    // ---- the cut: position < top_k and exclusive cumulative weight <= top_p * Z (utils.py:72-76)
    const uint64_t mine = tid < n_cand ? arr[tid] : 0ull;
    const uint32_t my_key = (uint32_t)(mine >> 32);
    const uint32_t my_idx = 0xffffffffu - (uint32_t)mine;
    const uint64_t w = tid < n_cand ? sample_fix(__uint_as_float(my_key)) : 0ull;
    uint64_t s_cand;
    const uint64_t excl = block_excl_scan_u64(sh, w, s_cand);
    const bool kept = tid < n_cand && (uint32_t)tid < k_eff && excl <= p_rem;
    const uint64_t packed = block_sum_u64(sh, kept ? ((w << 11) | 1ull) : 0ull);  // w <= 2^40, <= 1024 kept: 51 + 11 bits
    const int n_kept = (int)(packed & 0x7ffu);
    const uint64_t s_kept = packed >> 11;
    const bool sufficient = n_kept < n_cand || (uint32_t)n_cand >= k_eff || s_cand > p_rem || n_cand == vocab;
    if (!sufficient) return false;
    if (tid == 0) {
        if (n_kept_out) n_kept_out[row_id] = n_kept;
        if (kept_mass_out) kept_mass_out[row_id] = (float)((double)s_kept / (double)z);
    }
    if (s_kept == 0) {
        if (tid == 0) out_tokens[row_id] = arg_max;
        return true;
    }
    uint64_t target;
    {
        const double t = (double)__builtin_fminf(__builtin_fmaxf(u, 0.f), 1.f) * (double)s_kept;
        target = t >= (double)s_kept ? s_kept - 1 : (uint64_t)t;
        if (target >= s_kept) target = s_kept - 1;
    }
    // ---- inverse CDF over the kept entries in INDEX order: sort them by index, scan, pick
    __syncthreads();
    if (tid < npow) arr[tid] = kept ? (((uint64_t)my_idx << 32) | my_key) : ~0ull;
    __syncthreads();
    bitonic_sort_lds<false>(arr, npow);
    const uint64_t ent = tid < n_kept ? arr[tid] : 0ull;
    const uint64_t w2 = tid < n_kept ? sample_fix(__uint_as_float((uint32_t)ent)) : 0ull;
    uint64_t unused_total;
    const uint64_t excl2 = block_excl_scan_u64(sh, w2, unused_total);
    if (tid < n_kept && excl2 <= target && target < excl2 + w2) out_tokens[row_id] = (int64_t)(ent >> 32);
    return true;
}

template <int DT>
__global__ __launch_bounds__(kSampleThreads) void sample_kernel(
    const void* __restrict__ logits, int64_t row_stride, int vocab, const float* __restrict__ temperatures,
    const int32_t* __restrict__ top_ks, const float* __restrict__ top_ps, const float* __restrict__ uniforms,
    int probs_mode, int64_t* __restrict__ out_tokens, int32_t* __restrict__ n_kept_out,
    float* __restrict__ kept_mass_out, int use_candidates) {
    __shared__ SampleShared sh;
    const int row_id = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int elem = DT == 2 ? 4 : 2;
    const char* row = reinterpret_cast<const char*>(logits) + (int64_t)row_id * row_stride * elem;
    const bool vec = (reinterpret_cast<uintptr_t>(row) & (uintptr_t)(4 * elem - 1)) == 0;
    const bool greedy_all = top_ks == nullptr;
    const int top_k = greedy_all ? 1 : top_ks[row_id];
    const float temperature = (greedy_all || probs_mode) ? 1.0f : temperatures[row_id];

    // ---- pass 0: row maximum and its (lowest) index
    uint64_t best = 0;
    for_each_quad<DT, kSampleUnroll>(row, vocab, vec, [&](int i0, const float (&v)[4], int n) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < n) {
                const uint64_t key = ((uint64_t)ordered_key(sample_x(v[k], temperature, probs_mode)) << 32) |
                                     (uint32_t)(0xffffffffu - (uint32_t)(i0 + k));
                best = key > best ? key : best;
            }
        }
    });
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint64_t o = shfl_xor_u64(best, off);
        best = o > best ? o : best;
    }
    if (lane == 0) sh.wave_u64[wave] = best;
    __syncthreads();
    best = sh.wave_u64[0];
#pragma unroll
    for (int w = 1; w < kSampleWaves; ++w) best = sh.wave_u64[w] > best ? sh.wave_u64[w] : best;
    __syncthreads();  // wave_u64 is reused below
    const int64_t arg_max = (int64_t)(0xffffffffu - (uint32_t)best);
    const float m = ordered_key_inv((uint32_t)(best >> 32));
    if (top_k == 1) {  // greedy row (executor.py:103-104; a top_k of 1 keeps only the first sorted entry)
        if (tid == 0) {
            out_tokens[row_id] = arg_max;
            if (n_kept_out) n_kept_out[row_id] = 1;
            if (kept_mass_out) kept_mass_out[row_id] = 0.f;
        }
        return;
    }
    const float top_p = top_ps[row_id];
    const float u = uniforms[row_id];

    // ---- short candidate list first (peaked rows); the radix descent below is the general path
    if (use_candidates && vocab < (1 << 21) &&
        sample_candidates<DT>(sh, row, vocab, vec, temperature, m, probs_mode, top_k, top_p, u, row_id, arg_max,
                              out_tokens, n_kept_out, kept_mass_out))
        return;
    __syncthreads();  // the candidate path's LDS (sh.sum, counters) is about to be reused

    // ---- radix descent for (tau, c_keep)
    uint32_t k_rem = top_k <= 0 ? 0xffffffffu : (uint32_t)top_k;  // <= 0: no top-k limit
    uint64_t p_rem = 0;
    uint32_t tau = 0, c_above = 0, c_keep = 0xffffffffu;
    uint64_t s_above = 0, e_tau = 0, z_total = 0;
    bool keep_all = false;
    sample_hist_level<DT>(sh, row, vocab, vec, temperature, m, probs_mode, 20, false, 0u);
    sample_find_boundary(sh, k_rem, p_rem, true, top_p);
    z_total = sh.total;
    if (!sh.b_found) {
        keep_all = true;
    } else {
        uint32_t prefix = sh.b_bin;
        c_above = sh.b_cexcl;
        s_above = sh.b_sexcl;
        k_rem -= sh.b_cexcl;
        p_rem -= sh.b_sexcl;
#pragma unroll 1
        for (int shift = 10; shift >= 0; shift -= 10) {
            __syncthreads();  // every thread has read the previous level's result
            sample_hist_level<DT>(sh, row, vocab, vec, temperature, m, probs_mode, shift, true, prefix);
            sample_find_boundary(sh, k_rem, p_rem, false, top_p);
            // the parent bin met the condition as a whole, so one of its children does
            prefix = (prefix << 10) | sh.b_bin;
            c_above += sh.b_cexcl;
            s_above += sh.b_sexcl;
            k_rem -= sh.b_cexcl;
            p_rem -= sh.b_sexcl;
        }
        tau = prefix;
        const uint32_t c_tau = sh.b_cnt;
        e_tau = c_tau ? sh.b_sum / c_tau : 0;  // all entries of the bin share one key, hence one weight
        // tie j (0-based) is kept iff j < k_rem and s_above + j * e_tau <= p, i.e. j <= p_rem / e_tau
        uint64_t c_p = e_tau ? p_rem / e_tau + 1 : (uint64_t)c_tau;
        c_p = c_p < c_tau ? c_p : c_tau;
        c_keep = (uint32_t)(c_p < k_rem ? c_p : k_rem);
    }
    const uint64_t s_kept = keep_all ? z_total : s_above + (uint64_t)c_keep * e_tau;
    if (tid == 0) {
        if (n_kept_out) n_kept_out[row_id] = keep_all ? vocab : (int32_t)(c_above + c_keep);
        if (kept_mass_out) kept_mass_out[row_id] = z_total ? (float)((double)s_kept / (double)z_total) : 0.f;
    }
    if (s_kept == 0) {  // no positive weight at all (NaN row / all-zero probabilities)
        if (tid == 0) out_tokens[row_id] = arg_max;
        return;
    }
    uint64_t target;
    {
        const double t = (double)__builtin_fminf(__builtin_fmaxf(u, 0.f), 1.f) * (double)s_kept;
        target = t >= (double)s_kept ? s_kept - 1 : (uint64_t)t;
        if (target >= s_kept) target = s_kept - 1;
    }

    // ---- inverse CDF in index order.  Pass A: kept mass and tie count of each wave's contiguous
    // segment; then the one wave whose segment holds `target` rescans it with prefix sums.
    const int seg_len = ((vocab + kSampleWaves - 1) / kSampleWaves + 255) / 256 * 256;
    const int seg_begin = min(wave * seg_len, vocab), seg_end = min(seg_begin + seg_len, vocab);
    {
        uint64_t strict = 0;
        uint32_t ties = 0;
        for (int base = seg_begin; base < seg_end; base += 256) {
            const int i0 = base + lane * 4;
            if (i0 < seg_end) {
                float v[4];
                const int n = load4<DT>(row, i0, vocab, vec, v);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float e = sample_e(sample_x(v[k], temperature, probs_mode), m, probs_mode);
                    const uint32_t key = __float_as_uint(e);
                    if (k < n) {
                        if (key > tau) strict += sample_fix(e);
                        ties += key == tau ? 1u : 0u;
                    }
                }
            }
        }
        strict = wave_sum_u64(strict);
        ties = (uint32_t)wave_sum_u64((uint64_t)ties);
        if (lane == 0) {
            sh.seg_strict[wave] = strict;
            sh.seg_ties[wave] = ties;
        }
    }
    __syncthreads();
    int wsel = -1;
    uint64_t run = 0;        // kept mass before the selected segment
    uint32_t ties_run = 0;   // ties before it
    {
        uint64_t cum = 0;
        uint32_t tb = 0;
#pragma unroll
        for (int w = 0; w < kSampleWaves; ++w) {
            const uint32_t st = sh.seg_ties[w];
            const uint64_t left = (uint64_t)c_keep > (uint64_t)tb ? (uint64_t)c_keep - tb : 0ull;
            const uint64_t kt = left < st ? left : (uint64_t)st;
            const uint64_t s = sh.seg_strict[w] + kt * e_tau;
            if (wsel < 0 && target < cum + s) {
                wsel = w;
                run = cum;
                ties_run = tb;
            }
            cum += s;
            tb += st;
        }
    }
    if (wsel < 0) {  // unreachable (the segment sums add up to s_kept); keep the launch well-defined
        if (tid == 0) out_tokens[row_id] = arg_max;
        return;
    }
    if (wave != wsel) return;
    for (int base = seg_begin; base < seg_end; base += 256) {
        const int i0 = base + lane * 4;
        float v[4];
        const int n = i0 < seg_end ? load4<DT>(row, i0, vocab, vec, v) : 0;
        uint64_t w4[4];
        bool tie4[4];
        uint32_t t_l = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float e = sample_e(sample_x(v[k], temperature, probs_mode), m, probs_mode);
            const uint32_t key = __float_as_uint(e);
            tie4[k] = k < n && key == tau;
            w4[k] = (k < n && key > tau) ? sample_fix(e) : 0ull;
            t_l += tie4[k] ? 1u : 0u;
        }
        // exclusive prefix of the tie counts over the lanes (index order = lane order)
        uint32_t t_incl = t_l;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)t_incl, off, 64);
            if (lane >= off) t_incl += o;
        }
        uint32_t rank = ties_run + t_incl - t_l;
        uint64_t s_l = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (tie4[k]) {
                if (rank < c_keep) w4[k] = e_tau;
                ++rank;
            }
            s_l += w4[k];
        }
        uint64_t s_incl = s_l;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint64_t o = shfl_up_u64(s_incl, off);
            if (lane >= off) s_incl += o;
        }
        const uint64_t hit = __builtin_amdgcn_ballot_w64(run + s_incl > target);
        if (hit) {
            const int first = __builtin_ctzll(hit);
            if (lane == first) {
                uint64_t acc = run + s_incl - s_l;
                int pick = 0;
                bool done = false;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    acc += w4[k];
                    if (!done && acc > target) {
                        pick = k;
                        done = true;
                    }
                }
                out_tokens[row_id] = (int64_t)(i0 + pick);
            }
            return;
        }
        run += (uint64_t)(((uint64_t)(uint32_t)__shfl((int)(s_incl >> 32), 63, 64) << 32) |
                          (uint32_t)__shfl((int)(uint32_t)s_incl, 63, 64));
        ties_run += (uint32_t)__shfl((int)t_incl, 63, 64);
    }
    if (lane == 0) out_tokens[row_id] = arg_max;  // unreachable, see above
}

// logits[row, token] -= penalty[row] once per occurrence of `token` in the row's generated tokens
// (executor.py:89-102: index_add_ with duplicates = frequency penalty), rows with penalty <= 0
// untouched (the reference's `> 0` test).  All addends of one address are equal, so the atomic
// order cannot change the result.
__global__ __launch_bounds__(256) void frequency_penalty_kernel(float* __restrict__ logits, int64_t row_stride,
                                                                int vocab, const int32_t* __restrict__ tokens,
                                                                const int32_t* __restrict__ offsets,
                                                                const float* __restrict__ penalties) {
    const int row = blockIdx.x;
    const float p = penalties[row];
    if (!(p > 0.f)) return;
    const int beg = offsets[row], end = offsets[row + 1];
    float* lr = logits + (int64_t)row * row_stride;
    for (int i = beg + threadIdx.x; i < end; i += blockDim.x) {
        const int t = tokens[i];
        if (t >= 0 && t < vocab) atomicAdd(lr + t, -p);
    }
}

}  // namespace chitu

extern "C" int chitu_hip_frequency_penalty(float* logits, int64_t row_stride, int64_t rows, int32_t vocab,
                                           const int32_t* tokens, const int32_t* offsets,
                                           const float* penalties, void* stream) {
    CHITU_REQUIRE(logits && tokens && offsets && penalties && rows >= 0 && vocab > 0 && row_stride >= vocab);
    if (rows == 0) return CHITU_OK;
    hipLaunchKernelGGL(chitu::frequency_penalty_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream,
                       logits, row_stride, (int)vocab, tokens, offsets, penalties);
    CHITU_RETURN_LAUNCH_STATUS();
}

extern "C" int chitu_hip_sample(const void* logits, int act_dtype, int64_t row_stride, int64_t rows,
                                int32_t vocab, const float* temperatures, const int32_t* top_ks,
                                const float* top_ps, const float* uniforms, int32_t probs_mode,
                                int64_t* out_tokens, int32_t* n_kept_out, float* kept_mass_out,
                                void* stream) {
    CHITU_REQUIRE(logits && out_tokens && rows >= 0 && vocab > 0 && row_stride >= vocab);
    CHITU_REQUIRE(act_dtype >= 0 && act_dtype <= 2);
    CHITU_REQUIRE(probs_mode == 0 || probs_mode == 1);
    // greedy for every row: top_ks == NULL; otherwise all four per-row arrays are required
    if (top_ks != nullptr) CHITU_REQUIRE(top_ps && uniforms && (probs_mode || temperatures));
    if (rows == 0) return CHITU_OK;
    const int use_candidates = chitu::debug_option(chitu::kOptSampleRadix) > 0 ? 0 : 1;  // test / A-B override: radix descent only
#define LAUNCH(DT)                                                                                          \
    hipLaunchKernelGGL(chitu::sample_kernel<DT>, dim3((unsigned)rows), dim3(chitu::kSampleThreads), 0,      \
                       (hipStream_t)stream, logits, row_stride, (int)vocab, temperatures, top_ks, top_ps,   \
                       uniforms, (int)probs_mode, out_tokens, n_kept_out, kept_mass_out, use_candidates)
    if (act_dtype == 0) LAUNCH(0);
    else if (act_dtype == 1) LAUNCH(1);
    else LAUNCH(2);
#undef LAUNCH
    CHITU_RETURN_LAUNCH_STATUS();
}
