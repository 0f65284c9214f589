// In-graph tensor-parallel collectives over xGMI: one-shot all-reduce fused with what surrounds it on
// the decode step, and the logits all-gather.  No library call, no host round trip: plain kernel
// launches, so a decode step with its 2 x layers + 2 collectives is ONE hipGraph.
//
// Replaces (reference, read-only):
//   chitu/tensor_parallel.py:157-169        RowParallelLinear.forward -> dist.all_reduce (after wo / w2)
//   chitu/models/model_deepseek_v3.py:1010-1011  MoE: y + z, dist.all_reduce
//   chitu/tensor_parallel.py:94-102         ColumnParallelLinear gather_output -> all_gather_into_tensor
//   and the launches that sit around them on the decode step: the fused MoE's top-k sum
//   (fused_moe.py:1299-1305) in front, the residual add + RMSNorm (model_deepseek_v3.py:1107-1113,
//   models/model.py:29-78) + act-quant of the next fp8 linear (model_deepseek_v3.py:98-100) behind.
//
// Transport.  Every rank owns one UNCACHED device allocation (hipDeviceMallocUncached: MTYPE_UC, never
// held in an L2, so a peer's write that lands in HBM over xGMI is what the next load returns) mapped into
// every peer (hipIpcMemHandle across processes; a raw pointer for ranks that share a process).  The
// buffers have one layout:  flags[kind][src][slot] u32 | data_ar[parity][src][row][max_dim] bf16 |
// data_ag[parity][src][bytes].  A rank PUSHES: it writes its contribution into ITS slot of every peer's
// buffer (16-byte write-through stores, sc0 sc1), every storing wave drains (s_waitcnt vmcnt(0)), the
// workgroup meets, then one lane per peer stores the flag (system-scope relaxed atomic) -- the hand-off
// recipe of the CDNA4 guide (Guideline 16, R1) at system scope.  The receiver polls ITS OWN memory (one
// lane per source, relaxed system-scope loads, s_sleep between polls), then reads the slots with sc0 sc1
// loads and reduces in RANK ORDER in fp32 with one rounding: every rank computes bit-identical results,
// whatever the arrival order.  xGMI is point-to-point, a push is one hop, and the messages of a decode
// step are 14 KB (bs 1) to 458 KB (bs 32) per rank: one-shot (every rank sends its whole row block to
// all peers) beats a two-hop reduce-scatter + all-gather until the per-link bytes dominate the extra hop; from
// `two_shot_bytes` (256 KB per rank by default) on, the SAME launch runs the two-shot form: a row's chunks are dealt to
// the ranks in slices, hop 1 sends each chunk to its owner only, the owner reduces in rank order and hop 2 sends the
// reduced chunk to everybody -- 2/world of the bytes per link, one more flag hop, bit-identical results.
//
// Replay safety.  Kernel arguments are frozen in a hipGraph, so the call counter lives in device memory:
// epoch (local, ordinary memory) counts calls -- per ROW for the all-reduce (a row's data slot has a fixed address),
// per CALL for the all-gather (its data layout depends on the call's shape, so the parity must flip for the whole
// buffer at once); the flag value of a call is its epoch, the data slot is epoch & 1.  A rank
// can only be one call ahead of a peer on a slot (it needs the peer's flag of call c to finish call c), so
// a flag is awaited as `flag - epoch >= 0` and two data slots suffice: a writer of call c + 2 has seen
// every peer's flag of call c + 1, which a peer stores in a kernel that starts after its kernel of call c
// has finished reading.  All ranks must issue the same sequence of calls (they replay the same graphs).
//
// Failure.  Every spin is bounded (wall clock, `timeout_ms` at creation); a timeout sets a sticky error
// word that later waits test first, so a missing peer costs one timeout per process, not one per launch.
#include "common.h"
#include "norm_common.h"
#include <string.h>
#include <mutex>
#include <utility>
#include <vector>

namespace chitu {

constexpr int kCommMaxRanks = 8;
constexpr int kCommThreads = kNormWideThreads;  // one 8-element chunk per thread, the wide row form of norm_common.h
constexpr int kCommMaxTerms = kNormMaxTerms;
constexpr int kCommAux = 17;  // sc0 | sc1: system scope, write-through / L2 bypass

struct CommPeers {
    char* buf[kCommMaxRanks];
};

struct CommGeom {
    int rank, world, max_rows, max_dim, max_blocks;
    int64_t flags_ar, flags_ag, data_ar, data_ag, ag_bytes;
    int64_t flags_ar2, data_ar2;            // second hop of the two-shot all-reduce: flags[src][row], data[parity][row][max_dim]
    uint32_t data_ar_bytes, data_ag_bytes, data_ar2_bytes;  // whole regions (buffer descriptors)
    uint64_t timeout_ticks;
    uint32_t* host_err;  // pinned host word (device-visible): set together with the device error word, so the host can
                         // notice a timed-out collective between two steps WITHOUT synchronising the stream
};

struct Comm {
    CommGeom g;
    CommPeers peers;
    bool ipc_opened[kCommMaxRanks];
    int64_t two_shot_bytes;  // all-reduces of at least this many bytes per rank take the two-shot form (host-side choice)
    uint32_t* state;  // device, ordinary memory: epoch_ar[max_rows] | epoch_ag (word 0: the all-gather CALL counter) [max_blocks] | err[4]
    int64_t total;
    int64_t buf_bytes;  // size of this rank's exchange allocation (>= total when a larger parked buffer was reused)
};

__device__ __forceinline__ uint32_t sys_load(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void sys_store(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One lane: wait until *flag has reached `epoch` (peers may already be one call ahead).  Bounded.
__device__ __forceinline__ void wait_flag(const uint32_t* flag, uint32_t epoch, uint32_t* err, uint64_t timeout,
                                          uint32_t* host_err) {
    if ((int32_t)(sys_load(flag) - epoch) >= 0) return;
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;  // sticky
    const uint64_t t0 = wall_clock64();
    for (unsigned spins = 0;; ++spins) {
        if ((int32_t)(sys_load(flag) - epoch) >= 0) return;
        __builtin_amdgcn_s_sleep(2);
        if ((spins & 63) == 63) {
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
            if (wall_clock64() - t0 > timeout) {
                __hip_atomic_fetch_or(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sys_store(host_err, 1u);  // the copy the host polls between steps (XgmiComm.poll_error)
                return;
            }
        }
    }
}

// Every storing wave drains its write-through stores, the workgroup meets, one lane per peer raises this
// rank's flag there ...
__device__ __forceinline__ void signal_peers(const CommPeers& peers, const CommGeom& g, int64_t flags_off, int slots,
                                             int slot, uint32_t epoch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int tid = threadIdx.x;
    if (tid < g.world && tid != g.rank)
        sys_store(reinterpret_cast<uint32_t*>(peers.buf[tid] + flags_off) + (int64_t)g.rank * slots + slot, epoch);
}
// ... and one lane per source waits for that source's flag here, then the workgroup meets again.
__device__ __forceinline__ void await_peers(const CommPeers& peers, const CommGeom& g, int64_t flags_off, int slots,
                                            int slot, uint32_t epoch, uint32_t* err) {
    const int tid = threadIdx.x;
    if (tid < g.world && tid != g.rank)
        wait_flag(reinterpret_cast<const uint32_t*>(peers.buf[g.rank] + flags_off) + (int64_t)tid * slots + slot, epoch, err,
                  g.timeout_ticks, g.host_err);
    __syncthreads();
}

// `phase` of the two collectives: 0 = the whole collective in this launch; 1 = CONTRIBUTE only (push this rank's
// data, raise its flags, return); 2 = COMPLETE only (wait for the peers, reduce / gather, advance the epoch), to
// be issued after a phase-1 launch with the same arguments.  Split phases let a caller put work between the push
// and the wait, and let every rank of a test live on one stream: all contributions first, then all completions,
// no two kernels ever waiting for each other.

// ---------------------------------------------------------------------------------------------
// all-reduce (+ top-k sum in front, + residual add, RMSNorm, fp8 quant behind).  One workgroup per row,
// one 8-element chunk per thread (dim <= 8192): partial terms, residual and norm weight are requested
// up front, so a row costs one local memory round trip, one push and one hand-off.
//   part_r = bf16(sum_k float(part[row, k, :]))            (chitu_hip_moe_sum's rounding; terms == 1: as is)
//   a      = bf16(sum_r float(part_r)), r = 0 .. world-1   (rank order, fp32, one rounding)
//   v      = x ? bf16(x + a) : a          -> sum_out       (the reference's `x = x + attn(...)` in bf16)
//   y      = w ? rmsnorm(v) * w : -       -> y / q, qs     (chitu_hip_rmsnorm's arithmetic and quant modes)
template <int QMODE>
__global__ __launch_bounds__(kCommThreads) void allreduce_rmsnorm_kernel(
    CommPeers peers, CommGeom g, uint32_t* state, const bf16_t* part, int64_t part_stride, int terms,
    int64_t term_stride, const bf16_t* x, int64_t x_stride, bf16_t* sum_out, int64_t sum_stride,
    const bf16_t* __restrict__ w, bf16_t* y, int64_t y_stride, fp8_t* __restrict__ q, float* __restrict__ qs, int dim,
    float eps, float qeps, int phase, int two_shot, int tile_major) {
    __shared__ float red[kCommThreads / 64];
    const int row = blockIdx.x, tid = threadIdx.x;
    const int n_chunks = dim >> 3;
    const bool act = tid < n_chunks;
    const int c = min(tid, n_chunks - 1);
    uint32_t* epoch_ar = state;
    uint32_t* err = state + g.max_rows + g.max_blocks;
    const uint32_t epoch = epoch_ar[row] + 1;
    const int parity = (int)(epoch & 1u);

    i32x4 traw[kCommMaxTerms];
#pragma unroll
    for (int k = 0; k < kCommMaxTerms; ++k)
        traw[k] = *reinterpret_cast<const i32x4*>(part + (int64_t)row * part_stride + (int64_t)min(k, terms - 1) * term_stride + c * 8);
    i32x4 xraw = {0, 0, 0, 0}, wraw = {0, 0, 0, 0};
    if (x) xraw = *reinterpret_cast<const i32x4*>(x + (int64_t)row * x_stride + c * 8);
    if (w) wraw = *reinterpret_cast<const i32x4*>(w + c * 8);

    const i32x4 mine = terms == 1 ? traw[0] : sum_terms_bf16x8<kCommMaxTerms>(traw, terms);

    const uint32_t slot_off = (uint32_t)(((((int64_t)parity * kCommMaxRanks + g.rank) * g.max_rows + row) * g.max_dim + c * 8) * 2);
    i32x4 sraw;  // the all-reduced chunk (one rounding)
    auto reduce_rank_order = [&](const i32x4 (&theirs)[kCommMaxRanks]) {
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < kCommMaxRanks; ++s) {
            if (s < g.world) {
                const i32x4 t = (s == g.rank) ? mine : theirs[s];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t u = (uint32_t)t[i];
                    a[2 * i] += __uint_as_float(u << 16);
                    a[2 * i + 1] += __uint_as_float(u & 0xffff0000u);
                }
            }
        }
        i32x4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = (int)f32x2_to_bf16x2(a[2 * i], a[2 * i + 1]);
        return r;
    };
    auto load_contributions = [&](i32x4 (&theirs)[kCommMaxRanks]) {
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(peers.buf[g.rank] + g.data_ar, 0, g.data_ar_bytes, 0x00020000);
#pragma unroll
        for (int s = 0; s < kCommMaxRanks; ++s) {
            const int src = min(s, g.world - 1);
            const uint32_t off = (uint32_t)(((((int64_t)parity * kCommMaxRanks + src) * g.max_rows + row) * g.max_dim + c * 8) * 2);
            theirs[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, kCommAux);
        }
    };
    if (!two_shot) {
        // ---- ONE-SHOT: push this rank's whole row into its slot of every peer, wait, reduce everything locally
        if (phase != 2) {
#pragma unroll
            for (int p = 0; p < kCommMaxRanks; ++p) {
                if (p < g.world && p != g.rank && act) {
                    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(peers.buf[p] + g.data_ar, 0, g.data_ar_bytes, 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b128(mine, rsrc, slot_off, 0, kCommAux);
                }
            }
            signal_peers(peers, g, g.flags_ar, g.max_rows, row, epoch);
            if (phase == 1) return;
        }
        await_peers(peers, g, g.flags_ar, g.max_rows, row, epoch, err);
        i32x4 theirs[kCommMaxRanks];
        load_contributions(theirs);
        sraw = reduce_rank_order(theirs);
    } else {
        // ---- TWO-SHOT (reduce-scatter + all-gather inside the launch): the row's chunks are dealt to the ranks in
        // `world` contiguous slices; hop 1 sends every chunk only to the rank that owns its slice, the owner reduces it
        // in rank order (the SAME arithmetic as the one-shot form: results are bit-identical) and hop 2 sends the
        // reduced chunk to everybody.  Per link 2 * bytes / world instead of bytes: for messages whose transfer time,
        // not the flag hop, is what a collective costs.  phase 1 = hop 1 only, 3 = wait + reduce + hop 2 only,
        // 2 = wait + assemble + finish (split phases: all ranks on one stream in the tests).
        const int cps = n_chunks / g.world;  // chunks per slice (the launcher checked divisibility)
        const int owner = min(c / cps, g.world - 1);
        const bool own = act && owner == g.rank;
        const uint32_t off2 = (uint32_t)((((int64_t)parity * g.max_rows + row) * g.max_dim + c * 8) * 2);
        if (phase == 0 || phase == 1) {
            if (act && owner != g.rank) {
                const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(peers.buf[owner] + g.data_ar, 0, g.data_ar_bytes, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(mine, rsrc, slot_off, 0, kCommAux);
            }
            signal_peers(peers, g, g.flags_ar, g.max_rows, row, epoch);
            if (phase == 1) return;
        }
        i32x4 reduced = {0, 0, 0, 0};
        if (phase == 0 || phase == 3) {
            await_peers(peers, g, g.flags_ar, g.max_rows, row, epoch, err);
            if (own) {
                i32x4 theirs[kCommMaxRanks];
                load_contributions(theirs);
                reduced = reduce_rank_order(theirs);
#pragma unroll
                for (int p = 0; p < kCommMaxRanks; ++p) {
                    // (the split-phase form also parks the owner's copy in its own buffer: registers do not survive a launch)
                    if (p < g.world && (p != g.rank || phase == 3)) {
                        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(peers.buf[p] + g.data_ar2, 0, g.data_ar2_bytes, 0x00020000);
                        __builtin_amdgcn_raw_buffer_store_b128(reduced, rsrc, off2, 0, kCommAux);
                    }
                }
            }
            signal_peers(peers, g, g.flags_ar2, g.max_rows, row, epoch);
            if (phase == 3) return;
        }
        await_peers(peers, g, g.flags_ar2, g.max_rows, row, epoch, err);
        {
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(peers.buf[g.rank] + g.data_ar2, 0, g.data_ar2_bytes, 0x00020000);
            const i32x4 got = __builtin_amdgcn_raw_buffer_load_b128(rsrc, act ? off2 : 0u, 0, kCommAux);
            sraw = (own && phase == 0) ? reduced : got;
        }
    }
    float v[8];
    if (x) add_bf16x8(xraw, sraw, v, sraw);
    else unpack_bf16x8(sraw, v);
    if (act && sum_out) *reinterpret_cast<i32x4*>(sum_out + (int64_t)row * sum_stride + tid * 8) = sraw;
    if (tid == 0) epoch_ar[row] = epoch;
    if (!w) return;  // uniform
    rmsnorm_wide_finish<QMODE>(v, act, row, wraw, y, y_stride, q, qs, dim, eps, qeps, red, tile_major);
}

// ---------------------------------------------------------------------------------------------
// all-gather along the last dimension: out[row, r * cols + j] = in_r[row, j] (the rank-major concat of
// tensor_parallel.py:94-102), bf16 in, bf16 or f32 out (the `.float()` of the logits, model.py:475, rides
// along).  grid (chunks per row, rows); a workgroup moves 1024 x 8 elements of one row to every peer.
template <bool OUT_F32>
__global__ __launch_bounds__(kCommThreads) void allgather_kernel(CommPeers peers, CommGeom g, uint32_t* state,
                                                                 const bf16_t* in, int64_t in_stride, int cols, void* out,
                                                                 int phase) {
    const int row = blockIdx.y, tid = threadIdx.x;
    const int blk = blockIdx.y * gridDim.x + blockIdx.x;
    const int col = (blockIdx.x * kCommThreads + tid) * 8;
    const bool act = col < cols;
    // ONE counter per comm for the all-gather calls: every workgroup of a call reads the same value (the previous
    // call's kernel has finished), so the data-slot parity alternates per CALL whatever the (rows, cols) of the calls
    // are -- with a per-workgroup counter, calls of different shapes could map different workgroups with different
    // parities onto overlapping bytes.  The flags stay per workgroup; a workgroup index an earlier call did not use
    // holds an older epoch, which `flag - epoch >= 0` treats as "not there yet".
    uint32_t* epoch_ag = state + g.max_rows;
    uint32_t* err = state + g.max_rows + g.max_blocks;
    const uint32_t epoch = epoch_ag[0] + 1;
    const int parity = (int)(epoch & 1u);
    i32x4 mine = {0, 0, 0, 0};
    if (act) mine = *reinterpret_cast<const i32x4*>(in + (int64_t)row * in_stride + col);
    const uint32_t in_slot = (uint32_t)(((int64_t)row * cols + col) * 2);
    if (phase != 2) {
#pragma unroll
        for (int p = 0; p < kCommMaxRanks; ++p) {
            if (p < g.world && p != g.rank && act) {
                const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(peers.buf[p] + g.data_ag, 0, g.data_ag_bytes, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(mine, rsrc, (uint32_t)(((int64_t)parity * kCommMaxRanks + g.rank) * g.ag_bytes) + in_slot, 0, kCommAux);
            }
        }
        signal_peers(peers, g, g.flags_ag, g.max_blocks, blk, epoch);
        if (phase == 1) return;
    }
    await_peers(peers, g, g.flags_ag, g.max_blocks, blk, epoch, err);
    i32x4 theirs[kCommMaxRanks];
    {
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(peers.buf[g.rank] + g.data_ag, 0, g.data_ag_bytes, 0x00020000);
#pragma unroll
        for (int s = 0; s < kCommMaxRanks; ++s) {
            const int src = min(s, g.world - 1);
            theirs[s] = __builtin_amdgcn_raw_buffer_load_b128(
                rsrc, act ? (uint32_t)(((int64_t)parity * kCommMaxRanks + src) * g.ag_bytes) + in_slot : 0u, 0, kCommAux);
        }
    }
    // advanced by the call's LAST workgroup to get here (ticket in word 1; the next call is a later kernel)
    if (tid == 0) {
        const uint32_t n = gridDim.x * gridDim.y;
        if (__hip_atomic_fetch_add(&epoch_ag[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n - 1) {
            __hip_atomic_store(&epoch_ag[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&epoch_ag[0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (!act) return;
    const int64_t out_row = (int64_t)row * g.world * cols;
#pragma unroll
    for (int s = 0; s < kCommMaxRanks; ++s) {
        if (s < g.world) {
            const i32x4 t = (s == g.rank) ? mine : theirs[s];
            const int64_t o = out_row + (int64_t)s * cols + col;
            if (OUT_F32) {
                f32x4 lo, hi;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    lo[2 * i] = __uint_as_float(((uint32_t)t[i]) << 16);
                    lo[2 * i + 1] = __uint_as_float(((uint32_t)t[i]) & 0xffff0000u);
                    hi[2 * i] = __uint_as_float(((uint32_t)t[2 + i]) << 16);
                    hi[2 * i + 1] = __uint_as_float(((uint32_t)t[2 + i]) & 0xffff0000u);
                }
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + o) = lo;
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + o + 4) = hi;
            } else {
                *reinterpret_cast<i32x4*>(reinterpret_cast<bf16_t*>(out) + o) = t;
            }
        }
    }
}

static int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

// ---- the exchange buffer's memory is UNCACHED, and uncached memory must not be recycled as ordinary memory -----------
// Measured on MI355X / ROCm 7.2 (round 4, profiles/r04_graph_mismatch_probe_suite_run2.txt): device memory
// that was allocated with hipDeviceMallocUncached, freed, and handed out again by hipMalloc as ordinary (cached) memory can
// serve STALE L2 lines on one XCD -- lines left from the ordinary life the range had BEFORE it was uncached; neither the
// uncached life nor the free evicts them, and a later reader on that XCD hits them (one eighth of a GEMM's workgroups read
// an old activation row; a 64 MB fill, which evicts every L2, cures it for good).  So:
//   * a buffer a Comm no longer needs is PARKED for the rest of the process and handed to the next Comm of that size
//     (a process makes one Comm; the tests make a hundred) -- uncached memory never goes back to the allocator;
//   * a FRESH uncached allocation is followed by one sweep of ordinary traffic larger than all L2s, so that no line of the
//     range's earlier life is left to be hit through the uncached mapping's neighbours either.
static std::mutex g_park_mutex;
static std::vector<std::pair<void*, int64_t>> g_parked;  // (uncached buffer, bytes)

static void* take_parked(int64_t bytes, int64_t* got_bytes) {
    std::lock_guard<std::mutex> lock(g_park_mutex);
    // the smallest parked buffer that is large enough (an exact-size match only would let create / destroy cycles of
    // differing sizes grow the parked set without bound)
    long best = -1;
    for (size_t i = 0; i < g_parked.size(); ++i)
        if (g_parked[i].second >= bytes && (best < 0 || g_parked[i].second < g_parked[(size_t)best].second)) best = (long)i;
    if (best < 0) return nullptr;
    void* p = g_parked[(size_t)best].first;
    *got_bytes = g_parked[(size_t)best].second;
    g_parked.erase(g_parked.begin() + best);
    return p;
}

static void park(void* p, int64_t bytes) {
    std::lock_guard<std::mutex> lock(g_park_mutex);
    g_parked.emplace_back(p, bytes);
}

static hipError_t sweep_all_l2(void) {
    // 8 x the 8 x 4 MB of L2 in one buffer; when the HBM is nearly full (KV cache sized to fill it) a smaller buffer written
    // several times does the same job -- the sweep must not be the allocation that fails communicator creation
    constexpr size_t kSweepBytes = (size_t)256 << 20;
    void* t = nullptr;
    size_t size = kSweepBytes;
    hipError_t e = hipMalloc(&t, size);
    while (e != hipSuccess && size > ((size_t)32 << 20)) {
        (void)hipGetLastError();
        size >>= 1;
        e = hipMalloc(&t, size);
    }
    if (e != hipSuccess) return e;
    for (size_t done = 0; done < kSweepBytes && e == hipSuccess; done += size) e = hipMemset(t, (int)(done / size) & 1, size);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    (void)hipFree(t);
    return e;
}

}  // namespace chitu

using namespace chitu;

extern "C" int chitu_hip_comm_create(int32_t rank, int32_t world, int32_t max_rows, int32_t max_dim,
                                     int64_t gather_bytes, int32_t timeout_ms, void** comm_out) {
    CHITU_REQUIRE(comm_out && world >= 1 && world <= kCommMaxRanks && rank >= 0 && rank < world);
    CHITU_REQUIRE(max_rows >= 1 && max_dim >= 8 && max_dim % 8 == 0 && max_dim <= kCommThreads * 8 && gather_bytes >= 0);
    CHITU_REQUIRE(timeout_ms >= 1);
    Comm* cm = new Comm();
    cm->state = nullptr;
    CommGeom& g = cm->g;
    g.host_err = nullptr;
    g.rank = rank, g.world = world, g.max_rows = max_rows, g.max_dim = max_dim;
    g.ag_bytes = align_up(gather_bytes, 256);
    g.max_blocks = (int)(g.ag_bytes / (kCommThreads * 16)) + max_rows + 2;  // a row's last chunk may be partial; >= 2 state words
    const int64_t ar = (int64_t)2 * kCommMaxRanks * max_rows * max_dim * 2, ag = (int64_t)2 * kCommMaxRanks * g.ag_bytes;
    const int64_t ar2 = (int64_t)2 * max_rows * max_dim * 2;
    if (ar >= (1ll << 32) || ag >= (1ll << 32)) {
        delete cm;
        return CHITU_ERR_UNSUPPORTED;
    }
    g.flags_ar = 0;
    g.flags_ag = align_up(g.flags_ar + (int64_t)kCommMaxRanks * max_rows * 4, 256);
    g.data_ar = align_up(g.flags_ag + (int64_t)kCommMaxRanks * g.max_blocks * 4, 4096);
    g.data_ag = align_up(g.data_ar + ar, 4096);
    g.data_ar_bytes = (uint32_t)ar, g.data_ag_bytes = (uint32_t)ag, g.data_ar2_bytes = (uint32_t)ar2;
    g.flags_ar2 = align_up(g.data_ag + ag, 4096);
    g.data_ar2 = align_up(g.flags_ar2 + (int64_t)kCommMaxRanks * max_rows * 4, 4096);
    cm->total = align_up(g.data_ar2 + ar2, 4096);
    cm->two_shot_bytes = 256 << 10;
    g.timeout_ticks = (uint64_t)timeout_ms * 100000ull;  // wall_clock64: 100 MHz
    for (int i = 0; i < kCommMaxRanks; ++i) cm->peers.buf[i] = nullptr, cm->ipc_opened[i] = false;
    cm->buf_bytes = cm->total;
    void* p = take_parked(cm->total, &cm->buf_bytes);
    hipError_t e = hipSuccess;
    if (!p) {
        e = hipExtMallocWithFlags(&p, (size_t)cm->total, hipDeviceMallocUncached);
        if (e == hipSuccess) e = sweep_all_l2();
    }
    if (e == hipSuccess) e = hipMemset(p, 0, (size_t)cm->total);
    const size_t state_bytes = ((size_t)max_rows + g.max_blocks + 4) * 4;
    if (e == hipSuccess) e = hipMalloc((void**)&cm->state, state_bytes);
    if (e == hipSuccess) e = hipMemset(cm->state, 0, state_bytes);
    if (e == hipSuccess) e = hipHostMalloc((void**)&g.host_err, 64, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) *g.host_err = 0;
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        if (p) park(p, cm->buf_bytes);
        if (cm->state) (void)hipFree(cm->state);
        if (g.host_err) (void)hipHostFree(g.host_err);
        delete cm;
        return (int)e;
    }
    cm->peers.buf[rank] = (char*)p;
    *comm_out = cm;
    return CHITU_OK;
}

extern "C" int chitu_hip_comm_ipc_handle(void* comm, void* handle_out_64) {
    CHITU_REQUIRE(comm && handle_out_64);
    Comm* cm = (Comm*)comm;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size is part of the ABI");
    hipIpcMemHandle_t h;
    const hipError_t e = hipIpcGetMemHandle(&h, cm->peers.buf[cm->g.rank]);
    if (e != hipSuccess) return (int)e;
    memcpy(handle_out_64, &h, 64);
    return CHITU_OK;
}

extern "C" int chitu_hip_comm_local_ptr(void* comm, void** ptr_out) {
    CHITU_REQUIRE(comm && ptr_out);
    Comm* cm = (Comm*)comm;
    *ptr_out = cm->peers.buf[cm->g.rank];
    return CHITU_OK;
}

extern "C" int chitu_hip_comm_set_peer(void* comm, int32_t peer, void* ptr) {
    CHITU_REQUIRE(comm && ptr);
    Comm* cm = (Comm*)comm;
    CHITU_REQUIRE(peer >= 0 && peer < cm->g.world && peer != cm->g.rank && !cm->peers.buf[peer]);
    cm->peers.buf[peer] = (char*)ptr;
    return CHITU_OK;
}

extern "C" int chitu_hip_comm_open_peer(void* comm, int32_t peer, const void* handle_64) {
    CHITU_REQUIRE(comm && handle_64);
    Comm* cm = (Comm*)comm;
    CHITU_REQUIRE(peer >= 0 && peer < cm->g.world && peer != cm->g.rank && !cm->peers.buf[peer]);
    hipIpcMemHandle_t h;
    memcpy(&h, handle_64, 64);
    void* p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return (int)e;
    cm->peers.buf[peer] = (char*)p;
    cm->ipc_opened[peer] = true;
    return CHITU_OK;
}

extern "C" int chitu_hip_comm_status(void* comm, uint32_t* err_out) {
    CHITU_REQUIRE(comm && err_out);
    Comm* cm = (Comm*)comm;
    const hipError_t e = hipMemcpy(err_out, cm->state + cm->g.max_rows + cm->g.max_blocks, 4, hipMemcpyDeviceToHost);
    return e == hipSuccess ? CHITU_OK : (int)e;
}

// All-reduces of at least `min_bytes` per rank (rows * dim * 2) take the two-shot form from now on (0 = always, a huge value
// = never; default 256 KB).  Host-side and graph-static: call it before capturing, with the same value on every rank.
extern "C" int chitu_hip_comm_set_two_shot(void* comm, int64_t min_bytes) {
    CHITU_REQUIRE(comm && min_bytes >= 0);
    ((Comm*)comm)->two_shot_bytes = min_bytes;
    return CHITU_OK;
}

// Non-blocking: the pinned host copy of the error word (set by the kernel that timed out, visible as soon as that
// store has crossed the bus).  No stream is synchronised; 0 only means "nothing reported yet".
extern "C" int chitu_hip_comm_poll_error(void* comm, uint32_t* err_out) {
    CHITU_REQUIRE(comm && err_out);
    Comm* cm = (Comm*)comm;
    *err_out = *(volatile uint32_t*)cm->g.host_err;
    return CHITU_OK;
}

extern "C" int chitu_hip_comm_destroy(void* comm) {
    CHITU_REQUIRE(comm);
    Comm* cm = (Comm*)comm;
    (void)hipDeviceSynchronize();
    for (int i = 0; i < kCommMaxRanks; ++i)
        if (cm->ipc_opened[i]) (void)hipIpcCloseMemHandle(cm->peers.buf[i]);
    park(cm->peers.buf[cm->g.rank], cm->buf_bytes);  // never back to the allocator: see "uncached memory must not be recycled"
    (void)hipFree(cm->state);
    (void)hipHostFree(cm->g.host_err);
    delete cm;
    return CHITU_OK;
}

static bool comm_ready(const Comm* cm) {
    for (int i = 0; i < cm->g.world; ++i)
        if (!cm->peers.buf[i]) return false;
    return true;
}

extern "C" int chitu_hip_comm_allreduce_rmsnorm(void* comm, const void* part_bf16, int64_t part_row_stride,
                                                int32_t terms, int64_t term_stride, const void* x_bf16,
                                                int64_t x_row_stride, void* sum_out_bf16, int64_t sum_row_stride,
                                                const void* weight_bf16, void* y_bf16, int64_t y_row_stride,
                                                int64_t rows, int32_t dim, float eps, void* q_fp8, float* q_scales,
                                                int32_t quant_mode, float quant_eps, int32_t phase, void* stream) {
    CHITU_REQUIRE(comm && part_bf16 && rows >= 0 && dim >= 8 && dim % 8 == 0 && phase >= 0 && phase <= 3);
    Comm* cm = (Comm*)comm;
    CHITU_REQUIRE(comm_ready(cm));
    // two-shot (reduce-scatter + all-gather in the launch) from `two_shot_bytes` per rank on, when the row's 16-byte chunks
    // deal evenly to the ranks; bit-identical to the one-shot form, so the choice is pure speed (chitu_hip_comm_set_two_shot)
    const int two_shot = cm->g.world >= 2 && (dim / 8) % cm->g.world == 0 && rows * (int64_t)dim * 2 >= cm->two_shot_bytes ? 1 : 0;
    CHITU_REQUIRE(phase != 3 || two_shot);
    if (rows > cm->g.max_rows || dim > cm->g.max_dim) return CHITU_ERR_UNSUPPORTED;
    CHITU_REQUIRE(terms >= 1 && part_row_stride % 8 == 0 && (terms == 1 || term_stride % 8 == 0));
    if (terms > kCommMaxTerms) return CHITU_ERR_UNSUPPORTED;
    CHITU_REQUIRE(!x_bf16 || x_row_stride % 8 == 0);
    CHITU_REQUIRE(!sum_out_bf16 || sum_row_stride % 8 == 0);
    CHITU_REQUIRE(sum_out_bf16 || weight_bf16);
    const int tile_major = (quant_mode & 4) ? 1 : 0;  // quant_mode + 4: tile-major codes and scales, as chitu_hip_rmsnorm
    quant_mode &= 3;
    CHITU_REQUIRE(!tile_major || quant_mode != 0);
    if (weight_bf16) {
        CHITU_REQUIRE(y_bf16 || quant_mode != 0);
        CHITU_REQUIRE(!y_bf16 || y_row_stride % 8 == 0);
        if (quant_mode != 0) {
            CHITU_REQUIRE(q_fp8 && q_scales && (quant_mode == 1 || quant_mode == 2));
            if (dim % 128 != 0) return CHITU_ERR_UNSUPPORTED;
        }
    } else {
        CHITU_REQUIRE(quant_mode == 0 && !y_bf16);
    }
    if (rows == 0) return CHITU_OK;
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(QM)                                                                                                  \
    hipLaunchKernelGGL(allreduce_rmsnorm_kernel<QM>, dim3((unsigned)rows), dim3(kCommThreads), 0, st, cm->peers,    \
                       cm->g, cm->state, (const bf16_t*)part_bf16, part_row_stride, (int)terms, term_stride,        \
                       (const bf16_t*)x_bf16, x_row_stride, (bf16_t*)sum_out_bf16, sum_row_stride,                  \
                       (const bf16_t*)weight_bf16, (bf16_t*)y_bf16, y_row_stride, (fp8_t*)q_fp8, q_scales, (int)dim, \
                       eps, quant_eps, (int)phase, two_shot, tile_major)
    if (quant_mode == 0) LAUNCH(0);
    else if (quant_mode == 1) LAUNCH(1);
    else LAUNCH(2);
#undef LAUNCH
    CHITU_RETURN_LAUNCH_STATUS();
}

extern "C" int chitu_hip_comm_all_gather(void* comm, const void* in_bf16, int64_t in_row_stride, int64_t rows,
                                         int64_t cols, void* out, int32_t out_dtype, int32_t phase, void* stream) {
    CHITU_REQUIRE(comm && in_bf16 && out && rows >= 0 && cols >= 8 && phase >= 0 && phase <= 2);
    Comm* cm = (Comm*)comm;
    CHITU_REQUIRE(comm_ready(cm));
    CHITU_REQUIRE(out_dtype == 0 || out_dtype == 2);
    if (cols % 8 != 0 || in_row_stride % 8 != 0) return CHITU_ERR_UNSUPPORTED;
    const int64_t chunks = (cols + kCommThreads * 8 - 1) / (kCommThreads * 8);
    if (rows * cols * 2 > cm->g.ag_bytes || rows * chunks > cm->g.max_blocks || rows > 65535) return CHITU_ERR_UNSUPPORTED;
    if (rows == 0) return CHITU_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)chunks, (unsigned)rows);
    if (out_dtype == 2)
        hipLaunchKernelGGL(allgather_kernel<true>, grid, dim3(kCommThreads), 0, st, cm->peers, cm->g, cm->state,
                           (const bf16_t*)in_bf16, in_row_stride, (int)cols, out, (int)phase);
    else
        hipLaunchKernelGGL(allgather_kernel<false>, grid, dim3(kCommThreads), 0, st, cm->peers, cm->g, cm->state,
                           (const bf16_t*)in_bf16, in_row_stride, (int)cols, out, (int)phase);
    CHITU_RETURN_LAUNCH_STATUS();
}
