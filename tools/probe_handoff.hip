// Probe (round 3): what would a flag-chained persistent launch for the attention front end
// (wqkv_a -> [q_norm + wq_b | kv append] -> W_UK absorb) buy over the three graph-captured launches it replaces?
//
// Part A -- grid barriers across co-resident workgroups, three forms next to tools/probe_gridbar.hip's naive one
//   (every workgroup fetch_add + ACQUIRE-load spin on one word: 9.3 us at 256 workgroups):
//     A1  one counter, relaxed sc1 polls + s_sleep, ONE release fence before the arrive, ONE acquire fence after
//     A2  XCD-hierarchical: per-XCC counter, the XCC's last arriver goes to a root counter, the root's last arriver
//         bumps a generation word that everybody polls relaxed; one acquire fence on exit
//     A3  no barrier at all: a producer -> consumer FLAG (fan-in counter): consumers already resident
// Part B -- the pipeline itself with stand-in bodies of the real shapes (bs 16, R1 rank shard):
//     stage 1: 132 workgroups x 512 threads, 112 KB of weights each, reads 112 KB of activations (L2), writes 512 B
//     stage 2: 192 workgroups, 24 KB of weights each, reads ALL of stage 1's output (48 KB), writes 512 B
//     stage 3:  64 workgroups, 16 KB of weights each, reads 4 KB of stage 2's output, writes 4 KB
//   B0  three kernels per layer in a hipGraph (what the step does today)
//   B1  ONE launch per layer: stage k+1's workgroups issue their weight loads, THEN wait for stage k's fan-in counter
//       (relaxed polls by one lane, one agent acquire, plain loads); producers publish with write-through stores +
//       vmcnt(0) + one relaxed agent add (the CDNA4 guide's R1 hand-off in its counter form)
//   B2  as B1 but the weights are requested only AFTER the wait (prices the prefetch alone)
//   Weights advance through a 1.3 GB buffer from layer to layer (HBM-cold, like the step); every hand-off is
//   CHECKED (values depend on the layer), mismatches and timeouts are reported.  Every spin is bounded.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned gu32;

__device__ __forceinline__ unsigned rlx_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
    return x & 7u;
}
__device__ __forceinline__ bool spin_ge(const unsigned* p, unsigned target, int* err) {
    for (int spins = 0; (int)(rlx_load(p) - target) < 0; ++spins) {
        __builtin_amdgcn_s_sleep(1);
        if (spins > 2000000) { *err = 1; return false; }
    }
    return true;
}

// ------------------------------------------------------------------------------------------ part A
// A1: counter barrier, relaxed polls, one release + one acquire per workgroup
__device__ __forceinline__ bool barrier_counter(unsigned* counter, unsigned target, int* err) {
    __shared__ int ok_b;
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = spin_ge(counter, target, err);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        ok_b = ok;
    }
    __syncthreads();
    return ok_b != 0;
}
// A2: per-XCC counters (64-byte apart), root counter, generation word.  round r (1-based); n_xcc[x] = workgroups
// seen on XCC x (counted by a census kernel first: placement is observed, never assumed).
__device__ __forceinline__ bool barrier_xcd(unsigned* st, const unsigned* n_xcc, unsigned n_active_xcc, unsigned r, int* err) {
    __shared__ int ok_x;
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const unsigned x = xcc_id();
        unsigned* cx = st + 16 * (1 + x);
        unsigned* root = st + 16 * 9;
        unsigned* gen = st + 16 * 10;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned a = __hip_atomic_fetch_add(cx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a + 1 == r * n_xcc[x]) {  // this XCC's last arriver
            const unsigned b = __hip_atomic_fetch_add(root, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (b + 1 == r * n_active_xcc) __hip_atomic_store(gen, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ok = spin_ge(gen, r, err);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        ok_x = ok;
    }
    __syncthreads();
    return ok_x != 0;
}

__global__ void census_kernel(unsigned* n_xcc) {
    if (threadIdx.x == 0) atomicAdd(&n_xcc[xcc_id()], 1u);
}

template <int KIND>
__global__ __launch_bounds__(256) void barrier_kernel(unsigned* st, const unsigned* n_xcc, unsigned n_active, int rounds,
                                                      uint32_t* buf, int words_per_wg, int* err, unsigned long long* mism) {
    const unsigned G = gridDim.x;
    unsigned long long bad = 0;
    for (int r = 1; r <= rounds; ++r) {
        if (words_per_wg) {
            uint32_t* mine = buf + (size_t)blockIdx.x * words_per_wg;
            for (int i = threadIdx.x; i < words_per_wg; i += 256) mine[i] = (uint32_t)r * 1000003u + i;
        }
        bool ok = KIND == 1 ? barrier_counter(st, (unsigned)(2 * r - 1) * G, err) : barrier_xcd(st, n_xcc, n_active, 2 * r - 1, err);
        if (!ok) return;
        if (words_per_wg) {
            const uint32_t* other = buf + (size_t)((blockIdx.x + 37) % G) * words_per_wg;
            for (int i = threadIdx.x; i < words_per_wg; i += 256) bad += other[i] != (uint32_t)r * 1000003u + i;
        }
        ok = KIND == 1 ? barrier_counter(st, (unsigned)(2 * r) * G, err) : barrier_xcd(st, n_xcc, n_active, 2 * r, err);
        if (!ok) return;
    }
    if (bad) atomicAdd(mism, bad);
}

// A3: flag hand-off alone.  Producers (first half of the grid) publish 512 B each (sc1 stores, drain, relaxed add on
// one fan-in counter); consumers (second half) poll the counter, acquire, read one producer's 512 B and check it; then
// the roles' direction reverses through a second counter so that the next round can start: a round = two fan-in hops.
__global__ __launch_bounds__(256) void flag_kernel(unsigned* st, int rounds, uint32_t* buf, int* err, unsigned long long* mism) {
    const unsigned half = gridDim.x / 2;
    const bool producer = blockIdx.x < half;
    const unsigned me = producer ? blockIdx.x : blockIdx.x - half;
    unsigned* c_fwd = st;
    unsigned* c_back = st + 16;
    __shared__ int ok_f;
    unsigned long long bad = 0;
    for (int r = 1; r <= rounds; ++r) {
        for (int dir = 0; dir < 2; ++dir) {
            const bool writes = (dir == 0) == producer;
            unsigned* cnt = dir == 0 ? c_fwd : c_back;
            uint32_t* slab = buf + ((size_t)dir * half + me) * 128;
            if (writes) {
                if (threadIdx.x < 32) {
                    const i32x4 v = {(int)(r * 977u + me), (int)threadIdx.x, r, dir};
                    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(slab + threadIdx.x * 4), "v"(v) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                if (threadIdx.x == 0) {
                    ok_f = spin_ge(cnt, (unsigned)r * half, err);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                __syncthreads();
                if (!ok_f) return;
                const uint32_t* theirs = buf + ((size_t)dir * half + (me + 37) % half) * 128;
                if (threadIdx.x < 32) {
                    const i32x4 v = *reinterpret_cast<const i32x4*>(theirs + threadIdx.x * 4);
                    bad += v[0] != (int)(r * 977u + (me + 37) % half) || v[2] != r;
                }
            }
        }
    }
    if (bad) atomicAdd(mism, bad);
}

// ------------------------------------------------------------------------------------------ part B
constexpr int kThreads = 512;
struct Stage { int wgs, w16, in16, out16; };  // workgroups; 16-B weight loads per thread; 16-B input loads per thread; 16-B output chunks per wg
// bs 16: wqkv_a 132 x 16 rows x 7168 B | wq_b 192 x 16 x 1536 | w_uk 64 x 16 KB
constexpr Stage kS1 = {132, 14, 14, 32};   // 112 KB weights, 112 KB activations (fp8 x, L2), 512 B out
constexpr Stage kS2 = {192, 3, 6, 32};     // 24 KB weights, 48 KB = all of stage 1's bf16 output, 512 B out
constexpr Stage kS3 = {64, 2, 1, 256};     // 16 KB weights, 4 KB of stage 2's output (<= 256 threads load), 4 KB out

struct LayerPtrs {
    const i32x4* w1; const i32x4* w2; const i32x4* w3;  // this layer's weights
    const i32x4* x;        // stage 1's input (written before the launch)
    i32x4* o1; i32x4* o2; i32x4* o3;
    unsigned* st;          // words [0]: epoch, [16]: ticket, [32]: cnt1 (+ per-XCC leader / go words from +64), [512]: cnt2 (likewise)
    int* err; unsigned long long* mism;
    unsigned tag;          // layer tag folded into every output word (checked by the consumer)
};

template <int NW>
__device__ __forceinline__ void load_w(i32x4 (&w)[NW], const i32x4* base, int wg) {
#pragma unroll
    for (int i = 0; i < NW; ++i) w[i] = __builtin_nontemporal_load(base + ((size_t)wg * NW + i) * kThreads + threadIdx.x);
}
template <int NW>
__device__ __forceinline__ int fold_w(const i32x4 (&w)[NW]) {
    int a = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) a ^= w[i][0] ^ w[i][1] ^ w[i][2] ^ w[i][3];
    return a;  // the buffers hold zeros, so this is 0 at run time -- but data-dependent, so the loads stay
}
__device__ __forceinline__ void store_out(i32x4* out, int wg, int out16, unsigned tag, int wfold, bool write_through) {
    if ((int)threadIdx.x < out16) {
        const i32x4 v = {(int)tag + wfold, wg, (int)threadIdx.x, (int)tag};
        i32x4* dst = out + (size_t)wg * out16 + threadIdx.x;
        if (write_through) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
        else *dst = v;
    }
}
// consumer side check: every 16-B chunk of the producer's output carries (tag, wg, idx, tag)
template <int N16>
__device__ __forceinline__ int check_in(const i32x4* in, int chunks_total, unsigned tag, int per_wg16) {
    int bad = 0;
    i32x4 v[N16];
#pragma unroll
    for (int i = 0; i < N16; ++i) v[i] = in[(i * kThreads + threadIdx.x) % chunks_total];
#pragma unroll
    for (int i = 0; i < N16; ++i) {
        const int c = (i * kThreads + threadIdx.x) % chunks_total;
        bad += v[i][0] != (int)tag || v[i][3] != (int)tag || v[i][1] != c / per_wg16 || v[i][2] != c % per_wg16;
    }
    return bad;
}

// B0: the three stages as three kernels
template <int S>
__global__ __launch_bounds__(kThreads) void stage_kernel(LayerPtrs p) {
    const int wg = blockIdx.x;
    int bad = 0, f = 0;
    if (S == 1) {
        i32x4 w[kS1.w16]; load_w(w, p.w1, wg);
        i32x4 a[kS1.in16];
#pragma unroll
        for (int i = 0; i < kS1.in16; ++i) a[i] = p.x[i * kThreads + threadIdx.x];
        f = fold_w(w) ^ fold_w(a);
        store_out(p.o1, wg, kS1.out16, p.tag, f, false);
    } else if (S == 2) {
        i32x4 w[kS2.w16]; load_w(w, p.w2, wg);
        bad = check_in<kS2.in16>(p.o1, kS1.wgs * kS1.out16, p.tag, kS1.out16);
        f = fold_w(w);
        store_out(p.o2, wg, kS2.out16, p.tag, f, false);
    } else {
        i32x4 w[kS3.w16]; load_w(w, p.w3, wg);
        if (threadIdx.x < 256) {
            const i32x4 v = p.o2[((size_t)(wg % 24) * 8 * kS2.out16 + threadIdx.x) % (kS2.wgs * kS2.out16)];
            bad += v[0] != (int)p.tag;
        }
        f = fold_w(w);
        store_out(p.o3, wg, kS3.out16, p.tag, f, false);
    }
    if (bad) atomicAdd(p.mism, (unsigned long long)bad);
}

// B1 / B2: one launch; PREFETCH = weights requested before the dependency wait.
// B3 (HIER): as B1, but only ONE consumer workgroup per XCC polls the fan-in counter (the first of its XCC to arrive:
// CAS on a per-XCC "leader of this epoch" word); it then raises a per-XCC go word that the XCC's other consumers poll --
// a line their own L2 serves, so 8 pollers instead of 256 load the fabric while the producers stream.
template <bool PREFETCH, bool HIER>
__global__ __launch_bounds__(kThreads) void fused_kernel(LayerPtrs p) {
    const int b = blockIdx.x;
    unsigned* st = p.st;
    const unsigned epoch = st[0] + 1;  // same for every workgroup of the launch (the previous launch has finished)
    __shared__ int ok_s;
    int bad = 0, f = 0;
    auto publish = [&](unsigned* cnt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto await = [&](unsigned* cnt, unsigned n) -> bool {
        if (threadIdx.x == 0) {
            if (HIER) {
                const unsigned x = xcc_id();
                unsigned* lead = cnt + 64 + 32 * x;  // 128-byte apart per XCC
                unsigned* go = lead + 16;
                const unsigned old = atomicCAS(lead, epoch - 1, epoch);
                if (old == epoch - 1) {  // this XCC's poller
                    ok_s = spin_ge(cnt, epoch * n, p.err);
                    __hip_atomic_store(go, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    int ok = 1;
                    for (int spins = 0; (int)(rlx_load(go) - epoch) < 0; ++spins) {
                        __builtin_amdgcn_s_sleep(2);
                        if (spins > 2000000) { *p.err = 2; ok = 0; break; }
                    }
                    ok_s = ok;
                }
            } else {
                ok_s = spin_ge(cnt, epoch * n, p.err);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        return ok_s != 0;
    };
    if (b < kS1.wgs) {
        const int wg = b;
        i32x4 w[kS1.w16]; load_w(w, p.w1, wg);
        i32x4 a[kS1.in16];
#pragma unroll
        for (int i = 0; i < kS1.in16; ++i) a[i] = p.x[i * kThreads + threadIdx.x];
        f = fold_w(w) ^ fold_w(a);
        store_out(p.o1, wg, kS1.out16, p.tag, f, true);
        publish(st + 32);
    } else if (b < kS1.wgs + kS2.wgs) {
        const int wg = b - kS1.wgs;
        i32x4 w[kS2.w16];
        if (PREFETCH) load_w(w, p.w2, wg);
        if (!await(st + 32, kS1.wgs)) return;
        if (!PREFETCH) load_w(w, p.w2, wg);
        bad = check_in<kS2.in16>(p.o1, kS1.wgs * kS1.out16, p.tag, kS1.out16);
        f = fold_w(w);
        store_out(p.o2, wg, kS2.out16, p.tag, f, true);
        publish(st + 512);
    } else {
        const int wg = b - kS1.wgs - kS2.wgs;
        i32x4 w[kS3.w16];
        if (PREFETCH) load_w(w, p.w3, wg);
        if (!await(st + 512, kS2.wgs)) return;
        if (!PREFETCH) load_w(w, p.w3, wg);
        if (threadIdx.x < 256) {
            const i32x4 v = p.o2[((size_t)(wg % 24) * 8 * kS2.out16 + threadIdx.x) % (kS2.wgs * kS2.out16)];
            bad += v[0] != (int)p.tag;
        }
        f = fold_w(w);
        store_out(p.o3, wg, kS3.out16, p.tag, f, false);
    }
    if (bad) atomicAdd(p.mism, (unsigned long long)bad);
    // the launch's last workgroup advances the epoch (ticket; the next launch is a later kernel)
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned n = gridDim.x;
        if (__hip_atomic_fetch_add(st + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n - 1) {
            __hip_atomic_store(st + 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(st, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned* state; int* err; unsigned long long* mism; uint32_t* buf; unsigned* n_xcc;
    CK(hipMalloc(&state, 8192)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&mism, 8)); CK(hipMalloc(&buf, 64 << 20)); CK(hipMalloc(&n_xcc, 64));

    // ---- part A
    const int rounds = 2000;
    for (int G : {256, 512}) {
        CK(hipMemsetAsync(n_xcc, 0, 64, st));
        hipLaunchKernelGGL(census_kernel, dim3(G), dim3(256), 0, st, n_xcc);
        unsigned h_n[8]; CK(hipMemcpyAsync(h_n, n_xcc, 32, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
        unsigned active = 0; for (int i = 0; i < 8; ++i) active += h_n[i] != 0;
        printf("census G=%d: workgroups per XCC = %u %u %u %u %u %u %u %u\n", G, h_n[0], h_n[1], h_n[2], h_n[3], h_n[4], h_n[5], h_n[6], h_n[7]);
        for (int kind : {1, 2}) {
            for (int words : {0, 256, 4096}) {
                CK(hipMemsetAsync(state, 0, 8192, st)); CK(hipMemsetAsync(err, 0, 4, st)); CK(hipMemsetAsync(mism, 0, 8, st));
                CK(hipEventRecord(e0, st));
                if (kind == 1) hipLaunchKernelGGL(barrier_kernel<1>, dim3(G), dim3(256), 0, st, state, n_xcc, active, rounds, buf, words, err, mism);
                else hipLaunchKernelGGL(barrier_kernel<2>, dim3(G), dim3(256), 0, st, state, n_xcc, active, rounds, buf, words, err, mism);
                CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                int herr; unsigned long long hm; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hm, mism, 8, hipMemcpyDeviceToHost));
                printf("A%d %s G=%4d %5d B/WG exchanged: %.3f us per barrier  timeout=%d mismatches=%llu\n", kind,
                       kind == 1 ? "counter, relaxed poll + 1 release + 1 acquire" : "XCD-hierarchical                      ", G, words * 4,
                       ms * 1e3 / (2 * rounds), herr, hm);
            }
        }
    }
    for (int G : {128, 256, 512}) {
        CK(hipMemsetAsync(state, 0, 8192, st)); CK(hipMemsetAsync(err, 0, 4, st)); CK(hipMemsetAsync(mism, 0, 8, st));
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(flag_kernel, dim3(G), dim3(256), 0, st, state, rounds, buf, err, mism);
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        int herr; unsigned long long hm; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hm, mism, 8, hipMemcpyDeviceToHost));
        printf("A3 flag hand-off (%d producers -> %d consumers, 512 B each, fan-in counter): %.3f us per hop  timeout=%d mismatches=%llu\n",
               G / 2, G / 2, ms * 1e3 / (2 * rounds), herr, hm);
    }

    // ---- part B
    const int layers = 58;
    const size_t w1 = (size_t)kS1.wgs * kS1.w16 * kThreads, w2 = (size_t)kS2.wgs * kS2.w16 * kThreads, w3 = (size_t)kS3.wgs * kS3.w16 * kThreads;  // in 16-B units
    const size_t per_layer = w1 + w2 + w3;
    i32x4* weights; CK(hipMalloc(&weights, per_layer * layers * 16)); CK(hipMemset(weights, 0, per_layer * layers * 16));
    i32x4 *x, *o1, *o2, *o3;
    CK(hipMalloc(&x, (size_t)kS1.in16 * kThreads * 16)); CK(hipMemset(x, 0, (size_t)kS1.in16 * kThreads * 16));
    CK(hipMalloc(&o1, (size_t)kS1.wgs * kS1.out16 * 16)); CK(hipMalloc(&o2, (size_t)kS2.wgs * kS2.out16 * 16)); CK(hipMalloc(&o3, (size_t)kS3.wgs * kS3.out16 * 16));
    printf("part B: %d layers, %.1f MB of weights per layer (%.2f GB in all), stage grids %d / %d / %d x %d threads\n", layers,
           per_layer * 16 / 1e6, per_layer * layers * 16 / 1e9, kS1.wgs, kS2.wgs, kS3.wgs, kThreads);
    for (int variant = 0; variant < 4; ++variant) {
        CK(hipMemsetAsync(state, 0, 8192, st)); CK(hipMemsetAsync(err, 0, 4, st)); CK(hipMemsetAsync(mism, 0, 8, st));
        hipGraph_t graph; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int l = 0; l < layers; ++l) {
            LayerPtrs p;
            p.w1 = weights + per_layer * l; p.w2 = p.w1 + w1; p.w3 = p.w2 + w2;
            p.x = x; p.o1 = o1; p.o2 = o2; p.o3 = o3; p.st = state; p.err = err; p.mism = mism; p.tag = 1000u + l;
            if (variant == 0) {
                hipLaunchKernelGGL(stage_kernel<1>, dim3(kS1.wgs), dim3(kThreads), 0, st, p);
                hipLaunchKernelGGL(stage_kernel<2>, dim3(kS2.wgs), dim3(kThreads), 0, st, p);
                hipLaunchKernelGGL(stage_kernel<3>, dim3(kS3.wgs), dim3(kThreads), 0, st, p);
            } else if (variant == 1) {
                hipLaunchKernelGGL((fused_kernel<true, false>), dim3(kS1.wgs + kS2.wgs + kS3.wgs), dim3(kThreads), 0, st, p);
            } else if (variant == 2) {
                hipLaunchKernelGGL((fused_kernel<false, false>), dim3(kS1.wgs + kS2.wgs + kS3.wgs), dim3(kThreads), 0, st, p);
            } else {
                hipLaunchKernelGGL((fused_kernel<true, true>), dim3(kS1.wgs + kS2.wgs + kS3.wgs), dim3(kThreads), 0, st, p);
            }
        }
        CK(hipStreamEndCapture(st, &graph)); CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(exec, st));
        CK(hipStreamSynchronize(st));
        float best = 1e9f, sum = 0.f; const int reps = 10;
        for (int i = 0; i < reps; ++i) {
            CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(exec, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best; sum += ms;
        }
        int herr; unsigned long long hm; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hm, mism, 8, hipMemcpyDeviceToHost));
        printf("B%d %-62s: %.2f us per layer (best %.2f)  timeout=%d mismatches=%llu\n", variant,
               variant == 0 ? "three graph-captured launches per layer" : variant == 1 ? "ONE launch, weights prefetched before the dependency wait"
               : variant == 2 ? "ONE launch, weights requested after the wait" : "ONE launch, prefetch, one poller per XCC + XCC-local go word",
               sum / reps * 1e3 / layers, best * 1e3 / layers, herr, hm);
        CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
    }
    return 0;
}
