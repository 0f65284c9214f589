// Probe (round 4): the SMALL-fan-in last-arriver hand-off the round-3 review asked to be measured next to
// tools/probe_handoff.hip's 4.8-14.7 us flag hops (64-256 pollers): P producer workgroups, NOBODY polls -- each producer
// publishes its slice, takes a ticket, and the one that arrives last runs the tail in the same launch.  This is the shape of
// "router score GEMM + routing + moe_align in ONE launch" (gate.hip's gate_ticket form): P = 16 tiles x S K-splits producers
// of 1 KB of partial logits each, one sorter.
//
//   L  one launch per layer: producers stream `wbytes` of cold weights each, write 1 KB (sc1 write-through), drain, barrier,
//      one relaxed agent-scope fetch_add; the last arriver resets the ticket, acquires, reads all P KB (sc1 loads), reduces,
//      writes 4 KB.
//   K  two launches per layer in the same hipGraph: the producers (plain stores), then ONE workgroup doing the same tail.
// 58 layers per graph replay, weights advancing through a 1 GB buffer (HBM-cold, like the step); every tail result is
// checked.  Build: hipcc -O3 --offload-arch=gfx950 tools/probe_last_arriver.hip -o tools/probe_last_arriver.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 256;

// every producer: stream w16 x 16 B per thread of weights, fold them (zeros at run time, but data-dependent), write 1 KB
template <int W16>
__device__ __forceinline__ int stream_weights(const i32x4* w, int wg) {
    i32x4 r[W16];
#pragma unroll
    for (int i = 0; i < W16; ++i) r[i] = __builtin_nontemporal_load(w + ((size_t)wg * W16 + i) * kThreads + threadIdx.x);
    int a = 0;
#pragma unroll
    for (int i = 0; i < W16; ++i) a ^= r[i][0] ^ r[i][1] ^ r[i][2] ^ r[i][3];
    return a;
}

__device__ __forceinline__ void tail(const uint32_t* part, int P, uint32_t* out, unsigned tag, unsigned long long* mism, bool sc1) {
    // 256 threads: thread t sums word t of every producer's 256-word slice (P loads in flight per thread, clamped unroll)
    uint32_t acc = 0;
    for (int p0 = 0; p0 < P; p0 += 16) {
        uint32_t v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t* src = part + (size_t)min(p0 + i, P - 1) * 256 + threadIdx.x;
            // (agent-scope relaxed atomic load = global_load_dword ... sc1: bypasses this XCD's possibly stale L2 line)
            v[i] = sc1 ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *src;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += (p0 + i < P) ? v[i] : 0u;
    }
    // expected: sum_p (tag + p + t)
    const uint32_t want = (uint32_t)P * (tag + threadIdx.x) + (uint32_t)(P * (P - 1) / 2);
    if (acc != want) atomicAdd(mism, 1ull);
    for (int i = 0; i < 4; ++i) out[i * 256 + threadIdx.x] = acc + i;
}

template <int W16>
__global__ __launch_bounds__(kThreads) void fused_kernel(const i32x4* w, uint32_t* part, unsigned* ticket, uint32_t* out, unsigned tag,
                                                         unsigned long long* mism) {
    const int P = gridDim.x, wg = blockIdx.x;
    const int fold = stream_weights<W16>(w, wg);
    const uint32_t v = tag + wg + threadIdx.x + (uint32_t)fold;
    asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 1" ::"v"(part + (size_t)wg * 256 + threadIdx.x), "v"(v) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __shared__ unsigned last;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = t == (unsigned)P - 1;
        if (last) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    __syncthreads();
    if (!last) return;
    tail(part, P, out, tag, mism, true);
}

template <int W16>
__global__ __launch_bounds__(kThreads) void producer_kernel(const i32x4* w, uint32_t* part, unsigned tag) {
    const int wg = blockIdx.x;
    const int fold = stream_weights<W16>(w, wg);
    part[(size_t)wg * 256 + threadIdx.x] = tag + wg + threadIdx.x + (uint32_t)fold;
}
__global__ __launch_bounds__(kThreads) void tail_kernel(const uint32_t* part, int P, uint32_t* out, unsigned tag, unsigned long long* mism) {
    tail(part, P, out, tag, mism, false);
}

int main() {
    constexpr int kLayers = 58;
    constexpr int W16 = 4;  // 4 x 16 B x 256 threads = 16 KB of weights per producer (the router GEMM's 14 KB)
    const size_t wbytes_total = (size_t)1 << 30;
    i32x4* w;
    uint32_t *part, *out;
    unsigned* ticket;
    unsigned long long* mism;
    CK(hipMalloc(&w, wbytes_total));
    CK(hipMemset(w, 0, wbytes_total));
    CK(hipMalloc(&part, 256 * 256 * 4 * kLayers));
    CK(hipMalloc(&out, 4096 * kLayers));
    CK(hipMalloc(&ticket, 256));
    CK(hipMemset(ticket, 0, 256));
    CK(hipMalloc(&mism, 8));
    CK(hipMemset(mism, 0, 8));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (int P : {16, 64, 256}) {
        for (int variant = 0; variant < 2; ++variant) {
            hipGraph_t g;
            hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            for (int l = 0; l < kLayers; ++l) {
                const i32x4* wl = w + ((size_t)l * 256 * W16 * kThreads) % (wbytes_total / 16 - (size_t)256 * W16 * kThreads);
                uint32_t* pl = part + (size_t)l * 256 * 256;
                if (variant == 0) {
                    hipLaunchKernelGGL(fused_kernel<W16>, dim3(P), dim3(kThreads), 0, st, wl, pl, ticket, out + l * 1024, 1000u + l, mism);
                } else {
                    hipLaunchKernelGGL(producer_kernel<W16>, dim3(P), dim3(kThreads), 0, st, wl, pl, 1000u + l);
                    hipLaunchKernelGGL(tail_kernel, dim3(1), dim3(kThreads), 0, st, pl, P, out + l * 1024, 1000u + l, mism);
                }
            }
            CK(hipStreamEndCapture(st, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
            CK(hipStreamSynchronize(st));
            float best = 1e9f, sum = 0.f;
            const int reps = 10;
            for (int i = 0; i < reps; ++i) {
                CK(hipEventRecord(e0, st));
                CK(hipGraphLaunch(ge, st));
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
                sum += ms;
            }
            unsigned long long h_m = 0;
            CK(hipMemcpy(&h_m, mism, 8, hipMemcpyDeviceToHost));
            CK(hipMemset(mism, 0, 8));
            printf("%s P=%3d producers x 16 KB weights, 1 KB each -> one tail: %.2f us per layer (best %.2f)  mismatches=%llu\n",
                   variant == 0 ? "L one launch, last arriver runs the tail " : "K two launches (producers, then the tail)", P,
                   sum / reps / kLayers * 1e3f, best / kLayers * 1e3f, h_m);
            CK(hipGraphExecDestroy(ge));
            CK(hipGraphDestroy(g));
        }
    }
    return 0;
}
