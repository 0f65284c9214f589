// Probe: cost of a software grid barrier (all workgroups co-resident) on gfx950, with the
// agent-scope release/acquire needed for data written in one phase to be visible to every XCD in
// the next, vs the in-graph kernel boundary it would replace.  Every spin is bounded (no hangs).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, int* err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 4000000) { *err = 1; ok = false; break; }
        }
    }
    __syncthreads();
    return ok;
}

// MODE 0: barrier only.  MODE 1: each WG writes `bytes_per_wg` then after the barrier reads another WG's
// region (different XCD) and checks it.
template <int MODE>
__global__ __launch_bounds__(256) void bar_kernel(unsigned* counter, int rounds, uint32_t* buf, int words_per_wg,
                                                  int* err, unsigned long long* mism) {
    const unsigned G = gridDim.x;
    unsigned long long bad = 0;
    for (int r = 1; r <= rounds; ++r) {
        if (MODE == 1) {
            uint32_t* mine = buf + (size_t)blockIdx.x * words_per_wg;
            for (int i = threadIdx.x; i < words_per_wg; i += 256) mine[i] = (uint32_t)r * 1000003u + i;
        }
        if (!grid_barrier(counter, (unsigned)(2 * r - 1) * G, err)) return;
        if (MODE == 1) {
            const uint32_t* other = buf + (size_t)((blockIdx.x + 37) % G) * words_per_wg;
            for (int i = threadIdx.x; i < words_per_wg; i += 256) bad += other[i] != (uint32_t)r * 1000003u + i;
        }
        if (!grid_barrier(counter, (unsigned)(2 * r) * G, err)) return;
    }
    if (bad) atomicAdd(mism, bad);
}

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 12345) *p = 1; }
__global__ __launch_bounds__(256) void chain_kernel(uint32_t* buf, int words_per_wg, int r) {
    uint32_t* mine = buf + (size_t)blockIdx.x * words_per_wg;
    const uint32_t* other = buf + (size_t)((blockIdx.x + 37) % gridDim.x) * words_per_wg + words_per_wg * gridDim.x * (r & 1);
    uint32_t acc = 0;
    for (int i = threadIdx.x; i < words_per_wg; i += 256) acc += other[i];
    for (int i = threadIdx.x; i < words_per_wg; i += 256) mine[i + words_per_wg * gridDim.x * ((r + 1) & 1)] = acc + i;
}

int main() {
    unsigned* counter; int* err; unsigned long long* mism; uint32_t* buf;
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&mism, 8)); CK(hipMalloc(&buf, 64 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipStream_t st; CK(hipStreamCreate(&st));
    const int rounds = 2000;
    for (int G : {64, 256, 512, 1024}) {
        for (int mode = 0; mode < 3; ++mode) {
            const int words = mode == 2 ? 4096 : 256;   // 1 KB or 16 KB per WG
            CK(hipMemsetAsync(counter, 0, 4, st)); CK(hipMemsetAsync(err, 0, 4, st)); CK(hipMemsetAsync(mism, 0, 8, st));
            CK(hipEventRecord(e0, st));
            if (mode == 0) hipLaunchKernelGGL(bar_kernel<0>, dim3(G), dim3(256), 0, st, counter, rounds, buf, words, err, mism);
            else hipLaunchKernelGGL(bar_kernel<1>, dim3(G), dim3(256), 0, st, counter, rounds, buf, words, err, mism);
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            int herr; unsigned long long hm; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hm, mism, 8, hipMemcpyDeviceToHost));
            printf("G=%4d mode=%d (%s): %.3f us per barrier, timeout=%d mismatches=%llu\n", G, mode,
                   mode == 0 ? "barrier only" : mode == 1 ? "1 KB/WG exchange" : "16 KB/WG exchange", ms * 1e3 / (2 * rounds), herr, hm);
        }
    }
    // reference: the same exchange as a chain of dependent kernels in a hipGraph
    for (int G : {256, 1024}) {
        for (int words : {256, 4096}) {
            hipGraph_t graph; hipGraphExec_t exec;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            for (int r = 0; r < 200; ++r) hipLaunchKernelGGL(chain_kernel, dim3(G), dim3(256), 0, st, buf, words, r);
            CK(hipStreamEndCapture(st, &graph)); CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            CK(hipGraphLaunch(exec, st)); CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(exec, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("graph chain G=%4d %2d KB/WG: %.3f us per kernel\n", G, words * 4 / 1024, ms * 1e3 / 200);
        }
    }
    return 0;
}
