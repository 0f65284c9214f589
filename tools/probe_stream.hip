// Probe: HBM read bandwidth of the GEMM weight access pattern vs a linear stream (gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
// pattern 0: linear: wave reads its 16*K bytes region as consecutive 1 KB chunks
// pattern 1: rows: lane -> row (lane&15), 16 B at k = kb*128 + c*64 + (lane>>4)*16  (current kernels)
// pattern 2: rows128: lane -> row (lane>>3) (8 rows/instr), 16 B at (lane&7)*16 : 128 B contiguous per row
template <int PAT, int D>
__global__ __launch_bounds__(64) void k(const uint8_t* __restrict__ w, int K, int* sink) {
    const int lane = threadIdx.x;
    const size_t tile = blockIdx.x;               // 16 rows each
    const uint8_t* base = w + tile * 16 * (size_t)K;
    int acc = 0;
    const int KB = K / 128;
    i32x4 ring[D][2];
    auto ld = [&](int kb, i32x4 (&r)[2]) {
        if (PAT == 0) {
            r[0] = __builtin_nontemporal_load((const i32x4*)(base + (size_t)kb * 2048 + lane * 16));
            r[1] = __builtin_nontemporal_load((const i32x4*)(base + (size_t)kb * 2048 + 1024 + lane * 16));
        } else if (PAT == 1) {
            const uint8_t* p = base + (size_t)(lane & 15) * K + kb * 128 + (lane >> 4) * 16;
            r[0] = __builtin_nontemporal_load((const i32x4*)(p));
            r[1] = __builtin_nontemporal_load((const i32x4*)(p + 64));
        } else {
            const uint8_t* p = base + (size_t)(lane >> 3) * K + kb * 128 + (lane & 7) * 16;
            r[0] = __builtin_nontemporal_load((const i32x4*)(p));
            r[1] = __builtin_nontemporal_load((const i32x4*)(p + (size_t)8 * K));
        }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) if (d < KB) ld(d, ring[d]);
    for (int kb = 0; kb < KB; kb += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (kb + d < KB) {
                acc += ring[d][0][0] ^ ring[d][1][3] ^ ring[d][0][2] ^ ring[d][1][1];
                if (kb + d + D < KB) ld(kb + d + D, ring[d]);
            }
        }
    }
    if (acc == 0x12345678) sink[0] = acc;
}
template <int PAT, int D> float run(const uint8_t* w, int K, int tiles, int* sink) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<PAT, D><<<tiles, 64>>>(w, K, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) k<PAT, D><<<tiles, 64>>>(w + (size_t)i * tiles * 16 * K, K, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main() {
    const int K = 7168, tiles = 3300;   // ~378 MB per launch like moe_gemm1 at bs=16
    size_t bytes = (size_t)tiles * 16 * K;
    uint8_t* w; hipMalloc(&w, bytes * 6); hipMemset(w, 1, bytes * 6);
    int* sink; hipMalloc(&sink, 4);
    printf("bytes/launch %.1f MB\n", bytes / 1e6);
#define R(P, D) { float ms = run<P, D>(w, K, tiles, sink); printf("pattern %d depth %d: %.1f us  %.0f GB/s\n", P, D, ms * 1e3, bytes / ms / 1e6); }
    R(0, 2) R(0, 4) R(0, 8) R(1, 2) R(1, 4) R(1, 8) R(2, 2) R(2, 4) R(2, 8)
    return 0;
}
