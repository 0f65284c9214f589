// Probe: how fast can one CU fill LDS from the L2 (gfx950)?  LDS-DMA (global_load_lds_dwordx4) against global_load_dwordx4 into
// registers (+ ds_write_b128).  Every workgroup walks a window of a buffer that fits the XCD's L2 (2 MB, re-read), 8 KB per wave and
// iteration like the tiled GEMMs' staging.  Prints GB/s per CU and bytes per clock per CU (at the reported clock).
//   hipcc -O3 --offload-arch=gfx950 tools/probe_lds_fill.hip -o build_probe/probe_lds_fill && build_probe/probe_lds_fill
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void fill(const uint8_t* __restrict__ buf, uint32_t window, int iters, int* sink) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t ldsb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
    // window start differs per workgroup (so that workgroups of an XCD do not all hit one line at one moment)
    uint32_t off = (uint32_t)((blockIdx.x * 40960u + wave * 8192u) % window);
    i32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int p = 0; p < 8; ++p) glds16(buf + off, (uint32_t)(p * 1024 + lane * 16), ldsb + (uint32_t)(((it & 1) * WAVES + wave) * 8192 + p * 1024));
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // the previous iteration's pieces
        } else {
            i32x4 r[8];
#pragma unroll
            for (int p = 0; p < 8; ++p) r[p] = *(const i32x4*)(buf + off + p * 1024 + lane * 16);
            if (MODE == 1) {
#pragma unroll
                for (int p = 0; p < 8; ++p) acc ^= r[p];
            } else {
#pragma unroll
                for (int p = 0; p < 8; ++p) *(i32x4*)(lds + ((it & 1) * WAVES + wave) * 8192 + p * 1024 + lane * 16) = r[p];
            }
        }
        off += 8192u * WAVES;
        if (off + 8192u > window) off = (uint32_t)(wave * 8192u);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (MODE != 1) acc[0] ^= *(const int*)(lds + threadIdx.x * 4);
    if (acc[0] == 0x12345678 && acc[1] == 7) *sink = acc[2];
}

// The tiled GEMMs' staging pattern: one DMA piece = 8 rows x 128 B at row stride K bytes; a 4-wave workgroup brings 256 rows (two
// 128-row tiles) of K block kb per iteration and walks kb.  rows_total rows of K bytes stay L2-resident when small.  skew: workgroup
// w starts at K block w % KB instead of 0 (all workgroups of a GEMM start at 0 and move in step).
__global__ __launch_bounds__(256) void fill_rows(const uint8_t* __restrict__ buf, int K, int rows_total, int iters, int skew, int* sink) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t ldsb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
    const int KB = K / 128;
    const int row0 = (int)((blockIdx.x * 256u) % (unsigned)rows_total);
    uint32_t voff[8];
    for (int p = 0; p < 8; ++p) voff[p] = (uint32_t)(((wave * 8 + p) * 8 + (lane >> 3)) * K + (lane & 7) * 16);
    const uint8_t* base = buf + (size_t)row0 * K;
    int kb = skew ? (int)(blockIdx.x % (unsigned)KB) : 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 8; ++p) glds16(base + (size_t)kb * 128, voff[p], ldsb + (uint32_t)(((it & 1) * 4 + wave) * 8192 + p * 1024));
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        kb = kb + 1 == KB ? 0 : kb + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (*(const int*)(lds + threadIdx.x * 4) == 0x12345678) *sink = 1;
}

static void run_rows(const uint8_t* buf, int* sink, int wgs, int K, int rows_total, int skew, double mhz) {
    const int iters = 2000;
    const size_t ldsz = 2 * 4 * 8192;
    hipFuncSetAttribute((const void*)fill_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsz);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(fill_rows, dim3(wgs), dim3(256), ldsz, 0, buf, K, rows_total, 50, skew, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL(fill_rows, dim3(wgs), dim3(256), ldsz, 0, buf, K, rows_total, iters, skew, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)wgs * 4 * iters * 8192.0;
    const double cus = wgs < 256 ? wgs : 256;
    const double gbs_cu = bytes / (ms * 1e-3) / 1e9 / cus;
    printf("rows pattern  K %5d  rows %5d (%5.1f MB)  skew %d  wgs %4d  %8.3f ms  total %6.2f TB/s  per CU %6.1f GB/s = %5.1f B/clk\n", K, rows_total,
           (double)rows_total * K / 1e6, skew, wgs, ms, bytes / (ms * 1e-3) / 1e12, gbs_cu, gbs_cu * 1e3 / mhz);
}

template <int MODE, int WAVES>
static void run(const uint8_t* buf, int* sink, int wgs, const char* name, double mhz) {
    const int iters = 2000;
    const uint32_t window = 2u << 20;
    const size_t ldsz = 2 * WAVES * 8192;
    hipFuncSetAttribute((const void*)fill<MODE, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsz);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((fill<MODE, WAVES>), dim3(wgs), dim3(64 * WAVES), ldsz, 0, buf, window, 50, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((fill<MODE, WAVES>), dim3(wgs), dim3(64 * WAVES), ldsz, 0, buf, window, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)wgs * WAVES * iters * 8192.0;
    const double cus = wgs < 256 ? wgs : 256;
    const double gbs_cu = bytes / (ms * 1e-3) / 1e9 / cus;
    printf("%-34s waves/wg %d  wgs %4d  %8.3f ms  total %7.2f TB/s  per CU %6.1f GB/s = %5.1f B/clk at %.0f MHz\n", name, WAVES, wgs, ms,
           bytes / (ms * 1e-3) / 1e12, gbs_cu, gbs_cu * 1e3 / mhz, mhz);
}

int main() {
    uint8_t* buf; int* sink;
    hipMalloc(&buf, 64u << 20); hipMemset(buf, 1, 64u << 20); hipMalloc(&sink, 4);
    int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double mhz = khz / 1000.0;
    for (int wgs : {256}) {  // how many waves does it take to fill the path?
        run<0, 1>(buf, sink, wgs, "LDS-DMA dwordx4", mhz);
        run<0, 2>(buf, sink, wgs, "LDS-DMA dwordx4", mhz);
        run<1, 1>(buf, sink, wgs, "global_load_dwordx4 -> regs", mhz);
        run<1, 2>(buf, sink, wgs, "global_load_dwordx4 -> regs", mhz);
    }
    for (int wgs : {256, 512}) {
        run<0, 4>(buf, sink, wgs, "LDS-DMA dwordx4", mhz);
        run<0, 8>(buf, sink, wgs, "LDS-DMA dwordx4", mhz);
        run<1, 4>(buf, sink, wgs, "global_load_dwordx4 -> regs", mhz);
        run<1, 8>(buf, sink, wgs, "global_load_dwordx4 -> regs", mhz);
        run<2, 4>(buf, sink, wgs, "global_load_dwordx4 + ds_write_b128", mhz);
        run<2, 8>(buf, sink, wgs, "global_load_dwordx4 + ds_write_b128", mhz);
    }
    for (int wgs : {256, 512})
        for (int K : {7168, 2048})
            for (int rows : {512})
                for (int skew : {0, 1}) run_rows(buf, sink, wgs, K, rows, skew, mhz);
    return 0;
}
