// Probe: ds_read_b64_tr_b16 lane/element mapping and MFMA 16x16x32 bf16/fp8 fragment layouts on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k_tr(short* out, int stride_elems) {
    __shared__ short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = i;
    __syncthreads();
    // lane t in 16-lane group: row = t/4 (+ 4*group), chunk = t%4 ; address = row*stride + chunk*4 elements
    int t = threadIdx.x & 15, grp = threadIdx.x >> 4;
    int addr = (grp * 4 + t / 4) * stride_elems + (t % 4) * 4;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + addr));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
__device__ inline short f2bf(float f) { return (short)(__float_as_uint(f) >> 16); }
__global__ void k_mfma(float* out) {
    // A[i][k] = i + 0.01*k (bf16-rounded), B[k][j] = (k==j? 1:0)  => C[i][j] = A[i][j] for j<16... use k<32
    int l = threadIdx.x, i = l & 15, g = l >> 4;
    s16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        int k = g * 8 + e;
        a[e] = f2bf((float)(i * 32 + k));       // A[i][k]
        b[e] = f2bf((k == (l & 15) + 3) ? 1.0f : 0.0f);  // B[k][j] = delta(k, j+3)
    }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    short h[256];
    for (int stride : {16, 40}) {
        k_tr<<<1, 64>>>(d, stride);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("tr16_b64 stride=%d elems\n", stride);
        for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) { int v = h[l*4+j]; printf(" (r%d,c%d)", v / stride, v % stride); } printf("\n"); }
    }
    float* f; hipMalloc(&f, 64 * 4 * 4); float hf[256];
    k_mfma<<<1, 64>>>(f); hipMemcpy(hf, f, sizeof(hf), hipMemcpyDeviceToHost);
    printf("mfma C: lane -> 4 regs, value = i*32+k where B selects k=j+3 => C[i][j]=i*32+j+3\n");
    for (int l = 0; l < 64; l += 5) { printf("lane %2d:", l); for (int r = 0; r < 4; ++r) { int v = (int)hf[l*4+r]; printf(" (i%d,j%d)", v / 32, v % 32 - 3); } printf("\n"); }
    return 0;
}
