// LDS-DMA (global_load_lds_dwordx4) pieces shared by the LDS-staged gfx950 kernels (mla_prefill_flash.hip, the tiled GEMMs).
// One wave-instruction moves 1 KiB: lane i's 16 bytes land at LDS byte (lds_dst + 16 i) -- the destination is lane-linear,
// the SOURCE address is per lane, so a bank-conflict-free LDS image is built by permuting the source side (guide rule 21).
// hipcc does not count these loads (asm): the issuing kernel waits with its own `s_waitcnt vmcnt(0)` ahead of the barrier
// that publishes the tile.  s_nop 4: an SGPR written by VALU (readfirstlane) feeding a VMEM address; s_nop 0: M0 written by
// SALU feeding the LDS-DMA.
#pragma once
#include "common.h"

namespace chitu {

// a wave-uniform pointer the compiler cannot prove uniform (it depends on a loop-carried index, a value read from LDS, ...): an
// "s" asm operand fed from VGPRs does not assemble.  Free when the value already lives in SGPRs.
template <typename T>
__device__ __forceinline__ const T* uniform_ptr(const T* p) {
    const uint64_t a = (uint64_t)p;
    return (const T*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a));
}
// source = (wave-uniform base) + (per-lane 32-bit byte offset)
// NT: the `nt` (streaming) cache hint -- data read once per launch (a KV tile) should not displace the L2's resident lines
template <bool NT = false>
__device__ __forceinline__ void glds16_sbase(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    unsigned keep;
    if constexpr (NT)
        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(sbase), "s"(lds_dst)
                     : "memory");
    else
        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(sbase), "s"(lds_dst)
                     : "memory");
}
// Four consecutive 1 KiB pieces (LDS bytes lds_dst .. lds_dst + 4095) from one base with four per-lane offsets, as ONE asm statement:
// M0 is saved and restored once and stepped by s_add between the pieces, the VALU -> SGPR hazard nop is paid once -- 14 scalar
// instructions instead of 24 for four glds16_sbase calls (the tiled GEMMs issue ~5 scalar instructions per MFMA, rocprof round 6).
__device__ __forceinline__ void glds16x4_sbase(const void* sbase, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, uint32_t lds_dst) {
    unsigned keep;
    asm volatile(
        "s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(sbase), "s"(lds_dst)
        : "memory", "scc");
}
// 4 bytes per lane (256 B per wave-instruction, lane i -> LDS byte lds_dst + 4 i): per-row values that ride with a tile (block
// scales) -- brought this way they are counted by the same vmcnt as the tile and the K loop holds no load the compiler would
// wait for (its s_waitcnt counts in order, so a wait for ANY of its own loads also waits for the DMA issued before it)
__device__ __forceinline__ void glds4_sbase(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}
// source = per-lane 64-bit address
__device__ __forceinline__ void glds16_vaddr(const void* gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ void glds_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// all but the N most recent pieces have landed (a ring: the newest stage stays in flight)
template <int N>
__device__ __forceinline__ void glds_wait_leaving() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

// LDS byte offset of a __shared__ object (the low half of its flat address)
template <typename T>
__device__ __forceinline__ uint32_t lds_offset_of(T* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)(uint8_t*)p;
}

// ---- [rows][128 B] K-block tiles of the tiled GEMMs: row r = 8 chunks of 16 B, chunk c stored at c ^ ((r >> 1) & 7).
// Reader: lane (j = lane & 15, g = lane >> 4) of a 16-row MFMA tile takes chunks g and g + 4 of row j; under ds_read_b128's
// lane groups ({0-3, 12-15, 20-27}, ...) the sixteen (row, chunk) pairs of a group then hit sixteen different bank groups:
// 8 (r & 1) + (c ^ (r >> 1)) mod 16 is a bijection on them.  One DMA piece = 8 rows: lane i -> row 8 n + (i >> 3), position
// i & 7, i.e. source chunk (i & 7) ^ ((4 n + (i >> 4)) & 7).
__device__ __forceinline__ int kblock_src_chunk(int lane, int n) { return (lane & 7) ^ ((4 * n + (lane >> 4)) & 7); }
// byte offset inside a 16-row tile of lane (j, g)'s first fragment (chunk g); the second (chunk g + 4) is this ^ 64
__device__ __forceinline__ int kblock_frag_off(int j, int g) { return j * 128 + ((g ^ ((j >> 1) & 7)) << 4); }

}  // namespace chitu
