// Helpers shared by the two Winograd kernels (wino.hip, winoup.hip).  Internal to liblspf2f.so.
#pragma once
#include "device_common.h"

namespace lspf2f {

// two LDS-DMA pieces with unrelated LDS destinations, one descriptor and scalar offset (the raw patch: pieces w and w + 4, or the dump slot)
__device__ __forceinline__ void dma16_two(unsigned lds_a, unsigned lds_b, unsigned va, unsigned vb, i32x4 srd, int soff)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %3, %5, %6 offen lds\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %4, %5, %6 offen lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_a), "s"(lds_b), "v"(va), "v"(vb), "s"(srd), "s"(soff) : "memory");
}
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }


}  // namespace lspf2f
