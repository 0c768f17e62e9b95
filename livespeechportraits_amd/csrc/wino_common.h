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
// One U fragment (the B operand of 4 MFMAs: 16 bytes per lane) by a plain load into registers -- the UR form of the K loop (wino.hip).
// Inline asm like the LDS-DMA copies: the compiler does not see those, so its own vmcnt arithmetic would over-wait; the kernel counts.
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ void uload16(f32x4v &dst, unsigned voff, i32x4 srd, unsigned soff)
{
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(dst) : "v"(voff), "s"(srd), "s"(soff), "n"(OFF) : "memory");
}
template <int V> struct IntC { static constexpr int value = V; };

// float4 add / subtract as two v_pk_add_f32 (register pairs (x, y), (z, w): where a ds_read_b128 left them).  The input transforms are pure adds, and an
// fp32 VALU instruction costs the wave ~5 cycles of its MFMA stream: packed, a K-step's transform is 16 instructions instead of 32.
#ifdef LSPF2F_NO_PK      // A-B builds (tools/sessions/gpu_r4_pk.sh): the scalar form
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
#else
__device__ __forceinline__ float4 f4add(float4 a, float4 b)
{
    const v2f lo = v2f{a.x, a.y} + v2f{b.x, b.y}, hi = v2f{a.z, a.w} + v2f{b.z, b.w};
    return make_float4(lo.x, lo.y, hi.x, hi.y);
}
__device__ __forceinline__ float4 f4sub(float4 a, float4 b)
{
    const v2f lo = v2f{a.x, a.y} - v2f{b.x, b.y}, hi = v2f{a.z, a.w} - v2f{b.z, b.w};
    return make_float4(lo.x, lo.y, hi.x, hi.y);
}
#endif


}  // namespace lspf2f
