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
// the same with sc1 (past the L1, served by L2): what a workgroup reads of a tensor that OTHER workgroups of the same launch have just written through
// (wino3x3_chain; the guide's "16-B sc1 stores AND sc1 loads" hand-off form)
__device__ __forceinline__ void dma16_two_sc1(unsigned lds_a, unsigned lds_b, unsigned va, unsigned vb, i32x4 srd, int soff)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %3, %5, %6 offen sc1 lds\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %4, %5, %6 offen sc1 lds\n\t"
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

// Wave priority by K-loop progress (tune key `wino_prio`, round 5).  At batch 1 a Winograd layer is ONE round of 512 workgroups, two per CU, so every SIMD holds one wave of
// each; at equal priority the OLDER workgroup wins the matrix pipe on every SIMD (it leaves the loop after 28 000 cycles, the younger after 39 600, which then runs alone
// at a lone wave's issue rate: profiles/r05_wino_tail_stamps.txt).  Priority outranks age, so "the wave that is behind leads" keeps the two in step: level 3 - quarter of
// the loop done.  Schemes: 1 fair; 2 the one ahead leads (contrast arm); 3 fair, but the first half of the grid (the older workgroups of a two-per-CU round) keeps level 1
// through its last quarter, so that its epilogue overlaps the partner's last steps.  All operands are wave-uniform scalars: s_cmp / s_cbranch around one s_setprio.
struct ProgressPrio {
    int scheme, shift, floor_lvl;
    __device__ __forceinline__ ProgressPrio(int scheme_, int nsteps) : scheme(scheme_)
    {
        int sh = 0;
        while ((4 << sh) < nsteps) ++sh;                      // quarter = 1 << sh steps (nsteps is 8, 16 or 32 for every layer the generators build)
        shift = sh;
        floor_lvl = (scheme_ == 3 && blockIdx.x < (gridDim.x >> 1)) ? 1 : 0;
    }
    static __device__ __forceinline__ void set(int lvl)
    {
        if (lvl <= 0) __builtin_amdgcn_s_setprio(0); else if (lvl == 1) __builtin_amdgcn_s_setprio(1); else if (lvl == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3);
    }
    __device__ __forceinline__ void step(int t) const
    {
        if (!scheme || (t & ((1 << shift) - 1)) != 0) return;
        const int q = (t >> shift) > 3 ? 3 : (t >> shift);
        const int lvl = scheme == 2 ? q : 3 - q;
        set(lvl < floor_lvl ? floor_lvl : lvl);
    }
    __device__ __forceinline__ void done() const { if (scheme) __builtin_amdgcn_s_setprio(0); }
};

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
