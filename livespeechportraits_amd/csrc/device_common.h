// Device-side helpers shared by the gfx950 kernels (types, bf16 storage conversion, 4-channel load/store).
// Internal to liblspf2f.so; written for wave64 + MFMA only.
#pragma once
#include <hip/hip_runtime.h>

namespace lspf2f {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float v2f __attribute__((ext_vector_type(2)));   // pairs for v_pk_fma_f32
typedef unsigned short bf16_t;               // bf16 storage (round-to-nearest-even on store)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float((unsigned)v << 16); }
// round-to-nearest-even in hardware: gfx950 has v_cvt_pk_bf16_f32 (two floats -> one packed register); the integer sequence it replaces
// cost 5 VALU instructions per value in every bf16 epilogue
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi)
{
    const v2f v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }
// 4 consecutive channels of an activation row: load / store as float4 regardless of the storage type
template <typename T> __device__ __forceinline__ float4 load4(const T *p);
template <> __device__ __forceinline__ float4 load4<float>(const float *p) { return *reinterpret_cast<const float4 *>(p); }
template <> __device__ __forceinline__ float4 load4<bf16_t>(const bf16_t *p)
{
    const uint2 u = *reinterpret_cast<const uint2 *>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                       __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
// fp16 storage (dtype 2: the reference's opt.fp16 / autocast configuration, models/feature2face_G.py:28-30): IEEE binary16, round-to-nearest-even
// on store, fp32 accumulate and epilogue like the bf16 path.  _Float16 is a distinct type from bf16_t, which is what the templates dispatch on.
typedef _Float16 f16_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <> __device__ __forceinline__ float4 load4<f16_t>(const f16_t *p)
{
    const f32x4v f = __builtin_convertvector(__builtin_bit_cast(f16x4, *reinterpret_cast<const uint2 *>(p)), f32x4v);
    return make_float4(f.x, f.y, f.z, f.w);
}
template <typename T> __device__ __forceinline__ void store4(T *p, float4 v);
template <> __device__ __forceinline__ void store4<f16_t>(f16_t *p, float4 v)
{
    const f32x4v f = {v.x, v.y, v.z, v.w};
    *reinterpret_cast<uint2 *>(p) = __builtin_bit_cast(uint2, __builtin_convertvector(f, f16x4));
}
// 16-bit storage kernels written once for both types (F16 = IEEE half, else bf16): registers carry the 16-byte operand as bf16x8 whatever it holds
typedef float f32x4acc __attribute__((ext_vector_type(4)));
template <bool F16> __device__ __forceinline__ f32x16 mfma32_16b(bf16x8 a, bf16x8 b, f32x16 c)
{
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <bool F16> __device__ __forceinline__ f32x4acc mfma16_16b(bf16x8 a, bf16x8 b, f32x4acc c)
{
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
template <bool F16> __device__ __forceinline__ unsigned pack16x2(float lo, float hi)       // two floats -> one register of two 16-bit values, RNE
{
    if constexpr (F16) { const v2f v = {lo, hi}; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2)); }
    else return pack_bf16x2(lo, hi);
}
template <bool F16> __device__ __forceinline__ float lo16(unsigned w)
{
    if constexpr (F16) return (float)__builtin_bit_cast(f16x2, w).x; else return __uint_as_float(w << 16);
}
template <bool F16> __device__ __forceinline__ float hi16(unsigned w)
{
    if constexpr (F16) return (float)__builtin_bit_cast(f16x2, w).y; else return __uint_as_float(w & 0xffff0000u);
}
template <bool F16> struct St16 { typedef bf16_t type; };
template <> struct St16<true> { typedef f16_t type; };

// one element, whatever the storage type
template <typename T> __device__ __forceinline__ float ld1(const T *p);
template <> __device__ __forceinline__ float ld1<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float ld1<bf16_t>(const bf16_t *p) { return bf2f(*p); }
template <> __device__ __forceinline__ float ld1<f16_t>(const f16_t *p) { return (float)*p; }
template <typename T> __device__ __forceinline__ void st1(T *p, float v);
template <> __device__ __forceinline__ void st1<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void st1<bf16_t>(bf16_t *p, float v) { *p = f2bf(v); }
template <> __device__ __forceinline__ void st1<f16_t>(f16_t *p, float v) { *p = (f16_t)v; }
template <> __device__ __forceinline__ void store4<float>(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t *p, float4 v)
{
    uint2 u;
    u.x = pack_bf16x2(v.x, v.y);
    u.y = pack_bf16x2(v.z, v.w);
    *reinterpret_cast<uint2 *>(p) = u;
}

// buffer resource descriptor (raw buffer, stride 0) from wave-uniform values
__device__ __forceinline__ i32x4 make_srd(const void *base, unsigned bytes)
{
    const unsigned long long a = (unsigned long long)base;
    i32x4 r;
    r.x = (int)(unsigned)a;
    r.y = (int)((unsigned)(a >> 32) & 0xffffu);
    r.z = (int)bytes;
    r.w = 0x00020000;
    return r;
}

// One LDS-DMA piece: 64 lanes x 16 B -> LDS[lds_addr + lane*16]; out-of-range voffset lands as zeros.
// Issued from inline asm on purpose: a compiler-visible LDS-DMA makes hipcc wait vmcnt(0) before the
// next ds_read (it cannot disambiguate LDS addresses), which serialises the copy with the MFMAs.  The
// kernel waits for its DMAs itself (dma_wait) right before the barrier that publishes the buffer.
// M0 is compiler-reserved: saved and restored inside the statement.
__device__ __forceinline__ void dma16(unsigned lds_addr, unsigned voff, i32x4 srd, int soff)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %1\n\t"
                 "s_nop 4\n\t"
                 "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_addr), "v"(voff), "s"(srd), "s"(soff)
                 : "memory");
}
// N pieces of one operand in ONE statement (N = 1, 2, 4, 8): consecutive pieces land STRIDE bytes apart in LDS and share descriptor and scalar
// offset, so M0 is saved, advanced and restored once per group instead of once per piece (2 + 3N instead of 5N issue slots; the bf16 K loop
// is issue-bound).  One wait state between an M0 write and the LDS-DMA that uses it is what the hazard table asks for.
template <int N, int STRIDE>
__device__ __forceinline__ void dma16_group(unsigned lds_addr, const unsigned (&voff)[N], i32x4 srd, int soff)
{
    static_assert(N == 1 || N == 2 || N == 3 || N == 4 || N == 8, "piece counts of the shipped tiles");
    unsigned keep;
    if constexpr (N == 1) {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(lds_addr), "v"(voff[0]), "s"(srd), "s"(soff) : "memory");
    } else if constexpr (N == 2) {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %2, %4, %5 offen lds\n\ts_add_u32 m0, m0, %6\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %3, %4, %5 offen lds\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(lds_addr), "v"(voff[0]), "v"(voff[1]), "s"(srd), "s"(soff), "n"(STRIDE) : "memory", "scc");
    } else if constexpr (N == 3) {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %2, %5, %6 offen lds\n\ts_add_u32 m0, m0, %7\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %3, %5, %6 offen lds\n\ts_add_u32 m0, m0, %7\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %4, %5, %6 offen lds\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(lds_addr), "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "s"(srd), "s"(soff), "n"(STRIDE) : "memory", "scc");
    } else if constexpr (N == 4) {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %2, %6, %7 offen lds\n\ts_add_u32 m0, m0, %8\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %3, %6, %7 offen lds\n\ts_add_u32 m0, m0, %8\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %4, %6, %7 offen lds\n\ts_add_u32 m0, m0, %8\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %5, %6, %7 offen lds\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(lds_addr), "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(srd), "s"(soff), "n"(STRIDE) : "memory", "scc");
    } else {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %2, %10, %11 offen lds\n\ts_add_u32 m0, m0, %12\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %3, %10, %11 offen lds\n\ts_add_u32 m0, m0, %12\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %4, %10, %11 offen lds\n\ts_add_u32 m0, m0, %12\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %5, %10, %11 offen lds\n\ts_add_u32 m0, m0, %12\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %6, %10, %11 offen lds\n\ts_add_u32 m0, m0, %12\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %7, %10, %11 offen lds\n\ts_add_u32 m0, m0, %12\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %8, %10, %11 offen lds\n\ts_add_u32 m0, m0, %12\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %9, %10, %11 offen lds\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(lds_addr), "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "v"(voff[4]), "v"(voff[5]), "v"(voff[6]), "v"(voff[7]),
                       "s"(srd), "s"(soff), "n"(STRIDE) : "memory", "scc");
    }
}
// wait until at most N of this wave's DMA pieces are still in flight (they complete in order)
template <int N>
__device__ __forceinline__ void dma_wait()
{
    __builtin_amdgcn_sched_barrier(0);   // keep the MFMAs issued so far ABOVE the wait (they cover the copy)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LDS ring depth of the igemm pipeline (2; 3 if built with -DLSPF2F_STAGES=3 and it leaves >= 2 workgroups per CU)
#ifndef LSPF2F_STAGES
#define LSPF2F_STAGES 2   // measured: ring depth 3 is ~1 % slower than 2 on MI355X (DMA latency is not the limiter)
#endif
__host__ __device__ constexpr int igemm_stages(int bm, int bn, int g)
{
    return (LSPF2F_STAGES >= 3 && 3 * g * (bm + bn) * 128 <= 80 * 1024) ? 3 : 2;
}

// util.tensor2im (reference util/util.py:19-42) on one value: (x + 1) / 2 * 255 in float32, clip to
// [0, 255], truncate to uint8 -- the same operation order as the numpy expression it replaces.
__device__ __forceinline__ unsigned char to_u8(float v)
{
    float t = (v + 1.0f) / 2.0f * 255.0f;
    t = fminf(fmaxf(t, 0.0f), 255.0f);
    return (unsigned char)t;
}

}  // namespace lspf2f
