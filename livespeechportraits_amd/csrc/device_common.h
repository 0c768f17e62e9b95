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
__device__ __forceinline__ bf16_t f2bf(float f)
{
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
// 4 consecutive channels of an activation row: load / store as float4 regardless of the storage type
template <typename T> __device__ __forceinline__ float4 load4(const T *p);
template <> __device__ __forceinline__ float4 load4<float>(const float *p) { return *reinterpret_cast<const float4 *>(p); }
template <> __device__ __forceinline__ float4 load4<bf16_t>(const bf16_t *p)
{
    const uint2 u = *reinterpret_cast<const uint2 *>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                       __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
template <typename T> __device__ __forceinline__ void store4(T *p, float4 v);
template <> __device__ __forceinline__ void store4<float>(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t *p, float4 v)
{
    uint2 u;
    u.x = (unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16);
    u.y = (unsigned)f2bf(v.z) | ((unsigned)f2bf(v.w) << 16);
    *reinterpret_cast<uint2 *>(p) = u;
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
