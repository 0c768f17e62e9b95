// gfx950: InstanceNorm2d(affine=False, eps=1e-5) behind a convolution -- the norm_layer=nn.InstanceNorm2d variant of the
// generators (reference constructors: models/networks.py:459, :555; use_bias :494 / :590; ResidualBlock :650-668).
//
// Unlike eval-mode BatchNorm this is a run-time reduction over H*W per (frame, channel).  Three routes, chosen per layer by the
// planner (plan.cpp, InRoute):
//   fused   the igemm epilogue already produced per-wave sums of x and x^2 (xor-shuffles over the 8 lanes sharing a channel
//           quad, igemm.hip) -> in_finalize -> in_apply
//   reduce  split-K layers / extents the epilogue cannot partition: in_reduce_stats folds the partials (+ bias), writes the raw
//           tensor and 64-row partial sums -> in_finalize -> in_apply
//   small   H*W <= 1024: in_small, one workgroup per (frame, 32 channels), does fold + statistics + normalisation
// Partial sums are fp32 over <= 64 rows, combined in double in a fixed order (bit-reproducible); var = E[x^2] - mean^2 in double
// (ATen accumulates these statistics in double on the CPU as well).  The normalised tensor overwrites the raw one.
#include "device_common.h"
#include "kernels.h"

namespace lspf2f {

static constexpr double kInEps = 1e-5;      // nn.InstanceNorm2d default; the reference never overrides it

__device__ __forceinline__ double shfl_xor_d(double v, int o)
{
    int2 u = __builtin_bit_cast(int2, v);
    u.x = __shfl_xor(u.x, o);
    u.y = __shfl_xor(u.y, o);
    return __builtin_bit_cast(double, u);
}

__device__ __forceinline__ float4 fold_row(const InstNormParams &p, size_t e)      // element index of a channel quad: raw value
{
    if (p.splits <= 1) return *reinterpret_cast<const float4 *>(p.x + e);
    const size_t plane = (size_t)p.B * p.hw * p.C;
    float4 v = *reinterpret_cast<const float4 *>(p.partial + e);
    for (int z = 1; z < p.splits; ++z) {                       // ascending z: the order of splitk_reduce
        const float4 t = *reinterpret_cast<const float4 *>(p.partial + (size_t)z * plane + e);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    return v;
}

// grid (ceil(hw / 64), B), 256 threads: thread = (row lane, channel quad); partial sums of 64 rows per workgroup
__global__ __launch_bounds__(256) void in_reduce_stats(const InstNormParams p)
{
    __shared__ float4 red[2][256];
    const int cq = p.C >> 2;                                   // channel quads; the launcher guarantees cq <= 256
    const int rl_n = 256 / cq, tid = threadIdx.x;
    const int rl = tid / cq, q = tid - rl * cq;
    const int b = blockIdx.y, row0 = blockIdx.x * 64;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.splits > 1 && p.bias && rl < rl_n) bias = *reinterpret_cast<const float4 *>(p.bias + q * 4);
    if (rl < rl_n)
        for (int r = rl; r < 64 && row0 + r < p.hw; r += rl_n) {
            const size_t e = ((size_t)b * p.hw + row0 + r) * p.C + q * 4;
            float4 v = fold_row(p, e);
            if (p.splits > 1) {
                v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
                *reinterpret_cast<float4 *>(p.x + e) = v;
            }
            s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
            s2.x += v.x * v.x; s2.y += v.y * v.y; s2.z += v.z * v.z; s2.w += v.w * v.w;
        }
    red[0][tid] = s1;
    red[1][tid] = s2;
    __syncthreads();
    if (rl == 0) {
        for (int k = 1; k < rl_n; ++k) {                       // fixed order
            const float4 a = red[0][k * cq + q], c = red[1][k * cq + q];
            s1.x += a.x; s1.y += a.y; s1.z += a.z; s1.w += a.w;
            s2.x += c.x; s2.y += c.y; s2.z += c.z; s2.w += c.w;
        }
        const size_t g = ((size_t)b * p.groups + blockIdx.x) * p.C + q * 4;
        *reinterpret_cast<float4 *>(p.psum + g) = s1;
        *reinterpret_cast<float4 *>(p.psq + g) = s2;
    }
}

// one wave per (frame, channel quad): lanes stride over the groups, double accumulation, xor-shuffle tree
__global__ __launch_bounds__(256) void in_finalize(const InstNormParams p)
{
    const int cq = p.C >> 2;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= p.B * cq) return;
    const int b = wave / cq, q = wave - b * cq;
    double s[4] = {0, 0, 0, 0}, t[4] = {0, 0, 0, 0};
    for (int g = lane; g < p.groups; g += 64) {
        const size_t o = ((size_t)b * p.groups + g) * p.C + q * 4;
        const float4 a = *reinterpret_cast<const float4 *>(p.psum + o), c = *reinterpret_cast<const float4 *>(p.psq + o);
        s[0] += a.x; s[1] += a.y; s[2] += a.z; s[3] += a.w;
        t[0] += c.x; t[1] += c.y; t[2] += c.z; t[3] += c.w;
    }
#pragma unroll
    for (int o = 32; o; o >>= 1)
#pragma unroll
        for (int k = 0; k < 4; ++k) { s[k] += shfl_xor_d(s[k], o); t[k] += shfl_xor_d(t[k], o); }
    if (lane == 0) {
        float m[4], r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double mean = s[k] / p.hw;
            double var = t[k] / p.hw - mean * mean;             // biased variance, as instance_norm uses
            var = var < 0.0 ? 0.0 : var;
            m[k] = (float)mean;
            r[k] = (float)(1.0 / sqrt(var + kInEps));
        }
        *reinterpret_cast<float4 *>(p.mean + (size_t)b * p.C + q * 4) = make_float4(m[0], m[1], m[2], m[3]);
        *reinterpret_cast<float4 *>(p.rstd + (size_t)b * p.C + q * 4) = make_float4(r[0], r[1], r[2], r[3]);
    }
}

__device__ __forceinline__ float4 normalise(float4 v, float4 m, float4 r, const float *res, int relu)
{
    v.x = (v.x - m.x) * r.x; v.y = (v.y - m.y) * r.y; v.z = (v.z - m.z) * r.z; v.w = (v.w - m.w) * r.w;
    if (res) {
        const float4 t = *reinterpret_cast<const float4 *>(res);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    return v;
}

// streaming pass: one channel quad per thread, in place
__global__ __launch_bounds__(256) void in_apply(const InstNormParams p)
{
    const int cq = p.C >> 2;
    const size_t per_frame = (size_t)p.hw * cq;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= per_frame * p.B) return;
    const int b = (int)(i / per_frame);
    const int q = (int)(i % (size_t)cq);
    const float4 m = *reinterpret_cast<const float4 *>(p.mean + (size_t)b * p.C + q * 4);
    const float4 r = *reinterpret_cast<const float4 *>(p.rstd + (size_t)b * p.C + q * 4);
    float4 *px = reinterpret_cast<float4 *>(p.x) + i;
    *px = normalise(*px, m, r, p.residual ? p.residual + i * 4 : nullptr, p.relu);
}

// grid (C / 32, B), 256 threads = 32 row lanes x 8 channel quads: fold, statistics and normalisation of one 32-channel slab
__global__ __launch_bounds__(256) void in_small(const InstNormParams p)
{
    __shared__ float4 red[2][4][8];
    const int tid = threadIdx.x, q = tid & 7, rl = tid >> 3, wave = tid >> 6;
    const int b = blockIdx.y, c0 = blockIdx.x * 32 + q * 4;
    const bool live = c0 < p.C;                                // C % 32 != 0 cannot happen (ngf % 32 == 0), kept for safety
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live && p.splits > 1 && p.bias) bias = *reinterpret_cast<const float4 *>(p.bias + c0);
    if (live)
        for (int r = rl; r < p.hw; r += 32) {
            const size_t e = ((size_t)b * p.hw + r) * p.C + c0;
            float4 v = fold_row(p, e);
            if (p.splits > 1) {
                v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
                *reinterpret_cast<float4 *>(p.x + e) = v;      // re-read below by the SAME thread
            }
            s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
            s2.x += v.x * v.x; s2.y += v.y * v.y; s2.z += v.z * v.z; s2.w += v.w * v.w;
        }
    // the 8 row lanes of a wave that share a channel quad: xor-shuffle over lane bits 3..5, then the 4 waves through LDS
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
        s1.x += __shfl_xor(s1.x, o); s1.y += __shfl_xor(s1.y, o); s1.z += __shfl_xor(s1.z, o); s1.w += __shfl_xor(s1.w, o);
        s2.x += __shfl_xor(s2.x, o); s2.y += __shfl_xor(s2.y, o); s2.z += __shfl_xor(s2.z, o); s2.w += __shfl_xor(s2.w, o);
    }
    if ((tid & 63) < 8) { red[0][wave][q] = s1; red[1][wave][q] = s2; }
    __syncthreads();
    double s[4] = {0, 0, 0, 0}, t[4] = {0, 0, 0, 0};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float4 a = red[0][w][q], c = red[1][w][q];
        s[0] += a.x; s[1] += a.y; s[2] += a.z; s[3] += a.w;
        t[0] += c.x; t[1] += c.y; t[2] += c.z; t[3] += c.w;
    }
    float m[4], rs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double mean = s[k] / p.hw;
        double var = t[k] / p.hw - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        m[k] = (float)mean;
        rs[k] = (float)(1.0 / sqrt(var + kInEps));
    }
    if (!live) return;
    const float4 m4 = make_float4(m[0], m[1], m[2], m[3]), r4 = make_float4(rs[0], rs[1], rs[2], rs[3]);
    for (int r = rl; r < p.hw; r += 32) {
        const size_t e = ((size_t)b * p.hw + r) * p.C + c0;
        float4 *px = reinterpret_cast<float4 *>(p.x + e);
        *px = normalise(*px, m4, r4, p.residual ? p.residual + e : nullptr, p.relu);
    }
}

static bool in_shape_ok(const InstNormParams &p) { return p.B >= 1 && p.hw >= 1 && p.C >= 4 && p.C % 4 == 0; }

hipError_t launch_in_reduce_stats(const InstNormParams &p, hipStream_t s)
{
    if (!in_shape_ok(p) || p.C / 4 > 256 || p.groups != (p.hw + 63) / 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(in_reduce_stats, dim3((unsigned)p.groups, (unsigned)p.B), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_in_finalize(const InstNormParams &p, hipStream_t s)
{
    if (!in_shape_ok(p) || p.groups < 1) return hipErrorInvalidValue;
    const int waves = p.B * (p.C / 4);
    hipLaunchKernelGGL(in_finalize, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_in_apply(const InstNormParams &p, hipStream_t s)
{
    if (!in_shape_ok(p)) return hipErrorInvalidValue;
    const size_t total = (size_t)p.B * p.hw * (p.C / 4);
    hipLaunchKernelGGL(in_apply, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_in_small(const InstNormParams &p, hipStream_t s)
{
    if (!in_shape_ok(p)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(in_small, dim3((unsigned)((p.C + 31) / 32), (unsigned)p.B), dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace lspf2f
