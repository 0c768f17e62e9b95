// gfx950: InstanceNorm2d(affine=False, eps=1e-5) behind a convolution -- the norm_layer=nn.InstanceNorm2d variant of the
// generators (reference constructors: models/networks.py:459, :555; use_bias :494 / :590; ResidualBlock :650-668).
//
// Unlike eval-mode BatchNorm this is a run-time reduction over H*W per (frame, channel).  Three routes, chosen per layer by the
// planner (plan.cpp, InRoute):
//   fused   the igemm epilogue already produced per-wave shifted sums (xor-shuffles over the 8 lanes sharing a channel
//           quad, igemm.hip) -> in_finalize -> in_apply
//   reduce  split-K layers / extents the epilogue cannot partition: in_reduce_stats folds the partials (+ bias), writes the raw
//           tensor and 64-row partial sums -> in_finalize -> in_apply
//   small   H*W <= 1024: in_small, one workgroup per (frame, 32 channels), does fold + statistics + normalisation
// Numerics: a channel whose H*W values are nearly equal has var << mean^2, and 1/sqrt(var + 1e-5) then amplifies any error in
// var up to 316x (this is what happens at the 2x2 / 4x4 levels) -- so var is never formed as E[x^2] - mean^2 of raw values.
// Every group of <= 64 rows sums d = x - c with c = the group's first row (fp32; d is small where it matters), the groups are
// merged in double with the pairwise update of Chan et al. in a fixed order (bit-reproducible), and the one-launch route makes
// two passes (mean, then sum (x - mean)^2).  ATen accumulates these statistics in double on the CPU as well.
// The normalised tensor overwrites the raw one.
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
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1, c4 = s1;
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.splits > 1 && p.bias && rl < rl_n) bias = *reinterpret_cast<const float4 *>(p.bias + q * 4);
    if (rl < rl_n) {
        c4 = fold_row(p, ((size_t)b * p.hw + row0) * p.C + q * 4);          // the shift: this group's first row (every row lane reads it)
        if (p.splits > 1) { c4.x += bias.x; c4.y += bias.y; c4.z += bias.z; c4.w += bias.w; }
        for (int r = rl; r < 64 && row0 + r < p.hw; r += rl_n) {
            const size_t e = ((size_t)b * p.hw + row0 + r) * p.C + q * 4;
            float4 v = fold_row(p, e);
            if (p.splits > 1) {
                v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
                *reinterpret_cast<float4 *>(p.x + e) = v;
            }
            const float4 d = make_float4(v.x - c4.x, v.y - c4.y, v.z - c4.z, v.w - c4.w);
            s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
            s2.x += d.x * d.x; s2.y += d.y * d.y; s2.z += d.z * d.z; s2.w += d.w * d.w;
        }
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
        *reinterpret_cast<float4 *>(p.pshift + g) = c4;
    }
}

// running statistics of a set of values: count, mean, sum of squared deviations; merge = Chan, Golub & LeVeque's pairwise update
struct Moments { double n, mean, m2; };
__device__ __forceinline__ Moments merge(Moments a, Moments b)
{
    if (b.n == 0.0) return a;
    if (a.n == 0.0) return b;
    const double n = a.n + b.n, delta = b.mean - a.mean;
    return {n, a.mean + delta * (b.n / n), a.m2 + b.m2 + delta * delta * (a.n * b.n / n)};
}
__device__ __forceinline__ Moments group_moments(float s1, float s2, float c, double n)
{
    const double m = (double)s1 / n;                            // mean of d = x - c
    return {n, (double)c + m, (double)s2 - (double)s1 * m};     // sum (d - m)^2 = sum d^2 - (sum d)^2 / n
}

// one wave per (frame, channel quad): lanes stride over the groups, then an xor-shuffle merge tree; everything in double
__global__ __launch_bounds__(256) void in_finalize(const InstNormParams p)
{
    const int cq = p.C >> 2;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= p.B * cq) return;
    const int b = wave / cq, q = wave - b * cq;
    Moments acc[4] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    // a frame's rows are covered by hw / rows_per_group groups (x4 for the sub-pixel up-convs, whose groups are per parity and
    // cover hw / 4 rows each); only the reduce route can end on a short group
    const int full = p.hw / p.rows_per_group, tail = p.hw - full * p.rows_per_group;
    for (int g = lane; g < p.groups; g += 64) {
        const size_t o = ((size_t)b * p.groups + g) * p.C + q * 4;
        const float4 a = *reinterpret_cast<const float4 *>(p.psum + o), c = *reinterpret_cast<const float4 *>(p.psq + o);
        const float4 sh = *reinterpret_cast<const float4 *>(p.pshift + o);
        const double n = (tail && g == p.groups - 1) ? (double)tail : (double)p.rows_per_group;
        acc[0] = merge(acc[0], group_moments(a.x, c.x, sh.x, n));
        acc[1] = merge(acc[1], group_moments(a.y, c.y, sh.y, n));
        acc[2] = merge(acc[2], group_moments(a.z, c.z, sh.z, n));
        acc[3] = merge(acc[3], group_moments(a.w, c.w, sh.w, n));
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            Moments t = {shfl_xor_d(acc[k].n, o), shfl_xor_d(acc[k].mean, o), shfl_xor_d(acc[k].m2, o)};
            // both partners must compute the same value: merge in a canonical order (lower lane first)
            acc[k] = (lane & o) ? merge(t, acc[k]) : merge(acc[k], t);
        }
    if (lane == 0) {
        float m[4], r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double var = acc[k].m2 / acc[k].n;                  // biased variance, as instance_norm uses
            var = var < 0.0 ? 0.0 : var;
            m[k] = (float)acc[k].mean;
            r[k] = (float)(1.0 / sqrt(var + kInEps));
        }
        *reinterpret_cast<float4 *>(p.mean + (size_t)b * p.C + q * 4) = make_float4(m[0], m[1], m[2], m[3]);
        *reinterpret_cast<float4 *>(p.rstd + (size_t)b * p.C + q * 4) = make_float4(r[0], r[1], r[2], r[3]);
    }
}

__device__ __forceinline__ float4 normalise_v(float4 v, float4 m, float4 r, bool has_res, float4 t, int relu)
{
    v.x = (v.x - m.x) * r.x; v.y = (v.y - m.y) * r.y; v.z = (v.z - m.z) * r.z; v.w = (v.w - m.w) * r.w;
    if (has_res) { v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    return v;
}
__device__ __forceinline__ float4 normalise(float4 v, float4 m, float4 r, const float *res, int relu)
{
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (res) t = *reinterpret_cast<const float4 *>(res);
    return normalise_v(v, m, r, res != nullptr, t, relu);
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

// grid (C / 32, B), 256 threads = 32 row lanes x 8 channel quads: fold, statistics and normalisation of one 32-channel slab.
// R = rows per thread the instance holds in REGISTERS (hw <= 32 R; R = 0: the three-pass form of rounds 2-4, which re-reads the slab from memory for the variance and again
// for the normalisation).  With the rows resident the two-pass variance and the normalisation cost no further memory round trip: one read (+ the residual, requested together
// with it where the registers allow), one write -- same operations in the same order, so the same bits (round 5; `in_small_regs`).
template <int R>
__global__ __launch_bounds__(256) void in_small(const InstNormParams p)
{
    __shared__ float4 red[1][4][8];
    const int tid = threadIdx.x, q = tid & 7, rl = tid >> 3, wave = tid >> 6;
    const int b = blockIdx.y, c0 = blockIdx.x * 32 + q * 4;
    const bool live = c0 < p.C;                                // C % 32 != 0 cannot happen (ngf % 32 == 0), kept for safety
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live && p.splits > 1 && p.bias) bias = *reinterpret_cast<const float4 *>(p.bias + c0);
    // block-wide sum of a float4 per channel quad, in double: the 8 row lanes of a wave (xor-shuffle over lane bits 3..5), then the 4
    // waves through LDS in a fixed order.  Called by all 256 threads.
    auto block_sum = [&](float4 v, double (&out)[4]) {
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) { v.x += __shfl_xor(v.x, o); v.y += __shfl_xor(v.y, o); v.z += __shfl_xor(v.z, o); v.w += __shfl_xor(v.w, o); }
        __syncthreads();                                       // the previous round's readers are done with `red`
        if ((tid & 63) < 8) red[0][wave][q] = v;
        __syncthreads();
        out[0] = out[1] = out[2] = out[3] = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const float4 a = red[0][w][q]; out[0] += a.x; out[1] += a.y; out[2] += a.z; out[3] += a.w; }
    };
    constexpr int RR = R > 0 ? R : 1;
    constexpr bool PRE_RES = R > 0 && R <= 8;                  // the residual rows are requested with the tensor's own (registers allow it up to 8 rows per thread)
    float4 row[RR], resv[PRE_RES ? RR : 1];
    // pass 1: fold the split-K partials (+ bias), mean
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        if constexpr (R > 0) {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int r = rl + 32 * i;
                row[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < p.hw) {
                    const size_t e = ((size_t)b * p.hw + r) * p.C + c0;
                    row[i] = fold_row(p, e);
                    if constexpr (PRE_RES) resv[i] = p.residual ? *reinterpret_cast<const float4 *>(p.residual + e) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int i = 0; i < R; ++i)
                if (rl + 32 * i < p.hw) {
                    float4 &v = row[i];
                    if (p.splits > 1) { v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w; }
                    s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
                }
        } else {
            for (int r = rl; r < p.hw; r += 32) {
                const size_t e = ((size_t)b * p.hw + r) * p.C + c0;
                float4 v = fold_row(p, e);
                if (p.splits > 1) {
                    v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
                    *reinterpret_cast<float4 *>(p.x + e) = v;      // re-read below by the SAME thread
                }
                s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
            }
        }
    }
    double sum[4];
    block_sum(s1, sum);
    const float4 m4 = make_float4((float)(sum[0] / p.hw), (float)(sum[1] / p.hw), (float)(sum[2] / p.hw), (float)(sum[3] / p.hw));
    // pass 2: sum of squared deviations from that mean (two-pass variance: no cancellation)
    float4 s2 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        if constexpr (R > 0) {
#pragma unroll
            for (int i = 0; i < R; ++i)
                if (rl + 32 * i < p.hw) {
                    const float4 v = row[i];
                    const float4 d = make_float4(v.x - m4.x, v.y - m4.y, v.z - m4.z, v.w - m4.w);
                    s2.x += d.x * d.x; s2.y += d.y * d.y; s2.z += d.z * d.z; s2.w += d.w * d.w;
                }
        } else {
            for (int r = rl; r < p.hw; r += 32) {
                const float4 v = *reinterpret_cast<const float4 *>(p.x + ((size_t)b * p.hw + r) * p.C + c0);
                const float4 d = make_float4(v.x - m4.x, v.y - m4.y, v.z - m4.z, v.w - m4.w);
                s2.x += d.x * d.x; s2.y += d.y * d.y; s2.z += d.z * d.z; s2.w += d.w * d.w;
            }
        }
    }
    double ssq[4];
    block_sum(s2, ssq);
    if (!live) return;
    const float4 r4 = make_float4((float)(1.0 / sqrt(ssq[0] / p.hw + kInEps)), (float)(1.0 / sqrt(ssq[1] / p.hw + kInEps)),
                                  (float)(1.0 / sqrt(ssq[2] / p.hw + kInEps)), (float)(1.0 / sqrt(ssq[3] / p.hw + kInEps)));
    // pass 3: normalise (+ residual) (+ ReLU) in place
    if constexpr (R > 0) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int r = rl + 32 * i;
            if (r >= p.hw) continue;
            const size_t e = ((size_t)b * p.hw + r) * p.C + c0;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.residual) { if constexpr (PRE_RES) t = resv[i]; else t = *reinterpret_cast<const float4 *>(p.residual + e); }
            *reinterpret_cast<float4 *>(p.x + e) = normalise_v(row[i], m4, r4, p.residual != nullptr, t, p.relu);      // (the same helper as the three-pass form: the same roundings)
        }
    } else {
        for (int r = rl; r < p.hw; r += 32) {
            const size_t e = ((size_t)b * p.hw + r) * p.C + c0;
            float4 *px = reinterpret_cast<float4 *>(p.x + e);
            *px = normalise(*px, m4, r4, p.residual ? p.residual + e : nullptr, p.relu);
        }
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
    if (!in_shape_ok(p) || p.groups < 1 || p.rows_per_group < 1) return hipErrorInvalidValue;
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
    const dim3 grid((unsigned)((p.C + 31) / 32), (unsigned)p.B);
    // rows per thread held in registers (32 row lanes): the smallest instance that covers hw; beyond 1024 rows (never planned) or with in_small_regs = 0: the three-pass form
    if (p.three_pass || p.hw > 1024) hipLaunchKernelGGL(in_small<0>, grid, dim3(256), 0, s, p);
    else if (p.hw <= 32) hipLaunchKernelGGL(in_small<1>, grid, dim3(256), 0, s, p);
    else if (p.hw <= 64) hipLaunchKernelGGL(in_small<2>, grid, dim3(256), 0, s, p);
    else if (p.hw <= 256) hipLaunchKernelGGL(in_small<8>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(in_small<32>, grid, dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace lspf2f
