// gfx950: single-launch convolution for the <= 4x4 levels (pure weight streaming).  DESIGN.md section 4.2.
#include "device_common.h"
#include "kernels.h"

namespace lspf2f {

// ------------------------------------------------------------------------------------------
// Tiny-M convolution (<= 16 output pixels in the whole batch: the 4x4 and 2x2 levels at batch 1).
// These layers are pure weight streaming (9.4 MB of weights for <= 0.08 GFLOP) and were paying two
// launches each (split-K igemm + reduce, ~14 us).  Here one launch does the whole layer: every
// workgroup owns NC output channels over the FULL K, so no cross-workgroup reduction exists;
// all 256 CUs stream weight rows (issued first, one latency), the whole input tensor (<= 64 KB)
// is staged in LDS, each thread multiplies its K-slice for all pixels on the VALU, and the block
// reduces through LDS in a fixed order.  Same packed weights [Cout][tap][Cin] as the igemm.
//
// PF: the layers this kernel runs are a chain (14 launches in `large`), each waiting ~2 us for 9.4 MB of weights nobody has asked for yet.
// With PF the workgroup carries a fifth wave that does nothing but request this workgroup's share of the NEXT launch's weights (LDS-DMA into
// a 1-KB dump slot: no registers, its own vmcnt, so none of the four working waves ever waits for it), so that the next launch finds them on
// the chip -- in this XCD's L2 when the next launch maps the same rows to the same workgroup index (smallm -> smallm: block b owns rows
// 2b, 2b+1 in both), in the memory-side cache otherwise.  The wave takes part in the barriers and leaves once its requests have landed.
// MM = output pixels the instance is built for (batch folded in): 16, or 4 for the 2x2 level at batch 1, whose threads then do a quarter of the
// multiply-adds and LDS reads of the 16-pixel form.
// INF (InstanceNorm plans, fp32): the workgroup holds EVERY pixel of its channels, i.e. the whole population of InstanceNorm2d(affine=False, eps=1e-5) per (frame, channel) -- the
// normalisation (+ residual, ReLU; models/networks.py:587-590, 650-675) runs in the epilogue instead of a launch of its own (in_small): two-pass variance in double over the <= 16
// values of a frame, in pixel order.  Its own instances: the BatchNorm plans' kernel carries none of it.
template <typename T, int NC, bool PF, int MM, bool INF = false>
__global__ __launch_bounds__(PF ? 320 : 256) void conv3x3_smallm(const SmallMParams p)
{
    constexpr int NJ = 5;                                  // K/4 <= 5*256 float4 per weight row (Cin <= 512... 568)
    constexpr int RS = 264;                                // reduction row pitch (floats)
    extern __shared__ __attribute__((aligned(16))) float sm[];
    int *pixtab = reinterpret_cast<int *>(sm);             // [9][MM] source pixel of (tap, m); padding -> the zero pixel
    float *act = sm + 160;                                 // [B*Hs*Ws + 1][Cin] (last pixel = zeros); later red[MM*NC][RS]
    const int tid = threadIdx.x;
    const int n0 = blockIdx.x * NC;
    // one scalar-load round trip for the whole argument block (see igemm.hip), so nothing scalar sits between the weight requests and the
    // activation requests
    asm volatile("" :: "s"(p.src), "s"(p.w), "s"(p.scale), "s"(p.shift), "s"(p.residual), "s"(p.out), "s"(p.B), "s"(p.Hs), "s"(p.Ws), "s"(p.Ho),
                       "s"(p.Wo), "s"(p.Cin), "s"(p.Cout), "s"(p.stride), "s"(p.up), "s"(p.relu), "s"(p.M));
    const int C4 = p.Cin >> 2, K4 = 9 * C4;
    if (PF && tid >= 256) {
        typedef __attribute__((address_space(3))) float lds_float;
        const unsigned dump = (unsigned)(unsigned long long)(lds_float *)sm + p.pf_dump;
        const i32x4 srd = make_srd(p.pf, p.pf_bytes);
        const unsigned share = (p.pf_bytes / gridDim.x + 1023u) & ~1023u;       // whole 1-KB pieces; the tail past pf_bytes reads as zeros
        const unsigned lane_off = (unsigned)(tid - 256) * 16u;
        // the whole offset rides in the vector operand: that is the one the descriptor's range check sees
        for (unsigned o = 0; o < share; o += 1024u) dma16(dump, blockIdx.x * share + o + lane_off, srd, 0);
        __syncthreads();
        __syncthreads();
        __syncthreads();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // the dump slot belongs to this workgroup until the copies have landed
        return;
    }

    // 1. weights of this workgroup's channels: issue first
    float4 wv[NC][NJ];
#pragma unroll
    for (int nc = 0; nc < NC; ++nc)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int k4 = tid + 256 * j;
            wv[nc][j] = (k4 < K4 && n0 + nc < p.Cout)
                ? load4(static_cast<const T *>(p.w) + (size_t)(n0 + nc) * 9 * p.Cin + (size_t)k4 * 4)
                : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    // 2. input tensor -> LDS, (m, tap) -> source pixel table.  The epilogue's operands (folded BatchNorm, residual) of the threads that will write
    //    an output are requested here as well: asked for at the end they were a dependent L2 round trip per launch
    const int npix = p.B * p.Hs * p.Ws;
    const int nin4 = npix * C4;
    const int eo = tid >> 3, em = eo / NC, en = n0 + eo % NC;
    const bool ewrite = tid < MM * NC * 8 && (tid & 7) == 0 && em < p.M && en < p.Cout;
    float e_sc = 1.f, e_sh = 0.f, e_res = 0.f;
    if (ewrite) {
        if (p.scale) { e_sc = p.scale[en]; e_sh = p.shift[en]; }
        if (p.residual) e_res = ld1(static_cast<const T *>(p.residual) + (size_t)em * p.Cout + en);
    }
    bool dma = false;
    if constexpr (sizeof(T) == 4) {
        // fp32: LDS-DMA, 1 KB per wave and instruction, every piece in flight at once.  The register form below compiles to pairs of loads each waited for before the
        // next pair is issued (the ds_write needs the data): four dependent L2 round trips for the 32-KB tensor of a 4x4 level, inside a 6.6-us launch (round 5)
        if (!p.stage_regs) {
            dma = true;
            typedef __attribute__((address_space(3))) float lds_float;
            const unsigned lds_act = (unsigned)(unsigned long long)(lds_float *)act;
            const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane(tid >> 6), lane16 = ((unsigned)tid & 63u) * 16u;
            const unsigned total = (unsigned)nin4 * 16u;                // Cin % 256 == 0: whole 1-KB pieces
            const i32x4 srd = make_srd(p.src, total);
            for (unsigned o = wave * 1024u; o < total; o += 4096u) dma16(lds_act + o, o + lane16, srd, 0);
        }
    }
    if (!dma)
        for (int i = tid; i < nin4; i += 256)
            reinterpret_cast<float4 *>(act)[i] = load4(static_cast<const T *>(p.src) + (size_t)i * 4);   // LDS copy is fp32
    for (int i = tid; i < C4; i += 256) reinterpret_cast<float4 *>(act)[nin4 + i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < MM * 9) {
        const int t = tid / MM, m = tid - t * MM;
        int pix = npix;                                    // zero pixel: padding taps and rows past M
        if (m < p.M) {
            const int hw = p.Ho * p.Wo;
            const int b = m / hw, r = m - b * hw;
            const int oy = r / p.Wo, ox = r - oy * p.Wo;
            const int uy = oy * p.stride + t / 3 - 1, ux = ox * p.stride + t % 3 - 1;
            const int hl = p.up ? 2 * p.Hs : p.Hs, wl = p.up ? 2 * p.Ws : p.Ws;
            if (uy >= 0 && uy < hl && ux >= 0 && ux < wl)
                pix = (b * p.Hs + (p.up ? uy >> 1 : uy)) * p.Ws + (p.up ? ux >> 1 : ux);
        }
        pixtab[tid] = pix;
    }
    if (dma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's copies have landed (its weight rows too: needed next)
    __syncthreads();

    // 3. this thread's K-slice times every pixel.  k4 = tid + 256 j; a wave's 64 consecutive float4 stay
    //    inside one tap (Cin % 256 == 0), so the tap and the pixel validity are wave-uniform.
    float acc[MM][NC];
#pragma unroll
    for (int m = 0; m < MM; ++m)
#pragma unroll
        for (int nc = 0; nc < NC; ++nc) acc[m][nc] = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int k4 = tid + 256 * j;
        if (k4 >= K4) continue;
        const int tap = k4 / C4, c4 = k4 - tap * C4;
        // branch-free: 4 table reads, then 16 independent pixel reads, then the FMAs
        int pix[MM];
#pragma unroll
        for (int q = 0; q < MM / 4; ++q) {
            const int4 t4 = reinterpret_cast<const int4 *>(pixtab + tap * MM)[q];
            pix[4 * q] = t4.x; pix[4 * q + 1] = t4.y; pix[4 * q + 2] = t4.z; pix[4 * q + 3] = t4.w;
        }
        float4 a[MM];
#pragma unroll
        for (int m = 0; m < MM; ++m) a[m] = reinterpret_cast<const float4 *>(act)[pix[m] * C4 + c4];
#pragma unroll
        for (int m = 0; m < MM; ++m)
#pragma unroll
            for (int nc = 0; nc < NC; ++nc) {
                const float4 w4 = wv[nc][j];
                acc[m][nc] += a[m].x * w4.x + a[m].y * w4.y + a[m].z * w4.z + a[m].w * w4.w;
            }
    }
    __syncthreads();                                       // everyone is done with act
    // 4. block reduction: red[o][tid], then 8 threads per output sum 32 interleaved values each
    float *red = act;
#pragma unroll
    for (int m = 0; m < MM; ++m)
#pragma unroll
        for (int nc = 0; nc < NC; ++nc) red[(m * NC + nc) * RS + tid] = acc[m][nc];
    __syncthreads();
    float xraw = 0.f;
    if (tid < MM * NC * 8) {
        const int o = tid >> 3, part = tid & 7;
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) sum += red[o * RS + i * 8 + part];
        sum += __shfl_xor(sum, 1);
        sum += __shfl_xor(sum, 2);
        sum += __shfl_xor(sum, 4);
        if constexpr (INF) {
            // raw conv output (+ bias) of every (pixel, channel) of this workgroup -> LDS (the gather table's words are free since the barrier behind step 3)
            xraw = sum * e_sc + e_sh;
            if (part == 0) sm[o] = xraw;                      // xs[MM * NC]
        } else if (part == 0 && ewrite) {
            float v = sum * e_sc + e_sh + e_res;             // scale 1 / shift 0 / residual 0 where the layer has none: the same roundings as before
            if (p.relu) v = fmaxf(v, 0.f);
            st1(static_cast<T *>(p.out) + (size_t)em * p.Cout + en, v);
        }
    }
    if constexpr (INF) {
        __syncthreads();
        if (ewrite) {                                        // (tid < MM * NC * 8, part 0, a live output)
            const float *xs = sm;
            const int hw = p.Ho * p.Wo, f0 = (em / hw) * hw, nc = eo % NC;
            double s1 = 0.0;
            for (int r = 0; r < hw; ++r) s1 += (double)xs[(f0 + r) * NC + nc];
            const float m = (float)(s1 / hw);
            double s2 = 0.0;
            for (int r = 0; r < hw; ++r) { const float d = xs[(f0 + r) * NC + nc] - m; s2 += (double)(d * d); }
            const float rs = (float)(1.0 / sqrt(s2 / hw + 1e-5));
            float v = (xraw - m) * rs + e_res;
            if (p.relu) v = fmaxf(v, 0.f);
            st1(static_cast<T *>(p.out) + (size_t)em * p.Cout + en, v);
        }
    }
}

bool smallm_supported(const SmallMParams &p)
{
    const size_t act_bytes = (size_t)p.B * p.Hs * p.Ws * p.Cin * 4;
    return p.M <= 16 && p.Cin % 256 == 0 && 9 * (p.Cin / 4) <= 5 * 256 && act_bytes <= (p.dtype == 0 ? 128 : 64) * 1024 && p.Cout % 2 == 0;      // (fp32: staged by LDS-DMA, one workgroup per CU above 64 KB)
}

hipError_t launch_smallm(const SmallMParams &p, hipStream_t s)
{
    if (!smallm_supported(p)) return hipErrorInvalidValue;
    constexpr int NC = 2;
    const size_t act_bytes = ((size_t)p.B * p.Hs * p.Ws + 1) * p.Cin * 4;
    const size_t red_bytes = (size_t)16 * NC * 264 * 4;
    size_t smem = 160 * 4 + (act_bytes > red_bytes ? act_bytes : red_bytes);
    static AttrMask attr_mask;
    if (attr_needed_on_this_device(attr_mask)) {
        const int cap = 160 * 4 + 133 * 1024;       // 128 KB of input + the zero pixel + the prefetch wave's dump slot
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_smallm<float, NC, false, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_smallm<float, NC, true, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_smallm<float, NC, false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_smallm<float, NC, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_smallm<bf16_t, NC, false, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_smallm<f16_t, NC, false, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        if (e != hipSuccess) return e;
        attr_done_on_this_device(attr_mask);
    }
    const unsigned grid = (unsigned)((p.Cout + NC - 1) / NC);
    const bool four = p.M <= 4;                               // fp32 plans: the 2x2 level at batch 1
    if (p.in_fused) {                                         // InstanceNorm plans: normalisation in the epilogue (fp32; whole frames: M = B * Ho * Wo)
        if (p.dtype != 0 || p.M != p.B * p.Ho * p.Wo) return hipErrorInvalidValue;
        static AttrMask attr_in;
        if (attr_needed_on_this_device(attr_in)) {
            const int cap = 160 * 4 + 133 * 1024;
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_smallm<float, NC, false, 16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_smallm<float, NC, false, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            if (e != hipSuccess) return e;
            attr_done_on_this_device(attr_in);
        }
        if (four) hipLaunchKernelGGL((conv3x3_smallm<float, NC, false, 4, true>), dim3(grid), dim3(256), smem, s, p);
        else hipLaunchKernelGGL((conv3x3_smallm<float, NC, false, 16, true>), dim3(grid), dim3(256), smem, s, p);
        return hipGetLastError();
    }
    if (p.dtype == 2) hipLaunchKernelGGL((conv3x3_smallm<f16_t, NC, false, 16>), dim3(grid), dim3(256), smem, s, p);
    else if (p.dtype == 1) hipLaunchKernelGGL((conv3x3_smallm<bf16_t, NC, false, 16>), dim3(grid), dim3(256), smem, s, p);
    else if (p.pf != nullptr && p.pf_bytes >= 1024u * grid) {
        SmallMParams q = p;
        q.pf_dump = (unsigned)((smem + 15) & ~(size_t)15);                    // one 1-KB slot behind everything the working waves use
        smem = q.pf_dump + 1024;
        if (four) hipLaunchKernelGGL((conv3x3_smallm<float, NC, true, 4>), dim3(grid), dim3(320), smem, s, q);
        else hipLaunchKernelGGL((conv3x3_smallm<float, NC, true, 16>), dim3(grid), dim3(320), smem, s, q);
    }
    else if (four) hipLaunchKernelGGL((conv3x3_smallm<float, NC, false, 4>), dim3(grid), dim3(256), smem, s, p);
    else hipLaunchKernelGGL((conv3x3_smallm<float, NC, false, 16>), dim3(grid), dim3(256), smem, s, p);
    return hipGetLastError();
}

// Copy on the COMPUTE queue (lspf2f_memcpy): 16 bytes per lane, grid-stride.  For the render loop's transfers between pinned host memory and HBM (both are plain pointers to a
// kernel): a hipMemcpyAsync between two launches of one stream goes to the copy engine and back, two cross-queue hand-offs that the runtime resolves from a host thread -- 4-6 ms each on
// a busy host (tools/render_loop_profile.py) -- where a kernel is just the next packet of the same queue.
__global__ __launch_bounds__(256) void copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

hipError_t launch_copy16(void *dst, const void *src, size_t bytes, hipStream_t s)
{
    if (bytes == 0) return hipSuccess;
    if ((bytes & 15) || ((uintptr_t)dst & 15) || ((uintptr_t)src & 15)) return hipErrorInvalidValue;
    const size_t n16 = bytes >> 4;
    size_t blocks = (n16 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(copy16, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const uint4 *>(src), static_cast<uint4 *>(dst), n16);
    return hipGetLastError();
}

// Side branch of the forward's graph (tune key `tail_prefetch`, round 6): a few workgroups walk the byte range the <= 16x16 levels will stream (their weights do not depend on
// activations) towards the chip while the >= 32x32 levels compute, so the tail's first-byte latency is a memory-side cache hit instead of an HBM access.  Plain loads (policy 1) or
// non-temporal ones (policy 2: past the L2s of the XCDs the touching workgroups happen to sit on).  The XOR of everything read is stored only if it equals a value it cannot take.
__global__ __launch_bounds__(256) void touch_range(const uint4 *__restrict__ src, size_t n16, int nt, unsigned *sink)
{
    uint4 acc = make_uint4(0, 0, 0, 0);
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {          // four 16-byte loads per lane in flight
        typedef unsigned int v4u __attribute__((ext_vector_type(4)));
        const v4u *q = reinterpret_cast<const v4u *>(src);
        v4u a, b, c, d;
        if (nt) { a = __builtin_nontemporal_load(q + i); b = __builtin_nontemporal_load(q + i + stride); c = __builtin_nontemporal_load(q + i + 2 * stride); d = __builtin_nontemporal_load(q + i + 3 * stride); }
        else { a = q[i]; b = q[i + stride]; c = q[i + 2 * stride]; d = q[i + 3 * stride]; }
        acc.x ^= a.x ^ b.x ^ c.x ^ d.x; acc.y ^= a.y ^ b.y ^ c.y ^ d.y; acc.z ^= a.z ^ b.z ^ c.z ^ d.z; acc.w ^= a.w ^ b.w ^ c.w ^ d.w;
    }
    for (; i < n16; i += stride) { const uint4 a = src[i]; acc.x ^= a.x; acc.y ^= a.y; acc.z ^= a.z; acc.w ^= a.w; }
    if (sink && (acc.x & acc.y & acc.z & acc.w) == 0xffffffffu && (acc.x | acc.y) == 0u) *sink = acc.x;      // never true: keeps the loads
}

hipError_t launch_touch_range(const void *src, size_t bytes, int policy, int blocks, unsigned *sink, hipStream_t s)
{
    if (bytes < 16 || blocks < 1) return hipSuccess;
    if ((uintptr_t)src & 15) return hipErrorInvalidValue;
    hipLaunchKernelGGL(touch_range, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const uint4 *>(src), bytes >> 4, policy == 2 ? 1 : 0, sink);
    return hipGetLastError();
}

// bench.py's clock probe: s_memtime counts shader cycles, s_memrealtime the constant 100 MHz reference
__global__ __launch_bounds__(64) void clock_probe(unsigned long long *out, unsigned long long ticks)
{
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < ticks) {
        __builtin_amdgcn_s_sleep(64);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}

hipError_t launch_clock_probe(unsigned long long *out, unsigned duration_us, hipStream_t s)
{
    hipLaunchKernelGGL(clock_probe, dim3(1), dim3(64), 0, s, out, (unsigned long long)duration_us * 100ull);
    return hipGetLastError();
}

}  // namespace lspf2f
