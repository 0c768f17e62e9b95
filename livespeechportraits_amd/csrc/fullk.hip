// gfx950: single-launch 3x3 convolution for the 16x16 / 8x8 levels at small batch (fp32 plans).  DESIGN.md section 4.2.
//
// At batch 1 these layers have M = 256 / 64 output pixels for N = 512 channels and K = 4608 (9216 with a concat input): the
// implicit-GEMM kernel needs split-K (24-72 slices) to fill the chip, which costs a second launch (splitk_reduce), 6-13 MB of
// fp32 partial slabs through HBM and a dirty-slab kernel boundary -- 16-22 us per layer for 8-30 us-equivalent of arithmetic.
// Here ONE launch does the layer with no cross-workgroup reduction: a workgroup owns a (16*PB pixels) x (16 channels) output
// tile over the FULL K,
//   * its 4 waves split K by channel quarter and meet once at the end (LDS, fixed order -> bit-reproducible),
//   * v_mfma_f32_16x16x4_f32 (exact fp32; 32-cycle issue, two accumulators per pixel block hide the 40-cycle dependent latency),
//   * the B operand (weights) goes global -> VGPR directly, three taps (24 x 16 B per lane) ahead of use: the first loads are
//     in flight while the activation band is staged, so the HBM latency of a layer's weights is paid once, not per K-tile,
//   * the A operand is the tile's band of source rows (<= 4 rows x 16 px x 512 ch = 128 KB), staged once per source tensor
//     into LDS with a 16-byte pad per pixel (16 lanes x 16 B of a ds_read_b128 group land on 16 distinct bank slots); padding
//     taps and rows read a zero pixel, so the inner loop has no branches,
//   * workgroups that share an N-slice are dealt to one XCD (same weights, one L2), which keeps HBM weight traffic at 1x.
// Grid = tiles <= 512; fused epilogue (folded BatchNorm / bias, residual, ReLU) as in the igemm kernel.
// Reference semantics: Conv2d 3x3 s1 p1 (+ nearest x2 upsample in front, + cat) of models/networks.py:610-611, 663-667.
#include "device_common.h"
#include "kernels.h"

namespace lspf2f {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PB, int G>
__global__ __launch_bounds__(256, 1) void conv3x3_fullk(const FullKParams p)
{
    constexpr int CQ = G * 16;            // channels per wave (a quarter of a source tensor's channels)
    constexpr int CC = CQ * 4;            // channels per source tensor = one staged chunk
    constexpr int PST = CC + 4;           // LDS floats per band pixel (16 B pad)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;

    // tile: the N-slices of one XCD are contiguous, every M-tile of an N-slice lands on that XCD (blocks are dealt round-robin)
    const int x8 = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int nt = x8 * (p.ntn >> 3) + idx / p.ntm, mt = idx - (idx / p.ntm) * p.ntm;
    const int n0 = nt * 16;
    const int rpb = 16 / p.Wo;                               // output rows per 16-pixel block
    const int b = mt / p.tiles_per_img;
    const int r0 = (mt - b * p.tiles_per_img) * (PB * rpb);  // first output row of the tile
    const int nr = PB * rpb;
    int sy0, sy1;                                            // band of source rows [sy0, sy1)
    if (p.up) {
        const int u0 = r0 - 1 < 0 ? 0 : r0 - 1, u1 = r0 + nr > 2 * p.Hs - 1 ? 2 * p.Hs - 1 : r0 + nr;
        sy0 = u0 >> 1; sy1 = (u1 >> 1) + 1;
    } else {
        sy0 = r0 - 1 < 0 ? 0 : r0 - 1; sy1 = r0 + nr + 1 > p.Hs ? p.Hs : r0 + nr + 1;
    }
    const int npix = (sy1 - sy0) * p.Ws;                     // zero pixel sits at index npix

    // weights of this lane: row n0 + li, K offset of its wave quarter and k-quad; layout [Cout][tap][Cin]
    const int Cin = p.C0 + p.C1;
    const float *wrow = p.w + (size_t)(n0 + li) * 9 * Cin + wave * CQ + 4 * kq;
    float4 ring[3][G];
    auto load_tap = [&](int chunk, int tap, int slot) {
        const float *q = wrow + tap * Cin + chunk * CC;
#pragma unroll
        for (int g = 0; g < G; ++g) ring[slot][g] = *reinterpret_cast<const float4 *>(q + g * 16);
    };
    load_tap(0, 0, 0);
    load_tap(0, 1, 1);
    load_tap(0, 2, 2);

    // per lane: LDS float offset of the source pixel behind (pixel block pb, pixel li, tap), or of the zero pixel
    int aoff[PB][9];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        const int pl = pb * 16 + li;
        const int oy = r0 + pl / p.Wo, ox = pl - (pl / p.Wo) * p.Wo;
        const int hl = p.up ? 2 * p.Hs : p.Hs, wl = p.up ? 2 * p.Ws : p.Ws;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int uy = oy + t / 3 - 1, ux = ox + t % 3 - 1;
            int pix = npix;
            if (oy < p.Ho && uy >= 0 && uy < hl && ux >= 0 && ux < wl)
                pix = ((p.up ? uy >> 1 : uy) - sy0) * p.Ws + (p.up ? ux >> 1 : ux);
            aoff[pb][t] = pix * PST + wave * CQ + 4 * kq;
        }
    }

    f32x4 acc[PB][2];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) { acc[pb][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[pb][1] = acc[pb][0]; }

    const int nchunk = p.C1 ? 2 : 1;
    for (int ch = 0; ch < nchunk; ++ch) {
        // ---- stage the band of source tensor `ch`: rows [sy0, sy1) are one contiguous NHWC range
        if (ch) __syncthreads();                             // every wave is done with the previous band
        {
            const float *src = (ch ? p.src1 : p.src0) + ((size_t)b * p.Hs + sy0) * p.Ws * CC;
            const int n4 = npix * (CC / 4);
            for (int i = tid; i < n4; i += 256) {
                const int px = i / (CC / 4), c4 = i - px * (CC / 4);
                *reinterpret_cast<float4 *>(smem + px * PST + c4 * 4) = *reinterpret_cast<const float4 *>(src + (size_t)i * 4);
            }
            for (int i = tid; i < CC / 4; i += 256)
                *reinterpret_cast<float4 *>(smem + npix * PST + i * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        // ---- 9 taps x G channel groups, fully unrolled; ring slot = tap % 3, refilled three taps ahead (into the next source's
        // taps at the end).  One wave per SIMD has nobody to hide latency behind, so the order is pinned by hand: the LDS reads of
        // step s+1 are issued before the MFMAs of step s, the weight loads of tap t+3 right after tap t's last MFMA, and scheduling
        // barriers keep the compiler from sinking either to its point of use.
        float4 a_cur[PB], a_nxt[PB];
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) a_cur[pb] = *reinterpret_cast<const float4 *>(smem + aoff[pb][0]);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int slot = t % 3;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int tn = g + 1 < G ? t : t + 1, gn = g + 1 < G ? g + 1 : 0;       // the next step
                if (tn < 9) {
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) a_nxt[pb] = *reinterpret_cast<const float4 *>(smem + aoff[pb][tn] + gn * 16);
                }
                __builtin_amdgcn_sched_barrier(0);
                const float4 bq = ring[slot][g];
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) {
                    // two accumulators per pixel block, alternated: a dependent 16x16x4 MFMA needs 40 cycles, an independent one 32
                    f32x4 &c0 = acc[pb][0], &c1 = acc[pb][1];
                    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[pb].x, bq.x, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[pb].y, bq.y, c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[pb].z, bq.z, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[pb].w, bq.w, c1, 0, 0, 0);
                }
                if (g == G - 1) {
                    if (t + 3 < 9) load_tap(ch, t + 3, slot);
                    else if (ch + 1 < nchunk) load_tap(ch + 1, t + 3 - 9, slot);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) a_cur[pb] = a_nxt[pb];
            }
        }
    }

    // ---- the 4 waves' partial sums meet in LDS (the band is dead), summed in wave order; then the fused epilogue
    __syncthreads();
    float *red = smem;                                        // [4 waves][PB][4 regs][64 lanes]
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        const f32x4 s = acc[pb][0] + acc[pb][1];
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wave * PB + pb) * 4 + r) * 64 + lane] = s[r];
    }
    __syncthreads();
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        // thread -> (register r = wave, lane): C/D layout of the 16x16 MFMA: row (pixel) = 4 * (lane >> 4) + r, col (channel) = lane & 15
        const int r = wave;
        float v = red[((0 * PB + pb) * 4 + r) * 64 + lane];
        v += red[((1 * PB + pb) * 4 + r) * 64 + lane];
        v += red[((2 * PB + pb) * 4 + r) * 64 + lane];
        v += red[((3 * PB + pb) * 4 + r) * 64 + lane];
        const int pl = pb * 16 + 4 * kq + r;
        const int oy = r0 + pl / p.Wo, ox = pl - (pl / p.Wo) * p.Wo;
        const int n = n0 + li;
        if (oy >= p.Ho) continue;
        const size_t o = (((size_t)b * p.Ho + oy) * p.Wo + ox) * p.Cout + n;
        if (p.scale) v = v * p.scale[n] + p.shift[n];
        if (p.residual) v += p.residual[o];
        if (p.relu) v = fmaxf(v, 0.f);
        p.out[o] = v;
    }
}

bool fullk_supported(const FullKParams &p, int pb)
{
    if (p.Wo != 2 && p.Wo != 4 && p.Wo != 8 && p.Wo != 16) return false;
    if (p.Ho != p.Wo || p.Hs != p.Ws || (p.up ? 2 * p.Hs != p.Ho : p.Hs != p.Ho)) return false;
    if (p.C0 != 128 && p.C0 != 256 && p.C0 != 512) return false;          // G = C0 / 64 in {2, 4, 8}
    if (p.C1 != 0 && p.C1 != p.C0) return false;
    if (p.Cout % 128) return false;                                         // N-slices of 16 channels, a multiple of 8 of them
    if (pb != 1 && pb != 2) return false;
    if (pb == 2 && p.Wo != 16) return false;
    // band rows: <= nr + 2 source rows
    const int nr = pb * (16 / p.Wo);
    const int rows = (p.up ? nr / 2 + 2 : nr + 2) < p.Hs ? (p.up ? nr / 2 + 2 : nr + 2) : p.Hs;
    return ((size_t)rows * p.Ws + 1) * (p.C0 + 4) * sizeof(float) <= 150 * 1024;
}

template <int PB, int G>
static hipError_t launch_fullk_t(const FullKParams &p, size_t smem, hipStream_t s)
{
    static unsigned long long attr_mask = 0;
    if (attr_needed_on_this_device(attr_mask)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_fullk<PB, G>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((conv3x3_fullk<PB, G>), dim3(p.ntm * p.ntn), dim3(256), smem, s, p);
    return hipGetLastError();
}

hipError_t launch_fullk(const FullKParams &p_in, int pb, hipStream_t s)
{
    if (!fullk_supported(p_in, pb)) return hipErrorInvalidValue;
    FullKParams p = p_in;
    const int nr = pb * (16 / p.Wo);
    p.tiles_per_img = (p.Ho + nr - 1) / nr;
    p.ntm = p.B * p.tiles_per_img;
    p.ntn = p.Cout / 16;
    const int rows = (p.up ? nr / 2 + 2 : nr + 2) < p.Hs ? (p.up ? nr / 2 + 2 : nr + 2) : p.Hs;
    size_t smem = ((size_t)rows * p.Ws + 1) * (p.C0 + 4) * sizeof(float);
    const size_t red = (size_t)4 * pb * 4 * 64 * sizeof(float);
    if (smem < red) smem = red;
    const int g = p.C0 / 64;
    if (pb == 2) {
        if (g == 8) return launch_fullk_t<2, 8>(p, smem, s);
        if (g == 4) return launch_fullk_t<2, 4>(p, smem, s);
        return launch_fullk_t<2, 2>(p, smem, s);
    }
    if (g == 8) return launch_fullk_t<1, 8>(p, smem, s);
    if (g == 4) return launch_fullk_t<1, 4>(p, smem, s);
    return launch_fullk_t<1, 2>(p, smem, s);
}

}  // namespace lspf2f
