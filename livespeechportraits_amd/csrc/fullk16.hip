// gfx950: single-launch 3x3 convolution for the 8x8 / 4x4 / 2x2 levels of the 16-bit plans (bf16 | fp16 storage, fp32 accumulate) from 2 frames up.
// DESIGN.md section 4.5; the 16-bit twin of conv3x3_fullk (fullk.hip), which serves the same levels of the fp32 plans.
//
// BASELINE.json configs[2] (`normal`, 8 frames, bf16): these 11 layers have M = 32 .. 512 output pixels for N = 512 channels and K = 4608 (9216 behind a
// concat), so the implicit GEMM needs 8..18 K-splits to put 288..576 workgroups on the chip -- a second launch (splitk_reduce, 4.7 us each, 19 of them in
// the plan), fp32 partial slabs through HBM and a dirty-slab kernel boundary per layer: 15.4 us per 4x4 layer for 0.6 GFLOP and 4.7 MB of weights
// (profiles/r05_before_layers_normal_b8_bf16.txt).  Here ONE launch does the layer with no cross-workgroup reduction:
//   * a workgroup owns a (16 * PB pixels) x (16 channels) output tile over the FULL K; its 4 waves split K by channel eighth and meet once at the end in
//     LDS (fixed order -> bit-reproducible);
//   * v_mfma_f32_16x16x32_{bf16,f16}: one instruction consumes the 8 channels a lane holds of each operand (16 bytes), two accumulators alternate;
//   * the B operand (weights) goes global -> VGPR directly, in the tile-blocked order of pack_fullk16_weights() (one wave load = 1 KB contiguous), and a
//     whole source's share -- 9 taps x G loads of 16 B per lane -- is requested up front: every weight byte is read once per workgroup, so the stream is
//     latency-bound unless all of it is in flight (144 VGPRs at 512 channels; the kernel still fits two workgroups per CU);
//   * the A operand is the tile's band of source rows, moved by LDS-DMA (no registers) with a 16-byte pad per pixel; padding taps and rows read a zero
//     pixel, so the K loop has no branches; stride-2 bands (<= 5 rows x 16 px x 1 KB) fit LDS whole, which the fp32 kernel needed a K-split for;
//   * workgroups that share an N-slice are dealt to one XCD (same weights, one L2).
// Fused epilogue (folded BatchNorm, residual, ReLU, RNE store) as in the implicit-GEMM kernel.
// Reference semantics: Conv2d 3x3 p1 s1|s2 (+ nearest x2 upsample in front, + cat) of models/networks.py:594-595, 610-611, 663-667 under
// torch.cuda.amp.autocast (models/feature2face_G.py:28-30) for fp16 storage; bf16 storage is this repo's configs[2] (parity-unpinned, declared tolerance).
#include "device_common.h"
#include "kernels.h"

namespace lspf2f {

template <bool F16, int PB, int G, int NCH, bool WT>
__global__ __launch_bounds__(256, 2) void conv3x3_fullk16(const FullK16Params p)
{
    typedef typename St16<F16>::type T;
    constexpr int CC = G * 128;           // channels per source tensor
    constexpr int PSB = CC * 2 + 16;      // LDS bytes per band pixel (16 B pad: the 16 lanes of a ds_read_b128 group land on distinct bank slots)
    constexpr int NT = NCH * 9;           // taps over all sources
    constexpr int RD = 9;                 // weight ring depth in taps: one source's whole share in flight
    extern __shared__ __attribute__((aligned(16))) char smem16[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    // one scalar-load round trip for the whole argument block (see igemm.hip)
    asm volatile("" :: "s"(p.src0), "s"(p.src1), "s"(p.w), "s"(p.scale), "s"(p.shift), "s"(p.residual), "s"(p.out), "s"(p.B), "s"(p.Hs), "s"(p.Ws),
                       "s"(p.Ho), "s"(p.Wo), "s"(p.Cout), "s"(p.up), "s"(p.relu), "s"(p.stride), "s"(p.ntm), "s"(p.ntn), "s"(p.tiles_per_img), "s"(p.wo_log2));

    // tile: the N-slices of one XCD are contiguous, every M-tile of an N-slice lands on that XCD (blocks are dealt round-robin)
    const int x8 = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int nt = __builtin_amdgcn_readfirstlane(x8 * (p.ntn >> 3) + idx / p.ntm);
    const int mt = __builtin_amdgcn_readfirstlane(idx - (idx / p.ntm) * p.ntm);
    const int n0 = nt * 16;
    const int rpb = 16 >> p.wo_log2;                         // output rows per 16-pixel block (a 4x4 / 2x2 frame is one block, 16 / 4 of its pixels real)
    const int b = __builtin_amdgcn_readfirstlane(mt / p.tiles_per_img);
    const int r0 = (mt - b * p.tiles_per_img) * (PB * rpb);  // first output row of the tile
    const int nr = PB * rpb;
    const int S = (!p.up && p.stride == 2) ? 2 : 1;
    int sy0, sy1;                                            // band of source rows [sy0, sy1)
    if (p.up) {
        const int u0 = r0 - 1 < 0 ? 0 : r0 - 1, u1 = r0 + nr > 2 * p.Hs - 1 ? 2 * p.Hs - 1 : r0 + nr;
        sy0 = u0 >> 1; sy1 = (u1 >> 1) + 1;
    } else {
        const int lo = S * r0 - 1, hi = S * (r0 + nr - 1) + 2;
        sy0 = lo < 0 ? 0 : lo; sy1 = hi > p.Hs ? p.Hs : hi;
    }
    const int npix = (sy1 - sy0) * p.Ws;                     // the zero pixel sits at index npix

    // K assignment inside a source's CC channels, in 16-byte units (8 channels): unit(kq, wave, g) = kq * 4G + wave * G + g (any K permutation is fine as
    // long as A and B agree; the k-quad stride of 4G units = 256 B at 512 channels is what keeps a ds_read_b128 lane group conflict-free)
    const int unit0 = kq * 4 * G + wave * G;
    const int Cin = NCH * CC;
    const T *wq = static_cast<const T *>(p.w);
    bf16x8 ring[RD][G];
    auto load_b = [&](int Tt, int g) {
        if constexpr (WT) {
            const T *q = wq + ((((size_t)nt * NT + Tt) * 4 + wave) * G + g) * 512 + lane * 8;      // [nt][T][wave][g][64 lanes][8]
            ring[Tt % RD][g] = *reinterpret_cast<const bf16x8 *>(q);
        } else {
            const int ch = Tt / 9, tap = Tt - ch * 9;
            const T *q = wq + (size_t)(n0 + li) * 9 * Cin + tap * Cin + ch * CC + (unit0 + g) * 8;
            ring[Tt % RD][g] = *reinterpret_cast<const bf16x8 *>(q);
        }
    };

    // ---- the first tap's weights lead the queue (vector memory returns in order and the first MFMA needs both operands), then the activations:
    // every source's band of rows [sy0, sy1) is one contiguous NHWC range, a pixel is CC * 2 bytes = CC / 512 wave-wide DMA instructions
#pragma unroll
    for (int g = 0; g < G; ++g) load_b(0, g);
    typedef __attribute__((address_space(3))) char lds_char;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lds_char *)smem16);
    const int band = (npix + 1) * PSB;                       // bytes per source band incl. its zero pixel
    constexpr int PIECES = CC / 512;                          // 1-KB wave instructions per pixel
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const T *src = (ch ? static_cast<const T *>(p.src1) : static_cast<const T *>(p.src0)) + ((size_t)b * p.Hs + sy0) * p.Ws * CC;
        if constexpr (PIECES >= 1) {
            const i32x4 rs = make_srd(src, (unsigned)npix * CC * 2u);
            for (int px = __builtin_amdgcn_readfirstlane(wave); px < npix; px += 4)      // wave-uniform: the LDS base travels in M0
#pragma unroll
                for (int k = 0; k < PIECES; ++k)
                    dma16(lds0 + (unsigned)(ch * band + px * PSB + k * 1024), (unsigned)(px * CC * 2 + k * 1024 + lane * 16), rs, 0);
        } else {
            // 256 channels: a pixel (512 B) is shorter than one DMA piece, whose LDS destination is lane-linear -> through registers
            const int n16 = npix * (CC / 8);
            for (int i = tid; i < n16; i += 256) {
                const int px = i / (CC / 8), c8 = i - px * (CC / 8);
                *reinterpret_cast<uint4 *>(smem16 + ch * band + px * PSB + c8 * 16) = *reinterpret_cast<const uint4 *>(src + (size_t)px * CC + c8 * 8);
            }
        }
        for (int i = tid; i < CC / 8; i += 256)
            *reinterpret_cast<uint4 *>(smem16 + ch * band + npix * PSB + i * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
    // ---- then the weights of taps 1 .. RD - 2 (tap RD - 1 and the second source's taps follow one load per K step inside the loop)
#pragma unroll
    for (int Tt = 1; Tt < RD - 1; ++Tt)
#pragma unroll
        for (int g = 0; g < G; ++g) load_b(Tt, g);
    // (address arithmetic while the copies are in flight) per lane: LDS byte offset of the source pixel behind (pixel block, pixel li, tap), or of the zero pixel
    int aoff[PB][9];
    {
        const int hl = p.up ? 2 * p.Hs : p.Hs, wl = p.up ? 2 * p.Ws : p.Ws;
        const int zero = npix * PSB + unit0 * 16;
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            const int pl = pb * 16 + li;
            const int oy = r0 + (pl >> p.wo_log2), ox = pl & (p.Wo - 1);      // Wo is a power of two
            int rowoff[3], coloff[3];                                         // < 0: outside
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int uy = S * oy + d - 1, ux = S * ox + d - 1;
                rowoff[d] = (oy < p.Ho && (unsigned)uy < (unsigned)hl) ? ((p.up ? uy >> 1 : uy) - sy0) * p.Ws * PSB + unit0 * 16 : -1;
                coloff[d] = (unsigned)ux < (unsigned)wl ? (p.up ? ux >> 1 : ux) * PSB : -1;
            }
#pragma unroll
            for (int t = 0; t < 9; ++t)
                aoff[pb][t] = (rowoff[t / 3] | coloff[t % 3]) < 0 ? zero : rowoff[t / 3] + coloff[t % 3];
        }
    }
    // the DMA pieces (and tap 0) are OLDER than the loads of taps 1 .. RD - 2 and complete in order: wait until only those can be outstanding
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RD - 2) * G) : "memory");
    __syncthreads();

    f32x4acc acc[PB][2];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) { acc[pb][0] = (f32x4acc){0.f, 0.f, 0.f, 0.f}; acc[pb][1] = acc[pb][0]; }

    // ---- NT taps x G channel groups, fully unrolled: the LDS reads of step s + 1 are issued before the MFMAs of step s, ONE weight load (tap T + RD - 1,
    // same g: its slot was released by tap T - 1) after them; scheduling barriers keep the compiler from sinking either to its point of use
    bf16x8 a_cur[PB], a_nxt[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) a_cur[pb] = *reinterpret_cast<const bf16x8 *>(smem16 + aoff[pb][0]);
#pragma unroll
    for (int Tt = 0; Tt < NT; ++Tt) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int Tn = g + 1 < G ? Tt : Tt + 1, gn = g + 1 < G ? g + 1 : 0;       // the next step
            if (Tn < NT) {
                const int chn = Tn / 9, tn = Tn - chn * 9;
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) a_nxt[pb] = *reinterpret_cast<const bf16x8 *>(smem16 + chn * band + aoff[pb][tn] + gn * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 bq = ring[Tt % RD][g];
            const int par = (Tt * G + g) & 1;
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) acc[pb][par] = mfma16_16b<F16>(a_cur[pb], bq, acc[pb][par]);
            if (Tt + RD - 1 < NT) load_b(Tt + RD - 1, g);      // slot (Tt - 1) % RD: released by tap Tt - 1 (tap RD - 1 takes the one slot the prologue left free)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) a_cur[pb] = a_nxt[pb];
        }
    }

    // ---- the 4 waves' partial sums meet in LDS (the band is dead), summed in wave order; then the fused epilogue
    __syncthreads();
    float *red = reinterpret_cast<float *>(smem16);          // [4 waves][PB][4 regs][64 lanes]
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        const f32x4acc s = acc[pb][0] + acc[pb][1];
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wave * PB + pb) * 4 + r) * 64 + lane] = s[r];
    }
    __syncthreads();
    const float *sc = p.scale, *sh = p.shift;
    const T *resid = static_cast<const T *>(p.residual);
    T *outp = static_cast<T *>(p.out);
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        // thread -> (register r = wave, lane): C/D layout of the 16x16 MFMA: row (pixel) = 4 * (lane >> 4) + r, col (channel) = lane & 15
        const int r = wave;
        float v = red[((0 * PB + pb) * 4 + r) * 64 + lane];
        v += red[((1 * PB + pb) * 4 + r) * 64 + lane];
        v += red[((2 * PB + pb) * 4 + r) * 64 + lane];
        v += red[((3 * PB + pb) * 4 + r) * 64 + lane];
        const int pl = pb * 16 + 4 * kq + r;
        const int oy = r0 + (pl >> p.wo_log2), ox = pl & (p.Wo - 1);
        const int n = n0 + li;
        if (oy >= p.Ho) continue;
        const size_t o = (((size_t)b * p.Ho + oy) * p.Wo + ox) * p.Cout + n;
        if (sc) v = v * sc[n] + sh[n];
        if (resid) v += ld1<T>(resid + o);
        if (p.relu) v = fmaxf(v, 0.f);
        st1<T>(outp + o, v);
    }
}

// Host: 16-bit rows [Cout][9][Cin] (already narrowed, RNE) -> the tile-blocked operand of conv3x3_fullk16<..., WT = true>:
// [Cout/16][source][tap][wave][g][64 lanes][8], lane (li, kq) of block (nt, T, wave, g) holds channels 8 * (kq * 4G + wave * G + g) .. + 7 of row 16 nt + li.
void pack_fullk16_weights(const uint16_t *rows, int c0, int nch, int cout, uint16_t *out)
{
    const int G = c0 / 128, cin = nch * c0;
    for (int nt = 0; nt < cout / 16; ++nt)
        for (int T = 0; T < nch * 9; ++T)
            for (int w = 0; w < 4; ++w)
                for (int g = 0; g < G; ++g)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int li = lane & 15, kq = lane >> 4, ch = T / 9, tap = T % 9;
                        const uint16_t *src = rows + ((size_t)(nt * 16 + li) * 9 + tap) * cin + ch * c0 + 8 * (kq * 4 * G + w * G + g);
                        uint16_t *dst = out + (((((size_t)nt * nch * 9 + T) * 4 + w) * G + g) * 64 + lane) * 8;
                        for (int e = 0; e < 8; ++e) dst[e] = src[e];
                    }
}

static int fullk16_band_rows(const FullK16Params &p, int pb)
{
    const int S = (!p.up && p.stride == 2) ? 2 : 1;
    const int nr = pb * (16 / p.Wo);
    const int want = p.up ? nr / 2 + 2 : S * (nr - 1) + 3;
    return want < p.Hs ? want : p.Hs;
}

bool fullk16_supported(const FullK16Params &p, int pb)
{
    if (p.dtype != 1 && p.dtype != 2) return false;
    if (p.Wo != 2 && p.Wo != 4 && p.Wo != 8 && p.Wo != 16) return false;
    const int S = p.stride == 2 ? 2 : 1;
    if (S == 2 && (p.up || p.C1 != 0)) return false;                       // stride 2: one source
    if (p.Ho != p.Wo || p.Hs != p.Ws || (p.up ? 2 * p.Hs != p.Ho : p.Hs != S * p.Ho)) return false;
    if (p.C0 != 256 && p.C0 != 512) return false;                          // G = C0 / 128 in {2, 4}
    if (p.C1 != 0 && p.C1 != p.C0) return false;
    if (p.Cout % 128) return false;                                         // N-slices of 16 channels, a multiple of 8 of them
    if (pb != 1 && pb != 2) return false;
    if (pb == 2 && p.Wo != 16) return false;
    return (size_t)(p.C1 ? 2 : 1) * ((size_t)fullk16_band_rows(p, pb) * p.Ws + 1) * (p.C0 * 2 + 16) <= 150 * 1024;
}

template <bool F16, int PB, int G, int NCH, bool WT>
static hipError_t launch_fullk16_w(const FullK16Params &p, size_t smem, hipStream_t s)
{
    static AttrMask attr_mask;
    if (attr_needed_on_this_device(attr_mask)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_fullk16<F16, PB, G, NCH, WT>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return e;
        attr_done_on_this_device(attr_mask);
    }
    hipLaunchKernelGGL((conv3x3_fullk16<F16, PB, G, NCH, WT>), dim3(p.ntm * p.ntn), dim3(256), smem, s, p);
    return hipGetLastError();
}
template <bool F16, int PB, int G>
static hipError_t launch_fullk16_t(const FullK16Params &p, size_t smem, hipStream_t s)
{
    if (p.C1) return p.wtile ? launch_fullk16_w<F16, PB, G, 2, true>(p, smem, s) : launch_fullk16_w<F16, PB, G, 2, false>(p, smem, s);
    return p.wtile ? launch_fullk16_w<F16, PB, G, 1, true>(p, smem, s) : launch_fullk16_w<F16, PB, G, 1, false>(p, smem, s);
}
template <bool F16>
static hipError_t launch_fullk16_d(const FullK16Params &p, int pb, size_t smem, hipStream_t s)
{
    const int g = p.C0 / 128;
    if (pb == 2) return g == 4 ? launch_fullk16_t<F16, 2, 4>(p, smem, s) : launch_fullk16_t<F16, 2, 2>(p, smem, s);
    return g == 4 ? launch_fullk16_t<F16, 1, 4>(p, smem, s) : launch_fullk16_t<F16, 1, 2>(p, smem, s);
}

hipError_t launch_fullk16(const FullK16Params &p_in, int pb, hipStream_t s)
{
    if (!fullk16_supported(p_in, pb)) return hipErrorInvalidValue;
    FullK16Params p = p_in;
    const int nr = pb * (16 / p.Wo);
    p.tiles_per_img = (p.Ho + nr - 1) / nr;
    p.ntm = p.B * p.tiles_per_img;
    p.ntn = p.Cout / 16;
    p.wo_log2 = p.Wo == 16 ? 4 : p.Wo == 8 ? 3 : p.Wo == 4 ? 2 : 1;
    size_t smem = (size_t)(p.C1 ? 2 : 1) * ((size_t)fullk16_band_rows(p, pb) * p.Ws + 1) * (p.C0 * 2 + 16);
    const size_t red = (size_t)4 * pb * 4 * 64 * sizeof(float);
    if (smem < red) smem = red;
    return p.dtype == 2 ? launch_fullk16_d<true>(p, pb, smem, s) : launch_fullk16_d<false>(p, pb, smem, s);
}

}  // namespace lspf2f
