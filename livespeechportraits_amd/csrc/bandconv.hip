// gfx950: activation-stationary 3x3 convolution for the 512 -> 512 layers of the 16x16 and 8x8 levels in bf16 storage.  DESIGN.md section 4.7.
//
// At these levels a bf16 K-tile of the implicit GEMM is 4 MFMAs (128 cycles) per wave under ~80 instructions of copy issue, address
// arithmetic, waits and a barrier: the igemm runs them at 34 / 20 us per layer at 8 frames (9.7 / 2.4 GFLOP) plus a split-K reduce launch.
// Here nothing is staged per K-tile:
//   * a workgroup owns 32*PXB output pixels (whole rows of one frame) x 32 output channels over the FULL K = 9 * 512.  The band of source
//     rows it needs (<= 6 rows x 18 px x 512 ch = 108 KB) is copied ONCE by LDS-DMA, zero padding included (out-of-range pieces), in
//     1024-B pixel records whose 16-B chunks are XOR-swizzled on the low three bits of the chunk index;
//   * the four waves split K by channel quarter.  Each streams its 72 weight fragments (32 ch x 16 k = 1 KB per fragment, one coalesced
//     load per wave from a fragment-ordered copy in the blob) global -> VGPR, all requested up front, multiplies them with pixel fragments read
//     from the band (v_mfma_f32_32x32x16_bf16, A = weights, B = pixels), and never meets a barrier inside the K loop;
//   * the four partial tiles meet once in LDS (the band is dead by then) and are summed in wave order -> bit-reproducible; the
//     epilogue (scale / shift, residual, ReLU, bf16 rounding) writes 8 B per lane, 64 B contiguous per pixel.
// No split-K slabs, no reduce launch, one dependent launch per layer.
#include "device_common.h"
#include "kernels.h"

namespace lspf2f {

namespace {
constexpr unsigned kOOBb = 0x80000000u;
constexpr int BC_RING = 72;               // weight fragments requested ahead per wave: all of them (288 registers, one wave per SIMD anyway).  A ring
                                          // of 16 / 36 / 72 measured 18.1 / 17.1 / 16.2 us per launch in the graph (cold weights: every refill is an
                                          // HBM round trip), 4475 / 4541 / 4595 frames/s at `normal` batch 8
constexpr int BC_PATCH = 32 * 144;        // per wave and pixel block: 32 px x (128 B + 16 B pad), fp32
}  // namespace

// W = frame width (= height) of the level: 16 | 8.  PXB = pixel blocks of 32 per workgroup: a tile is TR = 32*PXB / W whole rows.
template <bool F16, int W, int PXB, bool RES, bool RELU>
__global__ __launch_bounds__(256) void bandconv512(const BandConvParams p)
{
    constexpr bool MULTI = W * W < 32 * PXB;          // 4x4 / 2x2 levels: a tile is FR whole frames (2 / 8), each with its own halo
    constexpr int FR = MULTI ? 32 * PXB / (W * W) : 1;
    constexpr int TR = MULTI ? W : 32 * PXB / W;      // output rows per tile and frame
    constexpr int BW = W + 2, BR = TR + 2;            // band extent per frame incl. halo
    constexpr int NSLOT = FR * BR * BW * 64;          // 16-B slots of the band (64 per 1024-B pixel record)
    constexpr int NPASS = (NSLOT + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef __attribute__((address_space(3))) float lds_float;
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_float *)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // K quarter: input channels wave*128 .. +127
    const int hi = lane >> 5, l31 = lane & 31;

    asm volatile("" :: "s"(p.src), "s"(p.w), "s"(p.scale), "s"(p.shift), "s"(p.residual), "s"(p.out), "s"(p.B), "s"(p.Cout), "s"(p.ntiles),
                       "s"(p.nblocks), "s"(p.div_tiles.m), "s"(p.div_tiles.s1), "s"(p.div_tiles.s2));

    // workgroup -> (channel slice, tile).  XCD-aware order: the slices of one XCD are contiguous, so every tile of a slice lands on the
    // XCD whose L2 already holds that slice's 288 KB of weights
    unsigned lin = blockIdx.x;
    {
        const unsigned total = (unsigned)p.nblocks, q = total >> 3, r = total & 7, x = lin & 7;
        lin = x * q + (x < r ? x : r) + (lin >> 3);
    }
    const int cs = (int)p.div_tiles.div(lin);
    const int tile = (int)(lin - (unsigned)cs * (unsigned)p.ntiles);
    constexpr int TPI = MULTI ? 1 : W / TR;            // tiles per frame
    const int b0 = MULTI ? tile * FR : tile / TPI;     // first frame of the tile
    const int y0 = MULTI ? 0 : (tile - b0 * TPI) * TR;
    const i32x4 srd_in = make_srd(p.src, (unsigned)(p.B * W * W) * 1024u);

    // ---- 1. band -> LDS (once).  Slot s = pass*256 + tid: band pixel bp = s >> 6 (frame bp / (BR*BW) of the tile, then row, column), slot
    // c = s & 63 holds chunk c ^ (bp & 7); the pixel is (y0 - 1 + row, column - 1) of that frame, zeros outside the frame / past the batch.
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
        const int s = q * 256 + tid, bp = s >> 6, c = s & 63;
        const int fr = bp / (BR * BW), rem = bp - fr * (BR * BW);
        const int br = rem / BW, bc = rem - br * BW;
        const int y = y0 - 1 + br, x = bc - 1, b = b0 + fr;
        const bool ok = s < NSLOT && b < p.B && (unsigned)y < (unsigned)W && (unsigned)x < (unsigned)W;
        dma16(lds0 + (unsigned)(q * 4096 + wave * 1024), ok ? (unsigned)(((b * W + y) * W + x) * 1024 + ((c ^ (bp & 7)) << 4)) : kOOBb, srd_in, 0);
    }

    // ---- 2. weight stream: fragment f = tap*8 + kc of this wave's K quarter, 1 KB per fragment, ring of BC_RING
    const bf16x8 *wp = reinterpret_cast<const bf16x8 *>(p.w) + ((size_t)(cs * 4 + wave) * 72) * 64 + lane;
    bf16x8 ring[BC_RING];
    constexpr int BC_PRE = BC_RING < 56 ? BC_RING : 56;      // requested before the band wait (vmcnt counts to 63)
#pragma unroll
    for (int f = 0; f < BC_PRE; ++f) ring[f] = wp[f * 64];

    // pixel fragment addressing: tile pixel pb*32 + l31 -> band pixel (row + ky, column + kx)
    unsigned bp0[PXB];
#pragma unroll
    for (int pb = 0; pb < PXB; ++pb) {
        const int tp = pb * 32 + l31, fr = tp / (TR * W), rem = tp - fr * (TR * W), tr = rem / W, tc = rem - tr * W;
        bp0[pb] = (unsigned)(fr * (BR * BW) + tr * BW + tc);
    }
    f32x16 acc[PXB];
#pragma unroll
    for (int pb = 0; pb < PXB; ++pb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[pb][r] = 0.f;

    // the band has landed once at most the BC_PRE weight loads issued after it are outstanding (loads retire in order): the K loop then starts on
    // the first fragments while the rest of the weights are still streaming in
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(BC_PRE) : "memory");
    __syncthreads();
#pragma unroll
    for (int f = BC_PRE; f < BC_RING; ++f) ring[f] = wp[f * 64];

    // ---- 3. K loop of this wave: 9 taps x 8 channel blocks of 16, no barrier
#pragma unroll
    for (int f = 0; f < 72; ++f) {
        const int t = f >> 3, kc = f & 7, ky = t / 3, kx = t - ky * 3;
        const bf16x8 a = ring[f % BC_RING];
        if (f + BC_RING < 72) ring[f % BC_RING] = wp[(f + BC_RING) * 64];
        const unsigned chunk = (unsigned)(wave * 16 + kc * 2 + hi);
#pragma unroll
        for (int pb = 0; pb < PXB; ++pb) {
            const unsigned bp = bp0[pb] + (unsigned)(ky * BW + kx);
            const bf16x8 bv = *reinterpret_cast<const bf16x8 *>(reinterpret_cast<const char *>(smem) + bp * 1024u + ((chunk ^ (bp & 7u)) << 4));
            acc[pb] = mfma32_16b<F16>(a, bv, acc[pb]);
        }
    }

    // ---- 4. the four K quarters meet in LDS (the band is dead once every wave is here), summed in wave order
    __syncthreads();
    float *patch = smem + (wave * PXB) * (BC_PATCH / 4);
#pragma unroll
    for (int pb = 0; pb < PXB; ++pb)
#pragma unroll
        for (int g = 0; g < 4; ++g)      // D layout: lane = pixel l31, register r = channel 8*(r >> 2) + 4*hi + (r & 3)
            *reinterpret_cast<float4 *>(reinterpret_cast<char *>(patch + pb * (BC_PATCH / 4)) + l31 * 144 + g * 32 + hi * 16) =
                make_float4(acc[pb][4 * g], acc[pb][4 * g + 1], acc[pb][4 * g + 2], acc[pb][4 * g + 3]);
    __syncthreads();
    const int px = tid >> 3, cq = tid & 7;           // this thread: pixel px of a block, channels cq*4 .. +3 of the slice
    const int n = cs * 32 + cq * 4;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.scale) { sc = *reinterpret_cast<const float4 *>(p.scale + n); sh = *reinterpret_cast<const float4 *>(p.shift + n); }
#pragma unroll
    for (int pb = 0; pb < PXB; ++pb) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) {
            const float4 t4 = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(smem) + (size_t)((w4 * PXB + pb) * BC_PATCH) + px * 144 + cq * 16);
            v.x += t4.x; v.y += t4.y; v.z += t4.z; v.w += t4.w;
        }
        v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
        const int tp = pb * 32 + px, fr = tp / (TR * W), rem = tp - fr * (TR * W), tr = rem / W, tc = rem - tr * W;
        const int b = b0 + fr;
        if (b >= p.B) continue;                        // a tile of whole frames past the batch
        const size_t e = ((size_t)(b * W + y0 + tr) * W + tc) * (size_t)p.Cout + n;
        if (RES) {
            const float4 rv = load4(static_cast<const typename St16<F16>::type *>(p.residual) + e);
            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
        }
        if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        store4(static_cast<typename St16<F16>::type *>(p.out) + e, v);
    }
}

void pack_bandconv_weights(const unsigned short *rows, unsigned short *out, int cout)
{
    // fragment (slice cs, K quarter q, tap t, kc): lane = channel cs*32 + (lane & 31), k = input channel q*128 + kc*16 + 8*(lane >> 5) .. +7 of tap t
    for (int cs = 0; cs < cout / 32; ++cs)
        for (int q = 0; q < 4; ++q)
            for (int t = 0; t < 9; ++t)
                for (int kc = 0; kc < 8; ++kc)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e)
                            out[(((((size_t)cs * 4 + q) * 9 + t) * 8 + kc) * 64 + lane) * 8 + e] =
                                rows[((size_t)(cs * 32 + (lane & 31)) * 9 + t) * 512 + q * 128 + kc * 16 + 8 * (lane >> 5) + e];
}

bool bandconv_supported(const BandConvParams &p)
{
    return p.B >= 1 && (p.W == 16 || p.W == 8 || p.W == 4 || p.W == 2) && p.Cout % 32 == 0 && p.Cout >= 32 &&
           (size_t)p.B * p.W * p.W * 1024 < 0x7fffffffull;
}

hipError_t launch_bandconv(const BandConvParams &p_in, hipStream_t s)
{
    if (!bandconv_supported(p_in)) return hipErrorInvalidValue;
    BandConvParams p = p_in;
    const int pxb = p.W == 16 ? 2 : 1;
    const int fr = p.W >= 8 ? 1 : 32 / (p.W * p.W);                  // frames per tile at the 4x4 / 2x2 levels
    const int tr = p.W >= 8 ? 32 * pxb / p.W : p.W;
    p.ntiles = p.W >= 8 ? p.B * (p.W / tr) : (p.B + fr - 1) / fr;
    p.nblocks = p.ntiles * (p.Cout / 32);
    p.div_tiles = FastDiv::make((unsigned)p.ntiles);
    const size_t band = (size_t)fr * (tr + 2) * (p.W + 2) * 1024;
    const size_t pass_bytes = ((band / 16 + 255) / 256) * 4096;      // the copy writes whole passes (filler slots land as zeros)
    const size_t red = (size_t)4 * pxb * BC_PATCH;
    const size_t smem = pass_bytes > red ? pass_bytes : red;
    typedef void (*kern_t)(const BandConvParams);
    static const kern_t kern[8][4] = {
        {bandconv512<false, 16, 2, false, false>, bandconv512<false, 16, 2, false, true>, bandconv512<false, 16, 2, true, false>, bandconv512<false, 16, 2, true, true>},
        {bandconv512<false, 8, 1, false, false>, bandconv512<false, 8, 1, false, true>, bandconv512<false, 8, 1, true, false>, bandconv512<false, 8, 1, true, true>},
        {bandconv512<false, 4, 1, false, false>, bandconv512<false, 4, 1, false, true>, bandconv512<false, 4, 1, true, false>, bandconv512<false, 4, 1, true, true>},
        {bandconv512<false, 2, 1, false, false>, bandconv512<false, 2, 1, false, true>, bandconv512<false, 2, 1, true, false>, bandconv512<false, 2, 1, true, true>},
        {bandconv512<true, 16, 2, false, false>, bandconv512<true, 16, 2, false, true>, bandconv512<true, 16, 2, true, false>, bandconv512<true, 16, 2, true, true>},
        {bandconv512<true, 8, 1, false, false>, bandconv512<true, 8, 1, false, true>, bandconv512<true, 8, 1, true, false>, bandconv512<true, 8, 1, true, true>},
        {bandconv512<true, 4, 1, false, false>, bandconv512<true, 4, 1, false, true>, bandconv512<true, 4, 1, true, false>, bandconv512<true, 4, 1, true, true>},
        {bandconv512<true, 2, 1, false, false>, bandconv512<true, 2, 1, false, true>, bandconv512<true, 2, 1, true, false>, bandconv512<true, 2, 1, true, true>}};
    static AttrMask attr_mask;
    if (attr_needed_on_this_device(attr_mask)) {
        for (int k = 0; k < 32; ++k) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern[k >> 2][k & 3]), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            if (e != hipSuccess) return e;
        }
        attr_done_on_this_device(attr_mask);
    }
    const int lvl = p.W == 16 ? 0 : p.W == 8 ? 1 : p.W == 4 ? 2 : 3;
    hipLaunchKernelGGL(kern[(p.dtype == 2 ? 4 : 0) + lvl][(p.residual ? 2 : 0) + (p.relu ? 1 : 0)], dim3(p.nblocks), dim3(256), smem, s, p);
    return hipGetLastError();
}

}  // namespace lspf2f
