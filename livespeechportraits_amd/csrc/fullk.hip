// gfx950: single-launch 3x3 convolution for the 16x16 / 8x8 levels at small batch (fp32 plans).  DESIGN.md section 4.2.
//
// At batch 1 these layers have M = 256 / 64 output pixels for N = 512 channels and K = 4608 (9216 with a concat input): the
// implicit-GEMM kernel needs split-K (24-72 slices) to fill the chip, which costs a second launch (splitk_reduce), 6-13 MB of
// fp32 partial slabs through HBM and a dirty-slab kernel boundary -- 16-22 us per layer for 8-30 us-equivalent of arithmetic.
// Here ONE launch does the layer with no cross-workgroup reduction: a workgroup owns a (16*PB pixels) x (16 channels) output
// tile over the FULL K,
//   * its 4 waves split K by channel quarter and meet once at the end (LDS, fixed order -> bit-reproducible),
//   * v_mfma_f32_16x16x4_f32 (exact fp32; 32-cycle issue, two accumulators per pixel block hide the 40-cycle dependent latency),
//   * the B operand (weights) goes global -> VGPR directly and a source tensor's whole share (72 x 16 B per lane) is requested
//     up front: every weight byte comes from HBM exactly once, so the stream is latency-bound unless nearly all of it is in
//     flight (three taps ahead measured 18 us per 16x16 layer; 288 of 512 VGPRs is the price),
//   * the A operand is the tile's band of source rows (<= 4 rows x 16 px x 512 ch = 128 KB per source), moved by LDS-DMA (no
//     registers) into LDS with a 16-byte pad per pixel (16 lanes x 16 B of a ds_read_b128 group land on 16 distinct bank
//     slots); padding taps and rows read a zero pixel, so the inner loop has no branches,
//   * workgroups that share an N-slice are dealt to one XCD (same weights, one L2), which keeps HBM weight traffic at 1x.
// Grid = tiles <= 512; fused epilogue (folded BatchNorm / bias, residual, ReLU) as in the igemm kernel.
// Reference semantics: Conv2d 3x3 s1 p1 (+ nearest x2 upsample in front, + cat) of models/networks.py:610-611, 663-667.
#include "device_common.h"
#include "kernels.h"

namespace lspf2f {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// phase timestamps for tools/time_conv.py --stamps (builds with -DLSPF2F_FULLK_STAMPS only)
#ifdef LSPF2F_FULLK_STAMPS
#define STAMP(i) do { if (p.stamps && lane == 0) p.stamps[((size_t)blockIdx.x * 4 + wave) * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

// WT: weights in the tile-blocked layout of pack_fullk_weights() (the shipped path: one wave load = 1 KB contiguous; the row layout
// [Cout][tap][Cin] costs a 16-B request per lane, 4 us per 16x16 layer) -- the row-layout variant serves the conv3x3 test hook.
// SPLIT: K in two halves over blockIdx.y (NCH == 1 then: a workgroup sees ONE source of CC channels -- source z of a concat input, or channels
// [z * CC, (z + 1) * CC) of a single source with 2 * CC channels per pixel); the second workgroup to finish a tile combines (ticket, fixed z order).
template <int PB, int G, int NCH, bool WT, bool SPLIT = false>
__global__ __launch_bounds__(256, 1) void conv3x3_fullk(const FullKParams p)
{
    static_assert(!SPLIT || NCH == 1, "a split workgroup works on one source");
    constexpr int CC = G * 64;            // channels per source tensor (per K half when SPLIT)
    constexpr int PST = CC + 4;           // LDS floats per band pixel (16 B pad)
    constexpr int NT = NCH * 9;           // taps over all sources
    constexpr int RD = 4;                 // weight ring depth in taps
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    STAMP(0);
    // one scalar-load round trip for the whole argument block (see igemm.hip): left alone the compiler fetches the source pointers in a second,
    // dependent round trip right before the first LDS-DMA piece
    asm volatile("" :: "s"(p.src0), "s"(p.src1), "s"(p.w), "s"(p.scale), "s"(p.shift), "s"(p.residual), "s"(p.out), "s"(p.B), "s"(p.Hs), "s"(p.Ws),
                       "s"(p.Ho), "s"(p.Wo), "s"(p.C0), "s"(p.C1), "s"(p.Cout), "s"(p.up), "s"(p.relu), "s"(p.ntm), "s"(p.ntn), "s"(p.tiles_per_img),
                       "s"(p.wo_log2));

    // tile: the N-slices of one XCD are contiguous, every M-tile of an N-slice lands on that XCD (blocks are dealt round-robin).
    // (integer division runs on the vector ALU: readfirstlane tells the compiler the results are wave-uniform, which the LDS-DMA
    // descriptors below need)
    const int x8 = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int nt = __builtin_amdgcn_readfirstlane(x8 * (p.ntn >> 3) + idx / p.ntm);
    const int mt = __builtin_amdgcn_readfirstlane(idx - (idx / p.ntm) * p.ntm);
    const int n0 = nt * 16;
    const int rpb = 16 >> p.wo_log2;                         // output rows per 16-pixel block
    const int b = __builtin_amdgcn_readfirstlane(mt / p.tiles_per_img);
    const int r0 = (mt - b * p.tiles_per_img) * (PB * rpb);  // first output row of the tile
    const int nr = PB * rpb;
    int sy0, sy1;                                            // band of source rows [sy0, sy1)
    if (p.up) {
        const int u0 = r0 - 1 < 0 ? 0 : r0 - 1, u1 = r0 + nr > 2 * p.Hs - 1 ? 2 * p.Hs - 1 : r0 + nr;
        sy0 = u0 >> 1; sy1 = (u1 >> 1) + 1;
    } else {
        // stride S: output rows r0 .. r0 + nr - 1 read source rows S r0 - 1 .. S (r0 + nr - 1) + 1
        const int S = p.stride == 2 ? 2 : 1;
        const int lo = S * r0 - 1, hi = S * (r0 + nr - 1) + 2;
        sy0 = lo < 0 ? 0 : lo; sy1 = hi > p.Hs ? p.Hs : hi;
    }
    const int npix = (sy1 - sy0) * p.Ws;                     // zero pixel sits at index npix

    // K assignment inside a source's CC channels, in 16-byte units (4 channels): unit(kq, wave, g) = kq * 4G + wave * G + g.  The
    // k-quad stride of 4G units is a multiple of 256 B for G >= 4, which is what keeps a ds_read_b128 lane group (it mixes k-quads:
    // lanes {0-3, 12-15, 20-27}, ...) on 16 distinct bank slots; any K permutation is fine as long as A and B agree.
    const int unit0 = kq * 4 * G + wave * G;
    const int z = SPLIT ? (int)blockIdx.y : 0;
    const int Cin = SPLIT ? 2 * CC : NCH * CC;                // input channels of the layer
    const int pstride = (SPLIT && !p.C1) ? 2 * CC : CC;       // floats between two pixels of the source this workgroup reads
    constexpr int NTW = SPLIT ? 18 : NT;                      // taps per N-slice in the weight layout
    const int T0 = SPLIT ? 9 * z : 0;
    float4 ring[RD][G];
    // tap index T runs over (source, tap); one B load = this lane's 4 channels of row n0 + li
    auto load_b = [&](int T, int g) {
        const int ch = T / 9, tap = T - ch * 9;
        if constexpr (WT) {
            const float *q = p.w + (((size_t)nt * NTW + T0 + T) * 4 + wave) * G * 256;  // [nt][T][wave][g][64 lanes][4]
            ring[T % RD][g] = *reinterpret_cast<const float4 *>(q + g * 256 + lane * 4);
        } else {
            const float *q = p.w + (size_t)(n0 + li) * 9 * Cin + tap * Cin + (SPLIT ? z : ch) * CC;
            ring[T % RD][g] = *reinterpret_cast<const float4 *>(q + (unit0 + g) * 4);
        }
    };

    // ---- the first tap's weights lead the queue (the vector-memory path returns in order, and the first MFMA needs both operands),
    // then the activations: every source's band of rows [sy0, sy1) is one contiguous NHWC range; LDS-DMA
    // moves it without registers, a pixel (CC * 4 bytes) per CC / 256 wave instructions, pixels dealt round-robin to the 4 waves
#pragma unroll
    for (int g = 0; g < G; ++g) load_b(0, g);
    typedef __attribute__((address_space(3))) float lds_float;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lds_float *)smem);
    const int band = (npix + 1) * PST;                      // floats per source band incl. its zero pixel
    constexpr int PIECES = CC / 256;                         // 1-KB wave instructions per pixel
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const float *src0z = SPLIT ? (p.C1 ? (z ? p.src1 : p.src0) : p.src0 + z * CC) : (ch ? p.src1 : p.src0);
        const float *src = src0z + ((size_t)b * p.Hs + sy0) * p.Ws * pstride;
        if constexpr (PIECES >= 1) {
            const i32x4 rs = make_srd(src, (unsigned)((npix - 1) * pstride + CC) * 4u);
            for (int px = __builtin_amdgcn_readfirstlane(wave); px < npix; px += 4)      // wave-uniform: the LDS base travels in M0
#pragma unroll
                for (int k = 0; k < PIECES; ++k)
                    dma16(lds0 + (unsigned)((ch * band + px * PST) * 4 + k * 1024), (unsigned)(px * pstride * 4 + k * 1024 + lane * 16), rs, 0);
        } else {
            // fewer than 256 channels: a pixel is shorter than one 1-KB DMA piece (whose LDS destination is lane-linear) -> through registers
            const int n4 = npix * (CC / 4);
            for (int i = tid; i < n4; i += 256) {
                const int px = i / (CC / 4), c4 = i - px * (CC / 4);
                *reinterpret_cast<float4 *>(smem + ch * band + px * PST + c4 * 4) = *reinterpret_cast<const float4 *>(src + (size_t)px * pstride + c4 * 4);
            }
        }
        for (int i = tid; i < CC / 4; i += 256)
            *reinterpret_cast<float4 *>(smem + ch * band + npix * PST + i * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    STAMP(1);
    // ---- then the weights of taps 1 .. RD - 2.  The CU's vector-memory path moves 64 B per clock, i.e. ONE 1-KB wave load per 64
    // cycles per wave with 4 waves issuing: everything requested here delays the first MFMA, so the rest of the stream is issued one
    // load per K step inside the loop, where the path is otherwise idle.
#pragma unroll
    for (int T = 1; T < RD - 1; ++T)
#pragma unroll
        for (int g = 0; g < G; ++g) load_b(T, g);
    // (address arithmetic while the copies are in flight)
    // per lane: LDS float offset of the source pixel behind (pixel block pb, pixel li, tap), or of the zero pixel.  Rows and columns are
    // classified once (3 + 3 values per pixel), a tap is then one add and one select.
    int aoff[PB][9];
    {
        const int hl = p.up ? 2 * p.Hs : p.Hs, wl = p.up ? 2 * p.Ws : p.Ws;
        const int S = (!p.up && p.stride == 2) ? 2 : 1;
        const int zero = npix * PST + unit0 * 4;
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            const int pl = pb * 16 + li;
            const int oy = r0 + (pl >> p.wo_log2), ox = pl & (p.Wo - 1);      // Wo is a power of two
            int rowoff[3], coloff[3];                                         // < 0: outside
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int uy = S * oy + d - 1, ux = S * ox + d - 1;
                rowoff[d] = (oy < p.Ho && (unsigned)uy < (unsigned)hl) ? ((p.up ? uy >> 1 : uy) - sy0) * p.Ws * PST + unit0 * 4 : -1;
                coloff[d] = (unsigned)ux < (unsigned)wl ? (p.up ? ux >> 1 : ux) * PST : -1;
            }
#pragma unroll
            for (int t = 0; t < 9; ++t)
                aoff[pb][t] = (rowoff[t / 3] | coloff[t % 3]) < 0 ? zero : rowoff[t / 3] + coloff[t % 3];
        }
    }
    STAMP(2);
    // the DMA pieces are OLDER than those loads and complete in order: wait until only they can be outstanding
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RD - 2) * G) : "memory");
    STAMP(3);
    __syncthreads();
    STAMP(4);

    f32x4 acc[PB][2];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) { acc[pb][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[pb][1] = acc[pb][0]; }

    // ---- NT taps x G channel groups, fully unrolled.  One wave per SIMD has nobody to hide latency behind, so the order is pinned by
    // hand: the LDS reads of step s+1 are issued before the MFMAs of step s, ONE weight load (tap T + RD - 1, same g) after them, and
    // scheduling barriers keep the compiler from sinking either to its point of use.
    float4 a_cur[PB], a_nxt[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) a_cur[pb] = *reinterpret_cast<const float4 *>(smem + aoff[pb][0]);
#pragma unroll
    for (int T = 0; T < NT; ++T) {
        const int ch = T / 9, t = T - ch * 9;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int Tn = g + 1 < G ? T : T + 1, gn = g + 1 < G ? g + 1 : 0;       // the next step
            if (Tn < NT) {
                const int chn = Tn / 9, tn = Tn - chn * 9;
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) a_nxt[pb] = *reinterpret_cast<const float4 *>(smem + chn * band + aoff[pb][tn] + gn * 4);
            }
            __builtin_amdgcn_sched_barrier(0);
            const float4 bq = ring[T % RD][g];
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                // two accumulators per pixel block, alternated: a dependent 16x16x4 MFMA needs 40 cycles, an independent one 32
                f32x4 &c0 = acc[pb][0], &c1 = acc[pb][1];
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[pb].x, bq.x, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[pb].y, bq.y, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[pb].z, bq.z, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[pb].w, bq.w, c1, 0, 0, 0);
            }
            if (T + RD - 1 < NT) load_b(T + RD - 1, g);      // its slot was released by tap T - 1
            if (g == G - 1 && T < 9) STAMP(5 + T);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) a_cur[pb] = a_nxt[pb];
        }
        (void)t;
    }

    STAMP(14);
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
    float vsum[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        // thread -> (register r = wave, lane): C/D layout of the 16x16 MFMA: row (pixel) = 4 * (lane >> 4) + r, col (channel) = lane & 15
        const int r = wave;
        float v = red[((0 * PB + pb) * 4 + r) * 64 + lane];
        v += red[((1 * PB + pb) * 4 + r) * 64 + lane];
        v += red[((2 * PB + pb) * 4 + r) * 64 + lane];
        v += red[((3 * PB + pb) * 4 + r) * 64 + lane];
        vsum[pb] = v;
    }
    if constexpr (SPLIT) {
        // partial tile -> slab [z][tile][pb][thread], written through (the other half may run on another XCD); every wave drains its stores, one
        // relaxed ticket per workgroup; the second arriver reads both slabs past its L1, adds them in z order and goes on to the epilogue
        // (cdna_hip_programming.md Guideline 16, counter form -- the protocol of the igemm / Winograd split-K combine)
        const __amdgpu_buffer_rsrc_t slab = __builtin_amdgcn_make_buffer_rsrc((void *)p.partial, 0, (int)((size_t)2 * gridDim.x * PB * 256 * 4), 0x00020000);
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vsum[pb]), slab, (unsigned)(((((size_t)z * gridDim.x + blockIdx.x) * PB + pb) * 256 + tid) * 4), 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned *flag = reinterpret_cast<unsigned *>(smem);
        if (tid == 0) flag[0] = __hip_atomic_fetch_add(p.tile_cnt + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (flag[0] != 1u) return;
        if (tid == 0) __hip_atomic_store(p.tile_cnt + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            const float a0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(slab, (unsigned)(((((size_t)0 * gridDim.x + blockIdx.x) * PB + pb) * 256 + tid) * 4), 0, 16));
            const float a1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(slab, (unsigned)(((((size_t)1 * gridDim.x + blockIdx.x) * PB + pb) * 256 + tid) * 4), 0, 16));
            vsum[pb] = a0 + a1;
        }
    }
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        const int r = wave;
        float v = vsum[pb];
        const int pl = pb * 16 + 4 * kq + r;
        const int oy = r0 + (pl >> p.wo_log2), ox = pl & (p.Wo - 1);
        const int n = n0 + li;
        if (oy >= p.Ho) continue;
        const size_t o = (((size_t)b * p.Ho + oy) * p.Wo + ox) * p.Cout + n;
        if (p.scale) v = v * p.scale[n] + p.shift[n];
        if (p.residual) v += p.residual[o];
        if (p.relu) v = fmaxf(v, 0.f);
        p.out[o] = v;
    }
    STAMP(15);
}

// Host: [Cout][9][Cin] rows -> the tile-blocked operand of conv3x3_fullk<..., WT = true>: [Cout/16][source][tap][wave][g][64 lanes][4],
// lane (li, kq) of block (nt, T, wave, g) holds channels 4 * (kq * 4G + wave * G + g) .. + 3 of row 16 nt + li.  One wave load = 1 KB
// contiguous, a workgroup's whole stream one contiguous 16 * 9 * Cin * 4 bytes.
void pack_fullk_weights(const float *rows, int c0, int nch, int cout, float *out)
{
    const int G = c0 / 64, cin = nch * c0;
    for (int nt = 0; nt < cout / 16; ++nt)
        for (int T = 0; T < nch * 9; ++T)
            for (int w = 0; w < 4; ++w)
                for (int g = 0; g < G; ++g)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int li = lane & 15, kq = lane >> 4, ch = T / 9, tap = T % 9;
                        const float *src = rows + ((size_t)(nt * 16 + li) * 9 + tap) * cin + ch * c0 + 4 * (kq * 4 * G + w * G + g);
                        float *dst = out + (((((size_t)nt * nch * 9 + T) * 4 + w) * G + g) * 64 + lane) * 4;
                        dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
                    }
}

bool fullk_supported(const FullKParams &p, int pb)
{
    if (p.Wo != 2 && p.Wo != 4 && p.Wo != 8 && p.Wo != 16) return false;
    const int S = p.stride == 2 ? 2 : 1;
    if (S == 2 && (p.up || p.C1 != 0 || p.split < 2)) return false;        // stride 2: one source, K-split form only
    if (p.Ho != p.Wo || p.Hs != p.Ws || (p.up ? 2 * p.Hs != p.Ho : p.Hs != S * p.Ho)) return false;
    if (p.C0 != 128 && p.C0 != 256 && p.C0 != 512) return false;          // G = C0 / 64 in {2, 4, 8}
    if (p.C1 != 0 && p.C1 != p.C0) return false;
    if (p.Cout % 128) return false;                                         // N-slices of 16 channels, a multiple of 8 of them
    if (pb != 1 && pb != 2) return false;
    if (pb == 2 && p.Wo != 16) return false;
    if (p.split > 1 && ((p.C1 == 0 && p.C0 < 256) || !p.partial || !p.tile_cnt)) return false;   // halves of >= 128 channels (G >= 2)
    // band rows: <= nr + 2 source rows
    const int nr = pb * (16 / p.Wo);
    const int want = p.up ? nr / 2 + 2 : S * (nr - 1) + 3;
    const int rows = want < p.Hs ? want : p.Hs;
    if (p.split > 1) return ((size_t)rows * p.Ws + 1) * ((p.C1 ? p.C0 : p.C0 / 2) + 4) * sizeof(float) <= 150 * 1024;      // one half-source per workgroup
    return (size_t)(p.C1 ? 2 : 1) * ((size_t)rows * p.Ws + 1) * (p.C0 + 4) * sizeof(float) <= 150 * 1024;
}

template <int PB, int G, int NCH, bool WT>
static hipError_t launch_fullk_w(const FullKParams &p, size_t smem, hipStream_t s)
{
    static AttrMask attr_mask;
    if (attr_needed_on_this_device(attr_mask)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_fullk<PB, G, NCH, WT>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return e;
        attr_done_on_this_device(attr_mask);
    }
    hipLaunchKernelGGL((conv3x3_fullk<PB, G, NCH, WT>), dim3(p.ntm * p.ntn), dim3(256), smem, s, p);
    return hipGetLastError();
}
template <int PB, int G, bool WT>
static hipError_t launch_fullk_split(const FullKParams &p, size_t smem, hipStream_t s)
{
    static AttrMask attr_mask;
    if (attr_needed_on_this_device(attr_mask)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_fullk<PB, G, 1, WT, true>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return e;
        attr_done_on_this_device(attr_mask);
    }
    hipLaunchKernelGGL((conv3x3_fullk<PB, G, 1, WT, true>), dim3(p.ntm * p.ntn, 2), dim3(256), smem, s, p);
    return hipGetLastError();
}
template <int PB>
static hipError_t launch_fullk_split_g(const FullKParams &p, int cc, size_t smem, hipStream_t s)
{
    switch (cc / 64) {
    case 8: return p.wtile ? launch_fullk_split<PB, 8, true>(p, smem, s) : launch_fullk_split<PB, 8, false>(p, smem, s);
    case 4: return p.wtile ? launch_fullk_split<PB, 4, true>(p, smem, s) : launch_fullk_split<PB, 4, false>(p, smem, s);
    case 2: return p.wtile ? launch_fullk_split<PB, 2, true>(p, smem, s) : launch_fullk_split<PB, 2, false>(p, smem, s);
    default: return hipErrorInvalidValue;
    }
}
template <int PB, int G>
static hipError_t launch_fullk_t(const FullKParams &p, size_t smem, hipStream_t s)
{
    if (p.C1) return p.wtile ? launch_fullk_w<PB, G, 2, true>(p, smem, s) : launch_fullk_w<PB, G, 2, false>(p, smem, s);
    return p.wtile ? launch_fullk_w<PB, G, 1, true>(p, smem, s) : launch_fullk_w<PB, G, 1, false>(p, smem, s);
}

hipError_t launch_fullk(const FullKParams &p_in, int pb, hipStream_t s)
{
    if (!fullk_supported(p_in, pb)) return hipErrorInvalidValue;
    FullKParams p = p_in;
    const int nr = pb * (16 / p.Wo);
    p.tiles_per_img = (p.Ho + nr - 1) / nr;
    p.ntm = p.B * p.tiles_per_img;
    p.ntn = p.Cout / 16;
    p.wo_log2 = p.Wo == 16 ? 4 : p.Wo == 8 ? 3 : p.Wo == 4 ? 2 : 1;
    const int S = p.stride == 2 ? 2 : 1;
    const int want_rows = p.up ? nr / 2 + 2 : S * (nr - 1) + 3;
    const int rows = want_rows < p.Hs ? want_rows : p.Hs;
    size_t smem = (size_t)(p.C1 ? 2 : 1) * ((size_t)rows * p.Ws + 1) * (p.C0 + 4) * sizeof(float);
    const size_t red = (size_t)4 * pb * 4 * 64 * sizeof(float);
    if (smem < red) smem = red;
    if (p.split > 1) {
        // one source of cc channels per workgroup: a concat input's source z, or half of a single source
        const int cc = p.C1 ? p.C0 : p.C0 / 2;
        smem = ((size_t)rows * p.Ws + 1) * (cc + 4) * sizeof(float);
        if (smem < red) smem = red;
        return pb == 2 ? launch_fullk_split_g<2>(p, cc, smem, s) : launch_fullk_split_g<1>(p, cc, smem, s);
    }
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
