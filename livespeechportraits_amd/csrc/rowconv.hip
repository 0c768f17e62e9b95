// gfx950: the row kernels of the bf16 plans -- weights-stationary 3x3 convolutions that sweep down the rows of a strip:
//   rowconv64    64 -> 64 channels (the 256x256 level)             described below
//   rowconv128   128 -> 128 channels (the 128x128 level)           288 weight registers per wave, shared input ring
//   rowlast128   the last conv in its GEMM form (N = 12)           16x16x32 MFMA, two 64-channel sources
//   rowup256     L1.up: Upsample x2 + conv over 2 x 128 channels   sub-pixel form, one output parity per wave
// All of them keep their weights in registers for the life of a workgroup, bring every input pixel into the CU once by LDS-DMA, feed several
// MFMAs from one LDS fragment read (the tap rows of consecutive output rows), and wait on counted vmcnt values.
//
// rowconv64.
// In bf16 these layers are HBM-bound (algorithmic: 128 B in + 128 B out per pixel for 73.7 kFLOP; the MFMA time is half the transfer time
// at 8 frames), and the implicit-GEMM kernel runs them at a fifth of the HBM roofline because im2col pulls every input pixel through the
// CU's vector-memory path nine times.  Here every input pixel enters the CU ONCE:
//   * a workgroup owns a strip of R output rows x 64 pixels; wave (pg, nb) computes pixels pg*32 .. +31 x channels nb*32 .. +31 of every row
//     and sweeps down the R + 2 input rows it needs.  Each wave copies ITS 34-pixel window of an input row global -> LDS by LDS-DMA into a
//     private ring (RC_PF rows ahead; 128-B pixel records, 16-B chunks XOR-swizzled exactly as the igemm's K-tiles).  Private rings cost the
//     second wave of a pixel group a re-read (an L1 hit), and buy a kernel without a single barrier: a wave waits on its own vmcnt only, and
//     the first fragments of the next row are read while the last MFMAs of this row run;
//   * the whole weight tensor (64 x 576 bf16 = 72 KB) lives in REGISTERS: a wave holds the 36 A-fragments of its 32 output channels
//     (144 VGPRs) for the lifetime of the workgroup and multiplies them with pixel fragments read from its ring
//     (v_mfma_f32_32x32x16_bf16, A = weights [32 ch x 16 k], B = pixels [16 k x 32 px]);
//   * one fragment read of input row i feeds three MFMAs -- tap rows ky = 0, 1, 2 go to the accumulators of output rows i, i-1, i-2 --
//     so LDS traffic is a third of a tap-by-tap loop; accumulators rotate by unrolling the row loop three times;
//   * when input row i has been consumed output row i-2 is complete: the wave parks its 32 px x 32 ch tile (fp32) in a private LDS patch
//     and finishes it DURING the next step, between that step's MFMAs (one wave per SIMD: nothing else would cover the epilogue): read
//     the patch transposed (a lane then owns 8 consecutive channels of a pixel), add the residual row (fetched by LDS-DMA into a per-wave
//     ring RC_PF steps earlier), scale / shift / ReLU in the igemm epilogue's operation order, store 16 B per lane.
// Every wave issues the same number of vector-memory operations per row step (5 input pieces, 2 residual pieces, 2 stores; out-of-range
// ones are addressed outside their buffer and dropped by the hardware), so the LDS-DMA waits are counted vmcnt waits (loads and stores
// retire in order on gfx9).  Same packed weights [Cout][9][Cin] as the igemm; same accumulation order (ky, kx, channel block) -> same bits.
#include "device_common.h"
#include "kernels.h"

namespace lspf2f {

namespace {
typedef short i16x2 __attribute__((ext_vector_type(2)));
constexpr int RC_TW = 64;                 // output pixels per strip row
constexpr int RC_ROWB = 5120;             // bytes per ring row of a wave: 40 pixel records of 128 B = 5 DMA pieces (34 are real)
constexpr int RC_NR = 5;                  // ring rows per wave; rows are fetched RC_NR - 1 steps ahead
constexpr int RC_PF = RC_NR - 1;
constexpr int RC_RES_SLOT = 2048;         // per-wave residual slot: 32 px x 64 B
constexpr int RC_RES_NR = RC_PF + 1;      // a residual row is fetched RC_PF steps before its epilogue
constexpr int RC_PATCH = 32 * 144;        // per-wave epilogue patch: 32 px x (128 B + 16 B pad), fp32
constexpr int RC_IN = 5, RC_RP = 2, RC_ST = 2;          // vector-memory operations per step and wave: input pieces, residual pieces, stores
constexpr int RC_OPS = RC_IN + RC_RP + RC_ST;
constexpr unsigned kOOB = 0x80000000u;

// Counted-wait bound for the rows issued by the prologue (RC_PF statements, statement k = input row k then residual row k - 3).  Row k
// is waited for after k complete steps; the bound is the number of operations issued after it (and after residual row k - 3, which the
// same step finishes), minimised over k < RC_PF (a smaller count only waits longer).
constexpr int rc_early_wait()
{
    int best = 1 << 20;
    for (int k = 0; k < RC_PF; ++k) {
        const int after_res = (RC_IN + RC_RP) * (RC_PF - 1 - k) + RC_OPS * k;      // the residual pieces close statement k
        if (after_res < best) best = after_res;
    }
    return best;
}
static_assert(RC_PF >= 3 && (RC_PF - 1) * RC_OPS < 64, "vmcnt is a 6-bit counter");

template <int N> __device__ __forceinline__ void vm_wait()
{
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
}  // namespace

template <bool F16, bool RES, bool RELU>
__global__ __launch_bounds__(256) void rowconv64(const RowConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef __attribute__((address_space(3))) float lds_float;
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_float *)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = wave & 1, pg = wave >> 1;           // this wave: output channels nb*32.., pixels pg*32.. of the strip
    const int hi = lane >> 5, l31 = lane & 31;

    asm volatile("" :: "s"(p.src), "s"(p.w), "s"(p.scale), "s"(p.shift), "s"(p.residual), "s"(p.out), "s"(p.B), "s"(p.H), "s"(p.W), "s"(p.R),
                       "s"(p.relu), "s"(p.wfrag), "s"(p.nsx), "s"(p.nsy), "s"(p.nblocks), "s"(p.div_sx.m), "s"(p.div_sx.s1), "s"(p.div_sx.s2),
                       "s"(p.div_sy.m), "s"(p.div_sy.s1), "s"(p.div_sy.s2));

    // strip of this workgroup; XCD-aware order: each XCD (private L2) works on vertically adjacent strips, which share their halo rows
    unsigned lin = blockIdx.x;
    {
        const unsigned total = (unsigned)p.nblocks, q = total >> 3, r = total & 7, x = lin & 7;
        lin = x * q + (x < r ? x : r) + (lin >> 3);
    }
    const unsigned t1 = p.div_sx.div(lin);
    const int sx = (int)(lin - t1 * (unsigned)p.nsx);
    const int b = (int)p.div_sy.div(t1);
    const int sy = (int)(t1 - (unsigned)b * (unsigned)p.nsy);
    const int x0 = sx * RC_TW, y0 = sy * p.R;
    const unsigned imgbytes = (unsigned)(p.H * p.W) * 128u;
    const i32x4 srd_in = make_srd(static_cast<const char *>(p.src) + (size_t)b * imgbytes, imgbytes);
    const i32x4 srd_res = make_srd(RES ? static_cast<const char *>(p.residual) + (size_t)b * imgbytes : static_cast<const char *>(p.src), RES ? imgbytes : 0u);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(static_cast<char *>(p.out) + (size_t)b * imgbytes, 0, (int)imgbytes, 0x00020000);

    // LDS map: per wave [RC_NR] input rows | per wave [RC_RES_NR] residual slots | per wave patch
    const unsigned lds_ring = lds0 + (unsigned)wave * (RC_NR * RC_ROWB);
    const unsigned lds_res = lds0 + 4 * RC_NR * RC_ROWB + (unsigned)wave * (RC_RES_NR * RC_RES_SLOT);
    float *patch = smem + (4 * RC_NR * RC_ROWB + 4 * RC_RES_NR * RC_RES_SLOT) / 4 + wave * (RC_PATCH / 4);

    // ---- DMA roles.  Input row: piece q moves record slots q*64 + lane (a slot = one 16-B chunk of a pixel record; record r = pixel
    // x0 + pg*32 - 1 + r); the lane fetches the chunk that belongs in its slot under the swizzle  slot c of record r <- chunk c ^ ((r >> 1) & 7).
    unsigned in_col[RC_IN];                  // byte offset of (x, chunk) inside an image row, or kOOB (padding column / filler record)
#pragma unroll
    for (int q = 0; q < RC_IN; ++q) {
        const int sl = q * 64 + lane, r = sl >> 3, c = sl & 7;
        const int x = x0 + pg * 32 - 1 + r;
        in_col[q] = (r < 34 && (unsigned)x < (unsigned)p.W) ? (unsigned)(x * 128 + ((c ^ ((r >> 1) & 7)) << 4)) : kOOB;
    }
    // Residual row (private ring): piece q, lane i <- pixel 16q + (i >> 2), 16-B chunk (i & 3) of this wave's 64 B per pixel --
    // the ownership the epilogue has after its transpose (lane i: pixel 16*pass + (i >> 2), channels (i & 3)*8 .. +7), so it reads slot + lane*16.
    unsigned res_col[RC_RP];
#pragma unroll
    for (int q = 0; q < RC_RP; ++q)
        res_col[q] = (unsigned)((x0 + pg * 32 + 16 * q + (lane >> 2)) * 128 + nb * 64 + (lane & 3) * 16);
    const unsigned rowbytes = (unsigned)p.W * 128u;

    // The 5 + 2 pieces of one step in ONE statement: M0 (LDS base of an LDS-DMA) is compiler-reserved, so it is saved and restored once
    // per step instead of once per piece (7 pieces: 23 instead of 35 issue slots, and this kernel is issue-bound, see the end of the header)
    auto dma_step = [&](int i, int j) {      // input row i of the strip (image row y0 - 1 + i) -> ring slot i % RC_NR; residual row j -> slot j % RC_RES_NR
        const int gyi = y0 - 1 + i, gyj = y0 + j;
        const bool oki = (unsigned)gyi < (unsigned)p.H && i < p.R + 2;
        const bool okj = RES && j >= 0 && j < p.R && gyj < p.H;
        const int soffi = oki ? gyi * (int)rowbytes : 0, soffj = okj ? gyj * (int)rowbytes : 0;
        const unsigned basei = lds_ring + (unsigned)(i % RC_NR) * RC_ROWB;
        const unsigned basej = lds_res + (unsigned)(((j % RC_RES_NR) + RC_RES_NR) % RC_RES_NR) * RC_RES_SLOT;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %1\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_dwordx4 %3, %8, %9 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x400\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_dwordx4 %4, %8, %9 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x400\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_dwordx4 %5, %8, %9 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x400\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_dwordx4 %6, %8, %9 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x400\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_dwordx4 %7, %8, %9 offen lds\n\t"
                     "s_mov_b32 m0, %2\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_dwordx4 %10, %12, %13 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x400\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_dwordx4 %11, %12, %13 offen lds\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(basei), "s"(basej),
                       "v"(oki ? in_col[0] : kOOB), "v"(oki ? in_col[1] : kOOB), "v"(oki ? in_col[2] : kOOB), "v"(oki ? in_col[3] : kOOB), "v"(oki ? in_col[4] : kOOB),
                       "s"(srd_in), "s"(soffi),
                       "v"(okj ? res_col[0] : kOOB), "v"(okj ? res_col[1] : kOOB), "s"(srd_res), "s"(soffj)
                     : "memory", "scc");
    };
    static_assert(RC_IN == 5 && RC_RP == 2, "dma_step is written out for 5 + 2 pieces");

    // ---- weights -> registers: A-fragment (tap, kc) of this wave's 32 channels: lane = channel nb*32 + l31, k = kc*16 + 8*hi .. +7
    bf16x8 wf[9][4];
    if (p.wfrag) {                           // fragment order: 1 KB per instruction, coalesced (72 KB per workgroup in ~1200 cycles)
        const bf16x8 *wp = reinterpret_cast<const bf16x8 *>(p.w) + (size_t)nb * 36 * 64 + lane;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc)
                wf[t][kc] = wp[(t * 4 + kc) * 64];
    } else {                                 // row layout [64][9][64] (single-layer entry point): 16 B per 1152-B row and lane
        const bf16_t *wrow = static_cast<const bf16_t *>(p.w) + (size_t)(nb * 32 + l31) * 576 + hi * 8;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc)
                wf[t][kc] = *reinterpret_cast<const bf16x8 *>(wrow + t * 64 + kc * 16);
    }
    // epilogue constants of this lane's 8 channels (after the transpose): nb*32 + (lane & 3)*8 ..
    float sc[8], sh[8];
    {
        const int c0 = nb * 32 + (lane & 3) * 8;
#pragma unroll
        for (int t = 0; t < 8; ++t) { sc[t] = p.scale ? p.scale[c0 + t] : 1.f; sh[t] = p.scale ? p.shift[c0 + t] : 0.f; }
    }
    // make sure the weights have landed before the counted waits start (they count every vector-memory operation of the wave)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // B-fragment addresses inside a ring row: pixel record r = l31 + kx, chunk kc*2 + hi, swizzled
    unsigned boff[3][4];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int r = l31 + kx;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) boff[kx][kc] = (unsigned)(r * 128 + (((kc * 2 + hi) ^ ((r >> 1) & 7)) << 4));
    }
    auto frag = [&](int i, int f) {          // fragment f = kx*4 + kc of input row i
        return *reinterpret_cast<const bf16x8 *>(reinterpret_cast<const char *>(smem) + (lds_ring - lds0) + (unsigned)(i % RC_NR) * RC_ROWB + boff[f >> 2][f & 3]);
    };

    f32x16 acc0, acc1, acc2;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; }

    // second half of the epilogue of output row j (its tile sits in the patch): one pass = 16 pixels of the tile
    auto finish = [&](int j, int pass) {
        const bool live = j >= 0 && j < p.R && y0 + j < p.H;
        const unsigned resbase = (lds_res - lds0) + (unsigned)(((j % RC_RES_NR) + RC_RES_NR) % RC_RES_NR) * RC_RES_SLOT;
        const int px = 16 * pass + (lane >> 2);
        const char *src = reinterpret_cast<const char *>(patch) + px * 144 + (lane & 3) * 32;
        const float4 v0 = *reinterpret_cast<const float4 *>(src), v1 = *reinterpret_cast<const float4 *>(src + 16);
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = v[t] * sc[t] + sh[t];
        if (RES) {
            const u32x4 rv = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(smem) + resbase + pass * 1024 + lane * 16);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                v[2 * t] += lo16<F16>(rv[t]);
                v[2 * t + 1] += hi16<F16>(rv[t]);
            }
        }
        // ReLU on the rounded pair: a bf16 is negative iff it is negative as a 16-bit integer, and rounding keeps sign and zero, so
        // max(., 0) on the packed halves (v_pk_max_i16) equals rounding fmaxf(v, 0) -- one instruction per pair instead of two
        unsigned o0[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            unsigned w = pack16x2<F16>(v[2 * t], v[2 * t + 1]);
            if (RELU) w = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, w), (i16x2){0, 0}));
            o0[t] = w;
        }
        const u32x4 o = {o0[0], o0[1], o0[2], o0[3]};
        const unsigned off = live ? (unsigned)(((y0 + j) * p.W + x0 + pg * 32 + px) * 128 + nb * 64 + (lane & 3) * 16) : kOOB;
        __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, off, 0, 0);
    };

    // One step = input row i: its three tap rows go to output rows i (an, started here), i-1 (am) and i-2 (ao, complete after this step);
    // the tile of output row i-3, parked in the patch by the previous step, is finished between the MFMAs.  The fragment registers are a
    // ring of 3 read two fragments ahead; the first two fragments of a row are read by the PREVIOUS step (f0, f1), after it has waited for
    // the row: input row i+1 and residual row i-2 were both issued RC_PF - 1 steps before this one, so once this step has issued all its
    // own operations, at most the operations of the last RC_PF - 1 steps may still be in flight.
    bf16x8 f0, f1;
    auto step = [&](int i, f32x16 &an, f32x16 &am, f32x16 &ao) {
        dma_step(i + RC_PF, i + RC_PF - 3);
        bf16x8 bfr[3];
        bfr[0] = f0; bfr[1] = f1;
#pragma unroll
        for (int f = 0; f < 12; ++f) {
            if (f + 2 < 12) bfr[(f + 2) % 3] = frag(i, f + 2);
            const int kx = f >> 2, kc = f & 3;
            ao = mfma32_16b<F16>(wf[6 + kx][kc], bfr[f % 3], ao);
            am = mfma32_16b<F16>(wf[3 + kx][kc], bfr[f % 3], am);
            an = mfma32_16b<F16>(wf[0 + kx][kc], bfr[f % 3], f == 0 ? zero16 : an);   // output row i starts here: C = 0
            if (f == 1) finish(i - 3, 0);
            if (f == 5) finish(i - 3, 1);
            if (f == 8) {
                if (i + 1 < RC_PF) vm_wait<rc_early_wait()>(); else vm_wait<(RC_PF - 1) * RC_OPS>();
            }
            if (f == 9) f0 = frag(i + 1, 0);
            if (f == 10) f1 = frag(i + 1, 1);
        }
        // park the tile of output row i - 2.  D layout: lane = pixel l31, register r = channel 8*(r >> 2) + 4*hi + (r & 3); the patch is
        // pixel-major fp32, so reading it back 32 B per lane transposes the tile
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4 *>(reinterpret_cast<char *>(patch) + l31 * 144 + g * 32 + hi * 16) = make_float4(ao[4 * g], ao[4 * g + 1], ao[4 * g + 2], ao[4 * g + 3]);
    };

    // ---- prologue: the first RC_PF input rows and the residual rows whose epilogue comes within the first RC_PF steps
    // (output row j is complete after step j + 2 and finished during step j + 3, so its residual is fetched at step j + 3 - RC_PF)
    // (issued as RC_PF full steps: the residual half of a statement that has no row to fetch is addressed out of range, like every filler)
#pragma unroll
    for (int i = 0; i < RC_PF; ++i) dma_step(i, i - 3);
    vm_wait<rc_early_wait()>();
    f0 = frag(0, 0); f1 = frag(0, 1);

    const int nsteps = p.R + 2;               // rounded up to a multiple of 3 by the rotation; the extra steps move zeros and store nothing
    int i = 0;
    for (; i < nsteps; i += 3) {
        step(i, acc0, acc2, acc1);
        step(i + 1, acc1, acc0, acc2);
        step(i + 2, acc2, acc1, acc0);
    }
    // the tile parked by the last step
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    finish(i - 3, 0);
    finish(i - 3, 1);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same idea for the 128 -> 128 layers (the 128x128 level).  72 A-fragments per wave (9 taps x 8 channel blocks = 288 registers: the kernel
// runs one wave per SIMD on the 512-register budget), wave = one of the 4 blocks of 32 output channels, all four waves on the SAME 32
// pixels of a row.  The 34-pixel input window (256-B pixel records) is therefore shared: one ring for the workgroup, each wave copying a
// quarter of a row, one barrier per row step; residual rows and the epilogue are per wave exactly as above.
namespace {
constexpr int RD_TW = 32;                 // output pixels per strip row
constexpr int RD_PITCH = 12288;           // bytes per ring row: 48 pixel records of 256 B = 3 DMA passes of the workgroup (34 are real)
constexpr int RD_NR = 6;                  // ring rows; rows are fetched RD_NR - 1 steps ahead (4 / 6 / 7 rows: 49.3 / 42.2 / 43.3 us per launch at 8 frames)
constexpr int RD_PF = RD_NR - 1;
constexpr int RD_RES_NR = RD_PF + 1;
constexpr int RD_IN = 3, RD_RP = 2, RD_ST = 2;
constexpr int RD_OPS = RD_IN + RD_RP + RD_ST;
static_assert((RD_PF - 1) * RD_OPS < 64, "vmcnt is a 6-bit counter");
}  // namespace

template <bool F16, bool RES, bool RELU>
__global__ __launch_bounds__(256) void rowconv128(const RowConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef __attribute__((address_space(3))) float lds_float;
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_float *)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int nb = __builtin_amdgcn_readfirstlane(tid >> 6);      // this wave: output channels nb*32 .. +31
    const int hi = lane >> 5, l31 = lane & 31;

    asm volatile("" :: "s"(p.src), "s"(p.w), "s"(p.scale), "s"(p.shift), "s"(p.residual), "s"(p.out), "s"(p.B), "s"(p.H), "s"(p.W), "s"(p.R),
                       "s"(p.relu), "s"(p.wfrag), "s"(p.nsx), "s"(p.nsy), "s"(p.nblocks), "s"(p.div_sx.m), "s"(p.div_sx.s1), "s"(p.div_sx.s2),
                       "s"(p.div_sy.m), "s"(p.div_sy.s1), "s"(p.div_sy.s2));

    unsigned lin = blockIdx.x;
    {
        const unsigned total = (unsigned)p.nblocks, q = total >> 3, r = total & 7, x = lin & 7;
        lin = x * q + (x < r ? x : r) + (lin >> 3);
    }
    const unsigned t1 = p.div_sx.div(lin);
    const int sx = (int)(lin - t1 * (unsigned)p.nsx);
    const int b = (int)p.div_sy.div(t1);
    const int sy = (int)(t1 - (unsigned)b * (unsigned)p.nsy);
    const int x0 = sx * RD_TW, y0 = sy * p.R;
    const unsigned imgbytes = (unsigned)(p.H * p.W) * 256u;
    const i32x4 srd_in = make_srd(static_cast<const char *>(p.src) + (size_t)b * imgbytes, imgbytes);
    const i32x4 srd_res = make_srd(RES ? static_cast<const char *>(p.residual) + (size_t)b * imgbytes : static_cast<const char *>(p.src), RES ? imgbytes : 0u);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(static_cast<char *>(p.out) + (size_t)b * imgbytes, 0, (int)imgbytes, 0x00020000);

    // LDS map: [RD_NR] shared input rows | per wave [RD_RES_NR] residual slots | per wave patch
    const unsigned lds_ring = lds0;
    const unsigned lds_res = lds0 + RD_NR * RD_PITCH + (unsigned)nb * (RD_RES_NR * RC_RES_SLOT);
    float *patch = smem + (RD_NR * RD_PITCH + 4 * RD_RES_NR * RC_RES_SLOT) / 4 + nb * (RC_PATCH / 4);

    // DMA roles.  Input row: pass q moves record slots q*256 + tid (16 slots per 256-B record; record r = pixel x0 - 1 + r), swizzled in the low
    // three bits of the chunk index like the 128-B records of the other kernels: slot c of record r <- chunk c ^ ((r >> 1) & 7).
    unsigned in_col[RD_IN];
#pragma unroll
    for (int q = 0; q < RD_IN; ++q) {
        const int sl = q * 256 + tid, r = sl >> 4, c = sl & 15;
        const int x = x0 - 1 + r;
        in_col[q] = (r < RD_TW + 2 && (unsigned)x < (unsigned)p.W) ? (unsigned)(x * 256 + ((c ^ ((r >> 1) & 7)) << 4)) : kOOB;
    }
    unsigned res_col[RD_RP];
#pragma unroll
    for (int q = 0; q < RD_RP; ++q)
        res_col[q] = (unsigned)((x0 + 16 * q + (lane >> 2)) * 256 + nb * 64 + (lane & 3) * 16);
    const unsigned rowbytes = (unsigned)p.W * 256u;

    auto dma_step = [&](int i, int j) {      // input row i -> ring slot i % RD_NR (this wave's quarter of each pass); residual row j -> slot j % RD_RES_NR
        const int gyi = y0 - 1 + i, gyj = y0 + j;
        const bool oki = (unsigned)gyi < (unsigned)p.H && i < p.R + 2;
        const bool okj = RES && j >= 0 && j < p.R && gyj < p.H;
        const int soffi = oki ? gyi * (int)rowbytes : 0, soffj = okj ? gyj * (int)rowbytes : 0;
        const unsigned basei = lds_ring + (unsigned)(i % RD_NR) * RD_PITCH + (unsigned)nb * 1024u;
        const unsigned basej = lds_res + (unsigned)(((j % RD_RES_NR) + RD_RES_NR) % RD_RES_NR) * RC_RES_SLOT;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %1\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_dwordx4 %3, %6, %7 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x1000\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_dwordx4 %4, %6, %7 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x1000\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_dwordx4 %5, %6, %7 offen lds\n\t"
                     "s_mov_b32 m0, %2\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_dwordx4 %8, %10, %11 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x400\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_dwordx4 %9, %10, %11 offen lds\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(basei), "s"(basej),
                       "v"(oki ? in_col[0] : kOOB), "v"(oki ? in_col[1] : kOOB), "v"(oki ? in_col[2] : kOOB), "s"(srd_in), "s"(soffi),
                       "v"(okj ? res_col[0] : kOOB), "v"(okj ? res_col[1] : kOOB), "s"(srd_res), "s"(soffj)
                     : "memory", "scc");
    };

    // weights -> registers: A-fragment (tap, kc) of this wave's 32 channels: lane = channel nb*32 + l31, k = kc*16 + 8*hi .. +7
    bf16x8 wf[9][8];
    if (p.wfrag) {
        const bf16x8 *wp = reinterpret_cast<const bf16x8 *>(p.w) + (size_t)nb * 72 * 64 + lane;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kc = 0; kc < 8; ++kc)
                wf[t][kc] = wp[(t * 8 + kc) * 64];
    } else {
        const bf16_t *wrow = static_cast<const bf16_t *>(p.w) + (size_t)(nb * 32 + l31) * 1152 + hi * 8;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kc = 0; kc < 8; ++kc)
                wf[t][kc] = *reinterpret_cast<const bf16x8 *>(wrow + t * 128 + kc * 16);
    }
    float sc[8], sh[8];
    {
        const int c0 = nb * 32 + (lane & 3) * 8;
#pragma unroll
        for (int t = 0; t < 8; ++t) { sc[t] = p.scale ? p.scale[c0 + t] : 1.f; sh[t] = p.scale ? p.shift[c0 + t] : 0.f; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // B-fragment addresses inside a ring row: record r = l31 + kx, chunk kc*2 + hi, swizzled; computed per use from three per-kx bases
    unsigned rbase[3], rswz[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) { const int r = l31 + kx; rbase[kx] = (unsigned)(r * 256); rswz[kx] = (unsigned)((r >> 1) & 7); }
    auto frag = [&](int i, int f) {          // fragment f = kx*8 + kc of input row i
        const int kx = f >> 3, kc = f & 7;
        const unsigned off = (unsigned)(i % RD_NR) * RD_PITCH + rbase[kx] + ((((unsigned)(kc * 2 + hi)) ^ rswz[kx]) << 4);
        return *reinterpret_cast<const bf16x8 *>(reinterpret_cast<const char *>(smem) + off);
    };

    f32x16 acc0, acc1, acc2;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; }

    auto finish = [&](int j, int pass) {     // second half of the epilogue of output row j (its tile sits in the patch): 16 pixels
        const bool live = j >= 0 && j < p.R && y0 + j < p.H;
        const unsigned resbase = (lds_res - lds0) + (unsigned)(((j % RD_RES_NR) + RD_RES_NR) % RD_RES_NR) * RC_RES_SLOT;
        const int px = 16 * pass + (lane >> 2);
        const char *src = reinterpret_cast<const char *>(patch) + px * 144 + (lane & 3) * 32;
        const float4 v0 = *reinterpret_cast<const float4 *>(src), v1 = *reinterpret_cast<const float4 *>(src + 16);
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = v[t] * sc[t] + sh[t];
        if (RES) {
            const u32x4 rv = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(smem) + resbase + pass * 1024 + lane * 16);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                v[2 * t] += lo16<F16>(rv[t]);
                v[2 * t + 1] += hi16<F16>(rv[t]);
            }
        }
        unsigned o0[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            unsigned w = pack16x2<F16>(v[2 * t], v[2 * t + 1]);
            if (RELU) w = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, w), (i16x2){0, 0}));
            o0[t] = w;
        }
        const u32x4 o = {o0[0], o0[1], o0[2], o0[3]};
        const unsigned off = live ? (unsigned)(((y0 + j) * p.W + x0 + px) * 256 + nb * 64 + (lane & 3) * 16) : kOOB;
        __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, off, 0, 0);
    };

    // One step = input row i (see rowconv64); the row and residual row i - 3 were issued RD_PF steps ago, so at most the operations of the
    // RD_PF - 1 steps since may be in flight (the first RD_PF rows come from the prologue: statement k = row k + residual row k - 3, 5 operations)
    auto step = [&](int i, f32x16 &an, f32x16 &am, f32x16 &ao) {
        if (i < RD_PF) vm_wait<(RD_IN + RD_RP) * (RD_PF - 1)>(); else vm_wait<(RD_PF - 1) * RD_OPS>();
        __syncthreads();                      // row i visible to every wave; ring slot (i - 1) % RD_NR released by every wave
        dma_step(i + RD_PF, i + RD_PF - 3);
        bf16x8 bfr[3];
        bfr[0] = frag(i, 0); bfr[1] = frag(i, 1);
#pragma unroll
        for (int f = 0; f < 24; ++f) {
            if (f + 2 < 24) bfr[(f + 2) % 3] = frag(i, f + 2);
            const int kx = f >> 3, kc = f & 7;
            ao = mfma32_16b<F16>(wf[6 + kx][kc], bfr[f % 3], ao);
            am = mfma32_16b<F16>(wf[3 + kx][kc], bfr[f % 3], am);
            an = mfma32_16b<F16>(wf[0 + kx][kc], bfr[f % 3], f == 0 ? zero16 : an);
            if (f == 3) finish(i - 3, 0);
            if (f == 11) finish(i - 3, 1);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4 *>(reinterpret_cast<char *>(patch) + l31 * 144 + g * 32 + hi * 16) = make_float4(ao[4 * g], ao[4 * g + 1], ao[4 * g + 2], ao[4 * g + 3]);
    };

#pragma unroll
    for (int i = 0; i < RD_PF; ++i) dma_step(i, i - 3);
    const int nsteps = p.R + 2;
    int i = 0;
    for (; i < nsteps; i += 3) {
        step(i, acc0, acc2, acc1);
        step(i + 1, acc1, acc0, acc2);
        step(i + 2, acc2, acc1, acc0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    finish(i - 3, 0);
    finish(i - 3, 1);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The last conv of bf16 plans in its GEMM form (3x3 conv on the LOW-res source with N = 4 parities x 3 channels = 12 and the pre-summed
// sub-pixel taps scattered into a 9-tap operand), as a row kernel: 12 (padded to 16) x 1152 weights = 144 registers per wave,
// v_mfma_f32_16x16x32_bf16 with A = weights [16 n x 32 k], B = pixels [32 k x 16 px].  All four waves hold the same weights and take 16 of
// the strip's 64 pixels each.  The two concatenated 64-channel sources keep their own rings of 128-B pixel records (an LDS-DMA piece has one
// buffer descriptor), swizzled like the igemm's K-tiles; one barrier per row step.  Output: fp32 [B][H][W][12] for pixel_shuffle_tanh.
namespace {
typedef f32x4acc f32x4v;
constexpr int RL_TW = 64;                 // low-res pixels per strip row
constexpr int RL_PITCH = 12288;           // bytes per ring row and source: 96 records of 128 B = 3 passes of the workgroup (66 are real)
constexpr int RL_NR = 5;
constexpr int RL_PF = RL_NR - 1;
constexpr int RL_OPS = 3 + 3 + 1;         // vector-memory operations per step and wave: 3 pieces per source, 1 store (2 in the fused form: + 1)
static_assert((RL_PF - 1) * (RL_OPS + 1) < 64, "vmcnt is a 6-bit counter");
}  // namespace

// FUSED (round 5): pixel shuffle + tanh in the epilogue -- the fp32 [B][H][W][12] intermediate (25 MB written and read back at 8 frames) and the pixel_shuffle_tanh
// launch (13 us) go.  The rows of the weight operand are packed in the order n' = ((py * cout + co) * 2 + px) (pack_rowlast_weights), so a lane's four accumulator registers
// are two (px = 0, px = 1) pairs of one (output row parity, channel) each: two 8-byte stores per lane, 16 lanes = 128 contiguous bytes of an NCHW output row.  Same tanhf,
// same fp32 values: bit-identical to the two-launch form, which stays for the uint8 (tensor2im) output.
template <bool F16, bool FUSED>
__global__ __launch_bounds__(256) void rowlast128(const RowLastParams p)
{
    constexpr int OPS = RL_OPS + (FUSED ? 1 : 0);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef __attribute__((address_space(3))) float lds_float;
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_float *)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // this wave: pixels wave*16 .. +15 of the strip
    const int l15 = lane & 15, g4 = lane >> 4;

    asm volatile("" :: "s"(p.src0), "s"(p.src1), "s"(p.w), "s"(p.out), "s"(p.B), "s"(p.H), "s"(p.W), "s"(p.R), "s"(p.nsx), "s"(p.nsy), "s"(p.nblocks),
                       "s"(p.div_sx.m), "s"(p.div_sx.s1), "s"(p.div_sx.s2), "s"(p.div_sy.m), "s"(p.div_sy.s1), "s"(p.div_sy.s2));
    unsigned lin = blockIdx.x;
    {
        const unsigned total = (unsigned)p.nblocks, q = total >> 3, r = total & 7, x = lin & 7;
        lin = x * q + (x < r ? x : r) + (lin >> 3);
    }
    const unsigned t1 = p.div_sx.div(lin);
    const int sx = (int)(lin - t1 * (unsigned)p.nsx);
    const int b = (int)p.div_sy.div(t1);
    const int sy = (int)(t1 - (unsigned)b * (unsigned)p.nsy);
    const int x0 = sx * RL_TW, y0 = sy * p.R;
    const unsigned imgbytes = (unsigned)(p.H * p.W) * 128u;
    const i32x4 srd0 = make_srd(static_cast<const char *>(p.src0) + (size_t)b * imgbytes, imgbytes);
    const i32x4 srd1 = make_srd(static_cast<const char *>(p.src1) + (size_t)b * imgbytes, imgbytes);
    const __amdgpu_buffer_rsrc_t rs_out = FUSED
        ? __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(p.out_nchw) + (size_t)b * p.cout * p.H * p.W * 16, 0, p.cout * p.H * p.W * 16, 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(p.out) + (size_t)b * p.H * p.W * 48, 0, p.H * p.W * 48, 0x00020000);

    // DMA roles (both sources alike): pass q moves record slots q*256 + tid, record r = pixel x0 - 1 + r, slot c <- chunk c ^ ((r >> 1) & 7)
    unsigned in_col[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int sl = q * 256 + tid, r = sl >> 3, c = sl & 7;
        const int x = x0 - 1 + r;
        in_col[q] = (r < RL_TW + 2 && (unsigned)x < (unsigned)p.W) ? (unsigned)(x * 128 + ((c ^ ((r >> 1) & 7)) << 4)) : kOOB;
    }
    const unsigned rowbytes = (unsigned)p.W * 128u;
    auto dma_step = [&](int i) {             // input row i of the strip (image row y0 - 1 + i) of both sources -> ring slot i % RL_NR
        const int gy = y0 - 1 + i;
        const bool ok = (unsigned)gy < (unsigned)p.H && i < p.R + 2;
        const int soff = ok ? gy * (int)rowbytes : 0;
        const unsigned base = lds0 + (unsigned)(i % RL_NR) * (2 * RL_PITCH) + (unsigned)wave * 1024u;
        const unsigned v[3] = {ok ? in_col[0] : kOOB, ok ? in_col[1] : kOOB, ok ? in_col[2] : kOOB};
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %1\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_dwordx4 %2, %5, %7 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x1000\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_dwordx4 %3, %5, %7 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x1000\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_dwordx4 %4, %5, %7 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x1000\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_dwordx4 %2, %6, %7 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x1000\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_dwordx4 %3, %6, %7 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x1000\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_dwordx4 %4, %6, %7 offen lds\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(base), "v"(v[0]), "v"(v[1]), "v"(v[2]), "s"(srd0), "s"(srd1), "s"(soff)
                     : "memory", "scc");
    };

    // weights -> registers: A-fragment (tap, kc): lane = output n = l15 (12 real), k = kc*32 + 8*g4 .. +7 of the tap
    bf16x8 wf[9][4];
    {
        const bf16x8 *wp = reinterpret_cast<const bf16x8 *>(p.w) + lane;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc)
                wf[t][kc] = wp[(t * 4 + kc) * 64];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // B-fragment (kx, kc): record r = wave*16 + l15 + kx of source kc >> 1, chunk (kc & 1)*4 + g4, swizzled
    unsigned boff[3][4];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int r = wave * 16 + l15 + kx;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc)
            boff[kx][kc] = (unsigned)((kc >> 1) * RL_PITCH + r * 128 + (((((kc & 1) * 4 + g4)) ^ ((r >> 1) & 7)) << 4));
    }
    f32x4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0;
    const f32x4v zero4 = {0.f, 0.f, 0.f, 0.f};

    auto step = [&](int i, f32x4v &an, f32x4v &am, f32x4v &ao) {
        if (i < RL_PF) vm_wait<6 * (RL_PF - 1)>(); else vm_wait<(RL_PF - 1) * OPS>();
        __syncthreads();
        dma_step(i + RL_PF);
        const char *row = reinterpret_cast<const char *>(smem) + (unsigned)(i % RL_NR) * (2 * RL_PITCH);
#pragma unroll
        for (int f = 0; f < 12; ++f) {
            const int kx = f >> 2, kc = f & 3;
            const bf16x8 bv = *reinterpret_cast<const bf16x8 *>(row + boff[kx][kc]);
            ao = mfma16_16b<F16>(wf[6 + kx][kc], bv, ao);
            am = mfma16_16b<F16>(wf[3 + kx][kc], bv, am);
            an = mfma16_16b<F16>(wf[0 + kx][kc], bv, f == 0 ? zero4 : an);
        }
        // output row j = i - 2 is complete.  D layout: lane = pixel l15, register r = output n = 4*g4 + r: 12 floats per pixel, lanes g4 < 3
        const int j = i - 2;
        if constexpr (FUSED) {
            // registers (2h, 2h + 1) = output pixels (2x, 2x + 1) of row 2y + py, channel co, with (py, co) = pair 2 g4 + h
            const bool rowok = j >= 0 && j < p.R && y0 + j < p.H;
            const int xg = x0 + wave * 16 + l15;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int q = 2 * g4 + h;
                const int py = q >= p.cout ? 1 : 0, co = q - py * p.cout;
                float v0 = ao[2 * h], v1 = ao[2 * h + 1];
                if (p.apply_tanh) { v0 = tanhf(v0); v1 = tanhf(v1); }
                const unsigned off = (rowok && q < 2 * p.cout) ? (unsigned)(((co * 2 * p.H + 2 * (y0 + j) + py) * 2 * p.W + 2 * xg) * 4) : kOOB;
                const v2f pr = {v0, v1};
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pr), rs_out, off, 0, 0);
            }
        } else {
            const bool live = j >= 0 && j < p.R && y0 + j < p.H && g4 < 3;
            const unsigned off = live ? (unsigned)(((y0 + j) * p.W + x0 + wave * 16 + l15) * 48 + g4 * 16) : kOOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ao), rs_out, off, 0, 0);
        }
    };

#pragma unroll
    for (int i = 0; i < RL_PF; ++i) dma_step(i);
    const int nsteps = p.R + 2;
    for (int i = 0; i < nsteps; i += 3) {
        step(i, acc0, acc2, acc1);
        step(i + 1, acc1, acc0, acc2);
        step(i + 2, acc2, acc1, acc0);
    }
}

void pack_rowlast_weights(const unsigned short *rows, unsigned short *out, int nout)
{
    // rows: bf16 [nout][9][128] (nout = 4 parities x cout <= 16); fragment (tap, kc): lane = n (lane & 15, zero rows past nout),
    // k = kc*32 + 8*(lane >> 4) .. +7 of the tap
    for (int t = 0; t < 9; ++t)
        for (int kc = 0; kc < 4; ++kc)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    // operand row n' = ((py * cout + co) * 2 + px) holds GEMM row (py * 2 + px) * cout + co: a lane's registers are (px 0, px 1) pairs (the fused epilogue)
                    const int np = lane & 15, cout = nout / 4, q = np >> 1, px = np & 1, py = cout ? q / cout : 0, co = cout ? q % cout : 0;
                    const int n = (py * 2 + px) * cout + co;
                    out[(((size_t)t * 4 + kc) * 64 + lane) * 8 + e] = np < nout ? rows[((size_t)n * 9 + t) * 128 + kc * 32 + 8 * (lane >> 4) + e] : 0;
                }
}

bool rowlast_supported(const RowLastParams &p)
{
    return p.B >= 1 && p.H >= 1 && p.W % RL_TW == 0 && p.R >= 1 && (size_t)p.H * p.W * 128 < 0x7fffffffull;
}

int rowlast_rows(int batch, int h, int w)
{
    const int cand[] = {32, 16, 8, 4, 2, 1};
    for (int r : cand)
        if ((long)batch * (w / RL_TW) * ((h + r - 1) / r) >= 256) return r;
    return 1;
}

hipError_t launch_rowlast(const RowLastParams &p_in, hipStream_t s)
{
    if (!rowlast_supported(p_in)) return hipErrorInvalidValue;
    RowLastParams p = p_in;
    p.nsx = p.W / RL_TW; p.nsy = (p.H + p.R - 1) / p.R;
    p.nblocks = p.B * p.nsx * p.nsy;
    p.div_sx = FastDiv::make((unsigned)p.nsx);
    p.div_sy = FastDiv::make((unsigned)p.nsy);
    const size_t smem = (size_t)RL_NR * 2 * RL_PITCH;
    static AttrMask attr_mask;
    if (attr_needed_on_this_device(attr_mask)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&rowlast128<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&rowlast128<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&rowlast128<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&rowlast128<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        attr_done_on_this_device(attr_mask);
    }
    if (p.out_nchw) {
        if (p.cout < 1 || p.cout > 4 || (size_t)p.cout * p.H * p.W * 16 >= 0x7fffffffull) return hipErrorInvalidValue;
        if (p.dtype == 2) hipLaunchKernelGGL((rowlast128<true, true>), dim3(p.nblocks), dim3(256), smem, s, p);
        else hipLaunchKernelGGL((rowlast128<false, true>), dim3(p.nblocks), dim3(256), smem, s, p);
        return hipGetLastError();
    }
    if (p.dtype == 2) hipLaunchKernelGGL((rowlast128<true, false>), dim3(p.nblocks), dim3(256), smem, s, p);
    else hipLaunchKernelGGL((rowlast128<false, false>), dim3(p.nblocks), dim3(256), smem, s, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// L1.up of the bf16 plans (Upsample x2 + conv3x3 over the concat of two 128-channel sources -> 64 channels, in sub-pixel form: 4 output
// parities x 2x2 taps over the LOW-res rows) as a row kernel.  A wave = one parity; a workgroup = one half (32) of the output channels, so a
// wave keeps 4 taps x 16 channel blocks = 64 A-fragments (256 registers).  Input row i of the low-res strip feeds tap row a = 0 of output
// row i - py and tap row a = 1 of output row i - py - 1: two accumulators per wave, rotating every step.  The four waves share the 34-pixel
// window of both sources (one ring, one barrier per step) and write disjoint pixels of the 2x-resolution output.
namespace {
constexpr int RU_TW = 32;                 // low-res pixels per strip row
constexpr int RU_PITCH = 12288;           // bytes per ring row and source: 48 records of 256 B = 3 passes of the workgroup (34 are real)
constexpr int RU_NR = 5;
constexpr int RU_PF = RU_NR - 1;
constexpr int RU_OPS = 6 + 2;             // vector-memory operations per step and wave: 3 pieces per source, 2 stores
static_assert((RU_PF - 1) * RU_OPS < 64, "vmcnt is a 6-bit counter");
}  // namespace

template <bool F16, bool RELU>
__global__ __launch_bounds__(256) void rowup256(const RowUpParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef __attribute__((address_space(3))) float lds_float;
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_float *)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int par = __builtin_amdgcn_readfirstlane(tid >> 6);     // this wave: output parity (py, px)
    const int py = par >> 1, px = par & 1;
    const int hi = lane >> 5, l31 = lane & 31;

    asm volatile("" :: "s"(p.src0), "s"(p.src1), "s"(p.w), "s"(p.scale), "s"(p.shift), "s"(p.out), "s"(p.B), "s"(p.H), "s"(p.W), "s"(p.R), "s"(p.nsx),
                       "s"(p.nsy), "s"(p.nblocks), "s"(p.div_sx.m), "s"(p.div_sx.s1), "s"(p.div_sx.s2), "s"(p.div_sy.m), "s"(p.div_sy.s1), "s"(p.div_sy.s2));
    unsigned lin = blockIdx.x;
    {
        const unsigned total = (unsigned)p.nblocks, q = total >> 3, r = total & 7, x = lin & 7;
        lin = x * q + (x < r ? x : r) + (lin >> 3);
    }
    const int nb = (int)(lin & 1u);                     // half of the output channels; the two halves of a strip are neighbours on one XCD
    const unsigned strip = lin >> 1;
    const unsigned t1 = p.div_sx.div(strip);
    const int sx = (int)(strip - t1 * (unsigned)p.nsx);
    const int b = (int)p.div_sy.div(t1);
    const int sy = (int)(t1 - (unsigned)b * (unsigned)p.nsy);
    const int x0 = sx * RU_TW, y0 = sy * p.R;
    const unsigned imgbytes = (unsigned)(p.H * p.W) * 256u;
    const i32x4 srd0 = make_srd(static_cast<const char *>(p.src0) + (size_t)b * imgbytes, imgbytes);
    const i32x4 srd1 = make_srd(static_cast<const char *>(p.src1) + (size_t)b * imgbytes, imgbytes);
    const unsigned outbytes = (unsigned)(4 * p.H * p.W) * 128u;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(static_cast<char *>(p.out) + (size_t)b * outbytes, 0, (int)outbytes, 0x00020000);
    float *patch = smem + (RU_NR * 2 * RU_PITCH) / 4 + par * (RC_PATCH / 4);

    unsigned in_col[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int sl = q * 256 + tid, r = sl >> 4, c = sl & 15;
        const int x = x0 - 1 + r;
        in_col[q] = (r < RU_TW + 2 && (unsigned)x < (unsigned)p.W) ? (unsigned)(x * 256 + ((c ^ ((r >> 1) & 7)) << 4)) : kOOB;
    }
    const unsigned rowbytes = (unsigned)p.W * 256u;
    auto dma_step = [&](int i) {             // low-res input row i of the strip (image row y0 - 1 + i) of both sources -> ring slot i % RU_NR
        const int gy = y0 - 1 + i;
        const bool ok = (unsigned)gy < (unsigned)p.H && i < p.R + 2;
        const int soff = ok ? gy * (int)rowbytes : 0;
        const unsigned base = lds0 + (unsigned)(i % RU_NR) * (2 * RU_PITCH) + (unsigned)par * 1024u;
        const unsigned v[3] = {ok ? in_col[0] : kOOB, ok ? in_col[1] : kOOB, ok ? in_col[2] : kOOB};
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %1\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_dwordx4 %2, %5, %7 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x1000\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_dwordx4 %3, %5, %7 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x1000\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_dwordx4 %4, %5, %7 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x1000\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_dwordx4 %2, %6, %7 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x1000\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_dwordx4 %3, %6, %7 offen lds\n\t"
                     "s_add_u32 m0, m0, 0x1000\n\t"
                     "s_nop 0\n\t"
                     "buffer_load_dwordx4 %4, %6, %7 offen lds\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(base), "v"(v[0]), "v"(v[1]), "v"(v[2]), "s"(srd0), "s"(srd1), "s"(soff)
                     : "memory", "scc");
    };

    // weights -> registers: A-fragment (tap t = a*2 + b, kc): lane = channel nb*32 + l31, k = concat channel kc*16 + 8*hi .. +7
    bf16x8 wf[4][16];
    {
        const bf16x8 *wp = reinterpret_cast<const bf16x8 *>(p.w) + ((size_t)(nb * 4 + par) * 64) * 64 + lane;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int kc = 0; kc < 16; ++kc)
                wf[t][kc] = wp[(t * 16 + kc) * 64];
    }
    float sc[8], sh[8];
    {
        const int c0 = nb * 32 + (lane & 3) * 8;
#pragma unroll
        for (int t = 0; t < 8; ++t) { sc[t] = p.scale ? p.scale[c0 + t] : 1.f; sh[t] = p.scale ? p.shift[c0 + t] : 0.f; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // B-fragment (b, kc): record r = l31 + px + b of source kc >> 3, chunk (kc & 7)*2 + hi, swizzled
    unsigned rb[2], rs[2];
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) { const int r = l31 + px + bb; rb[bb] = (unsigned)(r * 256); rs[bb] = (unsigned)((r >> 1) & 7); }

    f32x16 acc0, acc1;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

    // second half of the epilogue of low-res output row j of this parity (its tile sits in the patch): 16 pixels per pass
    auto finish = [&](int j, int pass) {
        const bool live = j >= 0 && j < p.R && y0 + j < p.H;
        const int pxl = 16 * pass + (lane >> 2);
        const char *src = reinterpret_cast<const char *>(patch) + pxl * 144 + (lane & 3) * 32;
        const float4 v0 = *reinterpret_cast<const float4 *>(src), v1 = *reinterpret_cast<const float4 *>(src + 16);
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int t = 0; t < 8; ++t) { v[t] = v[t] * sc[t] + sh[t]; if (RELU) v[t] = fmaxf(v[t], 0.f); }
        const u32x4 o = {pack16x2<F16>(v[0], v[1]), pack16x2<F16>(v[2], v[3]), pack16x2<F16>(v[4], v[5]), pack16x2<F16>(v[6], v[7])};
        const unsigned off = live ? (unsigned)(((2 * (y0 + j) + py) * (2 * p.W) + 2 * (x0 + pxl) + px) * 128 + nb * 64 + (lane & 3) * 16) : kOOB;
        __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, off, 0, 0);
    };

    // One step = low-res input row i.  Output row i - py - 1 is complete after it: its tile is parked in the patch and finished between the
    // MFMAs of the NEXT step, as in rowconv64.
    auto step = [&](int i, f32x16 &an, f32x16 &ao) {
        if (i < RU_PF) vm_wait<6 * (RU_PF - 1)>(); else vm_wait<(RU_PF - 1) * RU_OPS>();
        __syncthreads();
        dma_step(i + RU_PF);
        const char *row = reinterpret_cast<const char *>(smem) + (unsigned)(i % RU_NR) * (2 * RU_PITCH);
#pragma unroll
        for (int f = 0; f < 32; ++f) {
            const int bb = f >> 4, kc = f & 15;
            const bf16x8 bv = *reinterpret_cast<const bf16x8 *>(row + (kc >> 3) * RU_PITCH + rb[bb] + (((unsigned)((kc & 7) * 2 + hi) ^ rs[bb]) << 4));
            ao = mfma32_16b<F16>(wf[2 + bb][kc], bv, ao);                       // tap row a = 1: output row i - py - 1
            an = mfma32_16b<F16>(wf[0 + bb][kc], bv, f == 0 ? zero16 : an);     // tap row a = 0: output row i - py
            if (f == 3) finish(i - py - 2, 0);
            if (f == 13) finish(i - py - 2, 1);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)      // D layout: lane = pixel l31, register r = channel 8*(r >> 2) + 4*hi + (r & 3)
            *reinterpret_cast<float4 *>(reinterpret_cast<char *>(patch) + l31 * 144 + g * 32 + hi * 16) = make_float4(ao[4 * g], ao[4 * g + 1], ao[4 * g + 2], ao[4 * g + 3]);
    };

#pragma unroll
    for (int i = 0; i < RU_PF; ++i) dma_step(i);
    const int nsteps = p.R + 2;
    int i = 0;
    for (; i < nsteps; i += 2) {
        step(i, acc0, acc1);
        step(i + 1, acc1, acc0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    finish(i - py - 2, 0);                   // the tile parked by the last step
    finish(i - py - 2, 1);
}

void pack_rowup_weights(const unsigned short *rows, unsigned short *out)
{
    // rows: bf16 [par 4][cout 64][a 2][b 2][cin 256] (the igemm's sub-pixel layout); fragment (half nb, par, tap t = a*2 + b, kc):
    // lane = channel nb*32 + (lane & 31), k = kc*16 + 8*(lane >> 5) .. +7
    for (int nb = 0; nb < 2; ++nb)
        for (int par = 0; par < 4; ++par)
            for (int t = 0; t < 4; ++t)
                for (int kc = 0; kc < 16; ++kc)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e)
                            out[(((((size_t)nb * 4 + par) * 4 + t) * 16 + kc) * 64 + lane) * 8 + e] =
                                rows[(((size_t)par * 64 + nb * 32 + (lane & 31)) * 4 + t) * 256 + kc * 16 + 8 * (lane >> 5) + e];
}

bool rowup_supported(const RowUpParams &p)
{
    return p.B >= 1 && p.H >= 1 && p.W % RU_TW == 0 && p.R >= 1 && (size_t)4 * p.H * p.W * 128 < 0x7fffffffull;
}

int rowup_rows(int batch, int h, int w)
{
    const int cand[] = {32, 16, 8, 4, 2};          // even step counts (the two accumulators alternate): R + 2 even
    for (int r : cand)
        if ((long)batch * 2 * (w / RU_TW) * ((h + r - 1) / r) >= 256) return r;
    return 2;
}

hipError_t launch_rowup(const RowUpParams &p_in, hipStream_t s)
{
    if (!rowup_supported(p_in) || (p_in.R & 1)) return hipErrorInvalidValue;
    RowUpParams p = p_in;
    p.nsx = p.W / RU_TW; p.nsy = (p.H + p.R - 1) / p.R;
    p.nblocks = 2 * p.B * p.nsx * p.nsy;
    p.div_sx = FastDiv::make((unsigned)p.nsx);
    p.div_sy = FastDiv::make((unsigned)p.nsy);
    const size_t smem = (size_t)RU_NR * 2 * RU_PITCH + 4 * (size_t)RC_PATCH;
    static AttrMask attr_mask;
    if (attr_needed_on_this_device(attr_mask)) {
        hipError_t e = hipSuccess;
        const void *ks[4] = {reinterpret_cast<const void *>(&rowup256<false, true>), reinterpret_cast<const void *>(&rowup256<false, false>),
                             reinterpret_cast<const void *>(&rowup256<true, true>), reinterpret_cast<const void *>(&rowup256<true, false>)};
        for (int k = 0; k < 4 && e == hipSuccess; ++k) e = hipFuncSetAttribute(ks[k], hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        attr_done_on_this_device(attr_mask);
    }
    if (p.dtype == 2) {
        if (p.relu) hipLaunchKernelGGL((rowup256<true, true>), dim3(p.nblocks), dim3(256), smem, s, p);
        else hipLaunchKernelGGL((rowup256<true, false>), dim3(p.nblocks), dim3(256), smem, s, p);
    } else {
        if (p.relu) hipLaunchKernelGGL((rowup256<false, true>), dim3(p.nblocks), dim3(256), smem, s, p);
        else hipLaunchKernelGGL((rowup256<false, false>), dim3(p.nblocks), dim3(256), smem, s, p);
    }
    return hipGetLastError();
}

void pack_rowconv_weights(const unsigned short *rows, unsigned short *out, int c)
{
    // A-fragment (nb, tap, kc): lane = channel nb*32 + (lane & 31), k = kc*16 + 8*(lane >> 5) .. +7; c = 64 | 128 channels in and out
    const int nkc = c / 16;
    for (int nb = 0; nb < c / 32; ++nb)
        for (int t = 0; t < 9; ++t)
            for (int kc = 0; kc < nkc; ++kc)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e)
                        out[((((size_t)nb * 9 + t) * nkc + kc) * 64 + lane) * 8 + e] =
                            rows[((size_t)(nb * 32 + (lane & 31)) * 9 + t) * c + kc * 16 + 8 * (lane >> 5) + e];
}

bool rowconv_supported(const RowConvParams &p)
{
    if (p.C != 64 && p.C != 128) return false;
    const int tw = p.C == 64 ? RC_TW : RD_TW;
    return p.B >= 1 && p.H >= 1 && p.W % tw == 0 && p.R >= 1 && (size_t)p.H * p.W * p.C * 2 < 0x7fffffffull;
}

int rowconv_rows(int batch, int h, int w, int c)
{
    // rows per strip: the largest that still gives every CU a workgroup (one workgroup per CU fits: ~290-450 registers per lane)
    const int tw = c == 64 ? RC_TW : RD_TW;
    const int cand[] = {32, 16, 8, 4, 2, 1};
    for (int r : cand)
        if ((long)batch * (w / tw) * ((h + r - 1) / r) >= 256) return r;
    return 1;
}

hipError_t launch_rowconv(const RowConvParams &p_in, hipStream_t s)
{
    if (!rowconv_supported(p_in)) return hipErrorInvalidValue;
    RowConvParams p = p_in;
    const bool wide = p.C == 128;
    p.nsx = p.W / (wide ? RD_TW : RC_TW); p.nsy = (p.H + p.R - 1) / p.R;
    p.nblocks = p.B * p.nsx * p.nsy;
    p.div_sx = FastDiv::make((unsigned)p.nsx);
    p.div_sy = FastDiv::make((unsigned)p.nsy);
    const size_t smem = wide ? (size_t)RD_NR * RD_PITCH + 4 * ((size_t)RD_RES_NR * RC_RES_SLOT + (size_t)RC_PATCH)
                             : 4 * ((size_t)RC_NR * RC_ROWB + (size_t)RC_RES_NR * RC_RES_SLOT + (size_t)RC_PATCH);
    typedef void (*kern_t)(const RowConvParams);
    static const kern_t kern[16] = {rowconv64<false, false, false>, rowconv64<false, false, true>, rowconv64<false, true, false>, rowconv64<false, true, true>,
                                    rowconv128<false, false, false>, rowconv128<false, false, true>, rowconv128<false, true, false>, rowconv128<false, true, true>,
                                    rowconv64<true, false, false>, rowconv64<true, false, true>, rowconv64<true, true, false>, rowconv64<true, true, true>,
                                    rowconv128<true, false, false>, rowconv128<true, false, true>, rowconv128<true, true, false>, rowconv128<true, true, true>};
    static AttrMask attr_mask;
    if (attr_needed_on_this_device(attr_mask)) {
        for (int k = 0; k < 16; ++k) {
            const size_t need = (k & 7) < 4 ? 4 * ((size_t)RC_NR * RC_ROWB + (size_t)RC_RES_NR * RC_RES_SLOT + (size_t)RC_PATCH)
                                      : (size_t)RD_NR * RD_PITCH + 4 * ((size_t)RD_RES_NR * RC_RES_SLOT + (size_t)RC_PATCH);
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern[k]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)need);
            if (e != hipSuccess) return e;
        }
        attr_done_on_this_device(attr_mask);
    }
    hipLaunchKernelGGL(kern[(p.dtype == 2 ? 8 : 0) + (wide ? 4 : 0) + (p.residual ? 2 : 0) + (p.relu ? 1 : 0)], dim3(p.nblocks), dim3(256), smem, s, p);
    return hipGetLastError();
}

}  // namespace lspf2f
