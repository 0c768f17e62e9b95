// gfx950: the stride-1 3x3 convolutions of the 64x64 / 32x32 levels of the 16-bit plans (bf16 | fp16 storage, fp32 accumulate) from 8 frames up -- the ResidualBlock
// convs 256 -> 256 @ 64x64 and 512 -> 512 @ 32x32 of BASELINE.json configs[2], which ran on igemm3x3<128x128> at 0.33 of the dense 16-bit MFMA peak.  DESIGN.md 4.12.
//
// What bound the implicit GEMM there (round 6 ablation, profiles/r06_patch16_ab.txt (1)): a 128 x 128 tile pulls 8 LDS-DMA pieces per wave and K-tile through the
// CU's vector-memory path for 16 MFMAs (512 cycles) -- 64 KB per 1024 MFMA cycles of a SIMD, the path's own rate -- because im2col re-stages every input pixel
// once per tap.  This kernel stages each pixel ONCE per 64-channel block and reads it for all nine taps:
//   * a workgroup (8 waves) owns TR x TW = 256 output pixels of one frame x BN output channels; per 64-channel block `cb` the (TR + 2) x (TW + 2) halo patch of the
//     source arrives by LDS-DMA (128-byte pixel records, zero padding by out-of-range pieces), double-buffered: the pieces of block cb + 1 ride one per tap step
//     under the nine K-tiles of block cb.  Vector-memory traffic per K-tile: 16 KB of weights + 1/9 of a 50-KB patch = 21.6 KB instead of 64 KB;
//   * tap (ky, kx) of a K-tile is an address offset into the patch: the A fragment of output pixels (r, c .. c + 31) is patch pixels (r + ky, c + kx ..) -- 32
//     consecutive records, whose 16-byte k-slots are XOR-swizzled on the record index so every ds_read_b128 lane group covers the 64 banks once;
//   * weights: the implicit GEMM's own rows [Cout][9][Cin] (no second packed copy), BN x 128 B per K-tile through a 3-slot LDS ring, two K-tiles ahead;
//   * the eight waves form two groups of four (one wave of each per SIMD) that run half a K-tile step apart: while one group issues its 16 MFMAs
//     the other reads the next K-tile's fragments and issues its copies; two s_barrier per K-tile keep the groups in that alternation, every vmcnt wait is
//     counted (never 0 inside the loop) -- the schedule of the guide's 8-phase GEMM template carried over to an implicit GEMM;
//     Hazards by barrier interval (interval k lies between the k-th and (k+1)-th barrier a wave passes inside the loop; group 0 = waves 0-3, group 1 = waves 4-7):
//         group 0: load segment of K-tile t in interval 2t,     MFMA segment in 2t + 1        group 1: load segment in 2t + 1, MFMA segment in 2t + 2
//       WAR  the copies of K-tile t + 2 go into the ring slot of K-tile t - 1, issued in load segment t = interval >= 2t; the last read of t - 1 (group 1, interval 2t - 1)
//            completed (lgkmcnt(0)) in front of barrier 2t - 1.  The patch pieces of block cb + 1 go into the buffer block cb - 1 was read from: same argument.
//       RAW  K-tile t + 1 is read in interval >= 2t + 2; every wave waited (counted vmcnt: everything but this step's own pieces) for its share of it at the end of its load
//            segment t, in front of barrier <= 2t + 1.  The last patch piece (tap NPA - 1 <= 7) is older than the weights waited for at tap 8.
//       The epilogue's transpose patches live in the patch buffer of the LAST block: every read of it completed in front of the loop's last barrier, and the copies still in flight
//       then (zeros: out-of-range pieces of a block and K-tiles that do not exist) target the other buffer, its dump KB and the ring.
//   * K order: channel block outer, tap inner (the implicit GEMM: tap outer) -> same products, different fp32 summation order: the two kernels agree to one
//     16-bit ulp of the result, not bit for bit.
// Fused epilogue (folded BatchNorm scale / shift, residual, ReLU, RNE store) as in the implicit-GEMM kernel.
// Reference semantics: ResidualBlock's Conv2d 3x3 p1 s1 (models/networks.py:650-675) under torch.cuda.amp.autocast (models/feature2face_G.py:28-30) for fp16 storage;
// bf16 storage is this repo's configs[2] (parity-unpinned, declared tolerance).
#include "device_common.h"
#include "kernels.h"

namespace lspf2f {

static constexpr unsigned kOOBp = 0x80000000u;   // voffset beyond any num_records: the piece lands as zeros

// Ablation switches (-DLSPF2F_ABLATE builds, LSP_HIP_DBG of tools/time_conv.py; the shipped kernel has none of them): 1 no copies in the K loop, 2 no fragment reads, 4 no MFMAs,
// 16 no epilogue, 32 both wave groups in lockstep, 64 vmcnt(0) instead of the counted waits, 256 two extra barriers per step, 512 s_setprio 1 around the MFMAs,
// 1024 tap-invariant fragment addresses, 2048 no lgkmcnt(0) in front of the barrier
#ifdef LSPF2F_ABLATE
#define PABL(p, bit) ((p).dbg & (bit))
#else
#define PABL(p, bit) 0
#endif

// segment timers (tools/probes/patch16_stamps.py; builds with -DLSPF2F_PATCH_STAMPS only): s_memtime values are wave-uniform scalars; a sample is only USED behind a wait
// the schedule has anyway, so the instrumented loop keeps its waits
#ifdef LSPF2F_PATCH_STAMPS
#define PSTAMP(v) do { __builtin_amdgcn_sched_barrier(0); v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PSTAMP(v) do {} while (0)
#endif

// both barriers of a K-tile step: nothing moves across (the compiler sees a memory clobber, the scheduler a fence)
#define PATCH_BAR() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

template <bool F16, int TW, int TR, int BN>
__global__ __launch_bounds__(512) void conv3x3_patch16(const PatchConvParams p)
{
    typedef typename St16<F16>::type T;
    constexpr int PW = TW + 2, PPX = PW * (TR + 2);          // patch width / pixels
    constexpr int NPIECE = (PPX + 7) / 8;                    // 1-KB pieces (8 pixel records) per patch
    constexpr int NPA = (NPIECE + 7) / 8;                    // ... per wave
    constexpr int NPB = BN / 64;                             // weight pieces per wave and K-tile (BN rows x 128 B over 8 waves)
    constexpr int TN = BN / 64;                              // 32-channel MFMA blocks per wave (waves: 4 along pixels x 2 along channels)
    constexpr int PATCH_BYTES = (NPIECE + 1) * 1024;         // + one KB behind each buffer where the pieces past the patch's end land (zeros; piece slot NPIECE)
    constexpr int BRING = 2 * PATCH_BYTES;                   // 3 slots of BN x 128 B
    constexpr int BTILE = BN * 128;
    static_assert(TW * TR == 256 && TW % 32 == 0, "a wave's 32-pixel MFMA block is 32 consecutive pixels of one row");
    static_assert(NPA <= 8, "one patch piece per tap step, landed a step before the block is read");
    static_assert(8 * 32 * 36 * 4 <= PATCH_BYTES, "the epilogue's transpose patches live in the patch buffer of the last channel block");

    extern __shared__ __attribute__((aligned(16))) char smem_p16[];
    typedef __attribute__((address_space(3))) char lds_char;
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_char *)smem_p16;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, grp = wave >> 2;
    // one scalar-load round trip for the whole argument block (see igemm.hip)
    asm volatile("" :: "s"(p.src), "s"(p.w), "s"(p.scale), "s"(p.shift), "s"(p.residual), "s"(p.out), "s"(p.B), "s"(p.H), "s"(p.W), "s"(p.C), "s"(p.Cout),
                       "s"(p.relu), "s"(p.ntm), "s"(p.ntn), "s"(p.div_tpi.m), "s"(p.div_tpi.s1), "s"(p.div_tpi.s2), "s"(p.div_tx.m), "s"(p.div_tx.s1), "s"(p.div_tx.s2),
                       "s"(p.div_ntn.m), "s"(p.div_ntn.s1), "s"(p.div_ntn.s2));

    // ---- tile.  XCD-aware order: the dispatcher deals workgroups round-robin over the 8 XCDs; logical ids are handed out in 8 contiguous chunks, n fastest, so one
    // XCD's L2 sees a band of pixel tiles once (every channel tile of it) and each XCD streams the weights
    unsigned lin = blockIdx.x;
    {
        const unsigned total = gridDim.x, q = total >> 3, r = total & 7, x = lin & 7;
        lin = x * q + (x < r ? x : r) + (lin >> 3);
    }
    const int mt = (int)p.div_ntn.div(lin), nt = (int)lin - mt * p.ntn;
    const int fb = (int)p.div_tpi.div((unsigned)mt);                       // frame
    const int rem = mt - fb * p.tiles_per_img;
    const int ty = (int)p.div_tx.div((unsigned)rem), tx = rem - ty * p.tiles_x;
    const int y0 = ty * TR, x0 = tx * TW, n0 = nt * BN;
    const int NCB = p.C >> 6;                                              // 64-channel blocks
    const int Krow = 9 * p.C;                                              // elements per weight row

    // ---- copy descriptors (fixed for the whole K loop).  Patch piece q = j * 8 + wave: records q * 8 .. + 7, lane -> (record lane >> 3, 16-byte slot lane & 7);
    // slot s of record pp holds k-quad s ^ ((pp >> 1) & 7)
    unsigned voffA[NPA], voffB[NPB];
    unsigned dstA[NPA];                                                    // (wave-uniform) byte offset of piece j inside a patch buffer (the dump slot behind it for q >= NPIECE)
#pragma unroll
    for (int j = 0; j < NPA; ++j) {
        const int q = j * 8 + wave;
        const int pp = q * 8 + (lane >> 3), s = lane & 7;
        const int pr = pp / PW, pc = pp - pr * PW;
        const int y = y0 - 1 + pr, x = x0 - 1 + pc;
        const bool ok = pp < PPX && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        voffA[j] = ok ? (unsigned)((fb * p.H + y) * p.W + x) * (unsigned)(p.C * 2) + (unsigned)((s ^ ((pp >> 1) & 7)) << 4) : kOOBp;
        dstA[j] = (unsigned)((q < NPIECE ? q : NPIECE) * 1024);
    }
#pragma unroll
    for (int k = 0; k < NPB; ++k) {
        const int row = wave * 8 + k * 64 + (lane >> 3), s = lane & 7;
        voffB[k] = (unsigned)((n0 + row) * Krow * 2) + (unsigned)((s ^ ((row >> 1) & 7)) << 4);
    }
    const i32x4 rsa = make_srd(p.src, (unsigned)(p.B * p.H * p.W) * (unsigned)(p.C * 2));
    const i32x4 rsw = make_srd(p.w, (unsigned)(p.Cout * Krow * 2));
    const unsigned ldsB = lds0 + BRING + (unsigned)(wave * 8 * 128);        // this wave's first weight piece in ring slot 0

    // ---- fragment addresses.  Lane l supplies row (l & 31), k-quad (l >> 5) of each 8-wide k group (the implicit GEMM's operand order).
    const int l31 = lane & 31, hh = lane >> 5;
    int p0[2];                                                             // patch record of tap (0, 0) for this lane's pixel of MFMA block i
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = wm * 64 + i * 32 + l31;
        p0[i] = (m / TW) * PW + (m % TW);
    }
    unsigned boff[TN][4];                                                  // B fragment byte addresses in ring slot 0, per k-step (the slot is an immediate offset)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nl = wn * (BN / 2) + j * 32 + l31;
        const unsigned base = (unsigned)BRING + (unsigned)(nl * 128) + (unsigned)(((hh ^ (nl >> 1)) & 7) << 4);      // (BRING is a multiple of 1 KB: the k-step XOR below stays inside the record)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) boff[j][ks] = base ^ (unsigned)(ks << 5);
    }

    // ---- epilogue operands first in the queue (older than every copy: the counted waits below are unaffected).  Epilogue role of a lane: 8 consecutive channels
    // (16 bytes of output) of one pixel -- 4 lanes per 32-channel row segment, 16 rows per pass: half the store / residual-load instructions of the 4-channel form
    // (the store tail of such an epilogue is issue-bound, guide T21)
    const int erow = lane >> 2, ecol = (lane & 3) * 8;
    float4 scv[TN][2], shv[TN][2];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 32 + ecol;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            scv[j][q] = make_float4(1.f, 1.f, 1.f, 1.f); shv[j][q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.scale) {
                scv[j][q] = *reinterpret_cast<const float4 *>(p.scale + n + 4 * q);
                shv[j][q] = *reinterpret_cast<const float4 *>(p.shift + n + 4 * q);
            }
        }
    }

    // ---- prologue: patch of block 0, weights of K-tiles 0 and 1
#pragma unroll
    for (int j = 0; j < NPA; ++j) dma16(lds0 + dstA[j], voffA[j], rsa, 0);
    dma16_group<NPB, 64 * 128>(ldsB, voffB, rsw, 0);
    dma16_group<NPB, 64 * 128>(ldsB + BTILE, voffB, rsw, p.C * 2);                                // K-tile 1 = (block 0, tap 1)
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPB) : "memory");            // patch 0 and K-tile 0 have landed (this wave's share)
    PATCH_BAR();
    if (grp == 1 && !PABL(p, 32)) PATCH_BAR();                             // the second group runs half a step behind

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    bf16x8 fa[2][4], fb_[TN][4];
#ifdef LSPF2F_PATCH_STAMPS
    unsigned long long t_l0 = 0, t_l1 = 0, t_m0 = 0, t_m1 = 0, t_prev = 0, acc_l = 0, acc_b1 = 0, acc_m = 0, acc_b2 = 0, t_start = 0;
    PSTAMP(t_start);
#endif
    for (int cb = 0; cb < NCB; ++cb) {
        const unsigned abuf = (cb & 1) ? (unsigned)PATCH_BYTES : 0u, anext = (cb & 1) ? 0u : (unsigned)PATCH_BYTES;
        const bool more = cb + 1 < NCB;
        const bool last = !more;
        // the copies of tap step `tap` of this block: piece `tap` of the next block's patch (zeros past the last block) ...
        auto issue_a = [&](int tap) {
            if (tap < NPA) dma16(lds0 + anext + dstA[tap < NPA ? tap : 0], more ? voffA[tap < NPA ? tap : 0] : kOOBp, rsa, (cb + 1) * 128);
        };
        // ... and the weights of K-tile t + 2
        auto issue_b = [&](int tap) {
            const int tap2 = tap + 2 < 9 ? tap + 2 : tap + 2 - 9;
            const int cb2 = tap + 2 < 9 ? cb : cb + 1;
            const bool live = tap + 2 < 9 || more;
            unsigned vb[NPB];
#pragma unroll
            for (int k = 0; k < NPB; ++k) vb[k] = live ? voffB[k] : kOOBp;
            dma16_group<NPB, 64 * 128>(ldsB + (unsigned)(((tap + 2) % 3) * BTILE), vb, rsw, (tap2 * p.C + cb2 * 64) * 2);
        };
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const int slot = tap % 3;
            // ---- load segment: fragments of this K-tile ...
            PSTAMP(t_l0);
            if (!PABL(p, 2)) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
                        fb_[j][ks] = *reinterpret_cast<const bf16x8 *>(smem_p16 + slot * BTILE + boff[j][ks]);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int P = PABL(p, 1024) ? p0[i] : p0[i] + ky * PW + kx;
                    const unsigned base = abuf + (unsigned)(P << 7) + (unsigned)(((hh ^ (P >> 1)) & 7) << 4);
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
                        fa[i][ks] = *reinterpret_cast<const bf16x8 *>(smem_p16 + (base ^ (unsigned)(ks << 5)));
                }
            }
#ifndef LSPF2F_PATCH_COPIES_IN_M
            // ... and the copies: one piece of the next block's patch, the weights two K-tiles ahead (into the slot K-tile t - 1 has left); counted wait: K-tile t + 1
            // (and everything older: the patch pieces too) has landed, this step's own pieces may still fly
            if (!PABL(p, 1)) {
                issue_a(tap); issue_b(tap);
                if (PABL(p, 64)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (tap < NPA) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPB + 1) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPB) : "memory");
            }
#else
            // (A-B build -DLSPF2F_PATCH_COPIES_IN_M: the copies ride between the MFMAs instead.  Measured SLOWER: a 256 -> 256 @64x64 layer takes 0.79 instead of 0.74 of
            // the same box's implicit-GEMM time, the forward gains 4 % instead of 8 % (profiles/r06_patch16_ab.txt): a copy then has one and a half segments to land instead of two and the vmcnt(0) here waits for it.)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            if (!PABL(p, 2048)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PSTAMP(t_l1);
            PATCH_BAR();
            PSTAMP(t_m0);
            if (PABL(p, 256)) { PATCH_BAR(); PATCH_BAR(); }
            // ---- MFMA segment (the other group is in its load segment)
            if (PABL(p, 512)) __builtin_amdgcn_s_setprio(1);      // (measured: the MFMA group at raised priority is 0.8 us per layer SLOWER, profiles/r06_patch16_ab.txt)
            if (!PABL(p, 4)) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = mfma32_16b<F16>(fa[i][ks], fb_[j][ks], acc[i][j]);
#ifdef LSPF2F_PATCH_COPIES_IN_M
                    if (!PABL(p, 1)) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (ks == 0) issue_a(tap);
                        if (ks == 1) issue_b(tap);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#endif
                }
            }
#ifdef LSPF2F_PATCH_COPIES_IN_M
            else if (!PABL(p, 1)) { issue_a(tap); issue_b(tap); }
#endif
            if (PABL(p, 512)) __builtin_amdgcn_s_setprio(0);
#ifdef LSPF2F_PATCH_STAMPS
            PSTAMP(t_m1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (t_prev) acc_b2 += t_l0 - t_prev;
            acc_l += t_l1 - t_l0; acc_b1 += t_m0 - t_l1; acc_m += t_m1 - t_m0; t_prev = t_m1;
#endif
            // the second group skips the very last barrier: both groups then pass the same number, and the first one starts its epilogue under the second one's last MFMAs
            if (!(tap == 8 && last && grp == 1 && !PABL(p, 32))) PATCH_BAR();
        }
    }

    // ---- epilogue (the implicit GEMM's: each wave transposes its 32 x 32 tiles through a private LDS patch so a lane ends up with 4 consecutive channels of a pixel).
    // Every K-loop read of LDS is complete (the last barrier above); the patches live in the patch buffer of the LAST channel block, which no copy in flight targets.
#ifdef LSPF2F_PATCH_STAMPS
    unsigned long long t_end; PSTAMP(t_end);
    if (p.stamps && lane == 0) {
        unsigned long long *q = p.stamps + ((size_t)blockIdx.x * 8 + wave) * 8;
        q[0] = acc_l; q[1] = acc_b1; q[2] = acc_m; q[3] = acc_b2; q[4] = t_end - t_start; q[5] = t_start;
    }
#endif
    if (PABL(p, 16)) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    constexpr int EP = 36;
    float *patch = reinterpret_cast<float *>(smem_p16 + (((NCB - 1) & 1) ? PATCH_BYTES : 0)) + wave * (32 * EP);
    const int ccol = lane & 31, crow = 4 * (lane >> 5);
    T *outp = static_cast<T *>(p.out);
    const T *resp = static_cast<const T *>(p.residual);
    // output pixel of this lane in each (pixel block i, pass): fixed per lane; the residual rows are all requested before the first transpose
    size_t opix[2][2];
    u32x4 rres[TN][2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int m = wm * 64 + i * 32 + pass * 16 + erow;
            opix[i][pass] = (size_t)(fb * p.H + y0 + m / TW) * p.W + x0 + (m % TW);
        }
    if (resp) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pass = 0; pass < 2; ++pass)
                    rres[j][i][pass] = *reinterpret_cast<const u32x4 *>(resp + opix[i][pass] * p.Cout + n0 + wn * (BN / 2) + j * 32 + ecol);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 32 + ecol;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                patch[((r & 3) + 8 * (r >> 2) + crow) * EP + ccol] = acc[i][j][r];
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int row = pass * 16 + erow;
                float v[8];
                *reinterpret_cast<float4 *>(v) = *reinterpret_cast<const float4 *>(patch + row * EP + ecol);
                *reinterpret_cast<float4 *>(v + 4) = *reinterpret_cast<const float4 *>(patch + row * EP + ecol + 4);
                const float sc[8] = {scv[j][0].x, scv[j][0].y, scv[j][0].z, scv[j][0].w, scv[j][1].x, scv[j][1].y, scv[j][1].z, scv[j][1].w};
                const float sh[8] = {shv[j][0].x, shv[j][0].y, shv[j][0].z, shv[j][0].w, shv[j][1].x, shv[j][1].y, shv[j][1].z, shv[j][1].w};
                u32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float a = v[2 * q] * sc[2 * q] + sh[2 * q], b = v[2 * q + 1] * sc[2 * q + 1] + sh[2 * q + 1];
                    if (resp) { a += lo16<F16>(rres[j][i][pass][q]); b += hi16<F16>(rres[j][i][pass][q]); }
                    if (p.relu) { a = fmaxf(a, 0.f); b = fmaxf(b, 0.f); }
                    o[q] = pack16x2<F16>(a, b);
                }
                *reinterpret_cast<u32x4 *>(outp + opix[i][pass] * p.Cout + n) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------------------------------
// 64 channels per workgroup (the 32x32 level at 8 frames: 128-channel tiles would leave the chip half empty): a wave owns 64 pixels x 32 channels, ONE 32-channel block,
// so a K-tile is only 8 MFMAs (256 cycles) against 12 fragment reads and 2 copies -- in the form above the LOAD segment is then twice as long as the MFMA segment, and most
// of it is the issue cost of the two LDS-DMA statements (100-185 cycles each next to the reads).  Here the copies are issued BETWEEN the MFMAs, where a wave has issue
// slots to spare, and the weights run FOUR K-tiles ahead through a 6-slot ring (8 KB per slot at 64 channels: the LDS has room), so a copy has three and a half segments
// to land: the load segment is fragment reads + one counted wait.  (The same move with the 3-slot ring of the 128-channel tiles measured slower: one and a half segments
// are not enough, see conv3x3_patch16.)
//   ring slot of K-tile t = t mod 6 = (3 cb + tap) mod 6: a scalar per block;  patch pieces of block cb + 1: two per tap over taps 0..3 (all older than the copies of
//   taps 6 and 7, which are the ones still in flight when block cb + 1 is first read);
//   counted wait at the end of load segment tau: everything but the pieces issued in the two MFMA segments before it -- K-tile tau + 1, issued four segments ago, has landed.
template <bool F16, int TW, int TR>
__global__ __launch_bounds__(512) void conv3x3_patch16d(const PatchConvParams p)
{
    typedef typename St16<F16>::type T;
    constexpr int BN = 64, NSB = 6, DIST = 4;
    constexpr int PW = TW + 2, PPX = PW * (TR + 2);
    constexpr int NPIECE = (PPX + 7) / 8;
    constexpr int NPA = (NPIECE + 7) / 8;
    constexpr int PATCH_BYTES = (NPIECE + 1) * 1024;
    constexpr int BRING = 2 * PATCH_BYTES;
    constexpr int BTILE = BN * 128;
    static_assert(TW * TR == 256 && TW % 32 == 0, "a wave's 32-pixel MFMA block is 32 consecutive pixels of one row");
    static_assert(NPA <= 8, "two patch pieces per tap step over taps 0..3");
    static_assert(8 * 32 * 36 * 4 <= PATCH_BYTES, "the epilogue's transpose patches live in the patch buffer of the last channel block");

    extern __shared__ __attribute__((aligned(16))) char smem_p16[];
    typedef __attribute__((address_space(3))) char lds_char;
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_char *)smem_p16;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, grp = wave >> 2;
    asm volatile("" :: "s"(p.src), "s"(p.w), "s"(p.scale), "s"(p.shift), "s"(p.residual), "s"(p.out), "s"(p.B), "s"(p.H), "s"(p.W), "s"(p.C), "s"(p.Cout),
                       "s"(p.relu), "s"(p.ntm), "s"(p.ntn), "s"(p.div_tpi.m), "s"(p.div_tpi.s1), "s"(p.div_tpi.s2), "s"(p.div_tx.m), "s"(p.div_tx.s1), "s"(p.div_tx.s2),
                       "s"(p.div_ntn.m), "s"(p.div_ntn.s1), "s"(p.div_ntn.s2));

    unsigned lin = blockIdx.x;
    {
        const unsigned total = gridDim.x, q = total >> 3, r = total & 7, x = lin & 7;
        lin = x * q + (x < r ? x : r) + (lin >> 3);
    }
    const int mt = (int)p.div_ntn.div(lin), nt = (int)lin - mt * p.ntn;
    const int fb = (int)p.div_tpi.div((unsigned)mt);
    const int rem = mt - fb * p.tiles_per_img;
    const int ty = (int)p.div_tx.div((unsigned)rem), tx = rem - ty * p.tiles_x;
    const int y0 = ty * TR, x0 = tx * TW, n0 = nt * BN;
    const int NCB = p.C >> 6;
    const int Krow = 9 * p.C;

    unsigned voffA[NPA], dstA[NPA];
#pragma unroll
    for (int j = 0; j < NPA; ++j) {
        const int q = j * 8 + wave;
        const int pp = q * 8 + (lane >> 3), s = lane & 7;
        const int pr = pp / PW, pc = pp - pr * PW;
        const int y = y0 - 1 + pr, x = x0 - 1 + pc;
        const bool ok = pp < PPX && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        voffA[j] = ok ? (unsigned)((fb * p.H + y) * p.W + x) * (unsigned)(p.C * 2) + (unsigned)((s ^ ((pp >> 1) & 7)) << 4) : kOOBp;
        dstA[j] = (unsigned)((q < NPIECE ? q : NPIECE) * 1024);
    }
    unsigned voffB;                                                        // one weight piece per wave and K-tile: rows 8 wave .. + 7 of the 64
    {
        const int row = wave * 8 + (lane >> 3), s = lane & 7;
        voffB = (unsigned)((n0 + row) * Krow * 2) + (unsigned)((s ^ ((row >> 1) & 7)) << 4);
    }
    const i32x4 rsa = make_srd(p.src, (unsigned)(p.B * p.H * p.W) * (unsigned)(p.C * 2));
    const i32x4 rsw = make_srd(p.w, (unsigned)(p.Cout * Krow * 2));
    const unsigned ldsB = lds0 + BRING + (unsigned)(wave * 8 * 128);

    const int l31 = lane & 31, hh = lane >> 5;
    int p0[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = wm * 64 + i * 32 + l31;
        p0[i] = (m / TW) * PW + (m % TW);
    }
    unsigned boff[4];
    {
        const int nl = wn * 32 + l31;
        const unsigned base = (unsigned)BRING + (unsigned)(nl * 128) + (unsigned)(((hh ^ (nl >> 1)) & 7) << 4);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) boff[ks] = base ^ (unsigned)(ks << 5);
    }

    const int erow = lane >> 2, ecol = (lane & 3) * 8;
    float4 scv[2], shv[2];
    {
        const int n = n0 + wn * 32 + ecol;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            scv[q] = make_float4(1.f, 1.f, 1.f, 1.f); shv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.scale) {
                scv[q] = *reinterpret_cast<const float4 *>(p.scale + n + 4 * q);
                shv[q] = *reinterpret_cast<const float4 *>(p.shift + n + 4 * q);
            }
        }
    }

    // ---- prologue: patch of block 0, weights of K-tiles 0..3 (block 0, taps 0..3: NCB >= 2 and 9 taps per block)
#pragma unroll
    for (int j = 0; j < NPA; ++j) dma16(lds0 + dstA[j], voffA[j], rsa, 0);
#pragma unroll
    for (int t = 0; t < DIST; ++t) dma16(ldsB + (unsigned)(t * BTILE), voffB, rsw, t * p.C * 2);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DIST - 1) : "memory");        // patch 0 and K-tile 0 have landed
    PATCH_BAR();
    if (grp == 1) PATCH_BAR();

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    bf16x8 fa[2][4], fb_[4];
    int s0 = 0;                                                            // ring slot of this block's tap 0 = (9 cb) mod 6 = 0 | 3
    for (int cb = 0; cb < NCB; ++cb) {
        const unsigned abuf = (cb & 1) ? (unsigned)PATCH_BYTES : 0u, anext = (cb & 1) ? 0u : (unsigned)PATCH_BYTES;
        const bool more = cb + 1 < NCB;
        const bool last = !more;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            int sl = s0 + tap; if (sl >= NSB) sl -= NSB; if (sl >= NSB) sl -= NSB;
            const unsigned slb = (unsigned)(sl * BTILE);
            // ---- load segment: fragments only
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fb_[ks] = *reinterpret_cast<const bf16x8 *>(smem_p16 + (boff[ks] + slb));
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int P = p0[i] + ky * PW + kx;
                const unsigned base = abuf + (unsigned)(P << 7) + (unsigned)(((hh ^ (P >> 1)) & 7) << 4);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    fa[i][ks] = *reinterpret_cast<const bf16x8 *>(smem_p16 + (base ^ (unsigned)(ks << 5)));
            }
            // K-tile tau + 1 has landed: what may still fly are the pieces of the two MFMA segments before this one -- taps tau - 2 and tau - 1 (mod 9: the weights of a tap and,
            // for taps 0..3, two patch pieces)
            {
                const int t1 = (tap + 8) % 9, t2 = (tap + 7) % 9;
                const int n = 2 + (t1 < 4 ? 2 : 0) + (t2 < 4 ? 2 : 0);
                if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PATCH_BAR();
            // ---- MFMA segment with this step's copies between the MFMAs: two pieces of the next block's patch (taps 0..3) and the weights of K-tile t + 4
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i] = mfma32_16b<F16>(fa[i][ks], fb_[ks], acc[i]);
                __builtin_amdgcn_sched_barrier(0);
                if (ks == 0 && tap < 4) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int j = tap * 2 + k;
                        const unsigned va = (j < NPA && more) ? voffA[j < NPA ? j : 0] : kOOBp;
                        const unsigned dk = j < NPA ? dstA[j < NPA ? j : 0] : (unsigned)(NPIECE * 1024);
                        dma16(lds0 + anext + dk, va, rsa, (cb + 1) * 128);
                    }
                }
                if (ks == 1) {
                    const int tap2 = tap + DIST < 9 ? tap + DIST : tap + DIST - 9;
                    const int cb2 = tap + DIST < 9 ? cb : cb + 1;
                    const bool live = tap + DIST < 9 || more;
                    int sl2 = sl + DIST; if (sl2 >= NSB) sl2 -= NSB;
                    dma16(ldsB + (unsigned)(sl2 * BTILE), live ? voffB : kOOBp, rsw, (tap2 * p.C + cb2 * 64) * 2);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!(tap == 8 && last && grp == 1)) PATCH_BAR();
        }
        s0 = s0 == 0 ? 3 : 0;
    }

    // ---- epilogue
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    constexpr int EP = 36;
    float *patch = reinterpret_cast<float *>(smem_p16 + (((NCB - 1) & 1) ? PATCH_BYTES : 0)) + wave * (32 * EP);
    const int ccol = lane & 31, crow = 4 * (lane >> 5);
    T *outp = static_cast<T *>(p.out);
    const T *resp = static_cast<const T *>(p.residual);
    size_t opix[2][2];
    u32x4 rres[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int m = wm * 64 + i * 32 + pass * 16 + erow;
            opix[i][pass] = (size_t)(fb * p.H + y0 + m / TW) * p.W + x0 + (m % TW);
        }
    const int n = n0 + wn * 32 + ecol;
    if (resp) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pass = 0; pass < 2; ++pass)
                rres[i][pass] = *reinterpret_cast<const u32x4 *>(resp + opix[i][pass] * p.Cout + n);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            patch[((r & 3) + 8 * (r >> 2) + crow) * EP + ccol] = acc[i][r];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int row = pass * 16 + erow;
            float v[8];
            *reinterpret_cast<float4 *>(v) = *reinterpret_cast<const float4 *>(patch + row * EP + ecol);
            *reinterpret_cast<float4 *>(v + 4) = *reinterpret_cast<const float4 *>(patch + row * EP + ecol + 4);
            const float sc[8] = {scv[0].x, scv[0].y, scv[0].z, scv[0].w, scv[1].x, scv[1].y, scv[1].z, scv[1].w};
            const float sh[8] = {shv[0].x, shv[0].y, shv[0].z, shv[0].w, shv[1].x, shv[1].y, shv[1].z, shv[1].w};
            u32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float a = v[2 * q] * sc[2 * q] + sh[2 * q], b = v[2 * q + 1] * sc[2 * q + 1] + sh[2 * q + 1];
                if (resp) { a += lo16<F16>(rres[i][pass][q]); b += hi16<F16>(rres[i][pass][q]); }
                if (p.relu) { a = fmaxf(a, 0.f); b = fmaxf(b, 0.f); }
                o[q] = pack16x2<F16>(a, b);
            }
            *reinterpret_cast<u32x4 *>(outp + opix[i][pass] * p.Cout + n) = o;
        }
    }
}

template <bool F16, int TW, int TR>
static hipError_t launch_patch16d_t(const PatchConvParams &p_in, hipStream_t s)
{
    constexpr int PPX = (TW + 2) * (TR + 2), NPIECE = (PPX + 7) / 8;
    constexpr size_t smem = (size_t)2 * (NPIECE + 1) * 1024 + (size_t)6 * 64 * 128;
    static_assert(smem <= 160 * 1024, "LDS");
    PatchConvParams p = p_in;
    p.tiles_x = p.W / TW;
    p.tiles_per_img = (p.H / TR) * p.tiles_x;
    p.ntm = p.B * p.tiles_per_img; p.ntn = p.Cout / 64;
    p.div_tpi = FastDiv::make((unsigned)p.tiles_per_img);
    p.div_tx = FastDiv::make((unsigned)p.tiles_x);
    p.div_ntn = FastDiv::make((unsigned)p.ntn);
    static AttrMask attr_mask;
    if (attr_needed_on_this_device(attr_mask)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_patch16d<F16, TW, TR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        attr_done_on_this_device(attr_mask);
    }
    hipLaunchKernelGGL((conv3x3_patch16d<F16, TW, TR>), dim3((unsigned)(p.ntm * p.ntn)), dim3(512), smem, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------------------------------------------------
// The same structure for the sub-pixel up-convs (Upsample x2 + Conv3x3 over the concat of two equally wide sources, models/networks.py:610-611, in the implicit GEMM's
// up4 form: 4 output parities x 2x2 taps on the LOW-res source with pre-summed weights [4][Cout][2][2][Cin]).  A workgroup = (parity, 256 low-res pixels, BN channels):
// tap (a, b) of parity (py, px) reads patch records (r + py + a, c + px + b), so the same (TR + 2) x (TW + 2) patch serves every parity; the four workgroups of a
// tile sit next to each other in an XCD's chunk and share it in L2.  Per 64-channel block there are 4 K-tiles instead of 9: the ring slot of a K-tile is no longer a
// constant of the unrolled tap ((4 cb + tap) mod 3 = (cb + tap) mod 3: a scalar per block), and the next block's patch rides in ceil(NPA / 3) pieces per tap over
// taps 0..2.  Everything else -- two wave groups half a step apart, counted waits, hazards by barrier interval, epilogue -- as in conv3x3_patch16 above.
template <bool F16, int TW, int TR, int BN>
__global__ __launch_bounds__(512) void conv3x3_patchup16(const PatchConvParams p)
{
    typedef typename St16<F16>::type T;
    constexpr int PW = TW + 2, PPX = PW * (TR + 2);
    constexpr int NPIECE = (PPX + 7) / 8;
    constexpr int NPA = (NPIECE + 7) / 8;
    constexpr int PPS = (NPA + 2) / 3;                       // patch pieces per tap step (taps 0..2)
    constexpr int NPB = BN / 64;
    constexpr int TN = BN / 64;
    constexpr int PATCH_BYTES = (NPIECE + 1) * 1024;
    constexpr int BRING = 2 * PATCH_BYTES;
    constexpr int BTILE = BN * 128;
    static_assert(TW * TR == 256 && TW % 16 == 0, "a wave's 32-pixel MFMA block is 32 consecutive pixels of one row, or (16-pixel tiles) two half-rows: 2 of a ds_read_b128 group's 16 lanes then share a bank slot");
    static_assert(8 * 32 * 36 * 4 <= PATCH_BYTES, "the epilogue's transpose patches live in the patch buffer of the last channel block");

    extern __shared__ __attribute__((aligned(16))) char smem_p16[];
    typedef __attribute__((address_space(3))) char lds_char;
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_char *)smem_p16;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, grp = wave >> 2;
    asm volatile("" :: "s"(p.src), "s"(p.src1), "s"(p.w), "s"(p.scale), "s"(p.shift), "s"(p.residual), "s"(p.out), "s"(p.B), "s"(p.H), "s"(p.W), "s"(p.C), "s"(p.C1), "s"(p.Cout),
                       "s"(p.relu), "s"(p.ntm), "s"(p.ntn), "s"(p.div_tpi.m), "s"(p.div_tpi.s1), "s"(p.div_tpi.s2), "s"(p.div_tx.m), "s"(p.div_tx.s1), "s"(p.div_tx.s2),
                       "s"(p.div_ntn.m), "s"(p.div_ntn.s1), "s"(p.div_ntn.s2));

    // ---- tile: channel tile fastest, then the 4 parities of a pixel tile (they share its patch), then pixel tiles; 8 contiguous chunks, one per XCD
    unsigned lin = blockIdx.x;
    {
        const unsigned total = gridDim.x, q = total >> 3, r = total & 7, x = lin & 7;
        lin = x * q + (x < r ? x : r) + (lin >> 3);
    }
    const int t2 = (int)p.div_ntn.div(lin), nt = (int)lin - t2 * p.ntn;
    const int par = t2 & 3, mt = t2 >> 2;
    const int py = par >> 1, px = par & 1;
    const int fb = (int)p.div_tpi.div((unsigned)mt);
    const int rem = mt - fb * p.tiles_per_img;
    const int ty = (int)p.div_tx.div((unsigned)rem), tx = rem - ty * p.tiles_x;
    const int y0 = ty * TR, x0 = tx * TW, n0 = nt * BN;
    const int Cin = p.C + p.C1;
    const int NCB = Cin >> 6, NCB0 = p.C >> 6;
    const int Krow = 4 * Cin;

    unsigned voffA[NPA], voffB[NPB];
    unsigned dstA[NPA];
#pragma unroll
    for (int j = 0; j < NPA; ++j) {
        const int q = j * 8 + wave;
        const int pp = q * 8 + (lane >> 3), s = lane & 7;
        const int pr = pp / PW, pc = pp - pr * PW;
        const int y = y0 - 1 + pr, x = x0 - 1 + pc;
        const bool ok = pp < PPX && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        voffA[j] = ok ? (unsigned)((fb * p.H + y) * p.W + x) * (unsigned)(p.C * 2) + (unsigned)((s ^ ((pp >> 1) & 7)) << 4) : kOOBp;      // (both sources have p.C channels)
        dstA[j] = (unsigned)((q < NPIECE ? q : NPIECE) * 1024);
    }
#pragma unroll
    for (int k = 0; k < NPB; ++k) {
        const int row = wave * 8 + k * 64 + (lane >> 3), s = lane & 7;
        voffB[k] = (unsigned)((n0 + row) * Krow * 2) + (unsigned)((s ^ ((row >> 1) & 7)) << 4);
    }
    const unsigned src_bytes = (unsigned)(p.B * p.H * p.W) * (unsigned)(p.C * 2);
    const i32x4 rsa0 = make_srd(p.src, src_bytes);
    const i32x4 rsa1 = make_srd(p.C1 ? p.src1 : p.src, src_bytes);
    const i32x4 rsw = make_srd(static_cast<const char *>(p.w) + (size_t)par * p.Cout * Krow * 2, (unsigned)(p.Cout * Krow * 2));
    const unsigned ldsB = lds0 + BRING + (unsigned)(wave * 8 * 128);

    const int l31 = lane & 31, hh = lane >> 5;
    int p0[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = wm * 64 + i * 32 + l31;
        p0[i] = (m / TW + py) * PW + (m % TW) + px;                        // patch record of tap (0, 0) of THIS parity
    }
    unsigned boff[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nl = wn * (BN / 2) + j * 32 + l31;
        const unsigned base = (unsigned)BRING + (unsigned)(nl * 128) + (unsigned)(((hh ^ (nl >> 1)) & 7) << 4);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) boff[j][ks] = base ^ (unsigned)(ks << 5);
    }

    const int erow = lane >> 2, ecol = (lane & 3) * 8;
    float4 scv[TN][2], shv[TN][2];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 32 + ecol;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            scv[j][q] = make_float4(1.f, 1.f, 1.f, 1.f); shv[j][q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.scale) {
                scv[j][q] = *reinterpret_cast<const float4 *>(p.scale + n + 4 * q);
                shv[j][q] = *reinterpret_cast<const float4 *>(p.shift + n + 4 * q);
            }
        }
    }

    // ---- prologue: patch of block 0, weights of K-tiles 0 and 1 (ring slots 0 and 1)
#pragma unroll
    for (int j = 0; j < NPA; ++j) dma16(lds0 + dstA[j], voffA[j], rsa0, 0);
    dma16_group<NPB, 64 * 128>(ldsB, voffB, rsw, 0);
    dma16_group<NPB, 64 * 128>(ldsB + BTILE, voffB, rsw, Cin * 2);                                // K-tile 1 = (block 0, tap 1)
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPB) : "memory");
    PATCH_BAR();
    if (grp == 1) PATCH_BAR();

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    bf16x8 fa[2][4], fb_[TN][4];
    int s0 = 0;                                                            // ring slot of this block's tap 0 = (4 cb) mod 3 = cb mod 3
    for (int cb = 0; cb < NCB; ++cb) {
        const unsigned abuf = (cb & 1) ? (unsigned)PATCH_BYTES : 0u, anext = (cb & 1) ? 0u : (unsigned)PATCH_BYTES;
        const bool more = cb + 1 < NCB;
        const bool last = !more;
        const bool nsrc1 = cb + 1 >= NCB0;                                 // the NEXT block's source
        const i32x4 rsn = nsrc1 ? rsa1 : rsa0;
        const int nsoff = (nsrc1 ? cb + 1 - NCB0 : cb + 1) * 128;
#pragma unroll
        for (int tap = 0; tap < 4; ++tap) {
            int sl = s0 + tap; if (sl >= 3) sl -= 3; if (sl >= 3) sl -= 3;
            const unsigned slb = (unsigned)(sl * BTILE);
            // ---- load segment
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    fb_[j][ks] = *reinterpret_cast<const bf16x8 *>(smem_p16 + (boff[j][ks] + slb));
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int P = p0[i] + (tap >> 1) * PW + (tap & 1);
                const unsigned base = abuf + (unsigned)(P << 7) + (unsigned)(((hh ^ (P >> 1)) & 7) << 4);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    fa[i][ks] = *reinterpret_cast<const bf16x8 *>(smem_p16 + (base ^ (unsigned)(ks << 5)));
            }
            if (tap < 3) {                                                 // PPS pieces of the next block's patch (zeros into the dump slot where a wave has no piece left)
#pragma unroll
                for (int k = 0; k < PPS; ++k) {
                    const int j = tap * PPS + k;
                    const unsigned va = (j < NPA && more) ? voffA[j < NPA ? j : 0] : kOOBp;
                    const unsigned dk = j < NPA ? dstA[j < NPA ? j : 0] : (unsigned)(NPIECE * 1024);
                    dma16(lds0 + anext + dk, va, rsn, nsoff);
                }
            }
            {
                const int tap2 = tap + 2 < 4 ? tap + 2 : tap + 2 - 4;
                const int cb2 = tap + 2 < 4 ? cb : cb + 1;
                const bool live = tap + 2 < 4 || more;
                int sl2 = sl + 2; if (sl2 >= 3) sl2 -= 3;
                unsigned vb[NPB];
#pragma unroll
                for (int k = 0; k < NPB; ++k) vb[k] = live ? voffB[k] : kOOBp;
                dma16_group<NPB, 64 * 128>(ldsB + (unsigned)(sl2 * BTILE), vb, rsw, (tap2 * Cin + cb2 * 64) * 2);
            }
            if (tap < 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPB + PPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPB) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PATCH_BAR();
            // ---- MFMA segment
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = mfma32_16b<F16>(fa[i][ks], fb_[j][ks], acc[i][j]);
            if (!(tap == 3 && last && grp == 1)) PATCH_BAR();
        }
        s0 = s0 + 1 == 3 ? 0 : s0 + 1;
    }

    // ---- epilogue: output pixel (2 y + py, 2 x + px) of the [2H][2W] frame
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    constexpr int EP = 36;
    float *patch = reinterpret_cast<float *>(smem_p16 + (((NCB - 1) & 1) ? PATCH_BYTES : 0)) + wave * (32 * EP);
    const int ccol = lane & 31, crow = 4 * (lane >> 5);
    T *outp = static_cast<T *>(p.out);
    const T *resp = static_cast<const T *>(p.residual);
    size_t opix[2][2];
    u32x4 rres[TN][2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int m = wm * 64 + i * 32 + pass * 16 + erow;
            opix[i][pass] = ((size_t)(fb * 2 * p.H + 2 * (y0 + m / TW) + py) * (2 * p.W)) + 2 * (x0 + (m % TW)) + px;
        }
    if (resp) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pass = 0; pass < 2; ++pass)
                    rres[j][i][pass] = *reinterpret_cast<const u32x4 *>(resp + opix[i][pass] * p.Cout + n0 + wn * (BN / 2) + j * 32 + ecol);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 32 + ecol;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                patch[((r & 3) + 8 * (r >> 2) + crow) * EP + ccol] = acc[i][j][r];
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int row = pass * 16 + erow;
                float v[8];
                *reinterpret_cast<float4 *>(v) = *reinterpret_cast<const float4 *>(patch + row * EP + ecol);
                *reinterpret_cast<float4 *>(v + 4) = *reinterpret_cast<const float4 *>(patch + row * EP + ecol + 4);
                const float sc[8] = {scv[j][0].x, scv[j][0].y, scv[j][0].z, scv[j][0].w, scv[j][1].x, scv[j][1].y, scv[j][1].z, scv[j][1].w};
                const float sh[8] = {shv[j][0].x, shv[j][0].y, shv[j][0].z, shv[j][0].w, shv[j][1].x, shv[j][1].y, shv[j][1].z, shv[j][1].w};
                u32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float a = v[2 * q] * sc[2 * q] + sh[2 * q], b = v[2 * q + 1] * sc[2 * q + 1] + sh[2 * q + 1];
                    if (resp) { a += lo16<F16>(rres[j][i][pass][q]); b += hi16<F16>(rres[j][i][pass][q]); }
                    if (p.relu) { a = fmaxf(a, 0.f); b = fmaxf(b, 0.f); }
                    o[q] = pack16x2<F16>(a, b);
                }
                *reinterpret_cast<u32x4 *>(outp + opix[i][pass] * p.Cout + n) = o;
            }
        }
    }
}

template <bool F16, int TW, int TR, int BN>
static hipError_t launch_patchup16_t(const PatchConvParams &p_in, hipStream_t s)
{
    constexpr int PPX = (TW + 2) * (TR + 2), NPIECE = (PPX + 7) / 8;
    constexpr size_t smem = (size_t)2 * (NPIECE + 1) * 1024 + (size_t)3 * BN * 128;
    static_assert(smem <= 160 * 1024, "LDS");
    PatchConvParams p = p_in;
    p.tiles_x = p.W / TW;
    p.tiles_per_img = (p.H / TR) * p.tiles_x;
    p.ntm = p.B * p.tiles_per_img; p.ntn = p.Cout / BN;
    p.div_tpi = FastDiv::make((unsigned)p.tiles_per_img);
    p.div_tx = FastDiv::make((unsigned)p.tiles_x);
    p.div_ntn = FastDiv::make((unsigned)p.ntn);
    static AttrMask attr_mask;
    if (attr_needed_on_this_device(attr_mask)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_patchup16<F16, TW, TR, BN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        attr_done_on_this_device(attr_mask);
    }
    hipLaunchKernelGGL((conv3x3_patchup16<F16, TW, TR, BN>), dim3((unsigned)(p.ntm * p.ntn * 4)), dim3(512), smem, s, p);
    return hipGetLastError();
}

template <bool F16, int TW, int TR, int BN>
static hipError_t launch_patch16_t(const PatchConvParams &p_in, hipStream_t s)
{
    constexpr int PPX = (TW + 2) * (TR + 2), NPIECE = (PPX + 7) / 8;
    constexpr size_t smem = (size_t)2 * (NPIECE + 1) * 1024 + (size_t)3 * BN * 128;
    static_assert(smem <= 160 * 1024, "LDS");
    PatchConvParams p = p_in;
    p.tiles_x = p.W / TW;
    p.tiles_per_img = (p.H / TR) * p.tiles_x;
    p.ntm = p.B * p.tiles_per_img; p.ntn = p.Cout / BN;
    p.div_tpi = FastDiv::make((unsigned)p.tiles_per_img);
    p.div_tx = FastDiv::make((unsigned)p.tiles_x);
    p.div_ntn = FastDiv::make((unsigned)p.ntn);
    static AttrMask attr_mask;
    if (attr_needed_on_this_device(attr_mask)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_patch16<F16, TW, TR, BN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        attr_done_on_this_device(attr_mask);
    }
    hipLaunchKernelGGL((conv3x3_patch16<F16, TW, TR, BN>), dim3((unsigned)(p.ntm * p.ntn)), dim3(512), smem, s, p);
    return hipGetLastError();
}

bool patch16_supported(const PatchConvParams &p, int tw, int bn)
{
    if (p.dtype != 1 && p.dtype != 2) return false;
    if (!((tw == 64 && (bn == 128 || bn == 64)) || (tw == 32 && (bn == 128 || bn == 64)))) return false;
    const int tr = 256 / tw;
    if (p.B < 1 || p.W % tw || p.H % tr || p.C % 64 || p.C < 128 || p.Cout % bn) return false;
    // 32-bit buffer offsets with the top bit reserved as the out-of-range marker
    if ((size_t)p.B * p.H * p.W * p.C * 2 > 0x7fffffffull || (size_t)p.Cout * 9 * p.C * 2 > 0x7fffffffull) return false;
    return true;
}

bool patchup16_supported(const PatchConvParams &p, int tw, int bn)
{
    if (p.dtype != 1 && p.dtype != 2) return false;
    if ((tw != 64 && tw != 32 && tw != 16) || (bn != 128 && bn != 64) || (tw == 16 && bn != 64)) return false;
    const int tr = 256 / tw;
    if (p.B < 1 || p.W % tw || p.H % tr || p.C % 64 || (p.C1 != 0 && p.C1 != p.C) || p.C + p.C1 < 128 || p.Cout % bn) return false;
    if (p.C1 && !p.src1) return false;
    if ((size_t)p.B * p.H * p.W * p.C * 2 > 0x7fffffffull || (size_t)p.Cout * 4 * (p.C + p.C1) * 2 > 0x7fffffffull) return false;
    return true;
}

hipError_t launch_patchup16(const PatchConvParams &p, int tw, int bn, hipStream_t s)
{
    if (!patchup16_supported(p, tw, bn)) return hipErrorInvalidValue;
    if (p.dtype == 2) {
        if (tw == 16) return launch_patchup16_t<true, 16, 16, 64>(p, s);
        if (tw == 64) return bn == 128 ? launch_patchup16_t<true, 64, 4, 128>(p, s) : launch_patchup16_t<true, 64, 4, 64>(p, s);
        return bn == 128 ? launch_patchup16_t<true, 32, 8, 128>(p, s) : launch_patchup16_t<true, 32, 8, 64>(p, s);
    }
    if (tw == 16) return launch_patchup16_t<false, 16, 16, 64>(p, s);
    if (tw == 64) return bn == 128 ? launch_patchup16_t<false, 64, 4, 128>(p, s) : launch_patchup16_t<false, 64, 4, 64>(p, s);
    return bn == 128 ? launch_patchup16_t<false, 32, 8, 128>(p, s) : launch_patchup16_t<false, 32, 8, 64>(p, s);
}

hipError_t launch_patch16(const PatchConvParams &p, int tw, int bn, hipStream_t s)
{
    if (!patch16_supported(p, tw, bn)) return hipErrorInvalidValue;
    // 64 channels per workgroup: the deep-ring form (conv3x3_patch16d) unless the caller asks for the first one (deep == 0: A-B runs, tile_n 65 of lspf2f_conv3x3)
    if (bn == 64 && p.deep) {
        if (p.dtype == 2) return tw == 64 ? launch_patch16d_t<true, 64, 4>(p, s) : launch_patch16d_t<true, 32, 8>(p, s);
        return tw == 64 ? launch_patch16d_t<false, 64, 4>(p, s) : launch_patch16d_t<false, 32, 8>(p, s);
    }
    if (p.dtype == 2) {
        if (tw == 64) return bn == 128 ? launch_patch16_t<true, 64, 4, 128>(p, s) : launch_patch16_t<true, 64, 4, 64>(p, s);
        return bn == 128 ? launch_patch16_t<true, 32, 8, 128>(p, s) : launch_patch16_t<true, 32, 8, 64>(p, s);
    }
    if (tw == 64) return bn == 128 ? launch_patch16_t<false, 64, 4, 128>(p, s) : launch_patch16_t<false, 64, 4, 64>(p, s);
    return bn == 128 ? launch_patch16_t<false, 32, 8, 128>(p, s) : launch_patch16_t<false, 32, 8, 64>(p, s);
}

}  // namespace lspf2f
