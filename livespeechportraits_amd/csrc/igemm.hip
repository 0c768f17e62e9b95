// gfx950 (MI355X / CDNA4): the fused 3x3 convolution as an implicit GEMM on the matrix cores, plus the
// deterministic split-K reduction.  See DESIGN.md section 4.1.
//
//  igemm3x3<T,...>   im2col gather (stride 1|2, sub-pixel or 9-tap nearest-x2 upsample, optional second source =
//                    the never-materialised torch.cat) -> LDS-DMA staging -> MFMA (fp32: v_mfma_f32_32x32x2_f32,
//                    bf16: v_mfma_f32_32x32x16_bf16) -> epilogue (folded BatchNorm scale/shift, residual add, ReLU)
//                    or fp32 split-K partials.
//  splitk_reduce<T>  fixed-order reduction of the partials + the same epilogue.
//
// Reference semantics: models/networks.py:592-640 (level layout), :650-675 (ResidualBlock); BatchNorm eval
// folding validated in SURVEY.md 8c.
#include "device_common.h"
#include "kernels.h"
#include <cstdlib>


// Ablation switches (tools/ablate.sh builds with -DLSPF2F_ABLATE); the shipped kernel has none of them.
#ifdef LSPF2F_ABLATE
#define ABL(p, bit) ((p).dbg & (bit))
#else
#define ABL(p, bit) 0
#endif


namespace lspf2f {

// phase timestamps (tools/time_conv.py; builds with -DLSPF2F_IGEMM_STAMPS only): s_memtime values are wave-uniform scalars kept in
// registers and written once at the end, so the instrumented kernel keeps its control flow
#ifdef LSPF2F_IGEMM_STAMPS
#define ISTAMP(i) do { __builtin_amdgcn_sched_barrier(0); stamp_t[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define ISTAMP_DECL unsigned long long stamp_t[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define ISTAMP_FLUSH do { if (p.stamps && (threadIdx.x & 63) == 0) { unsigned long long *q_ = p.stamps + ((size_t)(blockIdx.x + blockIdx.y * gridDim.x) * 4 + (threadIdx.x >> 6)) * 16; \
    for (int i_ = 0; i_ < 16; ++i_) q_[i_] = stamp_t[i_]; } } while (0)
#else
#define ISTAMP(i) do {} while (0)
#define ISTAMP_DECL do {} while (0)
#define ISTAMP_FLUSH do {} while (0)
#endif

static constexpr unsigned kOOB = 0x80000000u;   // voffset beyond any num_records: buffer load returns 0

static constexpr int BK = 32;    // K-tile (floats); Cin % 32 == 0 so a K-tile never straddles a tap
static constexpr int LDK = 32;   // LDS row pitch in floats (128 B, unpadded: the tile is written by LDS-DMA,
                                 // whose destination is lane-linear).  Bank conflicts are avoided by an XOR
                                 // swizzle instead: 16-B slot s of row r holds k-quad s ^ ((r >> 1) & 7), which
                                 // puts the 16 rows of every ds_read_b128 lane group on 16 distinct bank slots.

// `small` U-Net plans (csrc/unet_plan.cpp, models/networks.py:737-767 of the reference): the stored skip tensor d of a down-conv is only ever read as
// leaky_relu(d, 0.2) by the next Conv2d(k4, s2, p1) -- through its space-to-depth image -- and as relu(d) by the up-conv (the in-place activations of the
// reference), so the producer writes those two tensors instead of d (what lspf2f_unet_prepare did in a pass of its own): row `orow` = pixel (b, y, x) of
// an Ho x Wo frame, channel quad n.  Ho, Wo even.
template <typename T>
__device__ __forceinline__ void unet_dual_store(const IgemmParams &p, unsigned orow, int n, float4 v)
{
    const unsigned hw = (unsigned)(p.Ho * p.Wo);
    const unsigned b = p.div_rhw.div(orow), r = orow - b * hw;
    const unsigned y = p.div_rw.div(r), x = r - y * (unsigned)p.Wo;
    store4(static_cast<T *>(p.relu_out) + (size_t)orow * p.Cout + n, make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)));
    const size_t opix = ((size_t)b * (unsigned)(p.Ho >> 1) + (y >> 1)) * (unsigned)(p.Wo >> 1) + (x >> 1);
    const float sl = p.slope;
    store4(static_cast<T *>(p.s2d_out) + opix * (size_t)(4 * p.Cout) + ((y & 1u) * 2u + (x & 1u)) * (unsigned)p.Cout + n,
           make_float4(v.x > 0.f ? v.x : sl * v.x, v.y > 0.f ? v.y : sl * v.y, v.z > 0.f ? v.z : sl * v.z, v.w > 0.f ? v.w : sl * v.w));
}

// G = K-tiles staged per pipeline step (one barrier per G tiles, G tiles of global loads in
// flight per thread).  G = 1 for long K loops; G = 4 turns a short split-K range (<= 4 tiles)
// into a single load -> LDS -> MFMA pass, which is what the latency-bound <= 8x8 levels need.
// UP = the 9-tap nearest-x2 gather form (only the small, weight-streaming up-convs use it).
// T = storage type of activations and weights: float (exact fp32 MFMA, v_mfma_f32_32x32x2_f32) or bf16_t
// (v_mfma_f32_32x32x16_bf16, fp32 accumulate and epilogue).  A K-tile is always 128 B of channels (32 fp32 /
// 64 bf16), so the LDS geometry, the DMA pieces and the swizzle are identical for both.
template <typename T, int BM, int BN, int WGM, int WGN, int G, bool UP, bool KM = false>      // KM: masked K (p.kmask), its own instances so that the others carry none of it
__global__ __launch_bounds__(64 * WGM * WGN) void igemm3x3(const IgemmParams p)
{
    constexpr int EB = (int)sizeof(T);               // element bytes
    constexpr int BKE = 128 / EB;                    // channels per K-tile
    constexpr int NT = 64 * WGM * WGN;
    constexpr int RPP = NT / 8;            // tile rows staged per pass (8 threads x float4 = one 32-float row)
    constexpr int PA = BM / RPP, PB = BN / RPP;
    constexpr int TM = BM / (32 * WGM), TN = BN / (32 * WGN);
    constexpr int TILE_A = BM * LDK, TILE_B = BN * LDK;
    static_assert(PA >= 1 && PB >= 1 && TM >= 1 && TN >= 1, "bad tile");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    // double-buffered unless the whole K range fits one step
    // LDS ring of NS pipeline steps (1 when the whole K range fits one step).  With LDS-DMA a deeper
    // ring costs no registers, so the copy of step t+2 is in flight while step t is multiplied.
    constexpr int NS = igemm_stages(BM, BN, G);
    const int nbuf = p.ktiles_per_split > G ? NS : 1;
    float *As = smem;                          // [nbuf][G][BM][LDK]
    float *Bs = smem + nbuf * G * TILE_A;      // [nbuf][G][BN][LDK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    ISTAMP_DECL;
    ISTAMP(0);
    // The argument block is read with scalar loads that miss the scalar cache once per launch and CU (~650 cycles each round trip).  Left alone,
    // the compiler sinks the loads next to their uses, i.e. into three dependent round trips spread over the prologue; naming the fields here
    // keeps all of them in the entry block -> one round trip (tools/time_conv.py stamps: -1300 cycles of ~4000 before the first fetch).
    asm volatile("" :: "s"(p.src0), "s"(p.src1), "s"(p.w), "s"(p.scale), "s"(p.shift), "s"(p.residual), "s"(p.out), "s"(p.partial));
    asm volatile("" :: "s"(p.B), "s"(p.Hs), "s"(p.Ws), "s"(p.Ho), "s"(p.Wo), "s"(p.C0), "s"(p.C1), "s"(p.Cin), "s"(p.Cout), "s"(p.stride),
                       "s"(p.up4), "s"(p.relu), "s"(p.M), "s"(p.Mout), "s"(p.ktiles_total), "s"(p.ktiles_per_split), "s"(p.splits));
    asm volatile("" :: "s"(p.div_rhw.m), "s"(p.div_rhw.s1), "s"(p.div_rhw.s2), "s"(p.div_rw.m), "s"(p.div_rw.s1), "s"(p.div_rw.s2),
                       "s"(p.div_gx.m), "s"(p.div_gx.s1), "s"(p.div_gx.s2), "s"(p.div_tile.m), "s"(p.div_tile.s1), "s"(p.div_tile.s2),
                       "s"(p.div_cin.m), "s"(p.div_cin.s1), "s"(p.div_cin.s2), "s"(p.ntm), "s"(p.ntn), "s"(p.gx), "s"(p.gy), "s"(p.psum), "s"(p.tile_cnt),
                       "s"(p.out_f32), "s"(p.xcd));

    // up4 (sub-pixel form of Upsample x2 + Conv3x3): blockIdx.x also enumerates the 4 output
    // parities (py, px); each is a 2x2-tap conv over the LOW-res source with pre-summed weights,
    // M counts low-res positions, and row m lands on output pixel (2y+py, 2x+px).
    // XCD-aware order: the dispatcher deals workgroups round-robin over the 8 XCDs (private L2s),
    // so logical tile ids are handed out in 8 contiguous chunks -- one XCD works on one band of
    // image rows (m-major: activations fetched once, weights by every XCD) or on one band of
    // output channels / K-splits (n-major, weight-heavy layers: weights fetched once).
    unsigned lin = blockIdx.x + blockIdx.y * (unsigned)p.gx;
    if (p.xcd) {
        const unsigned total = (unsigned)(p.gx * p.gy), q = total >> 3, r = total & 7, x = lin & 7;
        lin = x * q + (x < r ? x : r) + (lin >> 3);
    }
    const int z = (int)p.div_gx.div(lin);                // the scalar divisions of this prologue are multiplies by launch constants
    int bx = (int)(lin - (unsigned)(z * p.gx)), par = 0;
    if (p.up4) { par = bx & 3; bx >>= 2; }
    const int py = par >> 1, px = par & 1;
    int mt, nt;
    if (p.xcd == 2) { nt = (int)p.div_tile.div((unsigned)bx); mt = bx - nt * p.ntm; }
    else { mt = (int)p.div_tile.div((unsigned)bx); nt = bx - mt * p.ntn; }
    const int m0 = mt * BM, n0 = nt * BN;
    const int kt_begin = z * p.ktiles_per_split;
    int kt_end = kt_begin + p.ktiles_per_split;
    if (kt_end > p.ktiles_total) kt_end = p.ktiles_total;

    ISTAMP(8);
    const int lrow = tid >> 3;
    // staging: thread = (row lrow of each 8-row wave stripe, 16-B slot tid&7); it fetches the k-quad that
    // belongs in its slot
    const int lqb = ((tid & 7) ^ ((lrow >> 1) & 7)) * 16;   // byte offset of the k-slot this thread fetches

    // ---- per-thread im2col row descriptors (fixed for the whole K loop) ----
    // a_pix0: pixel index of tap (0,0) (may be "negative" at the border -- only used when the
    // tap's validity bit is set); a_mask: bit t = tap t reads inside the (virtually upsampled)
    // source; rows past M have mask 0.  All gathers are buffer loads whose voffset is forced
    // out of range for invalid taps, so padding costs no branch and no select: the hardware
    // returns zeros.
    const int tw = p.up4 ? 2 : 3;                       // taps per row
    const int ntap = tw * tw;
    const int hlim = UP ? 2 * p.Hs : p.Hs;
    const int wlim = UP ? 2 * p.Ws : p.Ws;
    const int rw = p.up4 ? p.Ws : p.Wo;                 // extent of the M index space
    const int rhw = p.up4 ? p.Hs * p.Ws : p.Ho * p.Wo;
    int a_pix0[PA], a_oy[PA], a_ox[PA];
    unsigned a_mask[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int m = m0 + i * RPP + lrow;
        a_mask[i] = 0; a_pix0[i] = 0; a_oy[i] = 0; a_ox[i] = 0;
        if (m < p.M) {
            // exact division by the launch-invariant extents via precomputed multipliers
            const int b = (int)p.div_rhw.div((unsigned)m);
            const int r = m - b * rhw;
            const int oy = (int)p.div_rw.div((unsigned)r);
            const int y0 = oy * p.stride - 1 + py;
            const int x0 = (r - oy * rw) * p.stride - 1 + px;
            a_oy[i] = y0; a_ox[i] = x0;
            a_pix0[i] = UP ? b * p.Hs * p.Ws : b * p.Hs * p.Ws + y0 * p.Ws + x0;
            unsigned mask = 0, bit = 0;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                if (ky >= tw) break;
                const bool oky = (unsigned)(y0 + ky) < (unsigned)hlim;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    if (kx >= tw) break;
                    mask |= (unsigned)(oky & ((unsigned)(x0 + kx) < (unsigned)wlim)) << bit;
                    ++bit;
                }
            }
            a_mask[i] = mask;
        }
    }
    ISTAMP(9);
    // byte offset of tap (0,0) of each staged row inside either source (modular arithmetic: a "negative" border pixel is only ever used in a
    // sum that is valid), so a K-tile's address is one add of a scalar tap offset instead of a multiply per piece
    unsigned a_b0[PA], a_b1[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        a_b0[i] = (unsigned)a_pix0[i] * ((unsigned)p.C0 * (unsigned)EB) + (unsigned)lqb;
        a_b1[i] = (unsigned)a_pix0[i] * ((unsigned)p.C1 * (unsigned)EB) + (unsigned)lqb;
    }
    const int K = KM ? p.ktiles_total * BKE : ntap * p.Cin;     // elements per weight row (masked K: only the live tiles are stored)
    const T *wbase = static_cast<const T *>(p.w) + (size_t)par * p.Cout * K;
    unsigned b_off[PB];                                  // byte offset of this thread's weight row, or OOB
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int n = n0 + i * RPP + lrow;
        b_off[i] = (n < p.Cout) ? (unsigned)(n * K) * (unsigned)EB + (unsigned)lqb : kOOB;
    }
    const unsigned plane = (unsigned)(p.B * p.Hs * p.Ws) * (unsigned)EB;
    const i32x4 rsw = make_srd(wbase, (unsigned)(p.Cout * K) * (unsigned)EB);

    // K-tile cursor of the NEXT tile to fetch: tap = ky*tw+kx, c = channel offset inside the
    // concatenated input
    int tap = (int)p.div_cin.div((unsigned)(kt_begin * BKE));
    int c = kt_begin * BKE - tap * p.Cin;
    // masked K (p.kmask: the space-to-depth form of a 4x4 / stride-2 conv): only the live (tap, channel quarter) pairs have K-tiles; the cursor steps over the others
    auto skip_dead = [&]() {
        while (tap < 9 && !((p.kmask >> (tap * 4 + c / p.kblk)) & 1ull)) {
            c += p.kblk;
            if (c >= p.Cin) { c = 0; ++tap; }
        }
    };
    if constexpr (KM) {
        tap = 0; c = 0;
        skip_dead();
        for (int i = 0; i < kt_begin; ++i) {           // (split-K slices: walk to the slice's first live tile)
            c += BKE;
            if (c == p.Cin) { c = 0; ++tap; }
            if (c % p.kblk == 0) skip_dead();
        }
    }
    int ky = p.up4 ? tap >> 1 : (tap * 11) >> 5, kx = tap - ky * tw;     // tap / tw for tap <= 8

    // Stage K-tiles kt .. kt+G-1 into LDS buffer `buf` with buffer_load ... lds (LDS-DMA): no staging
    // registers, no ds_write; invalid taps / ragged rows use an out-of-range voffset and land as zeros
    // (tools/probes/lds_dma_probe.hip verifies both properties on gfx950).  One instruction moves the
    // 8 rows x 128 B stripe of this wave: destination = wave-uniform base + lane * 16.
    typedef __attribute__((address_space(3))) float lds_float;
    const int wstripe = __builtin_amdgcn_readfirstlane((tid >> 6) * 8);   // first row of this wave's stripe in a pass
    const unsigned lds_a = (unsigned)(unsigned long long)(lds_float *)As;   // LDS byte addresses
    const unsigned lds_b = (unsigned)(unsigned long long)(lds_float *)Bs;
    auto fetch = [&](int kt, int buf) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const bool live = kt + g < kt_end;
            const bool first = c < p.C0;
            const void *sp = first ? p.src0 : p.src1;
            const int cs = first ? p.C0 : p.C1;
            const int soff = (first ? c : c - p.C0) * EB;
            const i32x4 rs = make_srd(sp, plane * (unsigned)cs);
            const int tapdelta = ky * p.Ws + kx;
            const unsigned csb = (unsigned)cs * (unsigned)EB;
            const unsigned A = lds_a + (unsigned)(((buf * G + g) * TILE_A + wstripe * LDK) * 4);
            const unsigned Bq = lds_b + (unsigned)(((buf * G + g) * TILE_B + wstripe * LDK) * 4);
            unsigned va[PA], vb[PB];
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                const bool ok = live && ((a_mask[i] >> tap) & 1u);
                if constexpr (UP) {
                    const int pix = a_pix0[i] + ((a_oy[i] + ky) >> 1) * p.Ws + ((a_ox[i] + kx) >> 1);
                    va[i] = ok ? (unsigned)pix * csb + (unsigned)lqb : kOOB;
                } else {
                    va[i] = ok ? (first ? a_b0[i] : a_b1[i]) + (unsigned)tapdelta * csb : kOOB;
                }
            }
#pragma unroll
            for (int i = 0; i < PB; ++i) vb[i] = live ? b_off[i] : kOOB;
            // one statement per operand: M0 saved / restored once per group (same session, grouped vs one statement per piece: fp32 batch 8
            // +1.3 %, bf16 `large` batch 8 +1.0 %, fp32 batch 1 +0.4 %)
            dma16_group<PA, RPP * LDK * 4>(A, va, rs, soff);
            dma16_group<PB, RPP * LDK * 4>(Bq, vb, rsw, (kt + g) * (BK * 4));
            if (live) {
                c += BKE;
                if (c == p.Cin) {
                    c = 0; ++tap; ++kx;
                    if (kx == tw) { kx = 0; ++ky; }
                }
                if (KM && c % p.kblk == 0) {
                    skip_dead();
                    ky = (tap * 11) >> 5; kx = tap - ky * 3;
                }
            }
        }
    };

    // Epilogue operands fetched before the K loop: the folded-BN scale / shift of this lane's channel quad and (tiles of <= 2 MFMA blocks per wave)
    // the residual rows.  Issued in the epilogue they are dependent global-load round trips with the MFMA pipe already idle (~2 us of a ~47-us
    // launch); issued here they ride under the first fetch.  They are OLDER than every LDS-DMA piece, and loads return in order, so the counted
    // vmcnt waits of the K loop are unaffected.
    const int erow = lane >> 3, ecol = (lane & 7) * 4;      // epilogue role of this lane: row within an 8-row pass, first channel of its quad
    constexpr bool PRE = TM * TN <= 2;
    const bool pre_res = PRE && p.residual != nullptr && p.splits == 1 && !p.up4;
    float4 scv[TN], shv[TN];
    float4 rpre[PRE ? TM : 1][PRE ? TN : 1][4];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + ecol;
        scv[j] = make_float4(1.f, 1.f, 1.f, 1.f); shv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.splits == 1 && p.scale && n < p.Cout) {
            scv[j] = *reinterpret_cast<const float4 *>(p.scale + n);
            shv[j] = *reinterpret_cast<const float4 *>(p.shift + n);
        }
        if constexpr (PRE) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    const int m = m0 + (wm * TM + i) * 32 + pass * 8 + erow;
                    rpre[i][j][pass] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (pre_res && m < p.M && n < p.Cout)
                        rpre[i][j][pass] = load4(static_cast<const T *>(p.residual) + (size_t)m * p.Cout + n);
                }
        }
    }
    ISTAMP(10);
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addressing: lane l supplies row (l&31), k-quad (l>>5) of each 8-wide k group.
    // The MFMA's k index (l>>5) then pairs k and k+4 -- any K permutation is fine as long as
    // A and B use the same one.
    const int frow = lane & 31;
    const int fsw = (lane >> 5) ^ ((frow >> 1) & 7);   // (k-quad low bit) ^ row swizzle; tile offsets are multiples of 32 rows
    // One "fragment step" = 8 k of one K-tile: TM + TN ds_read_b128, then TM*TN*4 MFMAs.  Fragment
    // registers are double-buffered so the LDS reads of step s+1 are issued BEFORE the MFMAs of step s
    // (an in-order wave otherwise exposes the full LDS latency once per step).
    constexpr int S = G * (BK / 8);            // fragment steps per pipeline step
    float4 fa[2][TM], fb[2][TN];
    auto read_frag = [&](int buf, int s, int set) {
        const int g = s / (BK / 8), kb = s % (BK / 8);
        const int qoff = ((kb * 2) ^ fsw) * 4;         // swizzled slot of k-quad kb*2 + (lane>>5)
        const float *A = As + (buf * G + g) * TILE_A + (wm * TM * 32 + frow) * LDK + qoff;
        const float *Bq = Bs + (buf * G + g) * TILE_B + (wn * TN * 32 + frow) * LDK + qoff;
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[set][i] = *reinterpret_cast<const float4 *>(A + i * 32 * LDK);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[set][j] = *reinterpret_cast<const float4 *>(Bq + j * 32 * LDK);
    };
    auto mfma_frag = [&](int set) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (EB == 4) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][i].x, fb[set][j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][i].y, fb[set][j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][i].z, fb[set][j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][i].w, fb[set][j].w, acc[i][j], 0, 0, 0);
                } else if constexpr (__is_same(T, f16_t)) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[set][i]), __builtin_bit_cast(f16x8, fb[set][j]),
                                                                       acc[i][j], 0, 0, 0);
                } else {
                    // the 16-B slot is 8 consecutive bf16 channels = this lane's K = 8*(lane>>5) .. +7 operand
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[set][i]),
                                                                        __builtin_bit_cast(bf16x8, fb[set][j]),
                                                                        acc[i][j], 0, 0, 0);
                }
            }
    };

    // ---- main loop.  Per pipeline step t (G K-tiles in LDS ring slot `cur`):
    //   issue the LDS-DMA of step t+NS-1 into the slot that step t-1 just released (no registers, no ds_write)
    //   | fragment steps 0..S-1: ds_read of fragment s+1 issued before the MFMAs of fragment s
    //   | counted vmcnt wait: step t+1 has landed | barrier (publishes it, releases slot `cur`)
    //   | first fragment read of step t+1 (its latency hides behind the next DMA issue).
    if (kt_begin < kt_end && !ABL(p, 32)) {
        constexpr int PIECES = G * (PA + PB);      // DMA instructions this wave issues per step
        const int nsteps = (kt_end - kt_begin + G - 1) / G;
        ISTAMP(1);
        fetch(kt_begin, 0);
        ISTAMP(2);
        if (NS > 2 && nsteps > 1) fetch(kt_begin + G, 1);
        if (NS > 2 && nsteps > 1) dma_wait<PIECES>(); else dma_wait<0>();
        __syncthreads();
        ISTAMP(3);
        int cur = 0;
        read_frag(0, 0, 0);
        for (int t = 0; t < nsteps; ++t) {

            // ring slot (cur + NS-1) % NS was last read in step t-1; every wave has passed that barrier
            const int ahead = t + NS - 1;
            const bool issue = ahead < nsteps && !ABL(p, 1);
            int slot = cur + NS - 1; if (slot >= NS) slot -= NS;
            if (issue) fetch(kt_begin + ahead * G, slot);
#pragma unroll
            for (int s = 0; s < S - 1; ++s) {
                read_frag(cur, s + 1, (s + 1) & 1);
                mfma_frag(s & 1);
            }
            mfma_frag((S - 1) & 1);
            // step t+1 must have landed: everything but the pieces issued in THIS iteration
            if (NS > 2 && issue) dma_wait<PIECES>(); else dma_wait<0>();
            if (!ABL(p, 4)) __syncthreads();
            if (!ABL(p, 8)) { if (++cur == NS) cur = 0; }
            if (t + 1 < nsteps) read_frag(cur, 0, 0);
        }
    }

    // ---- epilogue.  C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5),
    // i.e. a lane owns 16 rows of ONE column -- storing that directly means 4-byte accesses.  Each
    // wave instead transposes its 32x32 tile through a private 4.5-KB LDS patch (the K-loop buffers
    // are free after the last barrier) so every lane ends up with 4 consecutive channels of a row:
    // float4 residual loads / stores, 8 lanes per 128-B row segment, 4 instead of 16 memory
    // instructions per tile.  Same-wave LDS traffic needs no barrier (a wave's DS ops execute in order).
    ISTAMP(6);
    if (ABL(p, 16)) return;
    const __amdgpu_buffer_rsrc_t slab_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.partial, 0, p.tile_cnt ? (int)p.slab_bytes : 0, 0x00020000);
    constexpr int EP = 36;                                   // patch row pitch (floats), keeps rows 16-B aligned
    float *patch = smem + wave * (32 * EP);
    const int ccol = lane & 31, crow = 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nb = n0 + (wn * TN + j) * 32;              // first channel of the tile
        const int n = nb + ecol;
        const bool nok = n < p.Cout;                         // Cout % 4 == 0: a float4 is all-in or all-out
        const float4 sc = scv[j], sh = shv[j];
        // InstanceNorm plans: sums of (x - c) and (x - c)^2 over this lane's rows, c = the value in the wave's first row (a shift
        // close to the mean keeps var = E[d^2] - E[d]^2 free of cancellation; in_finalize merges the groups in double)
        float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = make_float4(0.f, 0.f, 0.f, 0.f), c4 = s1;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                patch[((r & 3) + 8 * (r >> 2) + crow) * EP + ccol] = acc[i][j][r];
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int row = pass * 8 + erow;
                float4 v = *reinterpret_cast<const float4 *>(patch + row * EP + ecol);
                const int m = m0 + (wm * TM + i) * 32 + row;
                const bool ok = m < p.M && nok;
                size_t orow = (size_t)m;                     // output pixel index (NHWC row)
                if (p.up4 && ok) {
                    const int b = (int)p.div_rhw.div((unsigned)m);
                    const int rr = m - b * rhw;
                    const int y = (int)p.div_rw.div((unsigned)rr), x = rr - y * rw;
                    orow = ((size_t)b * p.Ho + 2 * y + py) * p.Wo + 2 * x + px;
                }
                if (p.splits > 1) {
                    if (ok) {
                        const size_t e = ((size_t)z * p.Mout + orow) * p.Cout + n;
                        // fused combine: the slab is published to whichever workgroup arrives last at this tile, possibly on another XCD
                        // (private L2s) -> write-through (sc1) stores, no release fence needed (guide: Guideline 16, R1)
                        if (p.tile_cnt) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), slab_rsrc, (unsigned)(e * 4), 0, 16);
                        else *reinterpret_cast<float4 *>(p.partial + e) = v;
                    }
                } else {
                    v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                    if (pre_res) {
                        if constexpr (PRE) { const float4 rv = rpre[i][j][pass]; v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w; }
                    } else if (p.residual && ok) {
                        const float4 rv = load4(static_cast<const T *>(p.residual) + orow * p.Cout + n);
                        v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                    }
                    if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    if (ok) {
                        if (KM && p.s2d_out) unet_dual_store<T>(p, (unsigned)orow, n, v);       // (masked-K instances only: the `small` U-Net's down-convs)
                        else if (p.out_f32) store4(static_cast<float *>(p.out) + orow * p.Cout + n, v);
                        else store4(static_cast<T *>(p.out) + orow * p.Cout + n, v);
                    }
                    if (p.psum) {                            // wave-uniform
                        if (i == 0 && pass == 0) {           // row 0 of the wave's rows sits in the lanes with erow == 0
                            c4.x = __shfl(v.x, lane & 7); c4.y = __shfl(v.y, lane & 7);
                            c4.z = __shfl(v.z, lane & 7); c4.w = __shfl(v.w, lane & 7);
                        }
                        if (ok) {
                            const float4 d = make_float4(v.x - c4.x, v.y - c4.y, v.z - c4.z, v.w - c4.w);
                            s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
                            s2.x += d.x * d.x; s2.y += d.y * d.y; s2.z += d.z * d.z; s2.w += d.w * d.w;
                        }
                    }
                }
            }
        }
        if (p.psum) {
            // the 8 lanes that share a channel quad hold 4 rows each per tile -> xor-shuffle over lane bits 3..5
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) {
                s1.x += __shfl_xor(s1.x, o); s1.y += __shfl_xor(s1.y, o); s1.z += __shfl_xor(s1.z, o); s1.w += __shfl_xor(s1.w, o);
                s2.x += __shfl_xor(s2.x, o); s2.y += __shfl_xor(s2.y, o); s2.z += __shfl_xor(s2.z, o); s2.w += __shfl_xor(s2.w, o);
            }
            const int mw = m0 + wm * TM * 32;               // all TM*32 rows lie inside one frame (the planner checks divisibility)
            if (erow == 0 && nok && mw < p.M) {
                const int b = (int)p.div_rhw.div((unsigned)mw);
                const int gpp = rhw / (TM * 32);                                   // groups per frame and parity
                const size_t g = ((size_t)b * p.in_groups + par * gpp + (mw - b * rhw) / (TM * 32)) * p.Cout + n;
                *reinterpret_cast<float4 *>(p.psum + g) = s1;
                *reinterpret_cast<float4 *>(p.psq + g) = s2;
                *reinterpret_cast<float4 *>(p.pshift + g) = c4;
            }
        }
    }

    // ---- fused split-K combine (2 <= splits <= 8): the workgroup that arrives LAST at a tile sums the slabs in z order and runs the epilogue,
    // instead of a separate reduce launch.  Protocol of the guide (Guideline 16, counter form): write-through slab stores above -> every wave
    // drains its stores -> barrier -> one relaxed agent-scope ticket; the last arriver reads the slabs with sc1 loads (past its L1 / any stale
    // line), in a fixed order -> bit-reproducible, independent of which slice arrives last.  The counter is reset by the last arriver (and zeroed
    // once per workspace binding by the host).
    ISTAMP(7);
    ISTAMP_FLUSH;
    if (p.tile_cnt == nullptr || p.splits == 1) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                          // also: every wave is done with its epilogue patch (smem reused below)
    unsigned *flag = reinterpret_cast<unsigned *>(smem);
    const unsigned tile = (unsigned)(lin - (unsigned)(z * p.gx));          // logical (parity, M-tile, N-tile) id, the same for every z
    if (tid == 0) flag[0] = __hip_atomic_fetch_add(p.tile_cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (flag[0] != (unsigned)p.splits - 1u) return;
    if (tid == 0) __hip_atomic_store(p.tile_cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // tile = BM x BN fp32: thread -> (row r0 + 16 i... ) 16 float4 columns x NT / 16 rows per pass
    constexpr int C4 = BN / 4, RPP2 = NT / C4;               // float4 columns, rows per pass
    const int c4i = tid % C4, rr = tid / C4;
    const int ncol = n0 + c4i * 4;
    if (ncol >= p.Cout) return;
    float4 sc2 = make_float4(1.f, 1.f, 1.f, 1.f), sh2 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.scale) {
        sc2 = *reinterpret_cast<const float4 *>(p.scale + ncol);
        sh2 = *reinterpret_cast<const float4 *>(p.shift + ncol);
    }
#pragma unroll 2
    for (int r = rr; r < BM; r += RPP2) {
        const int m = m0 + r;
        if (m >= p.M) break;
        size_t orow = (size_t)m;
        if (p.up4) {
            const int b = (int)p.div_rhw.div((unsigned)m);
            const int q = m - b * rhw;
            const int y = (int)p.div_rw.div((unsigned)q), x = q - y * rw;
            orow = ((size_t)b * p.Ho + 2 * y + py) * p.Wo + 2 * x + px;
        }
        const size_t e = orow * p.Cout + ncol;
        float4 t[8];
#pragma unroll
        for (int zz = 0; zz < 8; ++zz)
            if (zz < p.splits)
                t[zz] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(slab_rsrc, (unsigned)(((size_t)zz * p.Mout * p.Cout + e) * 4), 0, 16));
        float4 v = t[0];
#pragma unroll
        for (int zz = 1; zz < 8; ++zz)
            if (zz < p.splits) { v.x += t[zz].x; v.y += t[zz].y; v.z += t[zz].z; v.w += t[zz].w; }
        v.x = v.x * sc2.x + sh2.x; v.y = v.y * sc2.y + sh2.y; v.z = v.z * sc2.z + sh2.z; v.w = v.w * sc2.w + sh2.w;
        if (p.residual) {
            const float4 rv = load4(static_cast<const T *>(p.residual) + e);
            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
        }
        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (KM && p.s2d_out) unet_dual_store<T>(p, (unsigned)orow, ncol, v);
        else store4(static_cast<T *>(p.out) + e, v);
    }
}

// out = epilogue(sum_z partial[z]) -- shared tail of both reduce kernels
template <typename T, bool DUAL = false>      // DUAL: unet_dual_store instead of `out` (its own instances: the reduce launches of the other plans carry none of it)
__device__ __forceinline__ void reduce_epilogue(const IgemmParams &p, unsigned i, float4 s)
{
    const unsigned n = (i * 4u) % (unsigned)p.Cout;
    if (p.scale) {
        const float4 sc = *reinterpret_cast<const float4 *>(p.scale + n);
        const float4 sh = *reinterpret_cast<const float4 *>(p.shift + n);
        s.x = s.x * sc.x + sh.x; s.y = s.y * sc.y + sh.y;
        s.z = s.z * sc.z + sh.z; s.w = s.w * sc.w + sh.w;
    }
    if (p.residual) {
        const float4 r = load4(static_cast<const T *>(p.residual) + (size_t)i * 4);
        s.x += r.x; s.y += r.y; s.z += r.z; s.w += r.w;
    }
    if constexpr (DUAL) {
        unet_dual_store<T>(p, (i * 4u) / (unsigned)p.Cout, (int)n, s);
        return;
    }
    if (p.relu) { s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f); }
    store4(static_cast<T *>(p.out) + (size_t)i * 4, s);
}

// Many splits (the <= 16x16 levels): block = 64 float4 columns x 4 z-lanes: z-lane y adds partials
// y, y+4, y+8, ... (ascending), then the four lane sums are added in lane order -- a fixed summation
// tree, so results are bit-reproducible run to run.
template <typename T, bool DUAL = false>
__global__ __launch_bounds__(256) void splitk_reduce(const IgemmParams p)
{
    __shared__ float4 red[3][64];
    const unsigned total4 = (unsigned)(((size_t)p.Mout * p.Cout) >> 2);
    const unsigned x = threadIdx.x & 63, y = threadIdx.x >> 6;
    const unsigned i = blockIdx.x * 64u + x;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < total4) {
        const float4 *pp = reinterpret_cast<const float4 *>(p.partial) + i;
        int zz = (int)y;
        for (; zz + 12 < p.splits; zz += 16) {
            const float4 t0 = pp[(size_t)zz * total4], t1 = pp[(size_t)(zz + 4) * total4];
            const float4 t2 = pp[(size_t)(zz + 8) * total4], t3 = pp[(size_t)(zz + 12) * total4];
            s.x += t0.x; s.y += t0.y; s.z += t0.z; s.w += t0.w;
            s.x += t1.x; s.y += t1.y; s.z += t1.z; s.w += t1.w;
            s.x += t2.x; s.y += t2.y; s.z += t2.z; s.w += t2.w;
            s.x += t3.x; s.y += t3.y; s.z += t3.z; s.w += t3.w;
        }
        for (; zz < p.splits; zz += 4) {
            const float4 t = pp[(size_t)zz * total4];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
    }
    if (y) red[y - 1][x] = s;
    __syncthreads();
    if (y || i >= total4) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float4 t = red[k][x];
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    reduce_epilogue<T, DUAL>(p, i, s);
}

// Few splits (2..4: the mid levels, megabytes of partials): one float4 per thread, all partial loads in
// flight at once, no LDS, no barrier -- a pure streaming pass.  z ascending.
template <typename T, int NS, bool DUAL = false>
__global__ __launch_bounds__(256) void splitk_reduce_few(const IgemmParams p)
{
    const unsigned total4 = (unsigned)(((size_t)p.Mout * p.Cout) >> 2);
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= total4) return;
    const float4 *pp = reinterpret_cast<const float4 *>(p.partial) + i;
    float4 t[NS];
#pragma unroll
    for (int z = 0; z < NS; ++z) t[z] = pp[(size_t)z * total4];
    float4 s = t[0];
#pragma unroll
    for (int z = 1; z < NS; ++z) { s.x += t[z].x; s.y += t[z].y; s.z += t[z].z; s.w += t[z].w; }
    reduce_epilogue<T, DUAL>(p, i, s);
}

template <typename T, int BM, int BN, int WGM, int WGN, int G, bool UP, bool KM = false>
static hipError_t launch_igemm_t(const IgemmParams &p, hipStream_t s)
{
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.Cout + BN - 1) / BN;
    const int npar = p.up4 ? 4 : 1;
    constexpr size_t smem_one = (size_t)G * (BM + BN) * LDK * sizeof(float);
    constexpr size_t smem_max = smem_one * igemm_stages(BM, BN, G);
    constexpr size_t smem_patch = (size_t)WGM * WGN * 32 * 36 * sizeof(float);   // epilogue transpose patches
    size_t smem = p.ktiles_per_split > G ? smem_max : smem_one;
    if (smem < smem_patch) smem = smem_patch;
    static AttrMask attr_mask;   // raise the dynamic-LDS cap once per instantiation and device
    if (smem_max > 64 * 1024 && attr_needed_on_this_device(attr_mask)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&igemm3x3<T, BM, BN, WGM, WGN, G, UP, KM>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_max);
        if (e != hipSuccess) return e;
        attr_done_on_this_device(attr_mask);
    }
    IgemmParams q = p;
    q.ntm = ntm; q.ntn = ntn; q.gx = ntm * ntn * npar; q.gy = p.splits;
    q.div_gx = FastDiv::make((unsigned)(ntm * ntn * npar));
    q.div_tile = FastDiv::make((unsigned)(p.xcd == 2 ? ntm : ntn));
    q.div_cin = FastDiv::make((unsigned)p.Cin);
    hipLaunchKernelGGL((igemm3x3<T, BM, BN, WGM, WGN, G, UP, KM>), dim3(ntm * ntn * npar, p.splits), dim3(64 * WGM * WGN),
                       smem, s, q);
    return hipGetLastError();
}

bool igemm_tile_supported(int bm, int bn)
{
    for (int i = 0; i < kNumTileConfigs; ++i)
        if (kTileConfigs[i].bm == bm && kTileConfigs[i].bn == bn) return true;
    return false;
}

bool igemm_group_supported(int bm, int bn, int g, bool up)
{
    if (up) return (bm == 64 && bn == 64 && (g == 1 || g == 4)) || (bm == 32 && bn == 64 && g == 4);
    if (g == 1) return igemm_tile_supported(bm, bn);
    if (g == 2) return (bm == 128 && bn == 64) || (bm == 64 && bn == 64);
    if (g == 4) return (bm == 64 && bn == 64) || (bm == 32 && bn == 64);
    return false;
}

// masked K (the space-to-depth form of a 4x4 / stride-2 conv, lspf2f_conv3x3 with k_group -4; the `small` U-Net's down-convs): no upsample; its own instances of the kernel,
// fp32 and -- for the opt.fp16 plan of that variant (include/lspunet.h) -- fp16 storage
template <typename T>
static hipError_t launch_igemm_masked_t(const IgemmParams &p, int bm, int bn, int g, hipStream_t s)
{
    if (g == 4) {
        if (bm == 64 && bn == 64) return launch_igemm_t<T, 64, 64, 2, 2, 4, false, true>(p, s);
        if (bm == 32 && bn == 64) return launch_igemm_t<T, 32, 64, 1, 2, 4, false, true>(p, s);
        return hipErrorInvalidValue;
    }
    if (g == 2) {
        if (bm == 128 && bn == 64) return launch_igemm_t<T, 128, 64, 2, 2, 2, false, true>(p, s);
        if (bm == 64 && bn == 64) return launch_igemm_t<T, 64, 64, 2, 2, 2, false, true>(p, s);
        return hipErrorInvalidValue;
    }
    if (bm == 128 && bn == 128) return launch_igemm_t<T, 128, 128, 2, 2, 1, false, true>(p, s);
    if (bm == 128 && bn == 64) return launch_igemm_t<T, 128, 64, 2, 2, 1, false, true>(p, s);
    if (bm == 64 && bn == 128) return launch_igemm_t<T, 64, 128, 2, 2, 1, false, true>(p, s);
    if (bm == 64 && bn == 64) return launch_igemm_t<T, 64, 64, 2, 2, 1, false, true>(p, s);
    if (bm == 32 && bn == 128) return launch_igemm_t<T, 32, 128, 1, 4, 1, false, true>(p, s);
    if (bm == 32 && bn == 64) return launch_igemm_t<T, 32, 64, 1, 2, 1, false, true>(p, s);
    return hipErrorInvalidValue;
}
static hipError_t launch_igemm_masked(const IgemmParams &p, int bm, int bn, int g, hipStream_t s)
{
    if ((p.dtype != 0 && p.dtype != 2) || p.up || p.up4) return hipErrorInvalidValue;
    if (p.s2d_out && (!p.relu_out || p.residual || p.relu || (p.Ho & 1) || (p.Wo & 1) || p.psum)) return hipErrorInvalidValue;   // dual store: see unet_dual_store
    return p.dtype == 2 ? launch_igemm_masked_t<f16_t>(p, bm, bn, g, s) : launch_igemm_masked_t<float>(p, bm, bn, g, s);
}

template <typename T>
static hipError_t launch_igemm_typed(const IgemmParams &p, int bm, int bn, int g, hipStream_t s)
{
    if (p.up) {
        if (bm == 64 && bn == 64 && g == 1) return launch_igemm_t<T, 64, 64, 2, 2, 1, true>(p, s);
        if (bm == 64 && bn == 64 && g == 4) return launch_igemm_t<T, 64, 64, 2, 2, 4, true>(p, s);
        if (bm == 32 && bn == 64 && g == 4) return launch_igemm_t<T, 32, 64, 1, 2, 4, true>(p, s);
        return hipErrorInvalidValue;
    }
    if (g == 4) {
        if (bm == 64 && bn == 64) return launch_igemm_t<T, 64, 64, 2, 2, 4, false>(p, s);
        if (bm == 32 && bn == 64) return launch_igemm_t<T, 32, 64, 1, 2, 4, false>(p, s);
        return hipErrorInvalidValue;
    }
    if (g == 2) {
        if (bm == 128 && bn == 64) return launch_igemm_t<T, 128, 64, 2, 2, 2, false>(p, s);
        if (bm == 64 && bn == 64) return launch_igemm_t<T, 64, 64, 2, 2, 2, false>(p, s);
        return hipErrorInvalidValue;
    }
    if (bm == 128 && bn == 32) return launch_igemm_t<T, 128, 32, 4, 1, 1, false>(p, s);     // narrow N: the last conv as a GEMM
    if (bm == 128 && bn == 128) return launch_igemm_t<T, 128, 128, 2, 2, 1, false>(p, s);
    if (bm == 128 && bn == 64) return launch_igemm_t<T, 128, 64, 2, 2, 1, false>(p, s);
    if (bm == 64 && bn == 128) return launch_igemm_t<T, 64, 128, 2, 2, 1, false>(p, s);
    if (bm == 64 && bn == 64) return launch_igemm_t<T, 64, 64, 2, 2, 1, false>(p, s);
    if (bm == 32 && bn == 128) return launch_igemm_t<T, 32, 128, 1, 4, 1, false>(p, s);
    if (bm == 32 && bn == 64) return launch_igemm_t<T, 32, 64, 1, 2, 1, false>(p, s);
    return hipErrorInvalidValue;
}

hipError_t launch_igemm(const IgemmParams &p_in, int bm, int bn, int g, hipStream_t s)
{
    IgemmParams p = p_in;
    p.div_rhw = FastDiv::make((unsigned)(p.up4 ? p.Hs * p.Ws : p.Ho * p.Wo));
    p.div_rw = FastDiv::make((unsigned)(p.up4 ? p.Ws : p.Wo));
    // 2 GiB per tensor: buffer-load offsets are 32-bit with the top bit reserved as the OOB marker
    const size_t lim = 0x7fffffffull;
    const size_t eb = p.dtype ? 2 : 4;
    if ((size_t)p.B * p.Hs * p.Ws * (size_t)(p.C0 > p.C1 ? p.C0 : p.C1) * eb > lim) return hipErrorInvalidValue;
    if ((p.C0 * eb) % 128 || (p.C1 * eb) % 128) return hipErrorInvalidValue;   // a K-tile is 128 B of channels
    {   // XCD chunking: partition the larger operand across the 8 L2s (xcd_force = 1 + mode overrides: `igemm_xcd` of lspf2f_create_tuned, tools only)
        const int forced = p.xcd_force - 1;
        const size_t act = (size_t)p.B * p.Hs * p.Ws * p.Cin * eb;
        const size_t wgt = (size_t)(p.up4 ? 16 : 9) * p.Cin * p.Cout * eb;
        int rule = wgt > act ? 2 : 1;
        // Measured exception, keyed on dtype like choose_tiling(): activation-heavy split-K layers keep dispatch order in
        // fp32 plans (m-major chunks cost them 4 %: fp32 batch 1 +0.8 % with the exception) but not in bf16 plans
        // (batch 8 -1.1 / -1.4 % with it); DESIGN.md 4.1.
        if (rule == 1 && p.splits > 1 && p.dtype == 0) rule = 0;
        p.xcd = forced >= 0 ? forced : rule;
    }
    if (p.s2d_out && !p.kmask) return hipErrorInvalidValue;      // the dual store lives in the masked-K instances only
    if (p.kmask) return launch_igemm_masked(p, bm, bn, g, s);
    if (p.dtype == 2) return launch_igemm_typed<f16_t>(p, bm, bn, g, s);
    return p.dtype == 1 ? launch_igemm_typed<bf16_t>(p, bm, bn, g, s) : launch_igemm_typed<float>(p, bm, bn, g, s);
}

template <typename T, bool DUAL = false>
static hipError_t launch_splitk_reduce_t(const IgemmParams &p, hipStream_t s)
{
    const size_t total4 = (size_t)p.Mout * p.Cout / 4;
    const dim3 few((unsigned)((total4 + 255) / 256));
    switch (p.splits) {
    case 2: hipLaunchKernelGGL((splitk_reduce_few<T, 2, DUAL>), few, dim3(256), 0, s, p); break;
    case 3: hipLaunchKernelGGL((splitk_reduce_few<T, 3, DUAL>), few, dim3(256), 0, s, p); break;
    case 4: hipLaunchKernelGGL((splitk_reduce_few<T, 4, DUAL>), few, dim3(256), 0, s, p); break;
    default: hipLaunchKernelGGL((splitk_reduce<T, DUAL>), dim3((unsigned)((total4 + 63) / 64)), dim3(256), 0, s, p); break;
    }
    return hipGetLastError();
}

hipError_t launch_splitk_reduce(const IgemmParams &p_in, hipStream_t s)
{
    if (p_in.s2d_out) {                       // the `small` U-Net's split-K down-convs (fp32 or fp16 storage): the two activated copies instead of `out`
        if ((p_in.dtype != 0 && p_in.dtype != 2) || !p_in.relu_out || (p_in.Ho & 1) || (p_in.Wo & 1) || p_in.up4) return hipErrorInvalidValue;
        IgemmParams p = p_in;
        p.div_rhw = FastDiv::make((unsigned)(p.Ho * p.Wo));
        p.div_rw = FastDiv::make((unsigned)p.Wo);
        return p.dtype == 2 ? launch_splitk_reduce_t<f16_t, true>(p, s) : launch_splitk_reduce_t<float, true>(p, s);
    }
    const IgemmParams &p = p_in;
    if (p.dtype == 2) return launch_splitk_reduce_t<f16_t>(p, s);
    return p.dtype == 1 ? launch_splitk_reduce_t<bf16_t>(p, s) : launch_splitk_reduce_t<float>(p, s);
}

}  // namespace lspf2f
