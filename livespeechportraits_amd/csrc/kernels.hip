// gfx950 (MI355X / CDNA4) kernels of the feature2face generator.  Written for wave64 + MFMA;
// no other target is supported.
//
//  igemm3x3_f32      fused 3x3 conv as implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32):
//                    im2col gather (stride 1|2, optional nearest-x2 upsample, optional second
//                    source = the never-materialised torch.cat) -> LDS -> MFMA -> epilogue
//                    (folded BatchNorm scale/shift, residual add, ReLU) or split-K partials.
//  splitk_reduce     deterministic z-ordered reduction of split-K partials + the same epilogue.
//  first_conv        13->ngf stride-2 conv reading the two NCHW API tensors, ReLU, NHWC out.
//  last_conv         2*ngf->3 conv with fused upsample + tanh, NHWC in, NCHW out.
//
// Reference semantics: models/networks.py:592-640 (level layout), :650-675 (ResidualBlock),
// :575-579 (tanh); BatchNorm eval folding validated in SURVEY.md 8c.
#include "kernels.h"

#include <cstdlib>

#ifndef LSPF2F_SWP
#define LSPF2F_SWP 1   // 1: LDS fragment reads issued one step ahead of the MFMAs
#endif

namespace lspf2f {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short bf16_t;               // bf16 storage (round-to-nearest-even on store)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ bf16_t f2bf(float f)
{
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
// 4 consecutive channels of an activation row: load / store as float4 regardless of the storage type
template <typename T> __device__ __forceinline__ float4 load4(const T *p);
template <> __device__ __forceinline__ float4 load4<float>(const float *p) { return *reinterpret_cast<const float4 *>(p); }
template <> __device__ __forceinline__ float4 load4<bf16_t>(const bf16_t *p)
{
    const uint2 u = *reinterpret_cast<const uint2 *>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                       __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
template <typename T> __device__ __forceinline__ void store4(T *p, float4 v);
template <> __device__ __forceinline__ void store4<float>(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t *p, float4 v)
{
    uint2 u;
    u.x = (unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16);
    u.y = (unsigned)f2bf(v.z) | ((unsigned)f2bf(v.w) << 16);
    *reinterpret_cast<uint2 *>(p) = u;
}

// LDS ring depth of the igemm pipeline (2; 3 if built with -DLSPF2F_STAGES=3 and it leaves >= 2 workgroups per CU)
#ifndef LSPF2F_STAGES
#define LSPF2F_STAGES 2   // measured: ring depth 3 is ~1 % slower than 2 on MI355X (DMA latency is not the limiter)
#endif
__host__ __device__ constexpr int igemm_stages(int bm, int bn, int g)
{
    return (LSPF2F_STAGES >= 3 && 3 * g * (bm + bn) * 128 <= 80 * 1024) ? 3 : 2;
}
static constexpr unsigned kOOB = 0x80000000u;   // voffset beyond any num_records: buffer load returns 0

static constexpr int BK = 32;    // K-tile (floats); Cin % 32 == 0 so a K-tile never straddles a tap
static constexpr int LDK = 32;   // LDS row pitch in floats (128 B, unpadded: the tile is written by LDS-DMA,
                                 // whose destination is lane-linear).  Bank conflicts are avoided by an XOR
                                 // swizzle instead: 16-B slot s of row r holds k-quad s ^ ((r >> 1) & 7), which
                                 // puts the 16 rows of every ds_read_b128 lane group on 16 distinct bank slots.

// G = K-tiles staged per pipeline step (one barrier per G tiles, G tiles of global loads in
// flight per thread).  G = 1 for long K loops; G = 4 turns a short split-K range (<= 4 tiles)
// into a single load -> LDS -> MFMA pass, which is what the latency-bound <= 8x8 levels need.
// UP = the 9-tap nearest-x2 gather form (only the small, weight-streaming up-convs use it).
// buffer resource descriptor (raw buffer, stride 0) from wave-uniform values
__device__ __forceinline__ i32x4 make_srd(const void *base, unsigned bytes)
{
    const unsigned long long a = (unsigned long long)base;
    i32x4 r;
    r.x = (int)(unsigned)a;
    r.y = (int)((unsigned)(a >> 32) & 0xffffu);
    r.z = (int)bytes;
    r.w = 0x00020000;
    return r;
}

// One LDS-DMA piece: 64 lanes x 16 B -> LDS[lds_addr + lane*16]; out-of-range voffset lands as zeros.
// Issued from inline asm on purpose: a compiler-visible LDS-DMA makes hipcc wait vmcnt(0) before the
// next ds_read (it cannot disambiguate LDS addresses), which serialises the copy with the MFMAs.  The
// kernel waits for its DMAs itself (dma_wait) right before the barrier that publishes the buffer.
// M0 is compiler-reserved: saved and restored inside the statement.
__device__ __forceinline__ void dma16(unsigned lds_addr, unsigned voff, i32x4 srd, int soff)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %1\n\t"
                 "s_nop 4\n\t"
                 "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_addr), "v"(voff), "s"(srd), "s"(soff)
                 : "memory");
}
// wait until at most N of this wave's DMA pieces are still in flight (they complete in order)
template <int N>
__device__ __forceinline__ void dma_wait()
{
    __builtin_amdgcn_sched_barrier(0);   // keep the MFMAs issued so far ABOVE the wait (they cover the copy)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// T = storage type of activations and weights: float (exact fp32 MFMA, v_mfma_f32_32x32x2_f32) or bf16_t
// (v_mfma_f32_32x32x16_bf16, fp32 accumulate and epilogue).  A K-tile is always 128 B of channels (32 fp32 /
// 64 bf16), so the LDS geometry, the DMA pieces and the swizzle are identical for both.
template <typename T, int BM, int BN, int WGM, int WGN, int G, bool UP>
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

    // up4 (sub-pixel form of Upsample x2 + Conv3x3): blockIdx.x also enumerates the 4 output
    // parities (py, px); each is a 2x2-tap conv over the LOW-res source with pre-summed weights,
    // M counts low-res positions, and row m lands on output pixel (2y+py, 2x+px).
    const int ntn = (p.Cout + BN - 1) / BN;
    int bx = blockIdx.x, par = 0;
    if (p.up4) { par = bx & 3; bx >>= 2; }
    const int py = par >> 1, px = par & 1;
    const int mt = bx / ntn, nt = bx - mt * ntn;
    const int m0 = mt * BM, n0 = nt * BN;
    const int z = blockIdx.y;
    const int kt_begin = z * p.ktiles_per_split;
    int kt_end = kt_begin + p.ktiles_per_split;
    if (kt_end > p.ktiles_total) kt_end = p.ktiles_total;

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
    const int K = ntap * p.Cin;
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
    int tap = (kt_begin * BKE) / p.Cin;
    int c = kt_begin * BKE - tap * p.Cin;
    int ky = tap / tw, kx = tap - ky * tw;

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
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                const bool ok = live && ((a_mask[i] >> tap) & 1u);
                const int pix = UP ? a_pix0[i] + ((a_oy[i] + ky) >> 1) * p.Ws + ((a_ox[i] + kx) >> 1)
                                   : a_pix0[i] + tapdelta;
                const unsigned voff = ok ? (unsigned)pix * csb + (unsigned)lqb : kOOB;
                dma16(A + i * RPP * LDK * 4, voff, rs, soff);
            }
#pragma unroll
            for (int i = 0; i < PB; ++i)
                dma16(Bq + i * RPP * LDK * 4, live ? b_off[i] : kOOB, rsw, (kt + g) * (BK * 4));
            if (live) {
                c += BKE;
                if (c == p.Cin) {
                    c = 0; ++tap; ++kx;
                    if (kx == tw) { kx = 0; ++ky; }
                }
            }
        }
    };

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
                } else {
                    // the 16-B slot is 8 consecutive bf16 channels = this lane's K = 8*(lane>>5) .. +7 operand
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[set][i]),
                                                                        __builtin_bit_cast(bf16x8, fb[set][j]),
                                                                        acc[i][j], 0, 0, 0);
                }
            }
    };

    // ---- main loop.  Per pipeline step (G K-tiles in LDS buffer `cur`):
    //   global loads of step t+1 -> registers | fragment steps 0..S-2 (LDS reads one step ahead)
    //   | registers -> LDS buffer cur^1 (its latency hides behind the last MFMA block)
    //   | last MFMA block | barrier | first fragment read of step t+1 (hidden behind the next fetch).
    if (kt_begin < kt_end && !(p.dbg & 32)) {
        constexpr int PIECES = G * (PA + PB);      // DMA instructions this wave issues per step
        const int nsteps = (kt_end - kt_begin + G - 1) / G;
        fetch(kt_begin, 0);
        if (NS > 2 && nsteps > 1) fetch(kt_begin + G, 1);
        if (NS > 2 && nsteps > 1) dma_wait<PIECES>(); else dma_wait<0>();
        __syncthreads();
        int cur = 0;
        read_frag(0, 0, 0);
        for (int t = 0; t < nsteps; ++t) {
            // ring slot (cur + NS-1) % NS was last read in step t-1; every wave has passed that barrier
            const int ahead = t + NS - 1;
            const bool issue = ahead < nsteps && !(p.dbg & 1);
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
            if (!(p.dbg & 4)) __syncthreads();
            if (!(p.dbg & 8)) { if (++cur == NS) cur = 0; }
            if (t + 1 < nsteps) read_frag(cur, 0, 0);
        }
    }

    // ---- epilogue.  C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5),
    // i.e. a lane owns 16 rows of ONE column -- storing that directly means 4-byte accesses.  Each
    // wave instead transposes its 32x32 tile through a private 4.5-KB LDS patch (the K-loop buffers
    // are free after the last barrier) so every lane ends up with 4 consecutive channels of a row:
    // float4 residual loads / stores, 8 lanes per 128-B row segment, 4 instead of 16 memory
    // instructions per tile.  Same-wave LDS traffic needs no barrier (a wave's DS ops execute in order).
    if (p.dbg & 16) return;
    constexpr int EP = 36;                                   // patch row pitch (floats), keeps rows 16-B aligned
    float *patch = smem + wave * (32 * EP);
    const int ccol = lane & 31, crow = 4 * (lane >> 5);
    const int erow = lane >> 3, ecol = (lane & 7) * 4;      // this lane's (row within 8-row pass, first channel)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nb = n0 + (wn * TN + j) * 32;              // first channel of the tile
        const int n = nb + ecol;
        const bool nok = n < p.Cout;                         // Cout % 4 == 0: a float4 is all-in or all-out
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.splits == 1 && p.scale && nok) {
            sc = *reinterpret_cast<const float4 *>(p.scale + n);
            sh = *reinterpret_cast<const float4 *>(p.shift + n);
        }
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
                if (m >= p.M || !nok) continue;
                size_t orow = (size_t)m;                     // output pixel index (NHWC row)
                if (p.up4) {
                    const int b = (int)p.div_rhw.div((unsigned)m);
                    const int rr = m - b * rhw;
                    const int y = (int)p.div_rw.div((unsigned)rr), x = rr - y * rw;
                    orow = ((size_t)b * p.Ho + 2 * y + py) * p.Wo + 2 * x + px;
                }
                if (p.splits > 1) {
                    *reinterpret_cast<float4 *>(p.partial + ((size_t)z * p.Mout + orow) * p.Cout + n) = v;
                } else {
                    v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                    if (p.residual) {
                        const float4 rv = load4(static_cast<const T *>(p.residual) + orow * p.Cout + n);
                        v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                    }
                    if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    store4(static_cast<T *>(p.out) + orow * p.Cout + n, v);
                }
            }
        }
    }
}

// out = epilogue(sum_z partial[z]).  Block = 64 float4 columns x 4 z-lanes: z-lane y adds
// partials y, y+4, y+8, ... (ascending), then the four lane sums are added in lane order --
// a fixed summation tree, so results are bit-reproducible run to run.
template <typename T>
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
    if (p.relu) { s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f); }
    store4(static_cast<T *>(p.out) + (size_t)i * 4, s);
}

template <typename T, int BM, int BN, int WGM, int WGN, int G, bool UP>
static hipError_t launch_igemm_t(const IgemmParams &p, hipStream_t s)
{
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.Cout + BN - 1) / BN;
    const int npar = p.up4 ? 4 : 1;
    constexpr size_t smem_one = (size_t)G * (BM + BN) * LDK * sizeof(float);
    constexpr size_t smem_max = smem_one * igemm_stages(BM, BN, G);
    constexpr size_t smem_patch = (size_t)WGM * WGN * 32 * 36 * sizeof(float);   // epilogue transpose patches
    size_t smem = p.ktiles_per_split > G ? smem_max : smem_one;
    if (smem < smem_patch) smem = smem_patch;
    static bool attr_done = false;   // raise the dynamic-LDS cap once per instantiation
    if (smem_max > 64 * 1024 && !attr_done) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&igemm3x3<T, BM, BN, WGM, WGN, G, UP>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_max);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((igemm3x3<T, BM, BN, WGM, WGN, G, UP>), dim3(ntm * ntn * npar, p.splits), dim3(64 * WGM * WGN),
                       smem, s, p);
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

hipError_t igemm_init() { return hipSuccess; }

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
    const size_t eb = p.dtype == 1 ? 2 : 4;
    if ((size_t)p.B * p.Hs * p.Ws * (size_t)(p.C0 > p.C1 ? p.C0 : p.C1) * eb > lim) return hipErrorInvalidValue;
    if ((p.C0 * eb) % 128 || (p.C1 * eb) % 128) return hipErrorInvalidValue;   // a K-tile is 128 B of channels
    return p.dtype == 1 ? launch_igemm_typed<bf16_t>(p, bm, bn, g, s) : launch_igemm_typed<float>(p, bm, bn, g, s);
}

hipError_t launch_splitk_reduce(const IgemmParams &p, hipStream_t s)
{
    const size_t total4 = (size_t)p.Mout * p.Cout / 4;
    if (p.dtype == 1) hipLaunchKernelGGL(splitk_reduce<bf16_t>, dim3((unsigned)((total4 + 63) / 64)), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(splitk_reduce<float>, dim3((unsigned)((total4 + 63) / 64)), dim3(256), 0, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Tiny-M convolution (<= 16 output pixels in the whole batch: the 4x4 and 2x2 levels at batch 1).
// These layers are pure weight streaming (9.4 MB of weights for <= 0.08 GFLOP) and were paying two
// launches each (split-K igemm + reduce, ~14 us).  Here one launch does the whole layer: every
// workgroup owns NC output channels over the FULL K, so no cross-workgroup reduction exists;
// all 256 CUs stream weight rows (issued first, one latency), the whole input tensor (<= 64 KB)
// is staged in LDS, each thread multiplies its K-slice for all pixels on the VALU, and the block
// reduces through LDS in a fixed order.  Same packed weights [Cout][tap][Cin] as the igemm.
template <typename T, int NC>
__global__ __launch_bounds__(256) void conv3x3_smallm(const SmallMParams p)
{
    constexpr int MM = 16;                                 // max output pixels (batch folded in)
    constexpr int NJ = 5;                                  // K/4 <= 5*256 float4 per weight row (Cin <= 512... 568)
    constexpr int RS = 264;                                // reduction row pitch (floats)
    extern __shared__ __attribute__((aligned(16))) float sm[];
    int *pixtab = reinterpret_cast<int *>(sm);             // [9][MM] source pixel of (tap, m); padding -> the zero pixel
    float *act = sm + 160;                                 // [B*Hs*Ws + 1][Cin] (last pixel = zeros); later red[MM*NC][RS]
    const int tid = threadIdx.x;
    const int n0 = blockIdx.x * NC;
    const int C4 = p.Cin >> 2, K4 = 9 * C4;

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
    // 2. input tensor -> LDS, (m, tap) -> source pixel table
    const int npix = p.B * p.Hs * p.Ws;
    const int nin4 = npix * C4;
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
    if (tid < MM * NC * 8) {
        const int o = tid >> 3, part = tid & 7;
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) sum += red[o * RS + i * 8 + part];
        sum += __shfl_xor(sum, 1);
        sum += __shfl_xor(sum, 2);
        sum += __shfl_xor(sum, 4);
        const int m = o / NC, n = n0 + o % NC;
        if (part == 0 && m < p.M && n < p.Cout) {
            float v = sum;
            if (p.scale) v = v * p.scale[n] + p.shift[n];
            if (p.residual) {
                if constexpr (sizeof(T) == 4) v += static_cast<const float *>(p.residual)[(size_t)m * p.Cout + n];
                else v += bf2f(static_cast<const bf16_t *>(p.residual)[(size_t)m * p.Cout + n]);
            }
            if (p.relu) v = fmaxf(v, 0.f);
            if constexpr (sizeof(T) == 4) static_cast<float *>(p.out)[(size_t)m * p.Cout + n] = v;
            else static_cast<bf16_t *>(p.out)[(size_t)m * p.Cout + n] = f2bf(v);
        }
    }
}

bool smallm_supported(const SmallMParams &p)
{
    const size_t act_bytes = (size_t)p.B * p.Hs * p.Ws * p.Cin * 4;
    return p.M <= 16 && p.Cin % 256 == 0 && 9 * (p.Cin / 4) <= 5 * 256 && act_bytes <= 64 * 1024 && p.Cout % 2 == 0;
}

hipError_t launch_smallm(const SmallMParams &p, hipStream_t s)
{
    if (!smallm_supported(p)) return hipErrorInvalidValue;
    constexpr int NC = 2;
    const size_t act_bytes = ((size_t)p.B * p.Hs * p.Ws + 1) * p.Cin * 4;
    const size_t red_bytes = (size_t)16 * NC * 264 * 4;
    const size_t smem = 160 * 4 + (act_bytes > red_bytes ? act_bytes : red_bytes);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_smallm<float, NC>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 4 + 66 * 1024);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_smallm<bf16_t, NC>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 4 + 66 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (p.dtype == 1) hipLaunchKernelGGL((conv3x3_smallm<bf16_t, NC>), dim3((p.Cout + NC - 1) / NC), dim3(256), smem, s, p);
    else hipLaunchKernelGGL((conv3x3_smallm<float, NC>), dim3((p.Cout + NC - 1) / NC), dim3(256), smem, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// First layer.  One thread = one output pixel x 32 output channels (blockIdx.y picks the
// channel slab).  Lanes run along ox, so the stride-2 NCHW reads of a wave cover one contiguous
// 512-B span per (ci, ky) that all three kx taps share; weights are broadcast from LDS.  All 9
// taps of a channel are loaded before any FMA (9 independent loads in flight per lane; the
// one-load-per-tap form was latency-bound at ~50 us).
// Reference: cat (feature2face_model.py:231) + Conv2d(13, ngf, 3, 2, 1, bias=False) + ReLU
// (networks.py:594, :603, :619 `down = [downconv, downrelu]`).
template <typename T>
__global__ __launch_bounds__(256) void first_conv(const FirstConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) float wsm[];   // [K][32]
    // channel range of this pass: the whole input, or -- when the candidate stack is shared by the batch --
    // only the candidate channels (pass 1, once: writes the pre-activation partial sums `base`) or only
    // the feature-map channels (pass 2, per frame: starts from `base`, applies ReLU)
    const int cbeg = p.ci_begin, cin = p.ci_end;
    const int K = (cin - cbeg) * 9;
    const int co0 = blockIdx.y * 32;
    for (int i = threadIdx.x; i < K * 32; i += blockDim.x) {
        const int k = i >> 5, j = i & 31;
        wsm[i] = p.w[(size_t)(cbeg * 9 + k) * p.Cout + co0 + j];
    }
    __syncthreads();

    const int Ho = p.H / 2, Wo = p.W / 2;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)p.B * Ho * Wo) return;
    const int b = (int)(gid / (Ho * Wo));
    const int r = (int)(gid - (long)b * Ho * Wo);
    const int oy = r / Wo, ox = r - oy * Wo;

    // tap offsets / validity are the same for every channel
    int toff[9];
    bool tok[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int iy = 2 * oy + t / 3 - 1, ix = 2 * ox + t % 3 - 1;
        tok[t] = (iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.W);
        toff[t] = tok[t] ? iy * p.W + ix : 0;
    }

    float acc[32];
    if (p.base) {
        // base is [1][Ho][Wo][Cout] (shared by every frame of the batch)
        const float4 *bp = reinterpret_cast<const float4 *>(p.base + (size_t)r * p.Cout + co0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 t = bp[j];
            acc[4 * j] = t.x; acc[4 * j + 1] = t.y; acc[4 * j + 2] = t.z; acc[4 * j + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 32; ++j) acc[j] = 0.f;
    }

    const size_t plane = (size_t)p.H * p.W;
    auto plane_of = [&](int ci) -> const float * {
        return (ci < p.feat_nc)
            ? p.feat + ((size_t)b * p.feat_nc + ci) * plane
            : p.cand + ((size_t)(p.cand_batch == 1 ? 0 : b) * p.cand_nc + (ci - p.feat_nc)) * plane;
    };
    // the 9 taps of channel ci+1 are in flight while channel ci is multiplied
    float v[9], vn[9];
    {
        const float *src = plane_of(cbeg);
#pragma unroll
        for (int t = 0; t < 9; ++t) v[t] = tok[t] ? src[toff[t]] : 0.f;
    }
#pragma unroll 1
    for (int ci = cbeg; ci < cin; ++ci) {
        if (ci + 1 < cin) {
            const float *src = plane_of(ci + 1);
#pragma unroll
            for (int t = 0; t < 9; ++t) vn[t] = tok[t] ? src[toff[t]] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float4 *wr = reinterpret_cast<const float4 *>(wsm + ((ci - cbeg) * 9 + t) * 32);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 w4 = wr[j];
                acc[4 * j + 0] += v[t] * w4.x; acc[4 * j + 1] += v[t] * w4.y;
                acc[4 * j + 2] += v[t] * w4.z; acc[4 * j + 3] += v[t] * w4.w;
            }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) v[t] = vn[t];
    }
    T *o = static_cast<T *>(p.out) + (size_t)gid * p.Cout + co0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        store4(o + 4 * j, p.relu ? make_float4(fmaxf(acc[4 * j], 0.f), fmaxf(acc[4 * j + 1], 0.f),
                                               fmaxf(acc[4 * j + 2], 0.f), fmaxf(acc[4 * j + 3], 0.f))
                                 : make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]));
}

// Feature-map-only pass of the first layer (the candidate share comes in through `base`): pure streaming,
// ~1 FLOP/byte.  16 lanes share a pixel and own 4 output channels each, so a wave reads and writes
// 4 pixels x Cout*4 B contiguously (Cout = 64: exactly 1 KB per instruction); the 9 x feat_nc tap values
// are broadcast loads, the lane's weights stay in registers.
template <typename T, int FN>
__global__ __launch_bounds__(256) void first_conv_feat(const FirstConvParams p)
{
    const int Ho = p.H / 2, Wo = p.W / 2;
    const int lpp = p.Cout / 4;                              // lanes per pixel (16 for ngf 64, 8 for ngf 32)
    const int ppw = 64 / lpp;                                // pixels per wave
    const int lane = threadIdx.x & 63;
    const int j = lane % lpp, sub = lane / lpp;
    float4 w[FN][9];
#pragma unroll
    for (int ci = 0; ci < FN; ++ci)
#pragma unroll
        for (int t = 0; t < 9; ++t)
            w[ci][t] = *reinterpret_cast<const float4 *>(p.w + (size_t)(ci * 9 + t) * p.Cout + j * 4);
    const long npix = (long)p.B * Ho * Wo;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    const size_t plane = (size_t)p.H * p.W;
    for (long g = wave * ppw + sub; g < npix; g += nwaves * ppw) {
        const int b = (int)(g / (Ho * Wo));
        const int r = (int)(g - (long)b * Ho * Wo);
        const int oy = r / Wo, ox = r - oy * Wo;
        float4 acc = *reinterpret_cast<const float4 *>(p.base + (size_t)r * p.Cout + j * 4);
#pragma unroll
        for (int ci = 0; ci < FN; ++ci) {
            const float *src = p.feat + ((size_t)b * p.feat_nc + ci) * plane;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int iy = 2 * oy + t / 3 - 1, ix = 2 * ox + t % 3 - 1;
                const bool ok = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
                const float v = ok ? src[(size_t)iy * p.W + ix] : 0.f;
                acc.x += v * w[ci][t].x; acc.y += v * w[ci][t].y; acc.z += v * w[ci][t].z; acc.w += v * w[ci][t].w;
            }
        }
        if (p.relu) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
        store4(static_cast<T *>(p.out) + (size_t)g * p.Cout + j * 4, acc);
    }
}

hipError_t launch_first_conv(const FirstConvParams &p, hipStream_t s)
{
    if (p.base && p.ci_begin == 0 && p.ci_end == 1 && p.feat_nc == 1 && 64 % (p.Cout / 4) == 0 && p.Cout <= 256) {
        const long npix = (long)p.B * (p.H / 2) * (p.W / 2);
        const int ppw = 64 / (p.Cout / 4);
        long blocks = (npix / ppw + 3) / 4;                  // one pixel group per wave ...
        if (blocks > 4096) blocks = 4096;                    // ... up to 16 blocks per CU, then grid-stride
        if (blocks < 1) blocks = 1;
        if (p.dtype == 1) hipLaunchKernelGGL((first_conv_feat<bf16_t, 1>), dim3((unsigned)blocks), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((first_conv_feat<float, 1>), dim3((unsigned)blocks), dim3(256), 0, s, p);
        return hipGetLastError();
    }
    const long total = (long)p.B * (p.H / 2) * (p.W / 2);
    const int K = (p.ci_end - p.ci_begin) * 9;
    if (p.dtype == 1 && p.out != nullptr && !(p.relu == 0 && p.base == nullptr && p.ci_begin > 0))
        hipLaunchKernelGGL(first_conv<bf16_t>, dim3((unsigned)((total + 255) / 256), p.Cout / 32), dim3(256),
                           (size_t)K * 32 * sizeof(float), s, p);
    else   // fp32 activations, or the candidate-share pass (its cache is always fp32)
        hipLaunchKernelGGL(first_conv<float>, dim3((unsigned)((total + 255) / 256), p.Cout / 32), dim3(256),
                           (size_t)K * 32 * sizeof(float), s, p);
    return hipGetLastError();
}

// util.tensor2im (reference util/util.py:19-42) on one value: (x + 1) / 2 * 255 in float32, clip to
// [0, 255], truncate to uint8 -- the same operation order as the numpy expression it replaces.
__device__ __forceinline__ unsigned char to_u8(float v)
{
    float t = (v + 1.0f) / 2.0f * 255.0f;
    t = fminf(fmaxf(t, 0.0f), 255.0f);
    return (unsigned char)t;
}

// ------------------------------------------------------------------------------------------
// Last layer, sub-pixel form.  Upsample(x2, nearest) + Conv3x3 over cat([src0, src1]) + tanh:
// output parity (py, px) only sees a 2x2 neighbourhood of the half-resolution source, so the
// packer pre-sums the aliasing taps (plan.cpp) and each output pixel costs 4 taps instead of 9.
// One thread = one output pixel, all CO (<= 4) channels.  Threads are ordered parity-major
// (b, py, px, y, x) with x fastest, so a wave reads 64 consecutive source pixels and all its lanes
// use the same weights (LDS broadcast); the concat is two base pointers; NCHW store.
// Reference: nn.Upsample(2,'nearest') + Conv2d(2*ngf, 3, 3, 1, 1, bias=False) (networks.py:610-611)
// + torch.tanh (networks.py:577).
template <typename T, int CO>
__global__ __launch_bounds__(256) void last_conv(const LastConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) float wsm[];   // [4 parities][CO][2][2][Cin]
    const int cin = p.C0 + p.C1;
    for (int i = threadIdx.x; i < 16 * CO * cin; i += blockDim.x) wsm[i] = p.w[i];
    __syncthreads();

    const int H = 2 * p.Hs, W = 2 * p.Ws;
    const long per_par = (long)p.Hs * p.Ws;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)p.B * 4 * per_par) return;
    const int b = (int)(gid / (4 * per_par));
    long rem = gid - (long)b * 4 * per_par;
    const int par = (int)(rem / per_par);
    rem -= (long)par * per_par;
    const int y = (int)(rem / p.Ws), x = (int)(rem - (long)y * p.Ws);
    const int py = par >> 1, px = par & 1;

    float acc[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) acc[co] = 0.f;

#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int sy = y + a - 1 + py;
        if (sy < 0 || sy >= p.Hs) continue;
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
            const int sx = x + bb - 1 + px;
            if (sx < 0 || sx >= p.Ws) continue;
            const size_t pix = ((size_t)b * p.Hs + sy) * p.Ws + sx;
            const float *wt = wsm + ((par * CO) * 4 + a * 2 + bb) * cin;      // + co*4*cin
            const T *s0 = static_cast<const T *>(p.src0) + pix * p.C0;
#pragma unroll 4
            for (int c4 = 0; c4 < p.C0 / 4; ++c4) {
                const float4 v = load4(s0 + c4 * 4);
#pragma unroll
                for (int co = 0; co < CO; ++co) {
                    const float4 w4 = *reinterpret_cast<const float4 *>(wt + co * 4 * cin + c4 * 4);
                    acc[co] += v.x * w4.x + v.y * w4.y + v.z * w4.z + v.w * w4.w;
                }
            }
            if (p.C1) {
                const T *s1 = static_cast<const T *>(p.src1) + pix * p.C1;
#pragma unroll 4
                for (int c4 = 0; c4 < p.C1 / 4; ++c4) {
                    const float4 v = load4(s1 + c4 * 4);
#pragma unroll
                    for (int co = 0; co < CO; ++co) {
                        const float4 w4 = *reinterpret_cast<const float4 *>(wt + co * 4 * cin + p.C0 + c4 * 4);
                        acc[co] += v.x * w4.x + v.y * w4.y + v.z * w4.z + v.w * w4.w;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int co = 0; co < CO; ++co)
    {
        const float v = p.apply_tanh ? tanhf(acc[co]) : acc[co];
        if (p.out) p.out[(((size_t)b * CO + co) * H + 2 * y + py) * W + 2 * x + px] = v;
        if (p.out_u8) p.out_u8[(((size_t)b * H + 2 * y + py) * W + 2 * x + px) * CO + co] = to_u8(v);
    }
}

// Fast path of the last layer for C0 == C1 <= 64*NCH: the 16 lanes of a DPP row share one output
// pixel and split its input channels (lane j owns channels 4j..4j+3 of every 64-channel slab), so a
// wave's load of 4 neighbouring pixels is one contiguous 1-KB read and the channel reduction is 4 DPP
// row rotations per output.  Each wave walks a contiguous run of pixel quads of ONE output parity
// (wave id & 3); that parity's 4-tap x CO weights sit in LDS (conflict-free: a row's 16 lanes read 16
// consecutive float4, the 4 rows broadcast), which keeps the kernel at ~64 VGPRs = 8 waves/SIMD --
// the loop is a load -> FMA -> DPP -> store chain and needs the occupancy to hide its latency.
template <typename T, int CO, int NCH>
__global__ __launch_bounds__(256) void last_conv_rows(const LastConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) float wsm[];   // [4 parities][2 src][NCH][4 taps][CO][16 lanes] float4
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, sub = lane >> 4;               // channel slot, pixel within the quad
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int par = wave & 3, py = par >> 1, px = par & 1;
    const int cin = p.C0 + p.C1;
    const int H = 2 * p.Hs, W = 2 * p.Ws;

    constexpr int WPP = 2 * NCH * 4 * CO * 16;              // float4 per parity
    for (int i = threadIdx.x; i < 4 * WPP; i += blockDim.x) {
        int r = i;
        const int jj = r & 15; r >>= 4;
        const int co = r % CO; r /= CO;
        const int t = r & 3; r >>= 2;
        const int c = r % NCH; r /= NCH;
        const int sidx = r & 1, pr = r >> 1;
        const int ch = (c * 16 + jj) * 4;
        const bool okc = ch < (sidx ? p.C1 : p.C0);
        reinterpret_cast<float4 *>(wsm)[i] = okc
            ? *reinterpret_cast<const float4 *>(p.w + ((size_t)(pr * CO + co) * 4 + t) * cin + (sidx ? p.C0 : 0) + ch)
            : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const float4 *wpar = reinterpret_cast<const float4 *>(wsm) + par * WPP + j;   // + ((sidx*NCH + c)*4 + t)*CO*16 + co*16

    const int qpr = (p.Ws + 3) / 4;                          // quads per source row
    const unsigned nquads = (unsigned)p.B * p.Hs * qpr;
    const unsigned wpp = (unsigned)nwaves >> 2;              // waves per parity
    const unsigned chunk = (nquads + wpp - 1) / wpp;
    unsigned q = (unsigned)(wave >> 2) * chunk;
    const unsigned qend = q + chunk < nquads ? q + chunk : nquads;
    if (q >= qend) return;
    int b = (int)(q / ((unsigned)p.Hs * qpr));
    int y, xq;
    { const unsigned r = q - (unsigned)b * p.Hs * qpr; y = (int)(r / qpr); xq = (int)(r - (unsigned)y * qpr); }

    const T *__restrict__ s0 = static_cast<const T *>(p.src0);
    const T *__restrict__ s1 = static_cast<const T *>(p.src1);
    float *__restrict__ outp = p.out;

    for (; q < qend; ++q) {
        asm volatile("" ::: "memory");   // keep the weight reads in LDS (LICM would pin 96 VGPRs)
        const int x = xq * 4 + sub;
        float4 v[2][4][NCH];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int sy = y + (t >> 1) - 1 + py, sx = x + (t & 1) - 1 + px;
            const bool ok = (x < p.Ws) & ((unsigned)sy < (unsigned)p.Hs) & ((unsigned)sx < (unsigned)p.Ws);
            const size_t pix = ok ? ((size_t)b * p.Hs + sy) * p.Ws + sx : 0;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int ch = (c * 16 + j) * 4;
                v[0][t][c] = (ok && ch < p.C0) ? load4(s0 + pix * p.C0 + ch)
                                               : make_float4(0.f, 0.f, 0.f, 0.f);
                v[1][t][c] = (ok && ch < p.C1) ? load4(s1 + pix * p.C1 + ch)
                                               : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        float acc[CO];
#pragma unroll
        for (int co = 0; co < CO; ++co) acc[co] = 0.f;
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx)
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float4 x4 = v[sidx][t][c];
#pragma unroll
                    for (int co = 0; co < CO; ++co) {
                        const float4 ww = wpar[(((sidx * NCH + c) * 4 + t) * CO + co) * 16];
                        acc[co] += x4.x * ww.x + x4.y * ww.y + x4.z * ww.z + x4.w * ww.w;
                    }
                }
        // sum over the 16 channel slots of the row: rotate-and-add (row_ror 8, 4, 2, 1)
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            float r = acc[co];
            r += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, r), 0x128, 0xf, 0xf, false));
            r += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, r), 0x124, 0xf, 0xf, false));
            r += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, r), 0x122, 0xf, 0xf, false));
            r += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, r), 0x121, 0xf, 0xf, false));
            acc[co] = r;
        }
        if (j < CO && x < p.Ws) {
            float r = acc[0];
#pragma unroll
            for (int co = 1; co < CO; ++co) r = (j == co) ? acc[co] : r;
            r = p.apply_tanh ? tanhf(r) : r;
            if (outp) outp[(((size_t)b * CO + j) * H + 2 * y + py) * W + 2 * x + px] = r;
            if (p.out_u8) p.out_u8[(((size_t)b * H + 2 * y + py) * W + 2 * x + px) * CO + j] = to_u8(r);
        }
        if (++xq == qpr) { xq = 0; if (++y == p.Hs) { y = 0; ++b; } }
    }
}

// Last layer, sliding-window form (C0 == C1 <= 64).  The rows kernel above re-reads every source pixel
// 16x (4 taps x 4 parities) through L1, which bounds it at ~1 TB/s of useful traffic.  Here a 16-lane
// group (lane = 4 input channels of each source) walks along one source row keeping the 3x3 source
// neighbourhood in registers: per step it loads ONE new column (3 rows x 2 sources) and emits all four
// output parities of that source pixel -- 1.5 loads per output instead of 8 -- with the pre-summed
// sub-pixel weights read conflict-free from LDS and the channel sum done by DPP row rotations.
template <typename T, int CO>
__global__ __launch_bounds__(256) void last_conv_strip(const LastConvParams p, int seg)
{
    extern __shared__ __attribute__((aligned(16))) float wsm[];   // [4 par][2 src][4 taps][CO][16 lanes] float4
    const int lane = threadIdx.x & 63, j = lane & 15;
    const int cin = p.C0 + p.C1;
    const int H = 2 * p.Hs, W = 2 * p.Ws;
    constexpr int WTOT = 4 * 2 * 4 * CO * 16;
    for (int i = threadIdx.x; i < WTOT; i += blockDim.x) {
        int r = i;
        const int jj = r & 15; r >>= 4;
        const int co = r % CO; r /= CO;
        const int t = r & 3; r >>= 2;
        const int sidx = r & 1, pr = r >> 1;
        const int ch = jj * 4;
        reinterpret_cast<float4 *>(wsm)[i] = ch < (sidx ? p.C1 : p.C0)
            ? *reinterpret_cast<const float4 *>(p.w + ((size_t)(pr * CO + co) * 4 + t) * cin + (sidx ? p.C0 : 0) + ch)
            : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const float4 *wl = reinterpret_cast<const float4 *>(wsm) + j;

    // one 16-lane group = one (frame, source row, column segment)
    const int nseg = (p.Ws + seg - 1) / seg;
    const long ngroups = (long)p.B * p.Hs * nseg;
    const long group = (((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4);
    if (group >= ngroups) return;          // whole 16-lane rows retire together (DPP rows stay intact)
    const int b = (int)(group / ((long)p.Hs * nseg));
    const int rr = (int)(group - (long)b * p.Hs * nseg);
    const int y = rr / nseg, x0 = (rr - y * nseg) * seg;
    const int x1 = x0 + seg < p.Ws ? x0 + seg : p.Ws;
    const bool chan_ok = j * 4 < p.C0;

    const T *__restrict__ s0 = static_cast<const T *>(p.src0);
    const T *__restrict__ s1 = static_cast<const T *>(p.src1);
    auto load_col = [&](int x, float4 (&col)[3][2]) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int sy = y + r - 1;
            const bool ok = chan_ok & ((unsigned)sy < (unsigned)p.Hs) & ((unsigned)x < (unsigned)p.Ws);
            const size_t pix = ok ? ((size_t)b * p.Hs + sy) * p.Ws + x : 0;
            col[r][0] = ok ? load4(s0 + pix * p.C0 + j * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            col[r][1] = ok ? load4(s1 + pix * p.C1 + j * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    float4 win[3][3][2];                       // [column x-1, x, x+1][row y-1, y, y+1][source]
    float4 nxt[3][2];
    load_col(x0 - 1, win[0]);
    load_col(x0, win[1]);
    load_col(x0 + 1, win[2]);
    for (int x = x0; x < x1; ++x) {
        asm volatile("" ::: "memory");         // keep the weights in LDS (see last_conv_rows)
        if (x + 1 < x1) load_col(x + 2, nxt);  // next step's column flies while this step computes
        float acc[4][CO];
#pragma unroll
        for (int par = 0; par < 4; ++par)
#pragma unroll
            for (int co = 0; co < CO; ++co) acc[par][co] = 0.f;
#pragma unroll
        for (int par = 0; par < 4; ++par) {
            const int py = par >> 1, px = par & 1;
            // all 8*CO weight reads of this parity are issued as one batch (one LDS latency per parity instead of
            // one per use: PMC showed 64 % of the wave time in s_waitcnt with just-in-time reads)
            float4 wq[2][4][CO];
#pragma unroll
            for (int sidx = 0; sidx < 2; ++sidx)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int co = 0; co < CO; ++co) wq[sidx][t][co] = wl[(((par * 2 + sidx) * 4 + t) * CO + co) * 16];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int sidx = 0; sidx < 2; ++sidx)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float4 v = win[(t & 1) + px][(t >> 1) + py][sidx];
#pragma unroll
                    for (int co = 0; co < CO; ++co) {
                        const float4 ww = wq[sidx][t][co];
                        acc[par][co] += v.x * ww.x + v.y * ww.y + v.z * ww.z + v.w * ww.w;
                    }
                }
        }
#pragma unroll
        for (int par = 0; par < 4; ++par) {
#pragma unroll
            for (int co = 0; co < CO; ++co) {
                float r = acc[par][co];
                r += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, r), 0x128, 0xf, 0xf, false));
                r += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, r), 0x124, 0xf, 0xf, false));
                r += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, r), 0x122, 0xf, 0xf, false));
                r += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, r), 0x121, 0xf, 0xf, false));
                acc[par][co] = r;
            }
        }
        if (j < CO) {
#pragma unroll
            for (int par = 0; par < 4; ++par) {
                float r = acc[par][0];
#pragma unroll
                for (int co = 1; co < CO; ++co) r = (j == co) ? acc[par][co] : r;
                r = p.apply_tanh ? tanhf(r) : r;
                const int Y = 2 * y + (par >> 1), X = 2 * x + (par & 1);
                if (p.out) p.out[(((size_t)b * CO + j) * H + Y) * W + X] = r;
                if (p.out_u8) p.out_u8[(((size_t)b * H + Y) * W + X) * CO + j] = to_u8(r);
            }
        }
        // slide the window
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int sidx = 0; sidx < 2; ++sidx) {
                win[0][r][sidx] = win[1][r][sidx];
                win[1][r][sidx] = win[2][r][sidx];
                win[2][r][sidx] = nxt[r][sidx];
            }
    }
}

template <typename T, int CO>
static hipError_t launch_last_conv_co(const LastConvParams &p, hipStream_t s)
{
    const long quads = (long)p.B * p.Hs * ((p.Ws + 3) / 4);
    // measured on MI355X (512x512 output): batch 1 rows 43 us / strip 55 us; batch 8 rows 270 us / strip 221 us
    const bool big = (long)p.B * p.Hs * p.Ws >= 4 * 65536;
    if (p.C0 == p.C1 && p.C0 % 4 == 0 && p.C0 <= 64 && (big || std::getenv("LSP_HIP_LASTCONV_STRIP")) &&
        !std::getenv("LSP_HIP_LASTCONV_ROWS") && !std::getenv("LSP_HIP_LASTCONV_GENERIC")) {
        // sliding-window kernel; segment length trades window priming (2 extra columns) for parallelism
        const int seg = 32;
        const long groups = (long)p.B * p.Hs * ((p.Ws + seg - 1) / seg);
        const size_t smem = (size_t)4 * 2 * 4 * CO * 16 * sizeof(float4);
        hipLaunchKernelGGL((last_conv_strip<T, CO>), dim3((unsigned)((groups + 15) / 16)), dim3(256), smem, s, p, seg);
        return hipGetLastError();
    }
    if (p.C0 == p.C1 && p.C0 % 4 == 0 && p.C0 <= 128 && !std::getenv("LSP_HIP_LASTCONV_GENERIC")) {
        // 4 parities x quads wave-iterations; 4 waves per block, parity = wave & 3
        long blocks = (quads + 3) / 4;                      // >= ~4 quads per wave
        if (blocks > 2048) blocks = 2048;                   // 8 blocks (32 waves) per CU, all resident
        if (blocks < 1) blocks = 1;
        const int nch = p.C0 <= 64 ? 1 : 2;
        const size_t smem = (size_t)4 * 2 * nch * 4 * CO * 16 * sizeof(float4);
        if (nch == 1) hipLaunchKernelGGL((last_conv_rows<T, CO, 1>), dim3((unsigned)blocks), dim3(256), smem, s, p);
        else hipLaunchKernelGGL((last_conv_rows<T, CO, 2>), dim3((unsigned)blocks), dim3(256), smem, s, p);
        return hipGetLastError();
    }
    const long total = (long)p.B * 4 * p.Hs * p.Ws;
    const size_t smem = (size_t)16 * p.Cout * (p.C0 + p.C1) * sizeof(float);
    hipLaunchKernelGGL((last_conv<T, CO>), dim3((unsigned)((total + 255) / 256)), dim3(256), smem, s, p);
    return hipGetLastError();
}

hipError_t launch_last_conv(const LastConvParams &p, hipStream_t s)
{
    if (p.dtype == 1) {
        switch (p.Cout) {
        case 1: return launch_last_conv_co<bf16_t, 1>(p, s);
        case 2: return launch_last_conv_co<bf16_t, 2>(p, s);
        case 3: return launch_last_conv_co<bf16_t, 3>(p, s);
        case 4: return launch_last_conv_co<bf16_t, 4>(p, s);
        default: return hipErrorInvalidValue;
        }
    }
    switch (p.Cout) {
    case 1: return launch_last_conv_co<float, 1>(p, s);
    case 2: return launch_last_conv_co<float, 2>(p, s);
    case 3: return launch_last_conv_co<float, 3>(p, s);
    case 4: return launch_last_conv_co<float, 4>(p, s);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace lspf2f
