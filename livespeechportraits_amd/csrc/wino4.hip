// gfx950 (MI355X / CDNA4): stride-1 3x3 convolution as Winograd F(4x4, 3x3) on the fp32 matrix cores -- 36 GEMMs of
// [tiles x Cin] . [Cin x Cout] (one per position of the 6x6 transformed tile): 36 multiplies per 16 outputs, 1/4 of the matrix FLOPs of the
// implicit GEMM (igemm.hip) and 9/16 of F(2x2, 3x3) (wino.hip).  fp32 in, fp32 accumulate, BatchNorm-folded scale / shift, residual add and
// ReLU in the epilogue like every other conv of this library.  DESIGN.md section 4.11; the fp32 error of this transform over the whole
// generator was measured before the kernel was written (tools/wino4_error.py, profiles/r04_wino4x4_error.txt: 1.9e-6 end to end on `large`,
// the same as F(2x2)).
//
// Reference semantics: the 3x3 / stride 1 / pad 1 / bias-free Conv2d calls of ResidualBlock, models/networks.py:650-675 (:663, :666),
// followed by BatchNorm2d in eval mode, the residual add and ReLU (:670-675).
//
//   Y = A^T [ sum_c (G g G^T) . (B^T d B) ] A      interpolation points 0, +-1, +-2, inf (Lavin & Gray)
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
//
// Workgroup = 4 x 8 Winograd tiles (16 x 32 output pixels of one frame) x 32 output channels, 4 waves, ONE workgroup per CU (a ring slot of
// transformed weights + raw patch is 56 KB).  Wave (a, b) owns the 3x3 block rows 3a..3a+2, columns 3b..3b+2 of the transformed tile: 9
// accumulators of 32 tiles x 32 channels (v_mfma_f32_32x32x2_f32, 144 registers).  It needs rows a..a+4 and columns b..b+4 of the raw 6x6
// patch only (25 ds_read_b128 per 8 channels), forms its nine V values in registers (rows, then columns: 48 four-channel operations) and
// feeds them straight to the matrix pipe -- the transformed input never exists in memory.  A wave runs alone on its SIMD, so nothing hides
// its latencies but its own instruction stream: the operands of step t + 1 (V and the nine U fragments) are read and transformed into a
// second register set WHILE the 36 MFMAs of step t issue, and the LDS-DMA of step t + 2 goes out between the first MFMAs.  One barrier per
// step.  The four waves meet in the epilogue: the 36 x 32 x 32 accumulator values go through LDS once, then every thread owns one tile x 4
// channels, applies A^T . A in registers and stores its 16 pixels.
#include "device_common.h"
#include "kernels.h"
#include "wino_common.h"

namespace lspf2f {

// phase timestamps (tools/wino4_stamps.py; builds with -DLSPF2F_WINO_STAMPS only): s_memtime values kept in registers, written once at the end
#ifdef LSPF2F_WINO_STAMPS
#define W4STAMP(i) do { __builtin_amdgcn_sched_barrier(0); stamp_t[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define W4STAMP_FLUSH do { if (p.stamps && lane == 0) { unsigned long long *q_ = p.stamps + ((size_t)blockIdx.x * 4 + wave) * 8; \
    for (int i_ = 0; i_ < 8; ++i_) q_[i_] = stamp_t[i_]; } } while (0)
#else
#define W4STAMP(i) do {} while (0)
#define W4STAMP_FLUSH do {} while (0)
#endif

static constexpr unsigned kOOB4 = 0x80000000u;           // voffset beyond any num_records: the LDS-DMA lands zeros (image border, unused chunks)

// ---- geometry shared with the host packer and tools/wino_model.py
static constexpr int kW4RawPieces = 20;                   // 18 x 34 pixels x 2 channel quads = 1224 chunks of 16 B in 20 pieces of 64 (56 chunks unused)
static constexpr int kW4UStage = 36 * 1024;               // bytes of U fragments per ring slot: 4 waves x 9 pieces
static constexpr int kW4RawStage = kW4RawPieces * 1024;
static constexpr int kW4Slot = kW4UStage + kW4RawStage;   // 57 344
static constexpr int kW4Patch = 36 * 32 * 32 * 4;         // epilogue: [position][tile][channel] fp32 = 147 456 B (> 2 ring slots)

// Raw patch in LDS: 16 planes by (py & 3, px & 3) so that the 32 tiles a fragment read touches (4 pixels apart) are consecutive 16-B chunks.
// Plane (pary, parx) holds hyn x hxn pixels x 2 quads: hyn = 5 for pary < 2 (patch rows 16, 17 exist), else 4; hxn = 9 for parx < 2, else 8.
// (Adjacent pixels of a plane are 32 B apart: a fragment read is a 2-way bank conflict, which the LDS has room for -- reads overlap the MFMAs.)
__host__ __device__ constexpr int w4_hyn(int pary) { return pary < 2 ? 5 : 4; }
__host__ __device__ constexpr int w4_hxn(int parx) { return parx < 2 ? 9 : 8; }
__host__ __device__ constexpr int w4_plane_base(int pary, int parx)
{
    return (pary == 0 ? 0 : pary == 1 ? 340 : pary == 2 ? 680 : 952) + (parx == 0 ? 0 : parx == 1 ? 18 : parx == 2 ? 36 : 52) * w4_hyn(pary);
}
// chunk of patch pixel (py, px) = (4 hy + pary, 4 hx + parx), channel quad q.  The two quads of a pixel are ADJACENT chunks: the copy's lanes 2i, 2i + 1
// then ask for 32 contiguous bytes of one cache line instead of 16 bytes of two (the raw pieces are line-count-bound, not byte-bound)
__host__ __device__ constexpr int w4_chunk(int py, int px, int q)
{
    return w4_plane_base(py & 3, px & 3) + ((py >> 2) * w4_hxn(px & 3) + (px >> 2)) * 2 + q;
}
// lane-dependent part of a fragment-read address, by plane class cls = 2 (pary >= 2) + (parx >= 2): lane (tile ty, tx; quad q)
__host__ __device__ constexpr int w4_cls(int dy, int dx) { return ((dy & 3) >= 2 ? 2 : 0) + ((dx & 3) >= 2 ? 1 : 0); }
// ... and the compile-time part for patch offset (dy, dx): the tile's pixel (4 ty + dy, 4 tx + dx)
__host__ __device__ constexpr int w4_imm(int dy, int dx)
{
    return 16 * (w4_plane_base(dy & 3, dx & 3) + ((dy >> 2) * w4_hxn(dx & 3) + (dx >> 2)) * 2);
}

// c * a + b as two v_pk_fma_f32 (see f4add in wino_common.h: a VALU instruction costs the wave ~5 cycles of its MFMA stream, packed or not)
__device__ __forceinline__ float4 f4fma(float c, float4 a, float4 b)
{
#ifdef LSPF2F_NO_PK
    return make_float4(fmaf(c, a.x, b.x), fmaf(c, a.y, b.y), fmaf(c, a.z, b.z), fmaf(c, a.w, b.w));
#else
    const v2f cc = {c, c};
    const v2f lo = __builtin_elementwise_fma(cc, v2f{a.x, a.y}, v2f{b.x, b.y}), hi = __builtin_elementwise_fma(cc, v2f{a.z, a.w}, v2f{b.z, b.w});
    return make_float4(lo.x, lo.y, hi.x, hi.y);
#endif
}

// Three rows of B^T applied to a 5-sample window w = d[H .. H+4] of the 6 patch samples (H = 0: rows 0, 1, 2; H = 1: rows 3, 4, 5): 6 operations
template <int H>
__device__ __forceinline__ void w4_bt3(const float4 (&w)[5], float4 &o0, float4 &o1, float4 &o2)
{
#ifdef W4_ABL_NOVALU          // ablation (results are garbage): no transform arithmetic, the operands are whatever was read
    o0 = w[0]; o1 = w[1]; o2 = w[2];
    return;
#endif
    if constexpr (H == 0) {
        o0 = f4fma(4.f, w[0], f4fma(-5.f, w[2], w[4]));                        // 4 d0 - 5 d2 + d4
        const float4 p = f4fma(-4.f, w[2], w[4]), q = f4fma(-4.f, w[1], w[3]);   // d4 - 4 d2, d3 - 4 d1
        o1 = f4add(p, q); o2 = f4sub(p, q);
    } else {
        const float4 c = f4sub(w[3], w[1]), g = f4sub(w[2], w[0]);               // d4 - d2, d3 - d1
        o0 = f4fma(2.f, g, c); o1 = f4fma(-2.f, g, c);
        o2 = f4fma(4.f, w[0], f4fma(-5.f, w[2], w[4]));                        // 4 d1 - 5 d3 + d5
    }
}

// "this value exists HERE": keeps the optimiser from sinking a transform into the block that consumes it one K-step later (no instruction)
__device__ __forceinline__ void w4_pin(float4 &a) { asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w)); }

struct W4Ops { float4 v[9]; float4 u[9]; };          // one K-step's operands of a wave: V (A side) and U fragments (B side), 4 channels per lane each

// One K-step of wave (A, B): the 36 MFMAs on `cur`; between them the reads + transform of the next step's operands from LDS slot `nslot` into
// `nxt` and (if `issue`) the LDS-DMA of step `ks_dma` into slot `dslot`.
template <int A, int B>
__device__ __forceinline__ void w4_step(f32x16 (&acc)[9], const W4Ops &cur, W4Ops &nxt, const char *smem_c, int nslot, const unsigned (&lp)[4], unsigned au,
                                        bool issue, int ks_dma, int dslot, unsigned lds_u, unsigned lds_r, const unsigned (&vraw)[5], const unsigned (&vu)[3],
                                        i32x4 srd_src, i32x4 srd_u, unsigned soff_u0)
{
    const char *ns = smem_c + nslot * kW4Slot;
    float4 d[2][5], t[3][5];
    auto rd_col = [&](int x, float4 (&dst)[5]) {
#ifdef W4_ABL_NOREAD          // ablation: no raw reads either
#pragma unroll
        for (int y = 0; y < 5; ++y) asm volatile("" : "+v"(dst[y].x), "+v"(dst[y].y), "+v"(dst[y].z), "+v"(dst[y].w));
        return;
#endif
#pragma unroll
        for (int y = 0; y < 5; ++y) dst[y] = *reinterpret_cast<const float4 *>(ns + lp[w4_cls(A + y, B + x)] + (kW4UStage + w4_imm(A + y, B + x)));
    };
    auto mm = [&](int i) {                                  // MFMA i of the step: channel component c = i / 9 of position f = i % 9
        const int c = i / 9, f = i % 9;
        const float a = c == 0 ? cur.v[f].x : c == 1 ? cur.v[f].y : c == 2 ? cur.v[f].z : cur.v[f].w;
        const float b = c == 0 ? cur.u[f].x : c == 1 ? cur.u[f].y : c == 2 ? cur.u[f].z : cur.u[f].w;
        acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[f], 0, 0, 0);
    };
    const unsigned dr = lds_r + (unsigned)(dslot * kW4Slot), du = lds_u + (unsigned)(dslot * kW4Slot);
    const int so = (int)(soff_u0 + (unsigned)ks_dma * 9216u);
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        // the copies of step t + 2 first (a wave-uniform branch: it ends a scheduling block, so the MFMAs of the piece sit behind it with the
        // vector work they are to cover): raw pieces 0-1, raw pieces 2-4, then the nine U fragments three at a time
        if (issue) {
#ifndef W4_ABL_NORAW
            if (k == 0) { const unsigned vv[2] = {vraw[0], vraw[1]}; dma16_group<2, 1024>(dr, vv, srd_src, ks_dma * 32); }
            if (k == 1) { const unsigned vv[3] = {vraw[2], vraw[3], vraw[4]}; dma16_group<3, 1024>(dr + 2048u, vv, srd_src, ks_dma * 32); }
#endif
#ifndef W4_ABL_NOU
            if (k >= 2 && k <= 4) dma16_group<3, 1024>(du + (unsigned)((k - 2) * 3072), vu, srd_u, so + (k - 2) * 3072);
#endif
        }
        mm(3 * k); mm(3 * k + 1); mm(3 * k + 2);
        if (k == 0) {
            rd_col(0, d[0]);
        } else if (k == 1) {
            rd_col(1, d[1]);
            w4_bt3<A>(d[0], t[0][0], t[1][0], t[2][0]);
            w4_pin(t[0][0]); w4_pin(t[1][0]); w4_pin(t[2][0]);
        } else if (k <= 4) {                                 // k = 2, 3, 4: raw column k, transform of column k - 1
            rd_col(k, d[k & 1]);
            w4_bt3<A>(d[(k - 1) & 1], t[0][k - 1], t[1][k - 1], t[2][k - 1]);
            w4_pin(t[0][k - 1]); w4_pin(t[1][k - 1]); w4_pin(t[2][k - 1]);
        } else if (k == 5) {
            w4_bt3<A>(d[0], t[0][4], t[1][4], t[2][4]);
            w4_pin(t[0][4]); w4_pin(t[1][4]); w4_pin(t[2][4]);
#pragma unroll
            for (int f = 0; f < 3; ++f) nxt.u[f] = *reinterpret_cast<const float4 *>(ns + au + f * 1024);
        } else if (k <= 8) {                                 // k = 6, 7, 8: column transform of row k - 6, U fragments 3 (k - 5) .. (k = 8: none left)
            w4_bt3<B>(t[k - 6], nxt.v[3 * (k - 6)], nxt.v[3 * (k - 6) + 1], nxt.v[3 * (k - 6) + 2]);
            w4_pin(nxt.v[3 * (k - 6)]); w4_pin(nxt.v[3 * (k - 6) + 1]); w4_pin(nxt.v[3 * (k - 6) + 2]);
            if (k < 8) {
#pragma unroll
                for (int f = 0; f < 3; ++f) nxt.u[3 * (k - 5) + f] = *reinterpret_cast<const float4 *>(ns + au + (3 * (k - 5) + f) * 1024);
            }
        }
        // inside a piece: one MFMA, a third of the piece's vector work, ... -- issued as a block the 24 transform operations leave the matrix pipe idle
        if (k >= 1 && k <= 8) {
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
            }
        }
        // pin the piece behind its three MFMAs: left alone, the scheduler sinks the whole transform below the step's barrier (its results are
        // only needed one step later), where nothing overlaps it
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int A, int B>
__device__ __forceinline__ void w4_loop(f32x16 (&acc)[9], const char *smem_c, unsigned lds0, int wave, int lane, const unsigned (&vraw)[5],
                                        i32x4 srd_src, i32x4 srd_u, unsigned soff_u0, int ks_begin, int ks_end, unsigned long long *stamp_t)
{
    const int r = lane & 31, q = lane >> 5, ty = r >> 3, tx = r & 7;
    unsigned lp[4];
#pragma unroll
    for (int cls = 0; cls < 4; ++cls) {
        const int hxn = (cls & 1) ? 8 : 9;
        lp[cls] = (unsigned)(16 * ((ty * hxn + tx) * 2 + q));
    }
    const unsigned au = (unsigned)(wave * 9216 + lane * 16);                 // this wave's nine fragments inside a slot's U region
    const unsigned vu[3] = {(unsigned)(lane * 16), (unsigned)(lane * 16 + 1024), (unsigned)(lane * 16 + 2048)};
    const unsigned lds_u = lds0 + (unsigned)(wave * 9216);
    const unsigned lds_r = lds0 + (unsigned)(kW4UStage + wave * 5 * 1024);   // this wave's five raw pieces
    auto fetch = [&](int ks, int slot) {
        const unsigned va[2] = {vraw[0], vraw[1]}, vb[3] = {vraw[2], vraw[3], vraw[4]};
        dma16_group<2, 1024>(lds_r + (unsigned)(slot * kW4Slot), va, srd_src, ks * 32);
        dma16_group<3, 1024>(lds_r + (unsigned)(slot * kW4Slot) + 2048u, vb, srd_src, ks * 32);
#pragma unroll
        for (int g = 0; g < 3; ++g)
            dma16_group<3, 1024>(lds_u + (unsigned)(slot * kW4Slot + g * 3072), vu, srd_u, (int)(soff_u0 + (unsigned)ks * 9216u) + g * 3072);
    };
    const int n = ks_end - ks_begin;
    if (n <= 0) return;
    // step 0 alone first: ISSUING a step's 14 pieces takes a wave ~1 500 cycles (the raw pieces are 64 scattered 16-byte requests each), so step 1 goes
    // out only once step 0 has landed -- its flight then rides under the first operands' reads and transform
    fetch(ks_begin, 0);
    dma_wait<0>();
    __syncthreads();
    W4STAMP(2);
    if (n > 1) fetch(ks_begin + 1, 1);
    W4Ops o0, o1;
    {   // operands of the first step: nothing to hide this behind
        const char *ns = smem_c;
        float4 t[3][5];
#pragma unroll
        for (int x = 0; x < 5; ++x) {
            float4 d[5];
#pragma unroll
            for (int y = 0; y < 5; ++y) d[y] = *reinterpret_cast<const float4 *>(ns + lp[w4_cls(A + y, B + x)] + (kW4UStage + w4_imm(A + y, B + x)));
            w4_bt3<A>(d, t[0][x], t[1][x], t[2][x]);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) w4_bt3<B>(t[i], o0.v[3 * i], o0.v[3 * i + 1], o0.v[3 * i + 2]);
#pragma unroll
        for (int f = 0; f < 9; ++f) o0.u[f] = *reinterpret_cast<const float4 *>(ns + au + f * 1024);
    }
    dma_wait<0>();                                    // step 1 has landed meanwhile (this wave's pieces; the barrier covers the others')
    __syncthreads();                                  // every wave holds step 0 in registers: slot 0 may be overwritten
    W4STAMP(3);
    for (int t = 0; t < n; t += 2) {
        // step t multiplies from o0; slot 1 (step t + 1) is read into o1; step t + 2 lands in slot 0 (free: consumed into o0 one step ago)
        w4_step<A, B>(acc, o0, o1, smem_c, 1, lp, au, t + 2 < n, ks_begin + t + 2, 0, lds_u, lds_r, vraw, vu, srd_src, srd_u, soff_u0);
        dma_wait<0>();
        __syncthreads();
#ifdef LSPF2F_WINO_STAMPS
        if (t == 0) W4STAMP(4);
        if (t == 2) W4STAMP(5);
#endif
        if (t + 1 < n) {
            w4_step<A, B>(acc, o1, o0, smem_c, 0, lp, au, t + 3 < n, ks_begin + t + 3, 1, lds_u, lds_r, vraw, vu, srd_src, srd_u, soff_u0);
            dma_wait<0>();
            __syncthreads();
        }
    }
}

// A^T applied to six values: (m0 + s1 + s2, d1 + 2 d2, s1 + 4 s2, d1 + 8 d2 + m5) with s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4
__device__ __forceinline__ void w4_at(const float4 (&m)[6], float4 &o0, float4 &o1, float4 &o2, float4 &o3)
{
    const float4 s1 = f4add(m[1], m[2]), d1 = f4sub(m[1], m[2]), s2 = f4add(m[3], m[4]), d2 = f4sub(m[3], m[4]);
    o0 = f4add(f4add(m[0], s1), s2);
    o1 = f4fma(2.f, d2, d1);
    o2 = f4fma(4.f, s2, s1);
    o3 = f4add(f4fma(8.f, d2, d1), m[5]);
}

__global__ __launch_bounds__(256, 1) void wino4_3x3(const WinoParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef __attribute__((address_space(3))) float lds_float;
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_float *)smem;
    const char *smem_c = reinterpret_cast<const char *>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef LSPF2F_WINO_STAMPS
    unsigned long long stamp_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#else
    unsigned long long *stamp_t = nullptr;
#endif
    W4STAMP(0);
    asm volatile("" :: "s"(p.src), "s"(p.u), "s"(p.scale), "s"(p.shift), "s"(p.residual), "s"(p.out), "s"(p.partial), "s"(p.tile_cnt));
    asm volatile("" :: "s"(p.B), "s"(p.H), "s"(p.W), "s"(p.C), "s"(p.N), "s"(p.relu), "s"(p.splits), "s"(p.steps_per_split), "s"(p.ntb), "s"(p.nng),
                       "s"(p.tbx), "s"(p.nmajor), "s"(p.xcd), "s"(p.div_plane.m), "s"(p.div_plane.s1), "s"(p.div_plane.s2), "s"(p.div_fast.m),
                       "s"(p.div_fast.s1), "s"(p.div_fast.s2), "s"(p.div_tbf.m), "s"(p.div_tbf.s1), "s"(p.div_tbf.s2), "s"(p.div_tbx.m), "s"(p.div_tbx.s1),
                       "s"(p.div_tbx.s2));

    // block -> (split z, tile-block tb, channel group ng), XCD-aware like wino3x3
    unsigned lin = blockIdx.x;
    if (p.xcd) {
        const unsigned total = gridDim.x, qq = total >> 3, rr = total & 7, x = lin & 7;
        lin = x * qq + (x < rr ? x : rr) + (lin >> 3);
    }
    const int z = (int)p.div_plane.div(lin);
    const unsigned rem = lin - (unsigned)z * (unsigned)(p.ntb * p.nng);
    int tb, ng;
    if (p.nmajor) { ng = (int)p.div_fast.div(rem); tb = (int)rem - ng * p.ntb; }
    else { tb = (int)p.div_fast.div(rem); ng = (int)rem - tb * p.nng; }
    const int b = (int)p.div_tbf.div((unsigned)tb);
    const int tbi = tb - b * (p.tby * p.tbx);
    const int by = (int)p.div_tbx.div((unsigned)tbi), bx = tbi - by * p.tbx;
    const int Y0 = by * 16, X0 = bx * 32;
    const int n0 = ng * 32;
    const int S = p.C >> 3;
    const int ks_begin = z * p.steps_per_split;
    int ks_end = ks_begin + p.steps_per_split;
    if (ks_end > S) ks_end = S;

    // raw-patch DMA: this wave's five pieces; lane -> chunk -> (patch pixel, channel quad) -> byte offset in the NHWC source or out of range
    unsigned vraw[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int ci = (wave * 5 + k) * 64 + lane;
        const int pary = ci >= 952 ? 3 : ci >= 680 ? 2 : ci >= 340 ? 1 : 0;
        const int rem1 = ci - (pary == 3 ? 952 : pary == 2 ? 680 : pary == 1 ? 340 : 0);
        const int hyn = pary < 2 ? 5 : 4;
        const int parx = rem1 >= 52 * hyn ? 3 : rem1 >= 36 * hyn ? 2 : rem1 >= 18 * hyn ? 1 : 0;
        const int rem2 = rem1 - (parx == 3 ? 52 : parx == 2 ? 36 : parx == 1 ? 18 : 0) * hyn;
        const int hxn = parx < 2 ? 9 : 8;
        const int qd = rem2 & 1, r3 = rem2 >> 1;
        const int hy = parx < 2 ? r3 / 9 : r3 >> 3, hx = r3 - hy * hxn;
        const int y = Y0 - 1 + 4 * hy + pary, x = X0 - 1 + 4 * hx + parx;
        const bool ok = ci < 1224 && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        vraw[k] = ok ? ((unsigned)((b * p.H + y) * p.W + x) * (unsigned)p.C + (unsigned)(qd * 4)) * 4u : kOOB4;
#ifdef W4_ABL_LINEAR_RAW      // ablation (tools/sessions/wino4_ablate_job.sh; results are garbage): every raw piece reads 1 KB of CONTIGUOUS memory instead of 64 scattered 16-B chunks
        vraw[k] = (unsigned)(((blockIdx.x * 20 + wave * 5 + k) * 1024 + lane * 16) % (p.H * p.W * p.C * 4 - 4096));
#endif
    }
    const i32x4 srd_src = make_srd(p.src, (unsigned)(p.B * p.H * p.W) * (unsigned)p.C * 4u);
    const i32x4 srd_u = make_srd(p.u, 36u * (unsigned)p.C * (unsigned)p.N * 4u);
    // U fragments: [n-block][wave][k-step][f 9][64 lanes][4]
    const unsigned soff_u0 = (unsigned)((n0 >> 5) * 4 + wave) * (unsigned)S * 9216u;

    W4STAMP(1);
    f32x16 acc[9];
#pragma unroll
    for (int f = 0; f < 9; ++f)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[f][e] = 0.f;

    switch (wave) {
    case 0: w4_loop<0, 0>(acc, smem_c, lds0, wave, lane, vraw, srd_src, srd_u, soff_u0, ks_begin, ks_end, stamp_t); break;
    case 1: w4_loop<0, 1>(acc, smem_c, lds0, wave, lane, vraw, srd_src, srd_u, soff_u0, ks_begin, ks_end, stamp_t); break;
    case 2: w4_loop<1, 0>(acc, smem_c, lds0, wave, lane, vraw, srd_src, srd_u, soff_u0, ks_begin, ks_end, stamp_t); break;
    default: w4_loop<1, 1>(acc, smem_c, lds0, wave, lane, vraw, srd_src, srd_u, soff_u0, ks_begin, ks_end, stamp_t); break;
    }
    // (the loop ends on a barrier: every wave is done with the ring slots, the patch below may overwrite them)
    W4STAMP(6);

    // ---- epilogue.  This thread's item: tile (tid >> 3) x channels 4 (tid & 7) .. +3.  Its folded-BN scale / shift and residual pixels are requested
    // first: they land while the accumulators cross LDS.
    const int tile = tid >> 3, cq = (tid & 7) * 4;
    const int oy = Y0 + 4 * (tile >> 3), ox = X0 + 4 * (tile & 7);
    const int n = n0 + cq;
    const bool single = p.splits == 1;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 rv[16];
    if (p.scale) {
        sc = *reinterpret_cast<const float4 *>(p.scale + n);
        sh = *reinterpret_cast<const float4 *>(p.shift + n);
    }
#pragma unroll
    for (int ij = 0; ij < 16; ++ij) {
        rv[ij] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (single && p.residual)
            rv[ij] = *reinterpret_cast<const float4 *>(p.residual + (((size_t)b * p.H + (size_t)(oy + (ij >> 2))) * p.W + (size_t)(ox + (ij & 3))) * p.N + n);
    }
    // accumulators -> patch [position (3a + i) * 6 + 3b + j][tile][channel].  C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
    {
        const int pa = wave >> 1, pb = wave & 1;
        const int ccol = lane & 31, crow = 4 * (lane >> 5);
#pragma unroll
        for (int f = 0; f < 9; ++f) {
            float *pz = smem + ((3 * pa + f / 3) * 6 + 3 * pb + f % 3) * 1024 + ccol;
#pragma unroll
            for (int e = 0; e < 16; ++e) pz[((e & 3) + 8 * (e >> 2) + crow) * 32] = acc[f][e];
        }
    }
    __syncthreads();
    W4STAMP(7);
    // Y = A^T M A for this thread's tile and channel quad: rows first (per column of M), then columns
    float4 zc[4][6];
    {
        const float *pm = smem + tile * 32 + cq;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            float4 m[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) m[r] = *reinterpret_cast<const float4 *>(pm + (r * 6 + c) * 1024);
            w4_at(m, zc[0][c], zc[1][c], zc[2][c], zc[3][c]);
        }
    }
    const __amdgpu_buffer_rsrc_t slab_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.partial, 0, p.splits > 1 ? (int)p.slab_bytes : 0, 0x00020000);
    const size_t npix = (size_t)p.B * p.H * p.W;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float4 y[4];
        w4_at(zc[i], y[0], y[1], y[2], y[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float4 v = y[j];
            const size_t pix = ((size_t)b * p.H + (size_t)(oy + i)) * p.W + (size_t)(ox + j);
            const size_t e = pix * p.N + n;
            if (!single) {
                // partial sums of this K slice; published write-through to whichever workgroup arrives last at the tile
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), slab_rsrc, (unsigned)(((size_t)z * npix * p.N + e) * 4), 0, 16);
            } else {
                v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                const float4 r4 = rv[i * 4 + j];
                v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
                if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *reinterpret_cast<float4 *>(p.out + e) = v;
            }
        }
    }
#ifdef LSPF2F_WINO_STAMPS
    __builtin_amdgcn_sched_barrier(0); stamp_t[1] = __builtin_amdgcn_s_memtime() - stamp_t[1]; __builtin_amdgcn_sched_barrier(0);    // slot 1: entry-of-loop .. stores issued, as a DURATION
#endif
    W4STAMP_FLUSH;
    if (single) return;

    // ---- split-K combine inside the launch (wino3x3's protocol): write-through slabs above -> every wave drains its stores -> barrier -> one relaxed
    // agent-scope ticket per tile; the last arriver sums the slabs in z order (bit-reproducible) with loads that bypass its L1 and runs the epilogue.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // the last arriver's residual pixels are requested by every slice BEFORE its ticket (in flight during the ticket's round trip; the others drop them)
#pragma unroll
    for (int ij = 0; ij < 16; ++ij)
        if (p.residual)
            rv[ij] = *reinterpret_cast<const float4 *>(p.residual + (((size_t)b * p.H + (size_t)(oy + (ij >> 2))) * p.W + (size_t)(ox + (ij & 3))) * p.N + n);
    unsigned *flag = reinterpret_cast<unsigned *>(smem);
    const unsigned tcnt = (unsigned)(tb * p.nng + ng);
    if (tid == 0) flag[0] = __hip_atomic_fetch_add(p.tile_cnt + tcnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (flag[0] != (unsigned)p.splits - 1u) return;
    if (tid == 0) __hip_atomic_store(p.tile_cnt + tcnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float4 tsl[4][8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {                         // one output row of the tile: all its slab loads in flight at once
            const size_t pix = ((size_t)b * p.H + (size_t)(oy + i)) * p.W + (size_t)(ox + j);
            const size_t e = pix * p.N + n;
#pragma unroll
            for (int sl = 0; sl < 8; ++sl)
                if (sl < p.splits)
                    tsl[j][sl] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(slab_rsrc, (unsigned)(((size_t)sl * npix * p.N + e) * 4), 0, 16));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t pix = ((size_t)b * p.H + (size_t)(oy + i)) * p.W + (size_t)(ox + j);
            const size_t e = pix * p.N + n;
            float4 v = tsl[j][0];
#pragma unroll
            for (int sl = 1; sl < 8; ++sl)
                if (sl < p.splits) { v.x += tsl[j][sl].x; v.y += tsl[j][sl].y; v.z += tsl[j][sl].z; v.w += tsl[j][sl].w; }
            v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
            const float4 r4 = rv[i * 4 + j];
            v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
            if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4 *>(p.out + e) = v;
        }
    }
}

bool wino4_supported(const WinoParams &p)
{
    if (p.B < 1 || p.H % 16 || p.W % 32 || p.C % 8 || p.C < 8 || p.N % 32) return false;
    const size_t lim = 0x7fffffffull;                      // 32-bit buffer offsets with the top bit reserved as the out-of-range marker
    if ((size_t)p.B * p.H * p.W * p.C * 4 > lim || (size_t)36 * p.C * p.N * 4 > lim) return false;
    if (p.splits < 1 || p.splits > 8) return false;
    if (p.splits > 1 && (!p.partial || !p.tile_cnt || (size_t)p.splits * p.B * p.H * p.W * p.N * 4 > lim)) return false;
    return true;
}

hipError_t launch_wino4(const WinoParams &p_in, hipStream_t s)
{
    if (!wino4_supported(p_in)) return hipErrorInvalidValue;
    WinoParams p = p_in;
    const int S = p.C / 8;
    p.steps_per_split = (S + p.splits - 1) / p.splits;
    if ((p.splits - 1) * p.steps_per_split >= S) return hipErrorInvalidValue;       // an empty split: planner bug
    p.tby = p.H / 16; p.tbx = p.W / 32;
    p.ntb = p.B * p.tby * p.tbx;
    p.nng = p.N / 32;
    if (p.splits > 1) p.slab_bytes = (size_t)p.splits * p.B * p.H * p.W * p.N * 4;
    // weight-heavy layers: an XCD keeps a slice of U in its L2 (channel groups slowest); activation-heavy: a band of tile-blocks
    const size_t act = (size_t)p.B * p.H * p.W * p.C * 4, wgt = (size_t)36 * p.C * p.N * 4;
    p.nmajor = wgt > act ? 1 : 0;
    p.xcd = 1;
    p.nopre = 0;
    p.div_plane = FastDiv::make((unsigned)(p.ntb * p.nng));
    p.div_fast = FastDiv::make((unsigned)(p.nmajor ? p.ntb : p.nng));
    p.div_tbf = FastDiv::make((unsigned)(p.tby * p.tbx));
    p.div_tbx = FastDiv::make((unsigned)p.tbx);
    static AttrMask attr_mask;
    if (attr_needed_on_this_device(attr_mask)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wino4_3x3), hipFuncAttributeMaxDynamicSharedMemorySize, kW4Patch);
        if (e != hipSuccess) return e;
        attr_done_on_this_device(attr_mask);
    }
    hipLaunchKernelGGL(wino4_3x3, dim3((unsigned)(p.ntb * p.nng * p.splits)), dim3(256), kW4Patch, s, p);
    return hipGetLastError();
}

// Host: OIHW [N][C][3][3] -> U = G g G^T (double, rounded once; G of the points 0, +-1, +-2, inf) in the order the waves read it:
// [n-block N/32][wave (a, b) 4][k-step C/8][f = 3 i + j, 9][lane 64][4] with position (3a + i, 3b + j); lane l holds output channel
// 32 nblock + (l & 31), input channels 8 s + 4 (l >> 5) + 0..3 -- one 1-KB piece = the B operand of 4 MFMAs.
void pack_wino4_weights(const float *oihw, int cin, int cout, float *out)
{
    static const double G[6][3] = {{1.0 / 4, 0.0, 0.0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
    const int S = cin / 8;
    for (int n = 0; n < cout; ++n)
        for (int c = 0; c < cin; ++c) {
            const float *g = oihw + ((size_t)n * cin + c) * 9;
            double tmp[6][3];
            for (int i = 0; i < 6; ++i)
                for (int b = 0; b < 3; ++b) tmp[i][b] = G[i][0] * (double)g[b] + G[i][1] * (double)g[3 + b] + G[i][2] * (double)g[6 + b];
            const int nblk = n >> 5, s = c >> 3, lane = (n & 31) + 32 * ((c & 7) >> 2), t = c & 3;
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 6; ++j) {
                    const double u = tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2];
                    const int wave = (i / 3) * 2 + j / 3, f = (i % 3) * 3 + j % 3;
                    out[(((((size_t)nblk * 4 + wave) * S + s) * 9 + f) * 64 + lane) * 4 + t] = (float)u;
                }
        }
}

}  // namespace lspf2f
