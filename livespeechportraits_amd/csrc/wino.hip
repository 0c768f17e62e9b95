// gfx950 (MI355X / CDNA4): stride-1 3x3 convolution as Winograd F(2x2, 3x3) on the fp32 matrix cores -- 16 GEMMs of
// [tiles x Cin] . [Cin x Cout] (one per position xi of the 4x4 transformed tile) instead of one GEMM with K = 9 Cin: 16 multiplies per
// 4 outputs instead of 36, i.e. 4/9 of the matrix FLOPs of the implicit GEMM (igemm.hip).  fp32 in, fp32 accumulate, BatchNorm-folded
// scale / shift, residual add and ReLU in the epilogue like every other conv of this library.  See DESIGN.md section 4.8.
//
// Reference semantics: the 3x3 / stride 1 / pad 1 / bias-free Conv2d calls of ResidualBlock, models/networks.py:650-675 (:663, :666),
// followed by BatchNorm2d in eval mode, the residual add and ReLU (:670-675).
//
//   Y = A^T [ sum_c (G g G^T) . (B^T d B) ] A     B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   A^T = [1 1 1 0; 0 1 -1 -1]
//
// Workgroup = 4 x 8 Winograd tiles (8 x 16 output pixels of one frame) x 32*NB output channels, 4 waves.  Wave i owns row i of the
// transformed tile (xi = 4i .. 4i+3): it needs rows (ra, rb) of the raw 4x4 patch only, computes its four V values in registers straight
// from the raw patch in LDS (8 ds_read_b128 + 32 adds per 8 channels: the transformed input never exists in memory), and keeps 4 x NB
// accumulators of 32 tiles x 32 channels (v_mfma_f32_32x32x2_f32).  The weights arrive pre-transformed (U = G g G^T, host packer, double)
// in the order the MFMA wants them: one 1-KB LDS-DMA piece = the B fragment of 4 MFMAs; a wave copies exactly the fragments it alone uses.
// Per 8 input channels a wave issues 16 NB MFMAs for 8 + 4 NB fragment reads.  The four waves meet once, in the epilogue: the column half of
// the output transform in registers, the row half across waves through an LDS patch.
// Round 4, one channel block per wave (the form batch-1 plans take): the U fragments skip LDS altogether -- one plain buffer_load_dwordx4 per fragment into the
// registers the MFMA reads, two K-steps ahead (wino_loop_ur below; the K loop then runs at 96 % of MFMA issue).  InstanceNorm plans: the epilogue also leaves the
// per-(frame, channel) sums of its 128 pixels (wino_stats).
#include "device_common.h"
#include "kernels.h"
#include "wino_common.h"

namespace lspf2f {

// phase timestamps (tools/wino_stamps.py; builds with -DLSPF2F_WINO_STAMPS only): s_memtime values kept in registers, written once at the end
#ifdef LSPF2F_WINO_STAMPS
#define WSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); stamp_t[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define WSTAMP_DECL unsigned long long stamp_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}
// slot 6 (the split-K ticket stamp) carries, when it is unused, where the wave ran: bit 63 | HW_REG_XCC_ID << 32 | HW_REG_HW_ID (wave 3:0, simd 5:4, cu 11:8, sh 12, se 15:13)
#define WSTAMP_FLUSH do { if (p.stamps && lane == 0) { unsigned long long *q_ = p.stamps + ((size_t)blockIdx.x * 4 + wave) * 8; \
    if (stamp_t[6] == 0) stamp_t[6] = (1ull << 63) | ((unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf) << 32) | (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4); \
    for (int i_ = 0; i_ < 8; ++i_) q_[i_] = stamp_t[i_]; } } while (0)
#else
#define WSTAMP(i) do {} while (0)
#define WSTAMP_DECL do {} while (0)
#define WSTAMP_FLUSH do {} while (0)
#endif

static constexpr unsigned kOOBw = 0x80000000u;   // voffset beyond any num_records: the LDS-DMA lands zeros (image border, unused chunks)

static constexpr int kRawPieces = 6;                       // 360 chunks of 16 B (10 x 18 pixels x 2 channel quads) in 6 pieces of 64
static constexpr int kRawStage = kRawPieces * 1024;        // bytes per ring slot
static constexpr int kWinoStatsScratch = (8 + 4 * 8 * 2) * 16;   // InstanceNorm plans (wino_stats below)

// LDS: [ns ring slots of U fragments][ns ring slots of raw patch][1 KB dump]; ns = ring depth (2, or 3 with one channel block per wave:
// its K-step is 1024 cycles of MFMA per wave, less than a loaded L2 round trip, so the copy of step t + 2 is in flight while step t multiplies)
__host__ __device__ constexpr int wino_u_stage(int nb) { return 4 * 4 * nb * 1024; }              // 4 waves x (4 j x nb) pieces
__host__ __device__ constexpr int wino_raw_base(int nb, int ns) { return ns * wino_u_stage(nb); }
__host__ __device__ constexpr int wino_dump(int nb, int ns) { return wino_raw_base(nb, ns) + ns * kRawStage; }
__host__ __device__ constexpr int wino_lds_bytes(int nb, int ns, bool ur = false)
{
    const int loop = (ur ? ns * kRawStage : wino_dump(nb, ns)) + 1024;     // ur: the raw ring alone (U fragments go to registers)
    const int patch = 4 * 2 * nb * 32 * 36 * 4;             // epilogue: [wave][b][nb][32 tiles][36]
    return (loop > patch ? loop : patch) + kWinoStatsScratch;
}
// behind the patch: the InstanceNorm statistics of a workgroup meet here -- the shift [8 quads] and the waves' partial sums [4][8][2], float4 each
__host__ __device__ constexpr int wino_stats_base(int nb, int ns, bool ur = false) { return wino_lds_bytes(nb, ns, ur) - kWinoStatsScratch; }

// The copies of one wave: two raw-patch pieces per K-step by LDS-DMA, and its own U fragments -- by LDS-DMA into its slice of the U ring, or (UR form,
// one channel block per wave) by plain loads into registers.  Built once in the kernel: the first step(s) are requested BEFORE the epilogue operands'
// address arithmetic, which then runs under their latency instead of ahead of it.
template <int NB, int NS, bool UR, bool SC1 = false>      // SC1: the raw patch is read past the L1 (wino3x3_chain: other workgroups of the launch wrote it)
struct WinoCopy {
    static constexpr int USTAGE = wino_u_stage(NB);
    static constexpr int RAWB = UR ? 0 : wino_raw_base(NB, NS);
    static constexpr int DUMP = UR ? NS * kRawStage : wino_dump(NB, NS);
    static constexpr int PIECES = 2 + 4 * NB;                               // loads of either kind this wave issues per step
    unsigned lds_u, lds_r0, lds_r1, vraw0, vraw1, soff_u0, soff_nb, vlane;
    i32x4 srd_src, srd_u;
    int wave;
    __device__ __forceinline__ void init(unsigned lds0, int wave_, int lane, unsigned vr0, unsigned vr1, i32x4 ssrc, i32x4 su, unsigned so0, unsigned sonb)
    {
        wave = wave_; vraw0 = vr0; vraw1 = vr1; srd_src = ssrc; srd_u = su; soff_u0 = so0; soff_nb = sonb;
        vlane = (unsigned)(lane * 16);
        lds_u = lds0 + (unsigned)(wave * (4 * NB * 1024));
        lds_r0 = lds0 + (unsigned)(RAWB + wave * 1024);
        lds_r1 = wave < 2 ? lds_r0 + 4096u : lds0 + (unsigned)DUMP;      // pieces 4, 5 exist for waves 0, 1 only; the others feed the dump slot
    }
    __device__ __forceinline__ void raw(int ks, int slot) const
    {
        if constexpr (SC1) dma16_two_sc1(lds_r0 + (unsigned)(slot * kRawStage), wave < 2 ? lds_r1 + (unsigned)(slot * kRawStage) : lds_r1, vraw0, vraw1, srd_src, ks * 32);
        else dma16_two(lds_r0 + (unsigned)(slot * kRawStage), wave < 2 ? lds_r1 + (unsigned)(slot * kRawStage) : lds_r1, vraw0, vraw1, srd_src, ks * 32);
    }
    // the wave's four j fragments of one channel block: consecutive 1-KB pieces in memory and in LDS (dma16_group advances M0; the per-piece
    // voffsets are registers rather than instruction offsets, which the LDS-DMA form would also add to the LDS address)
    __device__ __forceinline__ void u2(int ks, int slot, int nb, int h) const        // two of them: half h (0 | 1) of channel block nb
    {
        const unsigned vv[2] = {vlane + (unsigned)(2 * h) * 1024u, vlane + (unsigned)(2 * h + 1) * 1024u};
        dma16_group<2, 1024>(lds_u + (unsigned)(slot * USTAGE + nb * 4096 + h * 2048), vv, srd_u, (int)(soff_u0 + (unsigned)nb * soff_nb + (unsigned)ks * 4096u));
    }
    __device__ __forceinline__ void step(int ks, int slot) const                     // a whole step through LDS
    {
        raw(ks, slot);
        const unsigned vu[4] = {vlane, vlane + 1024u, vlane + 2048u, vlane + 3072u};
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            dma16_group<4, 1024>(lds_u + (unsigned)(slot * USTAGE + nb * 4096), vu, srd_u, (int)(soff_u0 + (unsigned)nb * soff_nb + (unsigned)ks * 4096u));
    }
    template <int H>
    __device__ __forceinline__ void ureg2(f32x4v (&dst)[4], int ks) const            // UR: half H of the step's four fragments into registers
    {
        const unsigned so = soff_u0 + (unsigned)ks * 4096u;
        uload16<2048 * H>(dst[2 * H], vlane, srd_u, so); uload16<2048 * H + 1024>(dst[2 * H + 1], vlane, srd_u, so);
    }
};

// The K loop of one wave.  ROW = its row i of B^T d B: t = d[ra] (+|-) d[rb] per patch column, then the column transform.
//   i = 0: d0 - d2     i = 1: d1 + d2     i = 2: d2 - d1     i = 3: d1 - d3
template <int NB, int NS, bool IL, bool ROT, int ROW, int EPL, class EpiLoads>
__device__ __forceinline__ void wino_loop(const WinoCopy<NB, NS, false> &cp, const EpiLoads &epi_loads, f32x16 (&acc)[4][NB], const char *smem_c, int wave, int lane,
                                          int ks_begin, int ks_end, unsigned long long *first_landed, int prio_mode)
{
    constexpr int RA = ROW == 0 ? 0 : 1, RB = ROW == 3 ? 3 : 2;
    constexpr int USTAGE = wino_u_stage(NB), RAWB = wino_raw_base(NB, NS);
    constexpr int PIECES = 2 + 4 * NB;                                      // LDS-DMA instructions this wave issues per step
    // fragment-read addresses: lane (tile r = l & 31 -> ty = r >> 3, tx = r & 7; channel quad q = l >> 5) reads patch pixel
    // (2 ty + dy, 2 tx + dx); chunk = ((pary * 2 + parx) * 2 + q) * 45 + hy * 9 + hx with (py, px) = (2 hy + pary, 2 hx + parx)
    const int r = lane & 31, q = lane >> 5, ty = r >> 3, tx = r & 7;
    unsigned araw[2][4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int dy = k == 0 ? RA : RB;
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            const int chunk = (((dy & 1) * 2 + (dx & 1)) * 2 + q) * 45 + (ty + (dy >> 1)) * 9 + tx + (dx >> 1);
            araw[k][dx] = (unsigned)(RAWB + chunk * 16);
        }
    }
    const unsigned au = (unsigned)(wave * (4 * NB * 1024) + lane * 16);     // this wave's fragments inside a U ring slot
    auto fetch_raw = [&](int ks, int slot) { cp.raw(ks, slot); };
    auto fetch_u2 = [&](int ks, int slot, int nb, int h) { cp.u2(ks, slot, nb, h); };
    auto fetch = [&](int ks, int slot) { cp.step(ks, slot); };

    // step 0 (and step 1 with a ring of 3) requested first; behind them the kernel's EPL epilogue-operand loads, whose address arithmetic then runs
    // under the copies' latency instead of ahead of it (loads return in order: the first wait allows for exactly those younger ones)
    const int nsteps = ks_end - ks_begin;
    fetch(ks_begin, 0);
    if (NS > 2 && nsteps > 1) fetch(ks_begin + 1, 1);
    epi_loads();
    if (NS > 2 && nsteps > 1) dma_wait<PIECES + EPL>(); else dma_wait<EPL>();
    __syncthreads();
#ifdef LSPF2F_WINO_STAMPS
    *first_landed = __builtin_amdgcn_s_memtime();
#endif
    int cur = 0;
    const ProgressPrio prio(prio_mode, nsteps);
    for (int t = 0; t < nsteps; ++t) {
        prio.step(t);
        const char *rawp = smem_c + cur * kRawStage;
        const char *up = smem_c + cur * USTAGE + au;
        // raw rows of this step and the first weight fragment: issued straight behind the barrier, their latency rides under the DMA issue below
        float4 d[2][4];
#ifndef WINO_ABL_NOHEAD
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) d[k][dx] = *reinterpret_cast<const float4 *>(rawp + araw[k][dx]);
#endif
        float4 u[2];
        u[0] = *reinterpret_cast<const float4 *>(up);
#ifdef WINO_ABL_NOHEAD       // ablation (tools/sessions/wino_ablate_job.sh): no raw reads / transform -- whatever the registers hold is multiplied
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) asm volatile("" : "+v"(d[k][dx].x), "+v"(d[k][dx].y), "+v"(d[k][dx].z), "+v"(d[k][dx].w));
#endif
        // ring slot (cur + NS - 1) % NS was last read in step t - 1; every wave has passed the barrier that ended it
        const int ahead = t + NS - 1;
#ifdef WINO_ABL_NODMA
        const bool issue = false;
#else
        const bool issue = ahead < nsteps;
#endif
        int slot = cur + NS - 1; if (slot >= NS) slot -= NS;
        if (!IL && issue) fetch(ks_begin + ahead, slot);
        float4 tt[4], v[4];
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            if constexpr (ROW == 0 || ROW == 3) tt[dx] = f4sub(d[0][dx], d[1][dx]);
            else if constexpr (ROW == 1) tt[dx] = f4add(d[0][dx], d[1][dx]);
            else tt[dx] = f4sub(d[1][dx], d[0][dx]);
        }
        v[0] = f4sub(tt[0], tt[2]); v[1] = f4add(tt[1], tt[2]); v[2] = f4sub(tt[2], tt[1]); v[3] = f4sub(tt[1], tt[3]);
        if constexpr (ROT) {
            // consecutive MFMAs on DIFFERENT accumulators: the four j of a channel block rotate, so no instruction waits for its predecessor's result
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                float4 uj[4];
                uj[0] = nb == 0 ? u[0] : *reinterpret_cast<const float4 *>(up + (nb * 4) * 1024);
#pragma unroll
                for (int j = 1; j < 4; ++j) uj[j] = *reinterpret_cast<const float4 *>(up + (nb * 4 + j) * 1024);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float a = c == 0 ? v[j].x : c == 1 ? v[j].y : c == 2 ? v[j].z : v[j].w;
                        const float b = c == 0 ? uj[j].x : c == 1 ? uj[j].y : c == 2 ? uj[j].z : uj[j].w;
                        acc[j][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j][nb], 0, 0, 0);
                    }
                    const int f = nb * 4 + c;
                    if (IL && issue) {
                        if (f == 0) fetch_raw(ks_begin + ahead, slot);
                        if (f >= 1 && f <= 2 * NB) fetch_u2(ks_begin + ahead, slot, (f - 1) >> 1, (f - 1) & 1);
                    }
                }
            }
        } else {
#pragma unroll
        for (int f = 0; f < 4 * NB; ++f) {                           // fragment f = nb * 4 + j, the order the pieces sit in LDS
            const int nb = f >> 2, j = f & 3;
            if (f + 1 < 4 * NB) u[(f + 1) & 1] = *reinterpret_cast<const float4 *>(up + (f + 1) * 1024);
            const float4 uu = u[f & 1];
            acc[j][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j].x, uu.x, acc[j][nb], 0, 0, 0);
            acc[j][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j].y, uu.y, acc[j][nb], 0, 0, 0);
            acc[j][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j].z, uu.z, acc[j][nb], 0, 0, 0);
            acc[j][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j].w, uu.w, acc[j][nb], 0, 0, 0);
            // IL: the copies of step t + NS - 1 go out in small groups BETWEEN this step's MFMA groups (each group of 4 MFMAs keeps the matrix pipe busy for
            // 256 cycles, a group of two pieces takes ~150 to issue), instead of as one ~600-cycle block ahead of them with the pipe idle
            if (IL && issue) {
                if (f == 0) fetch_raw(ks_begin + ahead, slot);
                if (f >= 1 && f <= 2 * NB) fetch_u2(ks_begin + ahead, slot, (f - 1) >> 1, (f - 1) & 1);
            }
        }
        }
        // step t + 1 must have landed (this wave's pieces; the barrier covers the other waves'): everything but the pieces issued in THIS iteration
        if (NS > 2 && issue) dma_wait<PIECES>(); else dma_wait<0>();
#ifndef WINO_ABL_NOBARRIER
        __syncthreads();
#endif
        if (++cur == NS) cur = 0;
    }
    prio.done();
}


// ---- UR form of the K loop (one channel block per wave, ring of 3): the wave's U fragments never touch LDS.  They are private to the wave anyway
// (fragment order, one 16-byte load per lane and MFMA quad), so each is ONE buffer_load_dwordx4 into the registers the MFMA reads, two steps ahead,
// instead of an LDS-DMA piece + a ds_read_b128: per step 4 plain loads replace 4 of the 6 LDS-DMA pieces (each of which holds the issuing wave for
// 40-150 cycles) and 4 of the 12 fragment reads, and the workgroup's LDS drops from 67 to 37 KB.  The loads are inline asm like the copies, for the
// same reason (the compiler does not see the copies, so its own vmcnt arithmetic would over-wait): the registers of step t + 2 are "defined" at the
// issue; their first use is an MFMA whose A operand comes from LDS reads behind the barrier that follows the counted wait of step t + 1, so it cannot
// move above their arrival.  What the compiler must NOT do is copy such a register while its load is in flight (a tied "+v" operand on the wait
// makes it do exactly that): tests/test_wino_cpu.py checks the generated code for moves out of the three register sets.

// NS = register sets = ring slots of the raw patch = how far ahead a step's operands are requested (NS - 1 steps): 3 is the shipped form, 4 (tune key
// `wino_ureg=2`, 16 more registers) an A-B arm of round 5 for the waves whose loads take longer than two steps.
// Gate (wino3x3_chain): what stands between a workgroup's weight requests and its first raw-patch request.  NoGate = the plain kernel.
struct NoGate {
    static constexpr bool active = false;
    static constexpr int LPL = 0;
    __device__ __forceinline__ void wait() const {}
    __device__ __forceinline__ void late_loads() const {}
};
template <int ROW, int EPL, int NS, class EpiLoads, bool SC1 = false, class Gate = NoGate>
__device__ __forceinline__ void wino_loop_ur(const WinoCopy<1, NS, true, SC1> &cp, const EpiLoads &epi_loads, f32x16 (&acc)[4][1], const char *smem_c, int lane,
                                             int ks_begin, int ks_end, unsigned long long *first_landed, int prio_mode, const Gate &gate = Gate())
{
    static_assert(NS == 3 || NS == 4, "register sets");
    constexpr int RA = ROW == 0 ? 0 : 1, RB = ROW == 3 ? 3 : 2;
    const int r = lane & 31, q = lane >> 5, ty = r >> 3, tx = r & 7;
    unsigned araw[2][4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int dy = k == 0 ? RA : RB;
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            const int chunk = (((dy & 1) * 2 + (dx & 1)) * 2 + q) * 45 + (ty + (dy >> 1)) * 9 + tx + (dx >> 1);
            araw[k][dx] = (unsigned)(chunk * 16);
        }
    }
    // The three register sets.  Requests, loop and uses stay inside this function (one per wave row): across the kernel's switch hipcc gives the sets
    // other registers per branch and copies them at the branch -- in flight.
    f32x4v ur[NS][4];
#pragma unroll
    for (int a = 0; a < NS; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "=v"(ur[a][j]));      // "defined" without an instruction: no write may trail the first load
    // steps 0 .. NS - 2 requested first; behind them the kernel's EPL epilogue-operand loads (see wino_loop)
    const int nsteps = ks_end - ks_begin;
    if constexpr (Gate::active) {
        // chain form: everything that does NOT depend on the previous layer of the launch goes out first (the U fragments of steps 0 and 1, scale / shift), then the
        // gate (arrival counters of the tile-blocks the raw patch reads; ends on a barrier), then the raw patch and the operands that may be fresh (LPL loads)
        static_assert(NS == 3, "the chain form has three register sets");
        cp.template ureg2<0>(ur[0], ks_begin); cp.template ureg2<1>(ur[0], ks_begin);
        if (nsteps > 1) { cp.template ureg2<0>(ur[1], ks_begin + 1); cp.template ureg2<1>(ur[1], ks_begin + 1); }
        epi_loads();
        gate.wait();
        cp.raw(ks_begin, 0);
        if (nsteps > 1) cp.raw(ks_begin + 1, 1);
        gate.late_loads();
        if (nsteps > 1) dma_wait<2 + Gate::LPL>(); else dma_wait<Gate::LPL>();
        __syncthreads();
    } else {
    cp.raw(ks_begin, 0);
    cp.template ureg2<0>(ur[0], ks_begin); cp.template ureg2<1>(ur[0], ks_begin);
    if (nsteps > 1) {
        cp.raw(ks_begin + 1, 1);
        cp.template ureg2<0>(ur[1], ks_begin + 1); cp.template ureg2<1>(ur[1], ks_begin + 1);
    }
    if constexpr (NS == 4) {
        if (nsteps > 2) {
            cp.raw(ks_begin + 2, 2);
            cp.template ureg2<0>(ur[2], ks_begin + 2); cp.template ureg2<1>(ur[2], ks_begin + 2);
        }
    }
    epi_loads();
    // step 0 has landed when only the younger loads can be outstanding: 6 per further requested step + the epilogue operands
    if (NS == 4 && nsteps > 2) dma_wait<12 + EPL>(); else if (nsteps > 1) dma_wait<6 + EPL>(); else dma_wait<EPL>();
    __syncthreads();
    }
#ifdef LSPF2F_WINO_STAMPS
    *first_landed = __builtin_amdgcn_s_memtime();
#endif
    const ProgressPrio prio(prio_mode, nsteps);
    // one K-step on register set S (= ring slot of the raw patch); the loads of step t + NS - 1 go into set (S + NS - 1) % NS, last read in step t - 1
    constexpr int AHEAD = NS - 1;
    auto step = [&](auto Sc, int t) {
        constexpr int S = decltype(Sc)::value, S2 = (S + AHEAD) % NS;
        prio.step(t);                                          // `wino_prio`: wave priority by K-loop progress (wino_common.h)
        const char *rawp = smem_c + S * kRawStage;
        float4 d[2][4];
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) d[k][dx] = *reinterpret_cast<const float4 *>(rawp + araw[k][dx]);
        const bool issue = t + AHEAD < nsteps;
        float4 tt[4], v[4];
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            if constexpr (ROW == 0 || ROW == 3) tt[dx] = f4sub(d[0][dx], d[1][dx]);
            else if constexpr (ROW == 1) tt[dx] = f4add(d[0][dx], d[1][dx]);
            else tt[dx] = f4sub(d[1][dx], d[0][dx]);
        }
        v[0] = f4sub(tt[0], tt[2]); v[1] = f4add(tt[1], tt[2]); v[2] = f4sub(tt[2], tt[1]); v[3] = f4sub(tt[1], tt[3]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = c == 0 ? v[j].x : c == 1 ? v[j].y : c == 2 ? v[j].z : v[j].w;
                acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, ur[S][j][c], acc[j][0], 0, 0, 0);
            }
            if (issue) {
                if (c == 0) cp.raw(ks_begin + t + AHEAD, S2);
                if (c == 1) cp.template ureg2<0>(ur[S2], ks_begin + t + AHEAD);
                if (c == 2) cp.template ureg2<1>(ur[S2], ks_begin + t + AHEAD);
            }
        }
        // step t + 1 has landed: everything but the loads of the steps behind it -- six per step (this wave's; the barrier covers the other waves' raw pieces)
        if constexpr (NS == 3) {
            if (issue) dma_wait<6>(); else dma_wait<0>();
        } else {
            if (issue) dma_wait<12>(); else if (t + 2 < nsteps) dma_wait<6>(); else dma_wait<0>();
        }
        __syncthreads();
    };
    for (int t = 0; t < nsteps; t += NS) {                  // (one exit: with a break per step hipcc copies the accumulators between register sets)
        step(IntC<0>{}, t);
        if (t + 1 < nsteps) step(IntC<1>{}, t + 1);
        if (t + 2 < nsteps) step(IntC<2>{}, t + 2);
        if constexpr (NS == 4) { if (t + 3 < nsteps) step(IntC<3>{}, t + 3); }
    }
    prio.done();
}


// InstanceNorm plans: the statistics of the 128 output pixels x 4 channels-per-thread a workgroup has just computed (v[0..3] = this thread's 2 x 2 pixels of
// one channel quad, raw conv output + bias), in the form in_finalize merges (instnorm.hip): sums of d = v - c and d^2 with c = the tile-block's first pixel, so that
// no E[x^2] - mean^2 of raw values is ever formed.  Fixed order: xor-shuffles over the wave's 8 tiles, then the four waves in order -> bit-reproducible.
__device__ __forceinline__ void wino_stats(const WinoParams &p, float4 *scratch, const float4 (&v)[4], int tid, size_t g)
{
    float4 *cs = scratch, *part = scratch + 8;
    const int qi = tid & 7, wave = tid >> 6, lane = tid & 63;
    __syncthreads();                                          // the previous channel block's readers are done with the scratch
    if (tid < 8) cs[qi] = v[0];                               // tile 0's pixel (0, 0): the tile-block's first pixel
    __syncthreads();
    const float4 c = cs[qi];
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float4 d = make_float4(v[k].x - c.x, v[k].y - c.y, v[k].z - c.z, v[k].w - c.w);
        s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
        s2.x += d.x * d.x; s2.y += d.y * d.y; s2.z += d.z * d.z; s2.w += d.w * d.w;
    }
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
        s1.x += __shfl_xor(s1.x, o); s1.y += __shfl_xor(s1.y, o); s1.z += __shfl_xor(s1.z, o); s1.w += __shfl_xor(s1.w, o);
        s2.x += __shfl_xor(s2.x, o); s2.y += __shfl_xor(s2.y, o); s2.z += __shfl_xor(s2.z, o); s2.w += __shfl_xor(s2.w, o);
    }
    if (lane < 8) { part[(wave * 8 + lane) * 2] = s1; part[(wave * 8 + lane) * 2 + 1] = s2; }
    __syncthreads();
    if (tid < 8) {
        float4 a = part[tid * 2], q2 = part[tid * 2 + 1];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float4 x = part[(w * 8 + tid) * 2], y = part[(w * 8 + tid) * 2 + 1];
            a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
            q2.x += y.x; q2.y += y.y; q2.z += y.z; q2.w += y.w;
        }
        *reinterpret_cast<float4 *>(p.psum + g) = a;
        *reinterpret_cast<float4 *>(p.psq + g) = q2;
        *reinterpret_cast<float4 *>(p.pshift + g) = c;
    }
}

// WT (tune key `out_wt`, A-B runs of round 5): the finished output leaves through write-through (sc1) buffer stores instead of plain ones, so that it is not left
// dirty in the XCD's L2 for the end-of-kernel write-back (the guide prices a dependent boundary at + B / 6 TB/s behind B dirty bytes)
template <int NB, int NS, bool IL, bool ROT, bool UR = false, bool WT = false>
__global__ __launch_bounds__(256, 2) void wino3x3(const WinoParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef __attribute__((address_space(3))) float lds_float;
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_float *)smem;
    const char *smem_c = reinterpret_cast<const char *>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    WSTAMP_DECL;
    WSTAMP(0);
    // one scalar-load round trip for the whole argument block (see igemm.hip)
    asm volatile("" :: "s"(p.src), "s"(p.u), "s"(p.scale), "s"(p.shift), "s"(p.residual), "s"(p.out), "s"(p.partial), "s"(p.tile_cnt));
    asm volatile("" :: "s"(p.B), "s"(p.H), "s"(p.W), "s"(p.C), "s"(p.N), "s"(p.relu), "s"(p.splits), "s"(p.steps_per_split), "s"(p.ntb), "s"(p.nng),
                       "s"(p.tbx), "s"(p.nmajor), "s"(p.xcd), "s"(p.div_plane.m), "s"(p.div_plane.s1), "s"(p.div_plane.s2), "s"(p.div_fast.m),
                       "s"(p.div_fast.s1), "s"(p.div_fast.s2), "s"(p.div_tbf.m), "s"(p.div_tbf.s1), "s"(p.div_tbf.s2), "s"(p.div_tbx.m), "s"(p.div_tbx.s1),
                       "s"(p.div_tbx.s2));

    // block -> (split z, tile-block tb, channel group ng).  XCD-aware like the igemm: the dispatcher deals blocks round-robin over the 8 XCDs,
    // so logical ids are handed out in 8 contiguous chunks; inside a split plane either the channel groups of a tile-block are adjacent
    // (activation-heavy layers) or the tile-blocks of a channel group (weight-heavy layers: an XCD's L2 then holds a slice of U).
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
    const int b = (int)p.div_tbf.div((unsigned)tb);               // frame; tile-blocks per frame = tby * tbx
    const int tbi = tb - b * (p.tby * p.tbx);
    const int by = (int)p.div_tbx.div((unsigned)tbi), bx = tbi - by * p.tbx;
    const int Y0 = by * 8, X0 = bx * 16;
    const int n0 = ng * 32 * NB;
    const int S = p.C >> 3;                                      // 8-channel K-steps
    const int ks_begin = z * p.steps_per_split;
    int ks_end = ks_begin + p.steps_per_split;
    if (ks_end > S) ks_end = S;

    // raw-patch DMA: this wave's two pieces; lane -> chunk -> (patch pixel, channel quad) -> byte offset in the NHWC source or out of range
    unsigned vraw[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int ci = (wave + 4 * k) * 64 + lane;
        const int par = ci / 90, rem2 = ci - par * 90, qd = rem2 / 45, r2 = rem2 - qd * 45, hy = r2 / 9, hx = r2 - hy * 9;
        const int y = Y0 - 1 + 2 * hy + (par >> 1), x = X0 - 1 + 2 * hx + (par & 1);
        const bool ok = ci < 360 && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        vraw[k] = ok ? ((unsigned)((b * p.H + y) * p.W + x) * (unsigned)p.C + (unsigned)(qd * 4)) * 4u : kOOBw;
    }
    const i32x4 srd_src = make_srd(p.src, (unsigned)(p.B * p.H * p.W) * (unsigned)p.C * 4u);
    const i32x4 srd_u = make_srd(p.u, 16u * (unsigned)p.C * (unsigned)p.N * 4u);
    // U fragments: [n-block][xi-row][k-step][j][64 lanes][4]
    const unsigned soff_nb = 4u * (unsigned)S * 4096u;                               // one n-block further
    const unsigned soff_u0 = (unsigned)((n0 >> 5) * 4 + wave) * (unsigned)S * 4096u;

    constexpr int EPL = 6 * NB;                                    // epilogue-operand loads below: always issued, so that the waits can count them
    WinoCopy<NB, NS, UR> cp;
    cp.init(lds0, wave, lane, vraw[0], vraw[1], srd_src, srd_u, soff_u0, soff_nb);

    // Epilogue operands requested BEFORE the K loop (the igemm's trick): folded-BN scale / shift of this thread's channel quads and its residual
    // pixels.  In the epilogue they would be dependent global round trips with the matrix pipe idle; here they ride under the first fetch (their
    // address arithmetic too: they are issued BEHIND it).  Buffer loads through descriptors whose range is zero when the operand is absent or not
    // wanted (split-K slices): always EPL instructions, zeros for the absent ones -- loads return in order, so the loop's first wait allows for
    // exactly EPL younger ones and every later wait is unaffected.
    const int trow = tid >> 3, cq = (tid & 7) * 4;                 // this thread's tile (ty = trow >> 3, tx = trow & 7) and channel quad
    const int oy = Y0 + 2 * (trow >> 3), ox = X0 + 2 * (trow & 7);
    const bool pre = p.splits == 1 && !p.nopre;
    const bool pre_sc = pre && p.scale != nullptr, pre_res = pre && p.residual != nullptr;
    const __amdgpu_buffer_rsrc_t rs_scale = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.scale), 0, pre_sc ? p.N * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_shift = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.shift), 0, pre_sc ? p.N * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.residual), 0, pre_res ? (int)((unsigned)(p.B * p.H * p.W) * (unsigned)p.N * 4u) : 0, 0x00020000);
    float4 scv[NB], shv[NB], rpre[NB][4];
    auto epi_loads = [&]() {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int n = n0 + nb * 32 + cq;
        scv[nb] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_scale, (unsigned)n * 4u, 0, 0));
        shv[nb] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_shift, (unsigned)n * 4u, 0, 0));
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) {
            const unsigned pix = (unsigned)((b * p.H + oy + (ab >> 1)) * p.W + ox + (ab & 1));
            rpre[nb][ab] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, (pix * (unsigned)p.N + (unsigned)n) * 4u, 0, 0));
        }
    }
    };
    WSTAMP(1);
#ifdef LSPF2F_WINO_STAMPS
    unsigned long long *fl = &stamp_t[2];
#else
    unsigned long long *fl = nullptr;
#endif

    f32x16 acc[4][NB];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][nb][e] = 0.f;

    if constexpr (UR) {
        static_assert(NB == 1 && (NS == 3 || NS == 4), "the register form exists for one channel block per wave");
        switch (wave) {
        case 0: wino_loop_ur<0, EPL, NS>(cp, epi_loads, acc, smem_c, lane, ks_begin, ks_end, fl, p.prio >= 4 ? p.prio - 3 : p.prio); break;
        case 1: wino_loop_ur<1, EPL, NS>(cp, epi_loads, acc, smem_c, lane, ks_begin, ks_end, fl, p.prio >= 4 ? p.prio - 3 : p.prio); break;
        case 2: wino_loop_ur<2, EPL, NS>(cp, epi_loads, acc, smem_c, lane, ks_begin, ks_end, fl, p.prio >= 4 ? p.prio - 3 : p.prio); break;
        default: wino_loop_ur<3, EPL, NS>(cp, epi_loads, acc, smem_c, lane, ks_begin, ks_end, fl, p.prio >= 4 ? p.prio - 3 : p.prio); break;
        }
    } else {
        switch (wave) {
        case 0: wino_loop<NB, NS, IL, ROT, 0, EPL>(cp, epi_loads, acc, smem_c, wave, lane, ks_begin, ks_end, fl, p.prio >= 4 ? p.prio - 3 : 0); break;
        case 1: wino_loop<NB, NS, IL, ROT, 1, EPL>(cp, epi_loads, acc, smem_c, wave, lane, ks_begin, ks_end, fl, p.prio >= 4 ? p.prio - 3 : 0); break;
        case 2: wino_loop<NB, NS, IL, ROT, 2, EPL>(cp, epi_loads, acc, smem_c, wave, lane, ks_begin, ks_end, fl, p.prio >= 4 ? p.prio - 3 : 0); break;
        default: wino_loop<NB, NS, IL, ROT, 3, EPL>(cp, epi_loads, acc, smem_c, wave, lane, ks_begin, ks_end, fl, p.prio >= 4 ? p.prio - 3 : 0); break;
        }
    }
    // (the loop ends on a barrier: every wave is done with the ring slots, the patch below may overwrite them)
    WSTAMP(3);

    // ---- output transform.  Column half in registers: Z[i][0] = M0 + M1 + M2, Z[i][1] = M1 - M2 - M3 (this wave's row i); row half across the
    // four waves through LDS: Y[0][b] = Z0 + Z1 + Z2, Y[1][b] = Z1 - Z2 - Z3.  C/D layout of the 32x32 MFMA: col = lane & 31,
    // row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5); the patch is [wave][b][nb][32 rows][36] so that a reader owns 4 consecutive channels of a tile.
    constexpr int EP = 36;
    {
        const int ccol = lane & 31, crow = 4 * (lane >> 5);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float *pz0 = smem + ((wave * 2 + 0) * NB + nb) * (32 * EP);
            float *pz1 = smem + ((wave * 2 + 1) * NB + nb) * (32 * EP);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + crow;
                pz0[row * EP + ccol] = (acc[0][nb][e] + acc[1][nb][e]) + acc[2][nb][e];
                pz1[row * EP + ccol] = (acc[1][nb][e] - acc[2][nb][e]) - acc[3][nb][e];
            }
        }
    }
    __syncthreads();
    WSTAMP(4);
    const __amdgpu_buffer_rsrc_t slab_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.partial, 0, p.splits > 1 ? (int)p.slab_bytes : 0, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, WT ? (int)((size_t)p.B * p.H * p.W * p.N * 4) : 0, 0x00020000);   // <= 2 GB: wino_supported
    const size_t npix = (size_t)p.B * p.H * p.W;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int n = n0 + nb * 32 + cq;
        float4 zz[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
                zz[i][bb] = *reinterpret_cast<const float4 *>(smem + ((i * 2 + bb) * NB + nb) * (32 * EP) + trow * EP + cq);
        float4 sc = scv[nb], sh = shv[nb];
        if (!pre_sc) { sc = make_float4(1.f, 1.f, 1.f, 1.f); sh = make_float4(0.f, 0.f, 0.f, 0.f); }
        if (!pre && p.splits == 1 && p.scale) {               // nopre (A-B runs): the operands are fetched here instead
            sc = *reinterpret_cast<const float4 *>(p.scale + n);
            sh = *reinterpret_cast<const float4 *>(p.shift + n);
        }
        float4 vst[4];                                       // InstanceNorm plans: the four finished pixels of this channel quad
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                float4 v = a == 0 ? f4add(f4add(zz[0][bb], zz[1][bb]), zz[2][bb]) : f4sub(f4sub(zz[1][bb], zz[2][bb]), zz[3][bb]);
                const size_t pix = ((size_t)b * p.H + (size_t)(oy + a)) * p.W + (size_t)(ox + bb);
                const size_t e = pix * p.N + n;
                if (p.splits > 1) {
                    // partial sums of this K slice; published write-through to whichever workgroup arrives last at the tile
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), slab_rsrc, (unsigned)(((size_t)z * npix * p.N + e) * 4), 0, 16);
                } else {
                    v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                    float4 rv = rpre[nb][a * 2 + bb];                 // zeros when the layer has no residual
                    if (!pre && p.residual) rv = *reinterpret_cast<const float4 *>(p.residual + e);
                    v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                    if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    if constexpr (WT) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), out_rsrc, (unsigned)(e * 4), 0, 16);
                    else *reinterpret_cast<float4 *>(p.out + e) = v;
                    vst[a * 2 + bb] = v;
                }
            }
        if (p.psum && p.splits == 1)
            wino_stats(p, reinterpret_cast<float4 *>(smem + wino_stats_base(NB, NS, UR) / 4), vst, tid, ((size_t)b * (size_t)(p.tby * p.tbx) + (size_t)tbi) * p.N + n);
    }
    WSTAMP(5);
    if (p.splits == 1) { WSTAMP_FLUSH; return; }

    // ---- split-K combine inside the launch (the igemm's protocol): write-through slabs above -> every wave drains its stores -> barrier -> one
    // relaxed agent-scope ticket per tile; the last arriver sums the slabs in z order (bit-reproducible) with loads that bypass its L1 and runs
    // the epilogue.  The counter is reset by the last arriver (zeroed once per workspace binding by the host).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // the combine's operands are requested by every slice BEFORE its ticket, so that for the slice that turns out to be last they are in flight
    // during the ticket's round trip instead of behind it (the others drop them)
    float4 sc2[NB], sh2[NB], rv2[NB][4];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int n = n0 + nb * 32 + cq;
        sc2[nb] = make_float4(1.f, 1.f, 1.f, 1.f); sh2[nb] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.scale) {
            sc2[nb] = *reinterpret_cast<const float4 *>(p.scale + n);
            sh2[nb] = *reinterpret_cast<const float4 *>(p.shift + n);
        }
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) {
            rv2[nb][ab] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.residual)
                rv2[nb][ab] = *reinterpret_cast<const float4 *>(p.residual + (((size_t)b * p.H + (size_t)(oy + (ab >> 1))) * p.W + (size_t)(ox + (ab & 1))) * p.N + n);
        }
    }
    unsigned *flag = reinterpret_cast<unsigned *>(smem);
    const unsigned tile = (unsigned)(tb * p.nng + ng);
    if (tid == 0) flag[0] = __hip_atomic_fetch_add(p.tile_cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    WSTAMP(6);
    if (flag[0] != (unsigned)p.splits - 1u) { WSTAMP_FLUSH; return; }
    if (tid == 0) __hip_atomic_store(p.tile_cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int n = n0 + nb * 32 + cq;
        const float4 sc = sc2[nb], sh = sh2[nb];
        float4 vst2[4];
        float4 tsl[4][8];
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) {                       // all slab loads of this channel quad in flight at once
            const size_t pix = ((size_t)b * p.H + (size_t)(oy + (ab >> 1))) * p.W + (size_t)(ox + (ab & 1));
            const size_t e = pix * p.N + n;
#pragma unroll
            for (int sl = 0; sl < 8; ++sl)
                if (sl < p.splits)
                    tsl[ab][sl] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(slab_rsrc, (unsigned)(((size_t)sl * npix * p.N + e) * 4), 0, 16));
        }
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) {
            const size_t pix = ((size_t)b * p.H + (size_t)(oy + (ab >> 1))) * p.W + (size_t)(ox + (ab & 1));
            const size_t e = pix * p.N + n;
            float4 v = tsl[ab][0];
#pragma unroll
            for (int sl = 1; sl < 8; ++sl)
                if (sl < p.splits) { v.x += tsl[ab][sl].x; v.y += tsl[ab][sl].y; v.z += tsl[ab][sl].z; v.w += tsl[ab][sl].w; }
            v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
            const float4 rv = rv2[nb][ab];
            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
            if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if constexpr (WT) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), out_rsrc, (unsigned)(e * 4), 0, 16);
            else *reinterpret_cast<float4 *>(p.out + e) = v;
            vst2[ab] = v;
        }
        if (p.psum) wino_stats(p, reinterpret_cast<float4 *>(smem + wino_stats_base(NB, NS, UR) / 4), vst2, tid, ((size_t)b * (size_t)(p.tby * p.tbx) + (size_t)tbi) * p.N + n);
    }
    WSTAMP(7);
    WSTAMP_FLUSH;
}


// ---- Round 6: 2..4 consecutive wino3x3<1> layers of one shape in ONE launch (the convs of one or two ResidualBlocks, models/networks.py:650-675: conv -> BN -> ReLU ->
// conv -> BN -> += x -> ReLU).  At one frame every such layer is ONE round of 512 workgroups and a dependent kernel boundary: ~10 000 of a launch's ~57 500 cycles pass
// outside any wave's lifetime, the prologue and the first landing (~4 000) overlap nothing.  Here the grid is nlayers x 512 workgroups; blockIdx / 512 is
// the layer.  A workgroup of layer k > 0 takes the slot a finished workgroup of layer k - 1 leaves, requests its U fragments and scale / shift (which depend on nothing),
// then polls the arrival counters of the <= 9 tile-blocks of layer k - 1 its 10 x 18 raw patch reads (one wave, one relaxed sc1 load per lane and counter, s_sleep between
// polls), then requests the patch with sc1 LDS-DMA loads -- the producers stored write-through (sc1) and drained (vmcnt(0) + barrier) before their one counter increment:
// the guide's "16-B sc1 stores AND sc1 loads" hand-off, no fence on either side.  Arithmetic and summation order are those of wino3x3<1, 3, .., UR, WT>: bit-identical.
//
// Progress (why a gate cannot wait for ever): the dispatcher hands out blocks in blockIdx order per XCD, so when a workgroup of layer k runs, every workgroup of a lower
// layer on its XCD's queue has been dispatched; the lowest unfinished layer m has nothing to wait for (its producers are finished), so its resident workgroups finish and
// free the slots its remaining ones need -- by induction every layer finishes, whatever else shares the chip.  HIP does not promise that order, so every gate is also bounded:
// after spin_limit polls it sets *fail and goes on (garbage out, never a hang); hosts that run the chain check the word in their tests.
// Counters: arrive[k][tb] += 1 per finished (tile-block tb, channel group) of layer k (split-K: by the last arriver, after its combine), so "ready" is >= nng; every
// workgroup of layer k + 1 that read it adds 1 on its way out and the last of them (the counter then reads nng + readers(tb)) stores 0: zero between launches, like tile_cnt.
// A buffer is written at most once per launch (the host keeps the chain's tensors apart), so no L2 can hold an older copy of a line a gate has released.
template <bool SC1, bool FENCE = true>      // FENCE = false with SC1 = false: measurement arm only (what a hand-off with free visibility would cost) -- never planned
struct ChainGate {
    static constexpr bool active = true;
    static constexpr int kArriveStride = kWinoArriveStride;     // one counter per 128-byte line: 512 pollers on 8 adjacent words were one memory channel's queue
    static constexpr int LPL = 4;                               // the residual's four pixels, requested behind the gate
    const unsigned *cnt;                                        // arrive[layer - 1] + (frame's first tile-block), nullptr for layer 0
    unsigned *fail;
    int by, bx, tby, tbx, wave, lane;
    unsigned target, limit;
    __amdgpu_buffer_rsrc_t rs_res;
    unsigned res_off[4];
    float4 *rpre;
#ifdef LSPF2F_WINO_STAMPS
    unsigned long long *t_gate;                                 // stamp slot 7 of a gated workgroup: when its gate opened (tools/probes/wino_chain_stamps.py)
#endif
    __device__ __forceinline__ bool neighbour(int &t) const      // lane l < 9 looks at tile-block (by + l / 3 - 1, bx + l % 3 - 1)
    {
        const int dy = lane / 3 - 1, dx = lane - (lane / 3) * 3 - 1;
        const int ny = by + dy, nx = bx + dx;
        const bool ok = lane < 9 && (unsigned)ny < (unsigned)tby && (unsigned)nx < (unsigned)tbx;
        t = ok ? ny * tbx + nx : by * tbx + bx;
        return ok;
    }
    __device__ __forceinline__ void wait() const
    {
        if (cnt) {
            if (wave == 0) {
                int t;
                bool pending = neighbour(t);                     // a lane stops polling the moment ITS counter is ready: the last polls touch one line, not nine
                const unsigned *c = cnt + (size_t)t * kArriveStride;
                unsigned spins = 0;
                for (;;) {
                    if (pending && __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) pending = false;
                    if (!__any(pending)) break;
                    if (++spins > limit) { if (lane == 0) __hip_atomic_fetch_or(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    if (spins < 4) __builtin_amdgcn_s_sleep(2); else if (spins < 16) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(32);
                }
                // the guide's consumer form: ONE relaxed poll -> ONE agent-scope acquire (buffer_inv sc1: this CU's L1) -> barrier -> plain loads.  The SC1 arm reads past the
                // L1 instead (every K-step then goes to L2: a 128-byte line of a pixel holds four K-steps, which the L1 serves to the plain loads)
                if constexpr (!SC1 && FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
#ifdef LSPF2F_WINO_STAMPS
            *t_gate = __builtin_amdgcn_s_memtime();
#endif
        }
    }
    __device__ __forceinline__ void late_loads() const          // always LPL loads (zero-range descriptor when the layer has no residual): the waits count them
    {
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) rpre[ab] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, res_off[ab], 0, SC1 ? 16 : 0));
    }
    // on the way out: this workgroup has read its <= 9 counters' tile-blocks; the last reader of a counter resets it
    __device__ __forceinline__ void release(int nng, int splits) const
    {
        if (cnt && wave == 0) {
            int t;
            const bool ok = neighbour(t);
            if (ok) {
                const int dy = lane / 3 - 1, dx = lane - (lane / 3) * 3 - 1;
                const int ny = by + dy, nx = bx + dx;
                const unsigned readers = (unsigned)((1 + (ny > 0) + (ny < tby - 1)) * (1 + (nx > 0) + (nx < tbx - 1)) * nng * splits);
                unsigned *c = const_cast<unsigned *>(cnt) + (size_t)t * kArriveStride;
                const unsigned old = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == target + readers - 1u) __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
};

template <bool SC1, bool FENCE = true>
__global__ __launch_bounds__(256, 2) void wino3x3_chain(const WinoChainParams pc)
{
    constexpr int NB = 1, NS = 3;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef __attribute__((address_space(3))) float lds_float;
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_float *)smem;
    const char *smem_c = reinterpret_cast<const char *>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const WinoParams &p = pc.c;
    WSTAMP_DECL;
    WSTAMP(0);
    // the layer: blockIdx order IS the dispatch order, which the progress argument above leans on -- so the layer comes from blockIdx itself, the XCD-aware
    // renumbering below only moves a workgroup inside its layer (wgs % 8 == 0: block b of the launch and block b % wgs of its layer sit on the same XCD)
    const int layer = (int)pc.div_wgs.div(blockIdx.x);
    const WinoChainLayer &L = pc.L[layer];
    const float *const l_src = L.src, *const l_u = L.u, *const l_scale = L.scale, *const l_shift = L.shift, *const l_res = L.residual;
    float *const l_out = L.out;
    const int l_relu = L.relu;
    const bool publish = layer + 1 < pc.nlayers;

    unsigned lin = blockIdx.x - (unsigned)layer * (unsigned)pc.wgs;
    if (p.xcd) {
        const unsigned total = (unsigned)pc.wgs, qq = total >> 3, rr = total & 7, x = lin & 7;
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
    const int Y0 = by * 8, X0 = bx * 16;
    const int n0 = ng * 32;
    const int S = p.C >> 3;
    const int ks_begin = z * p.steps_per_split;
    int ks_end = ks_begin + p.steps_per_split;
    if (ks_end > S) ks_end = S;

    unsigned vraw[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int ci = (wave + 4 * k) * 64 + lane;
        const int par = ci / 90, rem2 = ci - par * 90, qd = rem2 / 45, r2 = rem2 - qd * 45, hy = r2 / 9, hx = r2 - hy * 9;
        const int y = Y0 - 1 + 2 * hy + (par >> 1), x = X0 - 1 + 2 * hx + (par & 1);
        const bool ok = ci < 360 && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        vraw[k] = ok ? ((unsigned)((b * p.H + y) * p.W + x) * (unsigned)p.C + (unsigned)(qd * 4)) * 4u : kOOBw;
    }
    const i32x4 srd_src = make_srd(l_src, (unsigned)(p.B * p.H * p.W) * (unsigned)p.C * 4u);
    const i32x4 srd_u = make_srd(l_u, 16u * (unsigned)p.C * (unsigned)p.N * 4u);
    const unsigned soff_nb = 4u * (unsigned)S * 4096u;
    const unsigned soff_u0 = (unsigned)((n0 >> 5) * 4 + wave) * (unsigned)S * 4096u;

    constexpr int EPL = 2;                                         // scale, shift (ahead of the gate); the residual follows it (ChainGate::LPL)
    WinoCopy<NB, NS, true, SC1> cp;
    cp.init(lds0, wave, lane, vraw[0], vraw[1], srd_src, srd_u, soff_u0, soff_nb);

    const int trow = tid >> 3, cq = (tid & 7) * 4;
    const int oy = Y0 + 2 * (trow >> 3), ox = X0 + 2 * (trow & 7);
    const bool pre = p.splits == 1;
    const bool pre_sc = pre && l_scale != nullptr, pre_res = pre && l_res != nullptr;
    const __amdgpu_buffer_rsrc_t rs_scale = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(l_scale), 0, pre_sc ? p.N * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_shift = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(l_shift), 0, pre_sc ? p.N * 4 : 0, 0x00020000);
    const int nres_bytes = (int)((unsigned)(p.B * p.H * p.W) * (unsigned)p.N * 4u);
    float4 scv, shv, rpre[4];
    const int n = n0 + cq;
    auto epi_loads = [&]() {
        scv = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_scale, (unsigned)n * 4u, 0, 0));
        shv = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_shift, (unsigned)n * 4u, 0, 0));
    };
    ChainGate<SC1, FENCE> gate;
    gate.cnt = layer > 0 ? pc.arrive + ((size_t)(layer - 1) * (size_t)p.ntb + (size_t)b * (size_t)(p.tby * p.tbx)) * kWinoArriveStride : nullptr;
    gate.fail = pc.fail; gate.by = by; gate.bx = bx; gate.tby = p.tby; gate.tbx = p.tbx; gate.wave = wave; gate.lane = lane;
    gate.target = (unsigned)p.nng; gate.limit = pc.spin_limit;
    gate.rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(l_res), 0, pre_res ? nres_bytes : 0, 0x00020000);
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) {
        const unsigned pix = (unsigned)((b * p.H + oy + (ab >> 1)) * p.W + ox + (ab & 1));
        gate.res_off[ab] = (pix * (unsigned)p.N + (unsigned)n) * 4u;
    }
    gate.rpre = rpre;
#ifdef LSPF2F_WINO_STAMPS
    gate.t_gate = &stamp_t[7];
#endif
    WSTAMP(1);
#ifdef LSPF2F_WINO_STAMPS
    unsigned long long *fl = &stamp_t[2];
#else
    unsigned long long *fl = nullptr;
#endif

    f32x16 acc[4][NB];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][0][e] = 0.f;

    switch (wave) {
    case 0: wino_loop_ur<0, EPL, NS>(cp, epi_loads, acc, smem_c, lane, ks_begin, ks_end, fl, p.prio >= 4 ? p.prio - 3 : p.prio, gate); break;
    case 1: wino_loop_ur<1, EPL, NS>(cp, epi_loads, acc, smem_c, lane, ks_begin, ks_end, fl, p.prio >= 4 ? p.prio - 3 : p.prio, gate); break;
    case 2: wino_loop_ur<2, EPL, NS>(cp, epi_loads, acc, smem_c, lane, ks_begin, ks_end, fl, p.prio >= 4 ? p.prio - 3 : p.prio, gate); break;
    default: wino_loop_ur<3, EPL, NS>(cp, epi_loads, acc, smem_c, lane, ks_begin, ks_end, fl, p.prio >= 4 ? p.prio - 3 : p.prio, gate); break;
    }
    WSTAMP(3);

    // ---- output transform: as in wino3x3
    constexpr int EP = 36;
    {
        const int ccol = lane & 31, crow = 4 * (lane >> 5);
        float *pz0 = smem + ((wave * 2 + 0) * NB) * (32 * EP);
        float *pz1 = smem + ((wave * 2 + 1) * NB) * (32 * EP);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + crow;
            pz0[row * EP + ccol] = (acc[0][0][e] + acc[1][0][e]) + acc[2][0][e];
            pz1[row * EP + ccol] = (acc[1][0][e] - acc[2][0][e]) - acc[3][0][e];
        }
    }
    __syncthreads();
    WSTAMP(4);
    const __amdgpu_buffer_rsrc_t slab_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.partial, 0, p.splits > 1 ? (int)p.slab_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(l_out, 0, nres_bytes, 0x00020000);
    const size_t npix = (size_t)p.B * p.H * p.W;
    {
        float4 zz[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
                zz[i][bb] = *reinterpret_cast<const float4 *>(smem + ((i * 2 + bb) * NB) * (32 * EP) + trow * EP + cq);
        float4 sc = scv, sh = shv;
        if (!pre_sc) { sc = make_float4(1.f, 1.f, 1.f, 1.f); sh = make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                float4 v = a == 0 ? f4add(f4add(zz[0][bb], zz[1][bb]), zz[2][bb]) : f4sub(f4sub(zz[1][bb], zz[2][bb]), zz[3][bb]);
                const size_t pix = ((size_t)b * p.H + (size_t)(oy + a)) * p.W + (size_t)(ox + bb);
                const size_t e = pix * p.N + n;
                if (p.splits > 1) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), slab_rsrc, (unsigned)(((size_t)z * npix * p.N + e) * 4), 0, 16);
                } else {
                    v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                    const float4 rv = rpre[a * 2 + bb];                 // zeros when the layer has no residual
                    v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                    if (l_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), out_rsrc, (unsigned)(e * 4), 0, 16);
                }
            }
    }
    WSTAMP(5);
    // every wave's stores have left the chip before the counter says so (write-through + drained: nothing sits dirty in this XCD's L2)
    auto arrive = [&]() {
        if (publish) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) (void)__hip_atomic_fetch_add(pc.arrive + ((size_t)layer * (size_t)p.ntb + (size_t)tb) * kWinoArriveStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    if (p.splits == 1) { arrive(); gate.release(p.nng, p.splits); WSTAMP_FLUSH; return; }

    // ---- split-K combine inside the launch: the protocol of wino3x3
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float4 sc2 = make_float4(1.f, 1.f, 1.f, 1.f), sh2 = make_float4(0.f, 0.f, 0.f, 0.f), rv2[4];
    if (l_scale) {
        sc2 = *reinterpret_cast<const float4 *>(l_scale + n);
        sh2 = *reinterpret_cast<const float4 *>(l_shift + n);
    }
    const __amdgpu_buffer_rsrc_t rs_res2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(l_res), 0, l_res ? nres_bytes : 0, 0x00020000);
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) rv2[ab] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_res2, gate.res_off[ab], 0, 16));
    unsigned *flag = reinterpret_cast<unsigned *>(smem);
    const unsigned tile = (unsigned)(tb * p.nng + ng);
    if (tid == 0) flag[0] = __hip_atomic_fetch_add(p.tile_cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    WSTAMP(6);
    if (flag[0] != (unsigned)p.splits - 1u) { gate.release(p.nng, p.splits); WSTAMP_FLUSH; return; }
    if (tid == 0) __hip_atomic_store(p.tile_cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    {
        float4 tsl[4][8];
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) {
            const size_t pix = ((size_t)b * p.H + (size_t)(oy + (ab >> 1))) * p.W + (size_t)(ox + (ab & 1));
            const size_t e = pix * p.N + n;
#pragma unroll
            for (int sl = 0; sl < 8; ++sl)
                if (sl < p.splits)
                    tsl[ab][sl] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(slab_rsrc, (unsigned)(((size_t)sl * npix * p.N + e) * 4), 0, 16));
        }
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) {
            const size_t pix = ((size_t)b * p.H + (size_t)(oy + (ab >> 1))) * p.W + (size_t)(ox + (ab & 1));
            const size_t e = pix * p.N + n;
            float4 v = tsl[ab][0];
#pragma unroll
            for (int sl = 1; sl < 8; ++sl)
                if (sl < p.splits) { v.x += tsl[ab][sl].x; v.y += tsl[ab][sl].y; v.z += tsl[ab][sl].z; v.w += tsl[ab][sl].w; }
            v.x = v.x * sc2.x + sh2.x; v.y = v.y * sc2.y + sh2.y; v.z = v.z * sc2.z + sh2.z; v.w = v.w * sc2.w + sh2.w;
            const float4 rv = rv2[ab];
            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
            if (l_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), out_rsrc, (unsigned)(e * 4), 0, 16);
        }
    }
    arrive();
    gate.release(p.nng, p.splits);
    WSTAMP_FLUSH;
}

bool wino_supported(const WinoParams &p, int nb)
{
    if (nb != 1 && nb != 2) return false;
    if (p.B < 1 || p.H % 8 || p.W % 16 || p.C % 8 || p.C < 8 || p.N % (32 * nb)) return false;
    const size_t lim = 0x7fffffffull;                      // 32-bit buffer offsets with the top bit reserved as the out-of-range marker
    if ((size_t)p.B * p.H * p.W * p.C * 4 > lim || (size_t)16 * p.C * p.N * 4 > lim) return false;
    if ((size_t)p.B * p.H * p.W * p.N * 4 > lim) return false;          // the residual is read through a buffer descriptor with 32-bit offsets too
    if (p.splits < 1 || p.splits > 8) return false;
    if (p.splits > 1 && (!p.partial || !p.tile_cnt || (size_t)p.splits * p.B * p.H * p.W * p.N * 4 > lim)) return false;
    return true;
}

template <int NB, int NS, bool IL, bool ROT, bool UR = false, bool WT = false>
static hipError_t launch_wino_t(const WinoParams &q, hipStream_t s)
{
    constexpr int smem = wino_lds_bytes(NB, NS, UR);
    static AttrMask attr_mask;
    if (smem > 64 * 1024 && attr_needed_on_this_device(attr_mask)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wino3x3<NB, NS, IL, ROT, UR, WT>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        attr_done_on_this_device(attr_mask);
    }
    hipLaunchKernelGGL((wino3x3<NB, NS, IL, ROT, UR, WT>), dim3((unsigned)(q.ntb * q.nng * q.splits)), dim3(256), smem, s, q);
    return hipGetLastError();
}

hipError_t launch_wino(const WinoParams &p_in, int nb, hipStream_t s)
{
    if (!wino_supported(p_in, nb)) return hipErrorInvalidValue;
    WinoParams p = p_in;
    const int S = p.C / 8;
    p.steps_per_split = (S + p.splits - 1) / p.splits;
    if ((p.splits - 1) * p.steps_per_split >= S) return hipErrorInvalidValue;       // an empty split would publish garbage-free zeros but wastes a slab: planner bug
    p.tby = p.H / 8; p.tbx = p.W / 16;
    p.ntb = p.B * p.tby * p.tbx;
    p.nng = p.N / (32 * nb);
    if (p.splits > 1) p.slab_bytes = (size_t)p.splits * p.B * p.H * p.W * p.N * 4;
    // weight-heavy layers: an XCD keeps a slice of U in its L2 (channel groups slowest); activation-heavy: a band of tile-blocks
    const size_t act = (size_t)p.B * p.H * p.W * p.C * 4, wgt = (size_t)16 * p.C * p.N * 4;
    p.nmajor = wgt > act ? 1 : 0;
    p.xcd = 1;
    if (p.xcd_force == 1) p.xcd = 0;
    if (p.xcd_force == 2) p.nmajor = 0;
    if (p.xcd_force == 3) p.nmajor = 1;
    p.div_plane = FastDiv::make((unsigned)(p.ntb * p.nng));
    p.div_fast = FastDiv::make((unsigned)(p.nmajor ? p.ntb : p.nng));
    p.div_tbf = FastDiv::make((unsigned)(p.tby * p.tbx));
    p.div_tbx = FastDiv::make((unsigned)p.tbx);
    // tools only (A-B runs): no_il issues a step's copies as one block ahead of its MFMAs (ring of 2) instead of between them; no_rot runs the four
    // MFMAs of an accumulator back to back instead of rotating over the four accumulators of a channel block
    if (p.no_il) return nb == 2 ? launch_wino_t<2, 2, false, false>(p, s) : launch_wino_t<1, 2, false, false>(p, s);
    if (p.no_rot) return nb == 2 ? launch_wino_t<2, 2, true, false>(p, s) : launch_wino_t<1, 3, true, false>(p, s);
    if (p.out_wt) {
        if (nb == 1 && p.ureg) return launch_wino_t<1, 3, true, true, true, true>(p, s);
        return nb == 2 ? launch_wino_t<2, 2, true, true, false, true>(p, s) : launch_wino_t<1, 3, true, true, false, true>(p, s);
    }
    if (nb == 1 && p.ureg == 2) return launch_wino_t<1, 4, true, true, true>(p, s);       // four register sets (A-B arm)
    if (nb == 1 && p.ureg) return launch_wino_t<1, 3, true, true, true>(p, s);
    return nb == 2 ? launch_wino_t<2, 2, true, true>(p, s) : launch_wino_t<1, 3, true, true>(p, s);
}

bool wino_chain_supported(const WinoChainParams &pc)
{
    if (pc.nlayers < 1 || pc.nlayers > kWinoChainMax || !pc.arrive || !pc.fail) return false;
    if (!wino_supported(pc.c, 1)) return false;
    for (int k = 0; k < pc.nlayers; ++k) {
        const WinoChainLayer &l = pc.L[k];
        if (!l.src || !l.u || !l.out || ((l.scale == nullptr) != (l.shift == nullptr))) return false;
        // a buffer is written once per launch, and nothing a later layer writes may be something an earlier one still reads through a cache
        for (int j = 0; j < pc.nlayers; ++j) {
            if (j != k && pc.L[j].out == l.out) return false;
            if (j <= k && (pc.L[j].src == l.out || pc.L[j].residual == l.out)) return false;
        }
    }
    for (int k = 1; k < pc.nlayers; ++k)
        if (pc.L[k].src != pc.L[k - 1].out) return false;          // the gate orders exactly this edge (and, through it, every older one)
    return true;
}

hipError_t launch_wino_chain(const WinoChainParams &pc_in, hipStream_t s)
{
    if (!wino_chain_supported(pc_in)) return hipErrorInvalidValue;
    WinoChainParams pc = pc_in;
    WinoParams &p = pc.c;
    const int S = p.C / 8;
    p.steps_per_split = (S + p.splits - 1) / p.splits;
    if ((p.splits - 1) * p.steps_per_split >= S) return hipErrorInvalidValue;
    p.tby = p.H / 8; p.tbx = p.W / 16;
    p.ntb = p.B * p.tby * p.tbx;
    p.nng = p.N / 32;
    if (p.splits > 1) p.slab_bytes = (size_t)p.splits * p.B * p.H * p.W * p.N * 4;
    const size_t act = (size_t)p.B * p.H * p.W * p.C * 4, wgt = (size_t)16 * p.C * p.N * 4;
    p.nmajor = wgt > act ? 1 : 0;
    p.xcd = 1;
    if (p.xcd_force == 1) p.xcd = 0;
    if (p.xcd_force == 2) p.nmajor = 0;
    if (p.xcd_force == 3) p.nmajor = 1;
    p.div_plane = FastDiv::make((unsigned)(p.ntb * p.nng));
    p.div_fast = FastDiv::make((unsigned)(p.nmajor ? p.ntb : p.nng));
    p.div_tbf = FastDiv::make((unsigned)(p.tby * p.tbx));
    p.div_tbx = FastDiv::make((unsigned)p.tbx);
    pc.wgs = p.ntb * p.nng * p.splits;
    pc.div_wgs = FastDiv::make((unsigned)pc.wgs);
    if (!pc.spin_limit) pc.spin_limit = 1u << 18;
    if ((size_t)pc.wgs * (size_t)pc.nlayers > 0x7fffffffull) return hipErrorInvalidValue;
    constexpr int smem = wino_lds_bytes(1, 3, true);
    if (pc.sc1_loads == 2) hipLaunchKernelGGL((wino3x3_chain<false, false>), dim3((unsigned)(pc.wgs * pc.nlayers)), dim3(256), smem, s, pc);
    else if (pc.sc1_loads) hipLaunchKernelGGL(wino3x3_chain<true>, dim3((unsigned)(pc.wgs * pc.nlayers)), dim3(256), smem, s, pc);
    else hipLaunchKernelGGL(wino3x3_chain<false>, dim3((unsigned)(pc.wgs * pc.nlayers)), dim3(256), smem, s, pc);
    return hipGetLastError();
}

// Host: OIHW [N][C][3][3] -> U = G g G^T (double, rounded once) in the MFMA fragment order [n-block N/32][xi-row 4][k-step C/8][j 4][lane 64][4]:
// lane l holds output channel 32 nblock + (l & 31), input channels 8 s + 4 (l >> 5) + 0..3 -- one 1-KB piece = the B operand of 4 MFMAs.
void pack_wino_weights(const float *oihw, int cin, int cout, float *out)
{
    static const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    const int S = cin / 8;
    for (int n = 0; n < cout; ++n)
        for (int c = 0; c < cin; ++c) {
            const float *g = oihw + ((size_t)n * cin + c) * 9;
            double tmp[4][3];
            for (int i = 0; i < 4; ++i)
                for (int b = 0; b < 3; ++b) tmp[i][b] = G[i][0] * (double)g[b] + G[i][1] * (double)g[3 + b] + G[i][2] * (double)g[6 + b];
            const int nblk = n >> 5, s = c >> 3, lane = (n & 31) + 32 * ((c & 7) >> 2), t = c & 3;
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    const double u = tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2];
                    out[(((((size_t)nblk * 4 + i) * S + s) * 4 + j) * 64 + lane) * 4 + t] = (float)u;
                }
        }
}

}  // namespace lspf2f
