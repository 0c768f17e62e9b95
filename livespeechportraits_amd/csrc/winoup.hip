// gfx950 (MI355X / CDNA4): Upsample(x2, nearest) + Conv3x3 (+ folded BatchNorm, ReLU) over the never-materialised concat of two sources, as a
// Winograd form with 9 multiplies per 2x2 output tile -- against 16 in the sub-pixel form of igemm.hip and 36 in the literal one.
// fp32 in, fp32 accumulate.  See DESIGN.md section 4.9.
//
// Reference semantics: the up side of a skip block, models/networks.py:610-611 / :617-618 / :626-627 (nn.Upsample(scale_factor=2, 'nearest') -> Conv2d
// 3x3 pad 1 over cat([x, model(x)], 1), :646) followed by BatchNorm2d (eval) and ReLU (:619-620, :628-629).
//
// F(2x2, 3x3) on the UPSAMPLED image: the 4x4 patch under an output tile whose origin is even has rows (a, b, b, c) = source rows (y-1, y, y+1), so
//   B^T d  =  [a - b; 2b; 0; b - c]
// -- row 2 of the transformed tile vanishes identically and the others depend on the 3x3 SOURCE neighbourhood only; the same holds for columns.  A
// tile is therefore one source pixel, only the positions (i, j) in {0, 1, 3}^2 are multiplied (9 GEMMs of [source pixels x Cin] . [Cin x Cout]), and
// the output transform Y = A^T M A writes the pixel's 2x2 outputs.  The factors 2 are folded into the packed weights U' = c_i c_j (G g G^T)[i][j].
//
// Workgroup = 4 x 8 source pixels (8 x 16 output pixels) x 32 NB output channels, THREE waves: wave w owns row {0, 1, 3}[w] of the transformed tile
// (three accumulators per channel block), reads 3 or 6 raw pixels per tile and channel quad, and forms its three V values with at most 5 subtractions
// per channel in registers.  Raw patch (6 x 10 pixels x 8 channels per K-step: two LDS-DMA pieces) and weight fragments (3 NB pieces per wave and
// step, the wave's own) as in wino.hip; the K loop walks source 0's channels, then source 1's.
#include "device_common.h"
#include "kernels.h"
#include "wino_common.h"
#include <cstdlib>

namespace lspf2f {

static constexpr unsigned kOOBu = 0x80000000u;
static constexpr int kUpRawStage = 2 * 1024;                // 120 chunks (6 x 10 pixels x 2 channel quads) in 2 pieces of 64
static constexpr int kWinoUpStatsScratch = (8 + 3 * 8 * 2) * 16;   // InstanceNorm plans: the shift [8 quads] and the three waves' partial sums [3][8][2], float4 each (winoup_stats)

__host__ __device__ constexpr int winoup_u_stage(int nb) { return 3 * 3 * nb * 1024; }          // 3 waves x (3 j x nb) pieces
__host__ __device__ constexpr int winoup_raw_base(int nb) { return 2 * winoup_u_stage(nb); }
__host__ __device__ constexpr int winoup_dump(int nb) { return winoup_raw_base(nb) + 2 * kUpRawStage; }
__host__ __device__ constexpr int winoup_lds_bytes(int nb)
{
    const int loop = winoup_dump(nb) + 1024;
    const int patch = 3 * 2 * nb * 32 * 36 * 4;               // epilogue: [wave][b][nb][32 tiles][36]
    return (loop > patch ? loop : patch) + kWinoUpStatsScratch;
}
__host__ __device__ constexpr int winoup_stats_base(int nb) { return winoup_lds_bytes(nb) - kWinoUpStatsScratch; }

template <int NB, int ROW>      // ROW = 0, 1, 2: rows 0, 1, 3 of the transformed tile
__device__ __forceinline__ void winoup_loop(const WinoUpParams &p, f32x16 (&acc)[3][NB], const char *smem_c, unsigned lds0, int wave, int lane,
                                            unsigned vraw, i32x4 srd0, i32x4 srd1, i32x4 srd_u, unsigned soff_u0, unsigned soff_nb,
                                            int ks_begin, int ks_end)
{
    constexpr int USTAGE = winoup_u_stage(NB), RAWB = winoup_raw_base(NB), DUMP = winoup_dump(NB);
    constexpr int R0 = ROW == 0 ? 0 : 1;                                      // first raw row this wave reads (0: rows 0, 1; 1: row 1; 2: rows 1, 2)
    constexpr int NR = ROW == 1 ? 1 : 2;
    // fragment-read addresses: lane (tile r = l & 31 -> ty = r >> 3, tx = r & 7; channel quad q = l >> 5) reads raw pixel (ty + dy, tx + dx) of the
    // 6 x 10 patch: chunk = (q * 6 + py) * 10 + px
    const int r = lane & 31, q = lane >> 5, ty = r >> 3, tx = r & 7;
    unsigned araw[NR][3];
#pragma unroll
    for (int k = 0; k < NR; ++k)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) araw[k][dx] = (unsigned)(RAWB + ((q * 6 + ty + R0 + k) * 10 + tx + dx) * 16);
    const unsigned au = (unsigned)(wave * (3 * NB * 1024) + lane * 16);
    const unsigned vu2[2] = {(unsigned)(lane * 16), (unsigned)(lane * 16 + 1024)};
    const unsigned vu1[1] = {(unsigned)(lane * 16 + 2048)};
    const unsigned lds_u = lds0 + (unsigned)(wave * (3 * NB * 1024));
    const unsigned lds_r = wave < 2 ? lds0 + (unsigned)(RAWB + wave * 1024) : lds0 + (unsigned)DUMP;   // pieces 0, 1: waves 0, 1; wave 2 feeds the dump slot
    const int s0 = p.C0 >> 3;                                                  // K-steps of source 0

    auto fetch_raw = [&](int ks, int slot) {
        const bool first = ks < s0;
        const unsigned dst[1] = {vraw};
        dma16_group<1, 1024>(wave < 2 ? lds_r + (unsigned)(slot * kUpRawStage) : lds_r, dst, first ? srd0 : srd1, (first ? ks : ks - s0) * 32);
    };
    auto fetch_u = [&](int ks, int slot, int nb) {
        const unsigned base = lds_u + (unsigned)(slot * USTAGE + nb * 3072);
        const int soff = (int)(soff_u0 + (unsigned)nb * soff_nb + (unsigned)ks * 3072u);
        dma16_group<2, 1024>(base, vu2, srd_u, soff);
        dma16_group<1, 1024>(base + 2048u, vu1, srd_u, soff);
    };
    constexpr int PIECES = 1 + 3 * NB;

    const int nsteps = ks_end - ks_begin;
    if (nsteps <= 0) return;
    fetch_raw(ks_begin, 0);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) fetch_u(ks_begin, 0, nb);
    dma_wait<0>();
    __syncthreads();
    int cur = 0;
    const ProgressPrio prio(p.prio >= 4 ? p.prio - 3 : 0, nsteps);       // `wino_prio` >= 4: the up-conv form too (wino_common.h)
    for (int t = 0; t < nsteps; ++t) {
        prio.step(t);
        const char *rawp = smem_c + cur * kUpRawStage;
        const char *up = smem_c + cur * USTAGE + au;
        float4 d[NR][3];
#pragma unroll
        for (int k = 0; k < NR; ++k)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) d[k][dx] = *reinterpret_cast<const float4 *>(rawp + araw[k][dx]);
        const bool issue = t + 1 < nsteps;
        float4 tt[3], v[3];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            if constexpr (ROW == 1) tt[dx] = d[0][dx];                       // 2b: the factor lives in the weights
            else tt[dx] = f4sub(d[0][dx], d[1][dx]);                         // a - b  |  b - c
        }
        v[0] = f4sub(tt[0], tt[1]); v[1] = tt[1]; v[2] = f4sub(tt[1], tt[2]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float4 uj[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) uj[j] = *reinterpret_cast<const float4 *>(up + (nb * 3 + j) * 1024);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {                                // consecutive MFMAs on different accumulators
                    const float a = c == 0 ? v[j].x : c == 1 ? v[j].y : c == 2 ? v[j].z : v[j].w;
                    const float b = c == 0 ? uj[j].x : c == 1 ? uj[j].y : c == 2 ? uj[j].z : uj[j].w;
                    acc[j][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j][nb], 0, 0, 0);
                }
                // the copies of step t + 1 go out between the MFMA groups (ring of 2: the slot was released by the barrier that ended step t - 1)
                if (issue) {
                    if (nb == 0 && c == 0) fetch_raw(ks_begin + t + 1, cur ^ 1);
                    if (c == 1) fetch_u(ks_begin + t + 1, cur ^ 1, nb);
                }
            }
        }
        dma_wait<0>();
        __syncthreads();
        cur ^= 1;
    }
    prio.done();
    (void)PIECES;
}

// InstanceNorm plans: the statistics of the 128 output pixels a workgroup has just computed, in the form in_finalize merges (see wino_stats in wino.hip).  A thread
// holds one or two (source pixel, channel quad) items of a channel block -- tid and tid + 192 of 256, the same quad -- i.e. 4 or 8 finished output pixels.
__device__ __forceinline__ void winoup_stats(const WinoUpParams &p, float4 *scratch, const float4 (&v)[2][4], bool second, int tid, size_t g)
{
    float4 *cs = scratch, *part = scratch + 8;
    const int qi = tid & 7, wave = tid >> 6, lane = tid & 63;
    __syncthreads();                                          // the previous channel block's readers are done with the scratch
    if (tid < 8) cs[qi] = v[0][0];                            // item tid: source pixel 0 of the tile-block, output pixel (0, 0)
    __syncthreads();
    const float4 c = cs[qi];
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        if (it == 1 && !second) continue;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 d = make_float4(v[it][k].x - c.x, v[it][k].y - c.y, v[it][k].z - c.z, v[it][k].w - c.w);
            s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
            s2.x += d.x * d.x; s2.y += d.y * d.y; s2.z += d.z * d.z; s2.w += d.w * d.w;
        }
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
        for (int w = 1; w < 3; ++w) {
            const float4 x = part[(w * 8 + tid) * 2], y = part[(w * 8 + tid) * 2 + 1];
            a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
            q2.x += y.x; q2.y += y.y; q2.z += y.z; q2.w += y.w;
        }
        *reinterpret_cast<float4 *>(p.psum + g) = a;
        *reinterpret_cast<float4 *>(p.psq + g) = q2;
        *reinterpret_cast<float4 *>(p.pshift + g) = c;
    }
}

template <int NB, bool WT = false>      // WT: write-through output stores (tune key `out_wt`, see wino.hip)
__global__ __launch_bounds__(192, 2) void winoup3x3(const WinoUpParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef __attribute__((address_space(3))) float lds_float;
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_float *)smem;
    const char *smem_c = reinterpret_cast<const char *>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    asm volatile("" :: "s"(p.src0), "s"(p.src1), "s"(p.u), "s"(p.scale), "s"(p.shift), "s"(p.out), "s"(p.partial), "s"(p.tile_cnt));
    asm volatile("" :: "s"(p.B), "s"(p.Hs), "s"(p.Ws), "s"(p.C0), "s"(p.C1), "s"(p.N), "s"(p.relu), "s"(p.splits), "s"(p.steps_per_split), "s"(p.ntb), "s"(p.nng),
                       "s"(p.tbx), "s"(p.tby), "s"(p.nmajor), "s"(p.div_plane.m), "s"(p.div_plane.s1), "s"(p.div_plane.s2), "s"(p.div_fast.m),
                       "s"(p.div_fast.s1), "s"(p.div_fast.s2), "s"(p.div_tbf.m), "s"(p.div_tbf.s1), "s"(p.div_tbf.s2), "s"(p.div_tbx.m), "s"(p.div_tbx.s1),
                       "s"(p.div_tbx.s2));

    // block -> (split z, tile-block tb, channel group ng), handed out in per-XCD chunks like wino.hip
    unsigned lin = blockIdx.x;
    {
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
    const int Y0 = by * 4, X0 = bx * 8;                       // source pixel origin of the tile-block
    const int n0 = ng * 32 * NB;
    const int Cin = p.C0 + p.C1;
    const int S = Cin >> 3;
    const int ks_begin = z * p.steps_per_split;
    int ks_end = ks_begin + p.steps_per_split;
    if (ks_end > S) ks_end = S;

    // raw-patch DMA: waves 0 and 1 fetch one piece each; lane -> chunk (q * 6 + py) * 10 + px -> source pixel (Y0 - 1 + py, X0 - 1 + px), quad q.
    // Both sources have the same channel count (or there is only one), so one per-lane byte offset serves both descriptors.
    unsigned vraw;
    {
        const int ci = wave * 64 + lane;
        const int qd = ci / 60, rem2 = ci - qd * 60, py = rem2 / 10, px = rem2 - py * 10;
        const int y = Y0 - 1 + py, x = X0 - 1 + px;
        const bool ok = wave < 2 && ci < 120 && (unsigned)y < (unsigned)p.Hs && (unsigned)x < (unsigned)p.Ws;
        vraw = ok ? ((unsigned)((b * p.Hs + y) * p.Ws + x) * (unsigned)p.C0 + (unsigned)(qd * 4)) * 4u : kOOBu;
    }
    const unsigned src_bytes = (unsigned)(p.B * p.Hs * p.Ws) * (unsigned)p.C0 * 4u;
    const i32x4 srd0 = make_srd(p.src0, src_bytes);
    const i32x4 srd1 = make_srd(p.src1 ? p.src1 : p.src0, src_bytes);
    const i32x4 srd_u = make_srd(p.u, 9u * (unsigned)Cin * (unsigned)p.N * 4u);
    // U' fragments: [n-block][xi-row 3][k-step][j 3][64 lanes][4]
    const unsigned soff_nb = 3u * (unsigned)S * 3072u;
    const unsigned soff_u0 = (unsigned)((n0 >> 5) * 3 + wave) * (unsigned)S * 3072u;

    // epilogue operands requested before the K loop (see wino.hip)
    const bool pre = p.splits == 1;
    float4 scv[NB][2], shv[NB][2];                              // this thread's items: tid and tid + 192 (of 256 per channel block)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int item = tid + it * 192;
            const int n = n0 + nb * 32 + (item & 7) * 4;
            scv[nb][it] = make_float4(1.f, 1.f, 1.f, 1.f); shv[nb][it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pre && p.scale && item < 256) {
                scv[nb][it] = *reinterpret_cast<const float4 *>(p.scale + n);
                shv[nb][it] = *reinterpret_cast<const float4 *>(p.shift + n);
            }
        }

    f32x16 acc[3][NB];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][nb][e] = 0.f;

    switch (wave) {
    case 0: winoup_loop<NB, 0>(p, acc, smem_c, lds0, wave, lane, vraw, srd0, srd1, srd_u, soff_u0, soff_nb, ks_begin, ks_end); break;
    case 1: winoup_loop<NB, 1>(p, acc, smem_c, lds0, wave, lane, vraw, srd0, srd1, srd_u, soff_u0, soff_nb, ks_begin, ks_end); break;
    default: winoup_loop<NB, 2>(p, acc, smem_c, lds0, wave, lane, vraw, srd0, srd1, srd_u, soff_u0, soff_nb, ks_begin, ks_end); break;
    }

    // ---- output transform.  Columns in registers: Z[w][0] = M0 + M1, Z[w][1] = M1 - M3; rows across the three waves through LDS:
    // Y[0][b] = Z(row 0) + Z(row 1), Y[1][b] = Z(row 1) - Z(row 3).
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
                pz0[row * EP + ccol] = acc[0][nb][e] + acc[1][nb][e];
                pz1[row * EP + ccol] = acc[1][nb][e] - acc[2][nb][e];
            }
        }
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t slab_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.partial, 0, p.splits > 1 ? (int)p.slab_bytes : 0, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, WT ? (int)((size_t)p.B * 4 * p.Hs * p.Ws * p.N * 4) : 0, 0x00020000);   // <= 2 GB: launch_winoup
    const int Ho = 2 * p.Hs, Wo = 2 * p.Ws;
    const size_t npix = (size_t)p.B * Ho * Wo;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        float4 vkeep[2][4];                                      // InstanceNorm plans: this thread's finished pixels of the channel block
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int item = tid + it * 192;                     // 256 (tile, channel quad) items per channel block over 192 threads
            if (item >= 256) continue;
            const int trow = item >> 3, cq = (item & 7) * 4;
            const int n = n0 + nb * 32 + cq;
            const int oy = 2 * (Y0 + (trow >> 3)), ox = 2 * (X0 + (trow & 7));
            float4 zz[3][2];
#pragma unroll
            for (int w = 0; w < 3; ++w)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb)
                    zz[w][bb] = *reinterpret_cast<const float4 *>(smem + ((w * 2 + bb) * NB + nb) * (32 * EP) + trow * EP + cq);
            const float4 sc = scv[nb][it], sh = shv[nb][it];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    float4 v = a == 0 ? f4add(zz[0][bb], zz[1][bb]) : f4sub(zz[1][bb], zz[2][bb]);
                    const size_t pix = ((size_t)b * Ho + (size_t)(oy + a)) * Wo + (size_t)(ox + bb);
                    const size_t e = pix * p.N + n;
                    if (p.splits > 1) {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), slab_rsrc, (unsigned)(((size_t)z * npix * p.N + e) * 4), 0, 16);
                    } else {
                        v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                        if constexpr (WT) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), out_rsrc, (unsigned)(e * 4), 0, 16);
                        else *reinterpret_cast<float4 *>(p.out + e) = v;
                        vkeep[it][a * 2 + bb] = v;
                    }
                }
        }
        if (p.psum && p.splits == 1)
            winoup_stats(p, reinterpret_cast<float4 *>(smem + winoup_stats_base(NB) / 4), vkeep, tid + 192 < 256, tid,
                         ((size_t)b * (size_t)(p.tby * p.tbx) + (size_t)tbi) * p.N + (size_t)(n0 + nb * 32 + (tid & 7) * 4));
    }
    if (p.splits == 1) return;

    // ---- split-K combine inside the launch: the protocol of wino.hip / igemm.hip (write-through slabs, drain, ticket, last arriver sums in z order)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned *flag = reinterpret_cast<unsigned *>(smem);
    const unsigned tile = (unsigned)(tb * p.nng + ng);
    if (tid == 0) flag[0] = __hip_atomic_fetch_add(p.tile_cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (flag[0] != (unsigned)p.splits - 1u) return;
    if (tid == 0) __hip_atomic_store(p.tile_cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        float4 vkeep2[2][4];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int item = tid + it * 192;
            if (item >= 256) continue;
            const int trow = item >> 3, cq = (item & 7) * 4;
            const int n = n0 + nb * 32 + cq;
            const int oy = 2 * (Y0 + (trow >> 3)), ox = 2 * (X0 + (trow & 7));
            float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.scale) {
                sc = *reinterpret_cast<const float4 *>(p.scale + n);
                sh = *reinterpret_cast<const float4 *>(p.shift + n);
            }
#pragma unroll
            for (int ab = 0; ab < 4; ++ab) {
                const size_t pix = ((size_t)b * Ho + (size_t)(oy + (ab >> 1))) * Wo + (size_t)(ox + (ab & 1));
                const size_t e = pix * p.N + n;
                float4 tsl[8];
#pragma unroll
                for (int sl = 0; sl < 8; ++sl)
                    if (sl < p.splits)
                        tsl[sl] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(slab_rsrc, (unsigned)(((size_t)sl * npix * p.N + e) * 4), 0, 16));
                float4 v = tsl[0];
#pragma unroll
                for (int sl = 1; sl < 8; ++sl)
                    if (sl < p.splits) { v.x += tsl[sl].x; v.y += tsl[sl].y; v.z += tsl[sl].z; v.w += tsl[sl].w; }
                v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if constexpr (WT) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), out_rsrc, (unsigned)(e * 4), 0, 16);
                else *reinterpret_cast<float4 *>(p.out + e) = v;
                vkeep2[it][ab] = v;
            }
        }
        if (p.psum)
            winoup_stats(p, reinterpret_cast<float4 *>(smem + winoup_stats_base(NB) / 4), vkeep2, tid + 192 < 256, tid,
                         ((size_t)b * (size_t)(p.tby * p.tbx) + (size_t)tbi) * p.N + (size_t)(n0 + nb * 32 + (tid & 7) * 4));
    }
}

bool winoup_supported(const WinoUpParams &p, int nb)
{
    if (nb != 1 && nb != 2) return false;
    if (p.B < 1 || p.Hs % 4 || p.Ws % 8 || p.C0 % 8 || p.C0 < 8 || (p.C1 != 0 && p.C1 != p.C0) || p.N % (32 * nb)) return false;
    const size_t lim = 0x7fffffffull;
    if ((size_t)p.B * p.Hs * p.Ws * p.C0 * 4 > lim || (size_t)9 * (p.C0 + p.C1) * p.N * 4 > lim) return false;
    if (p.splits < 1 || p.splits > 8) return false;
    if (p.splits > 1 && (!p.partial || !p.tile_cnt || (size_t)p.splits * p.B * 4 * p.Hs * p.Ws * p.N * 4 > lim)) return false;
    return true;
}

template <int NB, bool WT = false>
static hipError_t launch_winoup_t(const WinoUpParams &q, hipStream_t s)
{
    constexpr int smem = winoup_lds_bytes(NB);
    static AttrMask attr_mask;
    if (smem > 64 * 1024 && attr_needed_on_this_device(attr_mask)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&winoup3x3<NB, WT>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        attr_done_on_this_device(attr_mask);
    }
    hipLaunchKernelGGL((winoup3x3<NB, WT>), dim3((unsigned)(q.ntb * q.nng * q.splits)), dim3(192), smem, s, q);
    return hipGetLastError();
}

hipError_t launch_winoup(const WinoUpParams &p_in, int nb, hipStream_t s)
{
    if (!winoup_supported(p_in, nb)) return hipErrorInvalidValue;
    WinoUpParams p = p_in;
    const int S = (p.C0 + p.C1) / 8;
    p.steps_per_split = (S + p.splits - 1) / p.splits;
    if ((p.splits - 1) * p.steps_per_split >= S) return hipErrorInvalidValue;
    p.tby = p.Hs / 4; p.tbx = p.Ws / 8;
    p.ntb = p.B * p.tby * p.tbx;
    p.nng = p.N / (32 * nb);
    if (p.splits > 1) p.slab_bytes = (size_t)p.splits * p.B * 4 * p.Hs * p.Ws * p.N * 4;
    const size_t act = (size_t)p.B * p.Hs * p.Ws * (p.C0 + p.C1) * 4, wgt = (size_t)9 * (p.C0 + p.C1) * p.N * 4;
    p.nmajor = wgt > act ? 1 : 0;
    p.div_plane = FastDiv::make((unsigned)(p.ntb * p.nng));
    p.div_fast = FastDiv::make((unsigned)(p.nmajor ? p.ntb : p.nng));
    p.div_tbf = FastDiv::make((unsigned)(p.tby * p.tbx));
    p.div_tbx = FastDiv::make((unsigned)p.tbx);
    if (p.out_wt && (size_t)p.B * 4 * p.Hs * p.Ws * p.N * 4 <= 0x7fffffffull) return nb == 2 ? launch_winoup_t<2, true>(p, s) : launch_winoup_t<1, true>(p, s);
    return nb == 2 ? launch_winoup_t<2>(p, s) : launch_winoup_t<1>(p, s);
}

// Host: OIHW [N][C][3][3] (C = C0 + C1 in concat order) -> U' = c_i c_j (G g G^T)[i][j] for i, j in {0, 1, 3}, c = (1, 2, 1), in the MFMA
// fragment order [n-block N/32][xi-row 3][k-step C/8][j 3][lane 64][4] (lane l: output channel 32 nblock + (l & 31), input channels 8 s + 4 (l >> 5) + 0..3)
void pack_winoup_weights(const float *oihw, int cin, int cout, float *out)
{
    static const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    static const int idx[3] = {0, 1, 3};
    static const double cs[3] = {1.0, 2.0, 1.0};
    const int S = cin / 8;
    for (int n = 0; n < cout; ++n)
        for (int c = 0; c < cin; ++c) {
            const float *g = oihw + ((size_t)n * cin + c) * 9;
            double tmp[4][3];
            for (int i = 0; i < 4; ++i)
                for (int b = 0; b < 3; ++b) tmp[i][b] = G[i][0] * (double)g[b] + G[i][1] * (double)g[3 + b] + G[i][2] * (double)g[6 + b];
            const int nblk = n >> 5, s = c >> 3, lane = (n & 31) + 32 * ((c & 7) >> 2), t = c & 3;
            for (int wi = 0; wi < 3; ++wi)
                for (int wj = 0; wj < 3; ++wj) {
                    const int i = idx[wi], j = idx[wj];
                    const double u = tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2];
                    out[(((((size_t)nblk * 3 + wi) * S + s) * 3 + wj) * 64 + lane) * 4 + t] = (float)(cs[wi] * cs[wj] * u);
                }
        }
}

}  // namespace lspf2f
