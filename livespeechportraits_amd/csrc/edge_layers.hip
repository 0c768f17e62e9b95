// gfx950: the two layers at the API edge.  First: cat([feature_map, cand_image]) -> Conv 3x3 s2 -> ReLU reading
// the NCHW fp32 API tensors; last: Upsample x2 + Conv 3x3 over the concat + tanh (+ fused util.tensor2im) writing
// NCHW fp32 and/or HWC uint8.  DESIGN.md section 4.3.
#include "device_common.h"
#include "kernels.h"

#include <cstdlib>
#include <type_traits>

namespace lspf2f {

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
    if (p.bias)
#pragma unroll
        for (int j = 0; j < 32; ++j) acc[j] += p.bias[co0 + j];
    T *o = static_cast<T *>(p.out) + (size_t)gid * p.Cout + co0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        store4(o + 4 * j, p.relu ? make_float4(fmaxf(acc[4 * j], 0.f), fmaxf(acc[4 * j + 1], 0.f),
                                               fmaxf(acc[4 * j + 2], 0.f), fmaxf(acc[4 * j + 3], 0.f))
                                 : make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]));
}

// Feature-map-only pass of the first layer (the candidate share comes in through `base`): pure streaming, ~1 FLOP/byte.  16 lanes share a pixel and own 4 output
// channels each, so a wave reads and writes 4 pixels x Cout*4 B contiguously (Cout = 64: exactly 1 KB per instruction); the lane's weights stay in registers.
// Round 5: a workgroup owns ONE output row segment of 64 pixels and walks the FRAMES inside.  (a) The 9 tap values come from LDS: the segment's 3 x 130 input
// window of up to 8 frames is staged with coalesced loads (zeros where the image ends, so no branches later) and read back as 4-address broadcasts -- as global loads
// every tap was a wave-wide instruction fetching 4 useful dwords, 1.18 M of them at 8 frames, and the kernel sat on the vector-memory ADDRESS path (44.7 us bf16 /
// 49.5 us fp32 at 8 frames whatever the byte count: profiles/r05_before_kernel_stats_normal_b8_bf16.txt).  (b) The candidate share `base` (fp32, one frame's worth) is
// read once per pixel, not once per frame.  Per output the operation order is unchanged (base, then the taps in order): bit-identical to the kernel it replaces.
template <typename T, int FN>
__global__ __launch_bounds__(256) void first_conv_feat(const FirstConvParams p)
{
    constexpr int SEG = 64, FB = 8, TW = 2 * SEG + 2;        // pixels per segment, frames per staging round, staged columns (2 ox0 - 1 .. 2 ox0 + 2 SEG)
    __shared__ float taps[FB][FN][3][TW];
    const int Ho = p.H / 2, Wo = p.W / 2;
    const int lpp = p.Cout / 4;                              // lanes per pixel (16 for ngf 64, 8 for ngf 32)
    const int ppw = 64 / lpp;                                // pixels per wave and step
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane % lpp, sub = lane / lpp;
    float4 w[FN][9];
#pragma unroll
    for (int ci = 0; ci < FN; ++ci)
#pragma unroll
        for (int t = 0; t < 9; ++t)
            w[ci][t] = *reinterpret_cast<const float4 *>(p.w + (size_t)(ci * 9 + t) * p.Cout + j * 4);
    const int segs = (Wo + SEG - 1) / SEG;
    const int oy = blockIdx.x / segs, ox0 = (blockIdx.x - oy * segs) * SEG;
    const size_t plane = (size_t)p.H * p.W;
    const long hw = (long)Ho * Wo;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bv = *reinterpret_cast<const float4 *>(p.bias + j * 4);
    for (int b0 = 0; b0 < p.B; b0 += FB) {
        const int nb = p.B - b0 < FB ? p.B - b0 : FB;
        if (b0) __syncthreads();                              // the previous round's readers are done
        for (int i = tid; i < nb * FN * 3 * TW; i += 256) {
            const int col = i % TW, row = (i / TW) % 3, ci = (i / (3 * TW)) % FN, fb = i / (3 * TW * FN);
            const int iy = 2 * oy + row - 1, ix = 2 * ox0 - 1 + col;
            const bool ok = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
            taps[fb][ci][row][col] = ok ? p.feat[((size_t)(b0 + fb) * p.feat_nc + ci) * plane + (size_t)iy * p.W + ix] : 0.f;
        }
        __syncthreads();
        for (int px = wave * ppw + sub; px < SEG; px += 4 * ppw) {
            const int ox = ox0 + px;
            if (ox >= Wo) break;
            const long r = (long)oy * Wo + ox;
            const float4 base4 = *reinterpret_cast<const float4 *>(p.base + (size_t)r * p.Cout + j * 4);
#pragma unroll 2
            for (int fb = 0; fb < nb; ++fb) {
                float4 acc = base4;
#pragma unroll
                for (int ci = 0; ci < FN; ++ci)
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const float v = taps[fb][ci][t / 3][2 * px + t % 3];
                        acc.x += v * w[ci][t].x; acc.y += v * w[ci][t].y; acc.z += v * w[ci][t].z; acc.w += v * w[ci][t].w;
                    }
                if (p.bias) { acc.x += bv.x; acc.y += bv.y; acc.z += bv.z; acc.w += bv.w; }
                if (p.relu) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
                store4(static_cast<T *>(p.out) + ((size_t)(b0 + fb) * hw + r) * p.Cout + j * 4, acc);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// First layer on the matrix cores.  The direct kernel above issues 0.98 GFLOP per frame on the vector ALU behind LDS weight
// broadcasts and sits at 36 us per frame (10 % of the HBM roofline of its 30 MB); as an implicit GEMM -- M = output pixels, N = Cout,
// K = 13 x 9 = 117 (padded to 118) -- the arithmetic is 6 us of v_mfma_f32_32x32x2_f32 and the layer becomes a streaming kernel.
//   * tile = 2 output rows x 64 columns (128 pixels, one 32-pixel run per wave); its 5 x 129 input window of every channel is
//     staged from the two NCHW API tensors into LDS once (zero padding written there: no branches later), the 117 x Cout weights too;
//   * A fragment: lane (pixel m, k parity) reads ONE float at [ci][2 r + ky][2 m + kx] -- a stride-2 ds_read_b32 (2-way conflict,
//     4 cycles against 64 x NH cycles of MFMA per K step); B fragment: weights [k][n], conflict-free;
//   * epilogue straight from the accumulators: for a fixed register the 32 lanes of a half-wave hold 32 consecutive channels of
//     one pixel -> 128-byte NHWC segments; + bias (InstanceNorm plans), ReLU, fp32 or bf16 store.
// Channel range [ci_begin, ci_end) as in the direct kernel (candidate-share pass of lspf2f_set_candidates: weights outside the
// range are zeroed in LDS, their inputs never read).
template <typename T, int NH>
__global__ __launch_bounds__(256) void first_conv_mfma(const FirstConvParams p)
{
    constexpr int CIN = 13, KS = (CIN * 9 + 1) / 2;          // 59 K steps of 2
    constexpr int LDW = 132;                                  // LDS row pitch of the input window (129 used)
    constexpr int N = NH * 32;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *win = sm;                                          // [13][5][LDW]
    float *wl = sm + CIN * 5 * LDW;                           // [118][N]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Ho = p.H / 2, Wo = p.W / 2;
    const int tiles_x = (Wo + 63) / 64, tiles_y = (Ho + 1) / 2;
    int t = blockIdx.x;
    const int b = t / (tiles_x * tiles_y);
    t -= b * tiles_x * tiles_y;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int oy0 = ty * 2, ox0 = tx * 64;

    // weights: blob layout [(ci * 9 + tap)][Cout]; rows outside [ci_begin, ci_end) and the pad row 117 are zero.  All loads of a
    // thread are issued before its LDS writes (a load -> write loop would be a chain of serial round trips).
    {
        constexpr int W4 = 2 * KS * N / 4, PER = (W4 + 255) / 256;
        float4 v[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = tid + u * 256, k = (i * 4) / N, ci = k / 9;
            const bool ok = i < W4 && k < CIN * 9 && ci >= p.ci_begin && ci < p.ci_end;
            v[u] = ok ? *reinterpret_cast<const float4 *>(p.w + (size_t)k * p.Cout + (i * 4 - k * N)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < PER; ++u)
            if (tid + u * 256 < W4) *reinterpret_cast<float4 *>(wl + (tid + u * 256) * 4) = v[u];
    }
    // input window: rows 2 oy0 - 1 .. + 3, columns 2 ox0 - 1 .. + 127 of every channel in range (others: zeros).  A wave takes the
    // (channel, row) lines q = wave, wave + 4, ...; a line is 129 floats = lanes 0..63 twice plus one
    {
        const size_t plane = (size_t)p.H * p.W;
        constexpr int LINES = CIN * 5, PERW = (LINES + 3) / 4;         // 17 lines per wave
        float v[PERW][3];
#pragma unroll
        for (int j = 0; j < PERW; ++j) {
            const int q = wave + 4 * j, ci = q / 5, r = q - ci * 5;
            const int iy = 2 * oy0 - 1 + r;
            const bool rok = q < LINES && ci >= p.ci_begin && ci < p.ci_end && (unsigned)iy < (unsigned)p.H;
            const float *src = nullptr;
            if (rok)
                src = (ci < p.feat_nc ? p.feat + ((size_t)b * p.feat_nc + ci) * plane
                                      : p.cand + ((size_t)(p.cand_batch == 1 ? 0 : b) * p.cand_nc + (ci - p.feat_nc)) * plane) + (size_t)iy * p.W;
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int c = lane + 64 * u, ix = 2 * ox0 - 1 + c;
                v[j][u] = (rok && c < 129 && (unsigned)ix < (unsigned)p.W) ? src[ix] : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < PERW; ++j) {
            const int q = wave + 4 * j;
            if (q < LINES)
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    if (lane + 64 * u < 129) win[q * LDW + lane + 64 * u] = v[j][u];
        }
    }
    __syncthreads();

    // wave -> 32 pixels: output row oy0 + (wave >> 1), columns ox0 + 32 (wave & 1) + m
    const int m = lane & 31, kp = lane >> 5;
    const int orow = wave >> 1, ocol = 32 * (wave & 1) + m;
    const float *abase = win + (2 * orow) * LDW + 2 * ocol;  // + (ci * 5 + ky) * LDW + kx
    f32x16 acc[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[h][r] = 0.f;
    // k = 2 s + kp -> (ci, ky, kx): both parities' window offsets are compile-time constants, the lane picks its own.  Operands of
    // step s + 1 are read from LDS before the MFMAs of step s are issued (pinned with scheduling barriers: the compiler otherwise
    // sinks each read to its use and every K step pays the LDS latency).
    auto a_off = [&](int s2) {
        const int k0 = 2 * s2, k1 = 2 * s2 + 1;
        const int o0 = ((k0 / 9) * 5 + (k0 % 9) / 3) * LDW + (k0 % 9) % 3;
        const int o1 = k1 < CIN * 9 ? ((k1 / 9) * 5 + (k1 % 9) / 3) * LDW + (k1 % 9) % 3 : 0;   // k = 117: weight row is zero
        return kp ? o1 : o0;
    };
    float a_cur = abase[a_off(0)], a_nxt = 0.f;
    float b_cur[NH], b_nxt[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) { b_cur[h] = wl[kp * N + h * 32 + m]; b_nxt[h] = 0.f; }
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) {
        if (s2 + 1 < KS) {
            a_nxt = abase[a_off(s2 + 1)];
#pragma unroll
            for (int h = 0; h < NH; ++h) b_nxt[h] = wl[(2 * (s2 + 1) + kp) * N + h * 32 + m];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < NH; ++h) acc[h] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur[h], acc[h], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        a_cur = a_nxt;
#pragma unroll
        for (int h = 0; h < NH; ++h) b_cur[h] = b_nxt[h];
    }
    // epilogue: C/D layout row (pixel) = (r & 3) + 8 (r >> 2) + 4 kp, col (channel) = m
    const int oy = oy0 + orow;
    if (oy >= Ho) return;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const int n = h * 32 + m;
        const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int px = 32 * (wave & 1) + (r & 3) + 8 * (r >> 2) + 4 * kp;
            const int ox = ox0 + px;
            if (ox >= Wo) continue;
            float v = acc[h][r] + bias;
            if (p.relu) v = fmaxf(v, 0.f);
            const size_t o = (((size_t)b * Ho + oy) * Wo + ox) * p.Cout + n;
            st1(static_cast<T *>(p.out) + o, v);
        }
    }
}


// The same layer with both operands staged by LDS-DMA (the shape feature2face_G.py builds: 1 feature-map channel + 12 candidate channels,
// W % 128 == 0, H % 4 == 0, Cout = 32 NH).  The kernel above spends ~1550 vector instructions per thread on staging (address arithmetic, 59
// loads into registers, 59 LDS writes) before its first MFMA; here a thread issues 17 copies and nothing passes through registers:
//   * the window is kept in 16-byte columns, x = 2 ox0 - 4 .. 2 ox0 + 127 (33 float4 per line: every global read is aligned, the left pad of
//     the first tile column and the rows above / below the image are out-of-range copies, which land as zeros), as two runs of whole 1-KB
//     pieces -- feature-map lines (5 x 33 slots in 3 pieces), candidate lines (60 x 33 slots in 31 pieces) -- because a piece has ONE
//     descriptor and the two tensors are separate allocations;
//   * the weights [117][N] are 30 pieces of the blob as it lies; rows outside [ci_begin, ci_end) and the pad row 117 are out-of-range lanes;
//   * piece P = 4 j + wave, so the window lands in channel order and the K loop starts on channels 0..3 while 4..8 and 9..12 are still in
//     flight (three counted waits + barriers instead of one);
//   * epilogue addresses are one base pointer + compile-time offsets.
#ifdef LSPF2F_ABLATE
#define FABL(p, bit) ((p).dbg & (bit))
#else
#define FABL(p, bit) 0
#endif
template <typename T, int NH>
__global__ __launch_bounds__(256) void first_conv_dma(const FirstConvParams p)
{
    constexpr int CIN = 13, KS = (CIN * 9 + 1) / 2;          // 59 K steps of 2
    constexpr int LW = 132;                                   // floats per window line (33 float4)
    constexpr int N = NH * 32;
    constexpr int WPIECES = (118 * N * 4 + 1023) / 1024;      // weight pieces (30 at N = 64)
    constexpr int WJ = (WPIECES + 3) / 4;                     // per wave
    constexpr int FPIECES = 3, CPIECES = 31, XJ = 9;          // window pieces: 3 + 31 = 34 -> 9 per wave (the last two are dummies)
    constexpr int WIN0 = WPIECES * 1024;                      // LDS byte offset of the window
    constexpr int DUMP = WIN0 + (FPIECES + CPIECES) * 1024;   // 1 KB: where the dummy pieces of waves that have fewer real ones go
    constexpr int CAND0 = FPIECES * 256;                      // float offset of the candidate lines inside the window
    extern __shared__ __attribute__((aligned(16))) float sm[];
    typedef __attribute__((address_space(3))) float lds_float;
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_float *)sm;
    const float *wl = sm;                                     // [118][N]
    const float *win = sm + WIN0 / 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Ho = p.H / 2, Wo = p.W / 2;
    const int tiles_x = Wo / 64, tiles_y = Ho / 2;
    int t = blockIdx.x;
    const int b = t / (tiles_x * tiles_y);
    t -= b * tiles_x * tiles_y;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int oy0 = ty * 2, ox0 = tx * 64;
    constexpr unsigned OOB = 0x80000000u;

    // ---- weights
    {
        const i32x4 srd = make_srd(p.w, (unsigned)(117 * N * 4));
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            const int P = 4 * j + wave;
            const int slot = 64 * P + lane, k = (slot * 4) / N, ci = k / 9;
            const bool ok = P < WPIECES && k < CIN * 9 && ci >= p.ci_begin && ci < p.ci_end && !FABL(p, 8);
            const unsigned voff[1] = {ok ? (unsigned)slot * 16u : OOB};
            dma16_group<1, 0>(lds0 + (unsigned)(P < WPIECES ? P * 1024 : DUMP), voff, srd, 0);
        }
    }
    // ---- window
    {
        const size_t plane = (size_t)p.H * p.W;
        const i32x4 srd_f = make_srd(p.feat + (size_t)b * p.feat_nc * plane, (unsigned)(plane * 4));
        const i32x4 srd_c = make_srd(p.cand ? p.cand + (size_t)(p.cand_batch == 1 ? 0 : b) * p.cand_nc * plane : p.feat, (unsigned)(p.cand ? 12 * plane * 4 : 0));
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
            const int P = 4 * j + wave;
            const bool isf = P < FPIECES;                                        // wave-uniform
            const int s = isf ? 64 * P + lane : 64 * (P - FPIECES) + lane;
            const int q = s / 33, c4 = s - q * 33;
            const int cl = isf ? 0 : q / 5, r = isf ? q : q - cl * 5;            // channel inside its tensor, window row
            const int ci = isf ? 0 : 1 + cl;
            const int iy = 2 * oy0 - 1 + r, x0 = 2 * ox0 - 4 + 4 * c4;
            const bool ok = P < FPIECES + CPIECES && s < (isf ? 5 * 33 : 60 * 33) && ci >= p.ci_begin && ci < p.ci_end &&
                            (unsigned)iy < (unsigned)p.H && x0 >= 0 && !FABL(p, 4);
            const unsigned voff[1] = {ok ? (unsigned)(((cl * p.H + iy) * p.W + x0) * 4) : OOB};
            const unsigned dst = lds0 + (unsigned)(P < FPIECES + CPIECES ? WIN0 + P * 1024 : DUMP);
            if (isf) dma16_group<1, 0>(dst, voff, srd_f, 0);
            else dma16_group<1, 0>(dst, voff, srd_c, 0);
        }
    }

    // wave -> 32 pixels: output row oy0 + (wave >> 1), columns ox0 + 32 (wave & 1) + m
    const int m = lane & 31, kp = lane >> 5;
    const int orow = wave >> 1, ocol = 32 * (wave & 1) + m;
    const float *abase = win + (2 * orow) * LW + 2 * ocol + 3;       // + line(ci, ky) * LW + kx; column 3 of the window is x = 2 ox0 - 1
    f32x16 acc[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[h][r] = 0.f;
    auto a_off = [&](int s2) {
        auto off = [](int k) { const int ci = k / 9, tap = k % 9; return (ci == 0 ? 0 : CAND0 + (ci - 1) * 5 * LW) + (tap / 3) * LW + tap % 3; };
        const int k0 = 2 * s2, k1 = 2 * s2 + 1;
        const int o0 = off(k0), o1 = k1 < CIN * 9 ? off(k1) : 0;   // k = 117: the weight row is zero
        return kp ? o1 : o0;
    };
    // K steps [S0, S1): operands of step s + 1 are read before the MFMAs of step s are issued, inside the range only (what lies past it may
    // not have landed yet)
    auto ksteps = [&](auto s0c, auto s1c) {
        constexpr int S0 = decltype(s0c)::value, S1 = decltype(s1c)::value;
        float a_cur = abase[a_off(S0)], a_nxt = 0.f;
        float b_cur[NH], b_nxt[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) { b_cur[h] = wl[(2 * S0 + kp) * N + h * 32 + m]; b_nxt[h] = 0.f; }
#pragma unroll
        for (int s2 = S0; s2 < S1; ++s2) {
            if (s2 + 1 < S1) {
                a_nxt = abase[a_off(s2 + 1)];
#pragma unroll
                for (int h = 0; h < NH; ++h) b_nxt[h] = wl[(2 * (s2 + 1) + kp) * N + h * 32 + m];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (FABL(p, 1)) { acc[0][s2 & 15] += a_cur * b_cur[0]; if (NH > 1) acc[NH - 1][s2 & 15] += a_cur * b_cur[NH - 1]; }      // (keeps the operand reads alive)
            else {
#pragma unroll
                for (int h = 0; h < NH; ++h) acc[h] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur[h], acc[h], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            a_cur = a_nxt;
#pragma unroll
            for (int h = 0; h < NH; ++h) b_cur[h] = b_nxt[h];
        }
    };
    // window pieces of a wave land in issue order: j <= 2 covers channels 0..3 (slots < 12 * 64), j <= 5 channels 0..8, j <= 8 everything
    dma_wait<6>();
    __syncthreads();
    ksteps(std::integral_constant<int, 0>{}, std::integral_constant<int, 18>{});       // k < 36
    dma_wait<3>();
    __syncthreads();
    ksteps(std::integral_constant<int, 18>{}, std::integral_constant<int, 40>{});      // k < 80 (k = 80 is the last tap of channel 8: next range)
    dma_wait<0>();
    __syncthreads();
    ksteps(std::integral_constant<int, 40>{}, std::integral_constant<int, KS>{});

    // epilogue: C/D layout row (pixel) = (r & 3) + 8 (r >> 2) + 4 kp, col (channel) = m
    const int oy = oy0 + orow;
    T *ob = static_cast<T *>(p.out) + (((size_t)b * Ho + oy) * Wo + ox0 + 32 * (wave & 1) + 4 * kp) * N + m;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const float bias = p.bias ? p.bias[h * 32 + m] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[h][r] + bias;
            if (p.relu) v = fmaxf(v, 0.f);
            if (FABL(p, 2) && v != 12345.678f) continue;      // (ablation: no stores, the value stays live)
            st1(ob + ((r & 3) + 8 * (r >> 2)) * N + h * 32, v);
        }
    }
}

template <typename T, int NH>
static hipError_t launch_first_conv_dma(const FirstConvParams &p, hipStream_t s)
{
    const int Ho = p.H / 2, Wo = p.W / 2;
    const long tiles = (long)p.B * (Wo / 64) * (Ho / 2);
    const size_t smem = (size_t)((118 * NH * 32 * 4 + 1023) / 1024 + 34 + 1) * 1024;
    static AttrMask attr_mask;
    if (smem > 64 * 1024 && attr_needed_on_this_device(attr_mask)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&first_conv_dma<T, NH>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        attr_done_on_this_device(attr_mask);
    }
    hipLaunchKernelGGL((first_conv_dma<T, NH>), dim3((unsigned)tiles), dim3(256), smem, s, p);
    return hipGetLastError();
}

template <typename T, int NH>
static hipError_t launch_first_conv_mfma(const FirstConvParams &p, hipStream_t s)
{
    const int Ho = p.H / 2, Wo = p.W / 2;
    const long tiles = (long)p.B * ((Wo + 63) / 64) * ((Ho + 1) / 2);
    const size_t smem = (size_t)(13 * 5 * 132 + 118 * NH * 32) * sizeof(float);
    static AttrMask attr_mask;
    if (smem > 64 * 1024 && attr_needed_on_this_device(attr_mask)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&first_conv_mfma<T, NH>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        attr_done_on_this_device(attr_mask);
    }
    hipLaunchKernelGGL((first_conv_mfma<T, NH>), dim3((unsigned)tiles), dim3(256), smem, s, p);
    return hipGetLastError();
}

hipError_t launch_first_conv(const FirstConvParams &p, hipStream_t s)
{
    if (p.base && p.ci_begin == 0 && p.ci_end == 1 && p.feat_nc == 1 && 64 % (p.Cout / 4) == 0 && p.Cout <= 256) {
        // one workgroup per output row segment of 64 pixels (it walks the frames inside): 1024 workgroups at 512x512
        const long blocks = (long)(p.H / 2) * ((p.W / 2 + 63) / 64);
        if (p.dtype == 2) hipLaunchKernelGGL((first_conv_feat<f16_t, 1>), dim3((unsigned)blocks), dim3(256), 0, s, p);
        else if (p.dtype == 1) hipLaunchKernelGGL((first_conv_feat<bf16_t, 1>), dim3((unsigned)blocks), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((first_conv_feat<float, 1>), dim3((unsigned)blocks), dim3(256), 0, s, p);
        return hipGetLastError();
    }
    // matrix-core form: no partial sums to start from, 13 input channels (the only count feature2face_G.py builds), Cout a multiple
    // of 32 up to 128; bf16 output only for the layer itself (the candidate cache is fp32)
    const bool cache_pass = p.relu == 0 && p.base == nullptr && p.ci_begin > 0;
    if (!p.base && p.force_direct != 1 && p.feat_nc + p.cand_nc == 13 && (p.Cout == 32 || p.Cout == 64 || p.Cout == 128)) {
        const int st = cache_pass ? 0 : p.dtype;            // storage type of what this pass writes
        // both operands by LDS-DMA for the shape the reference builds (force_direct == 2: the register-staged kernel, A-B runs)
        if (p.force_direct != 2 && p.feat_nc == 1 && p.cand_nc == 12 && p.W % 128 == 0 && p.H % 4 == 0 && p.Cout <= 64 &&
            (size_t)12 * p.H * p.W * 4 < 0x80000000ull) {
            if (p.Cout == 32) return st == 2 ? launch_first_conv_dma<f16_t, 1>(p, s) : st == 1 ? launch_first_conv_dma<bf16_t, 1>(p, s) : launch_first_conv_dma<float, 1>(p, s);
            return st == 2 ? launch_first_conv_dma<f16_t, 2>(p, s) : st == 1 ? launch_first_conv_dma<bf16_t, 2>(p, s) : launch_first_conv_dma<float, 2>(p, s);
        }
        switch (p.Cout / 32) {
        case 1: return st == 2 ? launch_first_conv_mfma<f16_t, 1>(p, s) : st == 1 ? launch_first_conv_mfma<bf16_t, 1>(p, s) : launch_first_conv_mfma<float, 1>(p, s);
        case 2: return st == 2 ? launch_first_conv_mfma<f16_t, 2>(p, s) : st == 1 ? launch_first_conv_mfma<bf16_t, 2>(p, s) : launch_first_conv_mfma<float, 2>(p, s);
        default: return st == 2 ? launch_first_conv_mfma<f16_t, 4>(p, s) : st == 1 ? launch_first_conv_mfma<bf16_t, 4>(p, s) : launch_first_conv_mfma<float, 4>(p, s);
        }
    }
    const long total = (long)p.B * (p.H / 2) * (p.W / 2);
    const int K = (p.ci_end - p.ci_begin) * 9;
    if (p.dtype == 2 && p.out != nullptr && !(p.relu == 0 && p.base == nullptr && p.ci_begin > 0))
        hipLaunchKernelGGL(first_conv<f16_t>, dim3((unsigned)((total + 255) / 256), p.Cout / 32), dim3(256),
                           (size_t)K * 32 * sizeof(float), s, p);
    else if (p.dtype == 1 && p.out != nullptr && !(p.relu == 0 && p.base == nullptr && p.ci_begin > 0))
        hipLaunchKernelGGL(first_conv<bf16_t>, dim3((unsigned)((total + 255) / 256), p.Cout / 32), dim3(256),
                           (size_t)K * 32 * sizeof(float), s, p);
    else   // fp32 activations, or the candidate-share pass (its cache is always fp32)
        hipLaunchKernelGGL(first_conv<float>, dim3((unsigned)((total + 255) / 256), p.Cout / 32), dim3(256),
                           (size_t)K * 32 * sizeof(float), s, p);
    return hipGetLastError();
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
        const float pre = acc[co] + (p.bias ? p.bias[co] : 0.f);
        const float v = p.apply_tanh ? tanhf(pre) : pre;
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
            if (p.bias) r += p.bias[j];
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
        // accumulate in float2 lanes so that the dot products compile to v_pk_fma_f32 (2 FMAs per VALU slot);
        // the pair is summed once at the end
        v2f acc2[4][CO];
#pragma unroll
        for (int par = 0; par < 4; ++par)
#pragma unroll
            for (int co = 0; co < CO; ++co) acc2[par][co] = v2f{0.f, 0.f};
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
                        acc2[par][co] += v2f{v.x, v.y} * v2f{ww.x, ww.y};
                        acc2[par][co] += v2f{v.z, v.w} * v2f{ww.z, ww.w};
                    }
                }
        }
        float acc[4][CO];
#pragma unroll
        for (int par = 0; par < 4; ++par)
#pragma unroll
            for (int co = 0; co < CO; ++co) acc[par][co] = acc2[par][co].x + acc2[par][co].y;
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
                if (p.bias) r += p.bias[j];
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


// ------------------------------------------------------------------------------------------
// Last layer on the matrix cores (fp32 plans, C0 == C1 == 64, Cout <= 4).  The vector-ALU kernels above re-read the 96 pre-summed weight
// vectors from LDS for every source pixel (the LDS pipe, not the ALU, bounds them: 185 us at batch 8, 36 us at batch 1).  Here the weights are a
// matrix operand.  v_mfma_f32_4x4x1_16b_f32 is 16 independent 4x4x1 products per instruction: block = (pixel group g of 4 neighbouring source
// pixels, output parity) -- rows = the group's 4 pixels, columns = the 3 output channels (+1 idle), K = one (tap, input channel) of that
// parity's 2x2 neighbourhood -- so a wave instruction advances 16 source pixels x 4 parities by one K step and only 1/4 of the columns idle,
// against 13/16 (N = 3 of 16) for the 16x16 shapes.  4 taps x 128 channels = 512 K steps per 16 pixels, 8 cycles each: 55 us of matrix time at
// batch 8, 7 us at batch 1, next to 42 / 5 us of HBM time for the 268 / 33.5 MB read once.
//   * workgroup = 8 x 32 source pixels; its 10 x 34 pixel halo tile is staged by LDS-DMA 32 channels at a time (4 stages: source 0 | 1, lower |
//     upper half), 128 B per pixel, two buffers.  Image borders are out-of-range copies (zeros).  A pixel's eight 16-byte quads are stored at slot
//     q ^ ((p >> 1) & 7) (p = pixel index in the tile, 40 per row): an A read is 16 lanes x 16 B of 16 DIFFERENT pixels at the same quad, and with
//     the swizzle and the row pitch of 40 (= 8 mod 16) every ds_read_b128 lane group lands on 16 distinct bank slots (tools/lastconv_model.py
//     checks all of them).  The swizzle is applied on the global side of the copy, where it permutes 16-byte pieces inside one 128-byte run.
//   * wave = 2 x 32 pixels = 4 MFMA tiles of 16; one B read (the weights of (parity, channel), broadcast over the 4 groups) serves the 4 tiles:
//     5 ds_read_b128 per 16 MFMAs.  Weights sit in LDS as [parity * 4 + channel][tap][128] with a row pitch of 516 floats (16 rows -> 16 bank slots).
//   * the accumulators (4 registers per tile: the group's 4 pixels) go through LDS once more so that the NCHW fp32 rows and the HWC uint8 rows
//     leave as full 16-byte / 4-byte coalesced stores; tanh and tensor2im as in the other kernels.
// NW = waves per workgroup: 4 (round 3: a wave = 2 tile rows = 4 MFMA tiles) or 8 (round 4: a wave = 1 tile row = 2 MFMA tiles; TWO waves per SIMD --
// a single wave per SIMD neither keeps the 2-pass v_mfma_f32_4x4x1 pipe busy (9-13 cycles per instruction measured) nor covers its own waits: profiles/r04_lastconv_ab.txt)
template <int NW>
__global__ __launch_bounds__(64 * NW) void last_conv_mfma(const LastConvParams p, int ntiles)
{
    constexpr int NT = 16 / NW;                               // MFMA tiles (16 pixels) per wave
    constexpr int NPC = (52 + NW - 1) / NW;                   // stage pieces per wave and step (50 real ones)
    constexpr int WP = 40;                                    // staged pixels per tile row (34 used)
    constexpr int LWB = 16 * 516 * 4;                         // weight bytes in LDS
    constexpr int STG = 10 * WP * 128;                        // one stage buffer
    constexpr int OTB = LWB + 2 * STG;                        // output tile [4][16][68] floats
    constexpr int DUMP = OTB + 4 * 16 * 68 * 4;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    typedef __attribute__((address_space(3))) float lds_float;
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_float *)sm;
    const char *smc = reinterpret_cast<const char *>(sm);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = p.Ws >> 5, tiles_y = p.Hs >> 3;
    const size_t frame = (size_t)p.Hs * p.Ws * 64;
    const int H = 2 * p.Hs, W = 2 * p.Ws;

    // ---- weights, once per workgroup: row (parity, n) = 512 floats = two 1-KB pieces; rows n >= Cout are out of range (zeros)
    {
        const i32x4 srd = make_srd(p.w, (unsigned)(4 * p.Cout * 512 * 4));
#pragma unroll
        for (int j = 0; j < 32 / NW; ++j) {
            const int Wp = NW * j + wave, row = Wp >> 1, par = row >> 2, n = row & 3;
            const unsigned voff[1] = {n < p.Cout ? (unsigned)((((par * p.Cout + n) * 512) + (Wp & 1) * 256 + lane * 4) * 4) : OOB};
            dma16_group<1, 0>(lds0 + (unsigned)(row * 2064 + (Wp & 1) * 1024), voff, srd, 0);
        }
    }
    // ---- stage copies of one tile: piece I = 8 consecutive tile pixels (row I / 5, columns 8 (I % 5) ..) x 8 slots; lane -> (pixel, slot), global
    // quad = slot ^ swizzle.  The copy side runs two stages ahead of the arithmetic and may already be in the workgroup's next tile.
    unsigned vst[NPC];
    i32x4 srd0, srd1;
    auto tile_origin = [&](int tile, int &b, int &y0, int &x0) {
        b = tile / (tiles_x * tiles_y);
        const int r = tile - b * tiles_x * tiles_y;
        const int ty = r / tiles_x;
        y0 = ty * 8; x0 = (r - ty * tiles_x) * 32;
    };
    auto plan_copies = [&](int tile) {
        int b, y0, x0;
        tile_origin(tile, b, y0, x0);
#pragma unroll
        for (int j = 0; j < NPC; ++j) {
            const int I = NW * j + wave;
            const int row = I / 5, cc = 8 * (I - row * 5) + (lane >> 3);
            const int pp = row * WP + cc, q = (lane & 7) ^ ((pp >> 1) & 7);
            const int y = y0 - 1 + row, x = x0 - 1 + cc;
            const bool ok = I < 50 && cc < 34 && (unsigned)y < (unsigned)p.Hs && (unsigned)x < (unsigned)p.Ws;
            vst[j] = ok ? (unsigned)(((y * p.Ws + x) * 64 + 4 * q) * 4) : OOB;
        }
        srd0 = make_srd(static_cast<const float *>(p.src0) + (size_t)b * frame, (unsigned)(frame * 4));
        srd1 = make_srd(static_cast<const float *>(p.src1) + (size_t)b * frame, (unsigned)(frame * 4));
    };
    auto issue = [&](int st, int buf) {
        const unsigned dst = lds0 + (unsigned)(LWB + buf * STG + wave * 1024);
        const int soff = (st & 1) * 128;
        const unsigned last = wave < 2 ? dst + 48 * 1024 : lds0 + (unsigned)DUMP;      // pieces 48, 49 exist; the others of that round go to the dump slot
        const i32x4 srd = st < 2 ? srd0 : srd1;
        if constexpr (NW == 4) {
            const unsigned v0[4] = {vst[0], vst[1], vst[2], vst[3]}, v1[4] = {vst[4], vst[5], vst[6], vst[7]}, v2[4] = {vst[8], vst[9], vst[10], vst[11]};
            const unsigned v3[1] = {vst[12]};
            dma16_group<4, 4096>(dst, v0, srd, soff); dma16_group<4, 4096>(dst + 16384, v1, srd, soff); dma16_group<4, 4096>(dst + 32768, v2, srd, soff);
            dma16_group<1, 0>(last, v3, srd, soff);
        } else {
            const unsigned v0[4] = {vst[0], vst[1], vst[2], vst[3]}, v1[2] = {vst[4], vst[5]}, v2[1] = {vst[6]};
            dma16_group<4, 8192>(dst, v0, srd, soff); dma16_group<2, 8192>(dst + 32768, v1, srd, soff);
            dma16_group<1, 0>(last, v2, srd, soff);
        }
    };

    // ---- operand addresses.  lane = (g, parity, i): as A the i-th pixel of group g, as B / D output channel n = i
    const int g = lane >> 4, par = (lane >> 2) & 3, mi = lane & 3;
    const int py = par >> 1, px = par & 1;
    unsigned abase[NT][4], aswz[NT][4];                      // [tile][tap]: byte offset of the pixel inside a stage buffer, its slot swizzle (<< 4)
#pragma unroll
    for (int tau = 0; tau < NT; ++tau)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int R = NW == 4 ? 2 * wave + (tau >> 1) : wave, Cb = NW == 4 ? 16 * (tau & 1) : 16 * tau;
            const int pp = (R + (t >> 1) + py) * WP + Cb + 4 * g + mi + (t & 1) + px;
            abase[tau][t] = (unsigned)(pp * 128);
            aswz[tau][t] = (unsigned)(((pp >> 1) & 7) << 4);
        }
    const unsigned brow = (unsigned)((par * 4 + mi) * 2064);
    f32x4acc acc[NT];
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) acc[tau] = f32x4acc{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int st, int buf) {
        const char *sb = smc + LWB + buf * STG;
        const char *wb = smc + brow + (unsigned)(((st >> 1) * 64 + (st & 1) * 32) * 4);
        float4 a_cur[NT], a_nxt[NT], b_cur, b_nxt;
        auto fetch = [&](int it, float4 (&a)[NT], float4 &bv) {
            const int t = it >> 3, q = it & 7;
            bv = *reinterpret_cast<const float4 *>(wb + t * 512 + q * 16);
#pragma unroll
            for (int tau = 0; tau < NT; ++tau) a[tau] = *reinterpret_cast<const float4 *>(sb + abase[tau][t] + (aswz[tau][t] ^ (unsigned)(q << 4)));
        };
#ifdef LC_ABL_NOMFMA                                         // timing builds of tools/sessions/lastconv_ablate.sh: copies only
        return;
#endif
#ifdef LC_PF2                                                // operands two iterations ahead (three register sets)
        float4 a3[3][NT], b3[3];
        fetch(0, a3[0], b3[0]);
        fetch(1, a3[1], b3[1]);
#pragma unroll
        for (int it = 0; it < 32; ++it) {
            if (it + 2 < 32) fetch(it + 2, a3[(it + 2) % 3], b3[(it + 2) % 3]);
            __builtin_amdgcn_sched_barrier(0);
            const float4 (&ac)[NT] = a3[it % 3];
            const float4 bc = b3[it % 3];
#pragma unroll
            for (int tau = 0; tau < NT; ++tau) acc[tau] = __builtin_amdgcn_mfma_f32_4x4x1f32(ac[tau].x, bc.x, acc[tau], 0, 0, 0);
#pragma unroll
            for (int tau = 0; tau < NT; ++tau) acc[tau] = __builtin_amdgcn_mfma_f32_4x4x1f32(ac[tau].y, bc.y, acc[tau], 0, 0, 0);
#pragma unroll
            for (int tau = 0; tau < NT; ++tau) acc[tau] = __builtin_amdgcn_mfma_f32_4x4x1f32(ac[tau].z, bc.z, acc[tau], 0, 0, 0);
#pragma unroll
            for (int tau = 0; tau < NT; ++tau) acc[tau] = __builtin_amdgcn_mfma_f32_4x4x1f32(ac[tau].w, bc.w, acc[tau], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
#endif
        fetch(0, a_cur, b_cur);
#pragma unroll
        for (int it = 0; it < 32; ++it) {
            if (it + 1 < 32) fetch(it + 1, a_nxt, b_nxt);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tau = 0; tau < NT; ++tau) acc[tau] = __builtin_amdgcn_mfma_f32_4x4x1f32(a_cur[tau].x, b_cur.x, acc[tau], 0, 0, 0);
#pragma unroll
            for (int tau = 0; tau < NT; ++tau) acc[tau] = __builtin_amdgcn_mfma_f32_4x4x1f32(a_cur[tau].y, b_cur.y, acc[tau], 0, 0, 0);
#pragma unroll
            for (int tau = 0; tau < NT; ++tau) acc[tau] = __builtin_amdgcn_mfma_f32_4x4x1f32(a_cur[tau].z, b_cur.z, acc[tau], 0, 0, 0);
#pragma unroll
            for (int tau = 0; tau < NT; ++tau) acc[tau] = __builtin_amdgcn_mfma_f32_4x4x1f32(a_cur[tau].w, b_cur.w, acc[tau], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tau = 0; tau < NT; ++tau) a_cur[tau] = a_nxt[tau];
            b_cur = b_nxt;
        }
    };

    // ---- steps n = (tile, stage): copy of step n + 2 goes out when step n's buffer is free.  A wave's copies land in issue order: weights (8), then
    // NPC per step, so "at most NPC in flight" = step n has landed.
    float *ot = sm + OTB / 4;
    const int nmine = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;       // tiles blockIdx.x, + gridDim.x, ...
    const int nsteps = 4 * nmine;
    int itile = blockIdx.x;                                   // the copy side's tile
    plan_copies(itile);
    issue(0, 0);
    issue(1, 1);
    int ist = 2;                                              // next stage the copy side issues
    int ctile = blockIdx.x;
    for (int n = 0; n < nsteps; ++n) {
        const int st = n & 3, buf = n & 1;
        if (n + 1 < nsteps) dma_wait<NPC>(); else dma_wait<0>();
        __syncthreads();
        compute(st, buf);
        if (st == 3 && mi < p.Cout) {
            // acc[tile][r] = output (row 2 R + py, column 2 (Cb + 4 g + r) + px) of channel mi -> LDS tile [n][16][68]
            const float bias = p.bias ? p.bias[mi] : 0.f;
#pragma unroll
            for (int tau = 0; tau < NT; ++tau) {
                const int Y = 2 * (NW == 4 ? 2 * wave + (tau >> 1) : wave) + py, Xb = 2 * ((NW == 4 ? 16 * (tau & 1) : 16 * tau) + 4 * g) + px;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pre = acc[tau][r] + bias;
                    ot[(mi * 16 + Y) * 68 + Xb + 2 * r] = p.apply_tanh ? tanhf(pre) : pre;
                }
            }
        }
        if (st == 3) {
#pragma unroll
            for (int tau = 0; tau < NT; ++tau) acc[tau] = f32x4acc{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();                                      // this step's buffer is free; the output tile is complete
        if (st == 3) {
            // coalesced rows: NCHW fp32 (16 B per lane) and / or HWC uint8 = tensor2im (4 B per lane)
            int b, y0, x0;
            tile_origin(ctile, b, y0, x0);
            if (p.out)
                for (int i = tid; i < p.Cout * 256; i += 64 * NW) {
                    const int nn = i >> 8, Y = (i >> 4) & 15, x4 = i & 15;
                    *reinterpret_cast<float4 *>(p.out + (((size_t)b * p.Cout + nn) * H + 2 * y0 + Y) * W + 2 * x0 + 4 * x4) =
                        *reinterpret_cast<const float4 *>(ot + (nn * 16 + Y) * 68 + 4 * x4);
                }
            if (p.out_u8) {
                const int wpr = 16 * p.Cout;                 // 4-byte words per tile row (64 pixels x Cout bytes)
                for (int i = tid; i < 16 * wpr; i += 64 * NW) {
                    const int Y = i / wpr, wd = i - Y * wpr;
                    unsigned pk = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int k = 4 * wd + e, X = k / p.Cout, nn = k - X * p.Cout;
                        pk |= (unsigned)to_u8(ot[(nn * 16 + Y) * 68 + X]) << (8 * e);
                    }
                    *reinterpret_cast<unsigned *>(p.out_u8 + (((size_t)b * H + 2 * y0 + Y) * W + 2 * x0) * p.Cout + 4 * wd) = pk;
                }
            }
            ctile += gridDim.x;
        }
#ifndef LC_ABL_NODMA                                         // timing builds: arithmetic on whatever the first two copies left in LDS
        if (n + 2 < nsteps) {
            if (ist == 4) { ist = 0; itile += gridDim.x; plan_copies(itile); }
            issue(ist, buf);
            ++ist;
        }
#endif
    }
}


static int device_cu_count()
{
    static int cu_count[64];                                  // per device, filled on first use (racing fills write the same value)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cu_count[dev] == 0) {
        int n = 0;
        cu_count[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cu_count[dev];
}

static bool last_conv_mfma_ok(const LastConvParams &p)
{
    return p.dtype == 0 && p.C0 == 64 && p.C1 == 64 && p.Cout >= 1 && p.Cout <= 4 && p.Hs % 8 == 0 && p.Ws % 32 == 0 &&
           (size_t)p.Hs * p.Ws * 256 < 0x80000000ull;
}

template <int NW>
static hipError_t launch_last_conv_mfma_t(const LastConvParams &p, hipStream_t s)
{
    const size_t smem = 16 * 516 * 4 + 2 * (10 * 40 * 128) + 4 * 16 * 68 * 4 + 1024;
    static AttrMask attr_mask;
    if (attr_needed_on_this_device(attr_mask)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&last_conv_mfma<NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        attr_done_on_this_device(attr_mask);
    }
    // one workgroup per CU (150 KB of LDS), each walks tiles blockIdx.x, + grid, ...: the weights are staged once per workgroup and a tile's first
    // copies fly under the previous tile's arithmetic
    const long tiles = (long)p.B * (p.Hs / 8) * (p.Ws / 32);
    const int cus = device_cu_count();
    const long grid = tiles < cus ? tiles : cus;
    hipLaunchKernelGGL(last_conv_mfma<NW>, dim3((unsigned)grid), dim3(64 * NW), smem, s, p, (int)tiles);
    return hipGetLastError();
}
static hipError_t launch_last_conv_mfma(const LastConvParams &p, hipStream_t s)
{
    return p.route == 4 ? launch_last_conv_mfma_t<4>(p, s) : launch_last_conv_mfma_t<8>(p, s);     // route 4: the four-wave form of round 3 (A-B runs, tests)
}

template <typename T, int CO>
static hipError_t launch_last_conv_co(const LastConvParams &p, hipStream_t s)
{
    const long quads = (long)p.B * p.Hs * ((p.Ws + 3) / 4);
    // measured on MI355X (512x512 output): batch 1 rows 43 us / strip 55 us; batch 8 rows 270 us / strip 221 us
    const bool big = (long)p.B * p.Hs * p.Ws >= 4 * 65536;
    // p.route: 0 / 5 = by size (the rule above), 1 strip, 2 rows, 3 generic (fixed per handle at create time, tests only)
    const bool by_size = p.route == 0 || p.route == 5;
    if (p.C0 == p.C1 && p.C0 % 4 == 0 && p.C0 <= 64 && ((big && by_size) || p.route == 1)) {
        // sliding-window kernel; segment length trades window priming (2 extra columns) for parallelism
        const int seg = 32;
        const long groups = (long)p.B * p.Hs * ((p.Ws + seg - 1) / seg);
        const size_t smem = (size_t)4 * 2 * 4 * CO * 16 * sizeof(float4);
        hipLaunchKernelGGL((last_conv_strip<T, CO>), dim3((unsigned)((groups + 15) / 16)), dim3(256), smem, s, p, seg);
        return hipGetLastError();
    }
    if (p.C0 == p.C1 && p.C0 % 4 == 0 && p.C0 <= 128 && p.route != 3) {
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
    // route 0 (by shape): the matrix-core kernel in its eight-wave form where it applies; 4 its four-wave form of round 3 "or fail" (A-B runs, tests);
    // 5: the vector-ALU kernels by size.  (A weights-stationary form with both operands in registers was built in round 4, parity-green and slower --
    // its 400 registers allow one wave per SIMD -- and removed: profiles/r04_lastconv_ab.txt.  Round 5: a vector-ALU form with LDS-broadcast weights, four source
    // pixels per lane and the K loop split over the waves -- parity-green, 25.9 / 139 us against 22.2 / 119 -- likewise: profiles/r05_lastconv_valu_lds.txt; its
    // source is archived in tools/sessions/experiments/last_conv_experiments.inc.)
    if ((p.route == 0 || p.route == 4) && last_conv_mfma_ok(p)) return launch_last_conv_mfma(p, s);
    if (p.route == 4) return hipErrorInvalidValue;
    if (p.dtype == 2) {
        switch (p.Cout) {
        case 1: return launch_last_conv_co<f16_t, 1>(p, s);
        case 2: return launch_last_conv_co<f16_t, 2>(p, s);
        case 3: return launch_last_conv_co<f16_t, 3>(p, s);
        case 4: return launch_last_conv_co<f16_t, 4>(p, s);
        default: return hipErrorInvalidValue;
        }
    }
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

// Pixel shuffle + tanh after the GEMM form of the last conv.  One thread per low-res pixel: 4*Cout values in, the 2x2
// output pixels of every channel out; lanes run along x so every store instruction writes one contiguous row segment.
__global__ __launch_bounds__(256) void pixel_shuffle_tanh(ShuffleParams p)
{
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)p.B * p.Hs * p.Ws;
    if (gid >= total) return;
    const int x = (int)(gid % p.Ws), y = (int)((gid / p.Ws) % p.Hs), b = (int)(gid / ((long)p.Ws * p.Hs));
    const int n = 4 * p.Cout, Ho = 2 * p.Hs, Wo = 2 * p.Ws;
    const float4 *g4 = reinterpret_cast<const float4 *>(p.g + gid * n);
    float v[16];
    for (int i = 0; i < n / 4; ++i) { const float4 t = g4[i]; v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w; }
    if (p.bias)                // (column -> output channel in either order)
        for (int i = 0; i < n; ++i) v[i] += p.bias[p.paired ? (i >> 1) % p.Cout : i % p.Cout];
    if (p.apply_tanh) for (int i = 0; i < n; ++i) v[i] = tanhf(v[i]);
    if (p.paired) {            // rowlast128's column order -> (py * 2 + px) * Cout + co
        float u[16];
        for (int i = 0; i < n; ++i) u[i] = v[i];
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px)
                for (int co = 0; co < p.Cout; ++co) v[(py * 2 + px) * p.Cout + co] = u[(py * p.Cout + co) * 2 + px];
    }
    for (int py = 0; py < 2; ++py) {
        if (p.out)
            for (int co = 0; co < p.Cout; ++co)
                *reinterpret_cast<float2 *>(p.out + (((long)b * p.Cout + co) * Ho + 2 * y + py) * Wo + 2 * x) =
                    make_float2(v[(py * 2) * p.Cout + co], v[(py * 2 + 1) * p.Cout + co]);
        if (p.out_u8) {
            unsigned char *o = p.out_u8 + (((long)b * Ho + 2 * y + py) * Wo + 2 * x) * p.Cout;
            for (int px = 0; px < 2; ++px)
                for (int co = 0; co < p.Cout; ++co) o[px * p.Cout + co] = to_u8(v[(py * 2 + px) * p.Cout + co]);
        }
    }
}

hipError_t launch_pixel_shuffle(const ShuffleParams &p, hipStream_t s)
{
    if (p.Cout < 1 || p.Cout > 4) return hipErrorInvalidValue;
    const long total = (long)p.B * p.Hs * p.Ws;
    hipLaunchKernelGGL(pixel_shuffle_tanh, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p);
    return hipGetLastError();
}

// One thread per (pixel, 4-channel group) of the SOURCE; writes its values into the s2d tensor and / or the relu copy.
__global__ __launch_bounds__(256) void unet_prepare(PrepareParams p)
{
    const int cg = (p.C + 3) / 4;
    const long total = (long)p.B * p.H * p.W * cg;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int g = (int)(gid % cg);
    const long pix = gid / cg;
    const int x = (int)(pix % p.W), y = (int)((pix / p.W) % p.H), b = (int)(pix / ((long)p.W * p.H));
    float v[4];
    for (int e = 0; e < 4; ++e) {
        const int c = g * 4 + e;
        v[e] = 0.f;
        if (c < p.C) v[e] = p.nchw ? p.src[(((long)b * p.C + c) * p.H + y) * p.W + x] : p.src[pix * p.C + c];
    }
    if (p.relu)
        for (int e = 0; e < 4; ++e)
            if (g * 4 + e < p.C) p.relu[pix * p.C + g * 4 + e] = fmaxf(v[e], 0.f);
    if (p.s2d) {
        const long opix = ((long)b * (p.H / 2) + (y >> 1)) * (p.W / 2) + (x >> 1);
        float *o = p.s2d + opix * p.s2d_c + ((y & 1) * 2 + (x & 1)) * p.C;
        for (int e = 0; e < 4; ++e)
            if (g * 4 + e < p.C) o[g * 4 + e] = v[e] > 0.f ? v[e] : p.slope * v[e];
        // zero the padding channels once per output pixel
        if (g == 0 && (y & 1) == 0 && (x & 1) == 0)
            for (int c = 4 * p.C; c < p.s2d_c; ++c) p.s2d[opix * p.s2d_c + c] = 0.f;
    }
}

hipError_t launch_unet_prepare(const PrepareParams &p, hipStream_t s)
{
    if (p.B < 1 || p.H < 2 || p.W < 2 || (p.H & 1) || (p.W & 1) || p.C < 1) return hipErrorInvalidValue;
    if (p.s2d && (p.s2d_c < 4 * p.C || (p.s2d_c & 3))) return hipErrorInvalidValue;
    const long total = (long)p.B * p.H * p.W * ((p.C + 3) / 4);
    hipLaunchKernelGGL(unet_prepare, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace lspf2f
