// gfx950: the `size == 'small'` generator (Feature2FaceGenerator_Unet, models/networks.py:680-769 of the reference) as a native plan behind include/lspunet.h:
// state dict -> packed blob, liveness-planned workspace, one launch per convolution, the whole forward replayed from a hipGraph.
//
// Effective dataflow (derived and checked bit for bit against the reference module by oracle/unet_small_oracle.py): with the in-place LeakyReLU / ReLU of the
// reference, the output d_k of down-conv k (after its BatchNorm) is only ever read as leaky_relu(d_k, 0.2) by down-conv k + 1 and as relu(d_k) by up-conv k, so
//   Y_0            = space-to-depth(cat(feature_map, cand_image))                                  unet_input_s2d (this file)
//   d_k            = BN_k(Conv2d(k4, s2, p1)(.)) = a 3x3 conv on Y_k walking its 16 live (tap, quarter) K blocks   igemm3x3<.., KM> (igemm.hip)
//   Y_{k+1}, R_k   = space-to-depth(leaky_relu(d_k)), relu(d_k)                                    written by that launch's epilogue (unet_dual_store)
//   U_k            = relu(BN(ConvTranspose2d(k4, s2, p1)(cat(R_k, U_{k+1}))))  in sub-pixel form (4 parities x 2x2 taps)   igemm3x3 (up4)
//   out            = tanh(ConvTranspose2d(cat(R_0, U_1)) + bias): a 3x3 GEMM with N = 4 parities x output_nc on the low-res source + pixel_shuffle_tanh
#include "../../include/lspunet.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "device_common.h"
#include "kernels.h"
#include "plan.h"

namespace lspf2f {

// ---- input pass: two NCHW sources -> the space-to-depth image of their concatenation ---------------------------------------------------------
// Y[b][i][j][(dy * 2 + dx) * C + c] = X[b][c][2i + dy][2j + dx], channels >= 4C zero (C = 23 -> 92 of 96).  A workgroup owns one output row i and up to 64
// output columns: it reads the 2 x C source row segments with coalesced loads (a row of an NCHW plane is contiguous), turns them through LDS and writes
// <= 64 x s2d_c contiguous floats.
struct UnetInputParams {
    const float *feat, *cand;      // [B][feat_nc][S][S], [cand_batch][C - feat_nc][S][S]
    void *out;                     // [B][S/2][S/2][s2d_c] in the plan's storage type
    int B, S, C, feat_nc, cand_bcast, s2d_c, cols;   // cols = output columns per workgroup (<= 64)
    int qc;                        // channels per quarter in the output: C (dense: quarter q at q * C, the tail zero) or C rounded up to a K-tile (quarter q at q * qc, its tail zero)
};

template <typename T>
__global__ __launch_bounds__(256) void unet_input_s2d(const UnetInputParams p)
{
    extern __shared__ float lds[];                       // [C][2][pitch]
    const int cw = 2 * p.cols, pitch = cw + 1;
    const int S2 = p.S >> 1;
    const int chunks = S2 / p.cols;
    const int i = blockIdx.x / chunks, j0 = (blockIdx.x - i * chunks) * p.cols, b = blockIdx.y;
    const int cand_nc = p.C - p.feat_nc;
    const size_t plane = (size_t)p.S * p.S;
    const int cw4 = cw >> 2;                                // 16-byte loads: a row segment starts at a multiple of 2 * cols >= 4 floats (cols even)
    for (int idx = threadIdx.x; idx < p.C * 2 * cw4; idx += 256) {
        const int c = idx / (2 * cw4), r = idx - c * 2 * cw4;
        const int dy = r / cw4, x = (r - dy * cw4) * 4;
        const float *src = c < p.feat_nc ? p.feat + ((size_t)b * p.feat_nc + c) * plane
                                         : p.cand + ((size_t)(p.cand_bcast ? 0 : b) * cand_nc + (c - p.feat_nc)) * plane;
        const float4 v = *reinterpret_cast<const float4 *>(src + (size_t)(2 * i + dy) * p.S + 2 * j0 + x);
        float *d = lds + (c * 2 + dy) * pitch + x;          // (odd pitch: the column gather below stays two-way conflicted at worst; four scalar stores)
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    const int q4 = p.s2d_c >> 2;
    T *dst = static_cast<T *>(p.out) + (((size_t)b * S2 + i) * S2 + j0) * p.s2d_c;
    for (int o = threadIdx.x; o < p.cols * q4; o += 256) {
        const int j = o / q4, ch0 = (o - j * q4) * 4;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ch = ch0 + e;
            v[e] = 0.f;
            const int q = ch / p.qc, c = ch - q * p.qc;
            if (q < 4 && c < p.C) v[e] = lds[(c * 2 + (q >> 1)) * pitch + 2 * j + (q & 1)];
        }
        store4(dst + (size_t)j * p.s2d_c + ch0, make_float4(v[0], v[1], v[2], v[3]));
    }
}

static hipError_t launch_unet_input(const UnetInputParams &p, int dtype, hipStream_t s)
{
    const int S2 = p.S / 2;
    if (p.B < 1 || p.S < 4 || (p.S & 3) || p.C < 1 || p.C > 48 || p.feat_nc < 1 || p.feat_nc > p.C || p.qc < p.C || p.s2d_c < 4 * p.qc || (p.s2d_c & 3) || p.cols < 2 || (p.cols & 1) || S2 % p.cols)
        return hipErrorInvalidValue;
    const size_t smem = (size_t)p.C * 2 * (2 * p.cols + 1) * sizeof(float);
    const dim3 grid((unsigned)(S2 * (S2 / p.cols)), (unsigned)p.B);
    if (dtype == 2) hipLaunchKernelGGL(unet_input_s2d<f16_t>, grid, dim3(256), smem, s, p);
    else hipLaunchKernelGGL(unet_input_s2d<float>, grid, dim3(256), smem, s, p);
    return hipGetLastError();
}

// ---- the <= 16-position levels: one weight-streaming launch per layer --------------------------------------------------------------------------
// The deepest levels at one frame (4x4 and 2x2 tensors, 512 channels) are pure weight streaming -- 16.8 / 33.5 MB of weights for <= 0.3 GFLOP -- and ran as a 64-way
// split-K implicit GEMM + a reduce launch (40-47 us each for 3-5 us of HBM time, profiles/r05_unet_small_native_first.txt).  Same structure as conv3x3_smallm
// (small_layers.hip): every workgroup owns NC weight rows over the FULL K (no cross-workgroup reduction), requests them first, stages the whole input tensor in LDS,
// multiplies on the vector ALU and reduces through LDS in a fixed order.  Two gathers:
//   down  Conv2d(k4, s2, p1) on the space-to-depth image: row n = [16 live (tap, quarter) slots][ci]; position m = output pixel; slot -> (s2d pixel, quarter)
//   up    ConvTranspose2d(k4, s2, p1) in sub-pixel form: row (parity, n) = [2][2][C0 + C1]; position m = SOURCE pixel, written to output pixel (2y + py, 2x + px)
// "vector" = the V contiguous floats one (tap, position) pair multiplies: V = ci (a quarter of an s2d pixel) or C0 + C1 (a pixel of the concatenation).
struct UnetTinyParams {
    const float *src0, *src1;      // down: the s2d image [B][Hs][Ws][4 ci]; up: [B][Hs][Ws][C0], [B][Hs][Ws][C1] (or nullptr, C1 = 0)
    const float *w, *scale, *shift;
    float *out;                    // down: [B][Hs][Ws][Cout] (innermost block, ReLU'd); up: [B][2Hs][2Ws][Cout]
    float *s2d_out, *relu_out;     // down: the two activated copies instead of `out` (unet_dual_store)
    float slope;
    int B, Hs, Ws, C0, C1, Cout, up, relu, M;
};

template <int NJ, int MM>
__global__ __launch_bounds__(256) void unet_tiny(const UnetTinyParams p)
{
    constexpr int NC = 2, RS = 264;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    int *tab = reinterpret_cast<int *>(sm);                // [ntap][MM] vector index of (tap, position); padding / rows past M -> the zero vector
    float *act = sm + 256;                                 // [NV + 1][V]; later red[MM * NC][RS]
    const int tid = threadIdx.x;
    asm volatile("" :: "s"(p.src0), "s"(p.src1), "s"(p.w), "s"(p.scale), "s"(p.shift), "s"(p.out), "s"(p.s2d_out), "s"(p.relu_out), "s"(p.B), "s"(p.Hs), "s"(p.Ws),
                       "s"(p.C0), "s"(p.C1), "s"(p.Cout), "s"(p.up), "s"(p.relu), "s"(p.M));
    const int V = p.up ? p.C0 + p.C1 : p.C0 >> 2, V4 = V >> 2;
    const int ntap = p.up ? 4 : 16, K4 = ntap * V4;
    const int rows_per_par = p.Cout / NC;
    const int par = p.up ? blockIdx.x / rows_per_par : 0;
    const int n0 = (blockIdx.x - par * rows_per_par) * NC;
    const int py = par >> 1, px = par & 1;
    // 1. this workgroup's weight rows: asked for first
    float4 wv[NC][NJ];
#pragma unroll
    for (int nc = 0; nc < NC; ++nc)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int k4 = tid + 256 * j;
            wv[nc][j] = k4 < K4 ? *reinterpret_cast<const float4 *>(p.w + ((size_t)par * p.Cout + n0 + nc) * (size_t)(K4 * 4) + (size_t)k4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    // 2. the input tensor(s) -> LDS as [vector][V]; the gather table; the epilogue's operands
    const int npix = p.B * p.Hs * p.Ws;
    const int nvec = p.up ? npix : npix * 4;
    const int eo = tid >> 3, em = eo / NC, en = n0 + eo % NC;
    const bool ewrite = tid < MM * NC * 8 && (tid & 7) == 0 && em < p.M;
    float e_sc = 1.f, e_sh = 0.f;
    if (ewrite && p.scale) { e_sc = p.scale[en]; e_sh = p.shift[en]; }
    // LDS-DMA, 1 KB per wave and instruction (no staging registers, no dependent load -> ds_write chain: the first version copied through registers and spent 10+ us
    // of a 128-KB tensor's staging on it); a vector of the concatenation is C0 floats of src0 then C1 of src1, both multiples of 256
    {
        typedef __attribute__((address_space(3))) float lds_float;
        const unsigned lds_act = (unsigned)(unsigned long long)(lds_float *)act;
        const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane(tid >> 6), lane16 = ((unsigned)tid & 63u) * 16u;      // (wave-uniform: the LDS base of a copy is a scalar)
        if (!p.up || p.C1 == 0) {
            const unsigned total = (unsigned)nvec * (unsigned)V * 4u;
            const i32x4 srd = make_srd(p.src0, total);
            for (unsigned o = wave * 1024u; o < total; o += 4096u) dma16(lds_act + o, o + lane16, srd, 0);
        } else {
            const unsigned ppv = (unsigned)V >> 8, pp0 = (unsigned)p.C0 >> 8;          // 1-KB pieces per vector, of which from src0
            const i32x4 srd0 = make_srd(p.src0, (unsigned)nvec * (unsigned)p.C0 * 4u), srd1 = make_srd(p.src1, (unsigned)nvec * (unsigned)p.C1 * 4u);
            for (unsigned piece = wave; piece < (unsigned)nvec * ppv; piece += 4u) {
                const unsigned v = piece / ppv, q = piece - v * ppv;
                if (q < pp0) dma16(lds_act + piece * 1024u, (v * pp0 + q) * 1024u + lane16, srd0, 0);
                else dma16(lds_act + piece * 1024u, (v * (ppv - pp0) + q - pp0) * 1024u + lane16, srd1, 0);
            }
        }
    }
    for (int i = tid; i < V4; i += 256) reinterpret_cast<float4 *>(act)[nvec * V4 + i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < ntap * MM) {
        const int t = tid / MM, m = tid - t * MM;
        int vec = nvec;
        if (m < p.M) {
            const int hw = p.Hs * p.Ws, b = m / hw, r = m - b * hw, y = r / p.Ws, x = r - y * p.Ws;
            if (p.up) {
                const int sy = y + py + (t >> 1) - 1, sx = x + px + (t & 1) - 1;
                if (sy >= 0 && sy < p.Hs && sx >= 0 && sx < p.Ws) vec = (b * p.Hs + sy) * p.Ws + sx;
            } else {
                // slot t of the live (tap, quarter) pairs in tap-major / quarter-minor order: tap row ty carries sub-rows {1}, {0, 1}, {0} for ty = 0, 1, 2 (columns alike)
                int ty = 0, tx = 0, dy = 0, dx = 0, n = 0;
                for (int a = 0; a < 3; ++a)
                    for (int c = 0; c < 3; ++c)
                        for (int d = 0; d < 2; ++d)
                            for (int e = 0; e < 2; ++e) {
                                const bool live = (a == 0 ? d == 1 : a == 2 ? d == 0 : true) && (c == 0 ? e == 1 : c == 2 ? e == 0 : true);
                                if (live) { if (n == t) { ty = a; tx = c; dy = d; dx = e; } ++n; }
                            }
                const int sy = y + ty - 1, sx = x + tx - 1;
                if (sy >= 0 && sy < p.Hs && sx >= 0 && sx < p.Ws) vec = ((b * p.Hs + sy) * p.Ws + sx) * 4 + dy * 2 + dx;
            }
        }
        tab[tid] = vec;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's copies have landed (and its weight rows)
    __syncthreads();
    // 3. this thread's K-slice times every position.  k4 = tid + 256 j; a wave's 64 consecutive float4 stay inside one tap (V % 256 == 0): the table reads are wave-uniform
    float acc[MM][NC];
#pragma unroll
    for (int m = 0; m < MM; ++m)
#pragma unroll
        for (int nc = 0; nc < NC; ++nc) acc[m][nc] = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int k4 = tid + 256 * j;
        if (k4 >= K4) continue;
        const int tap = k4 / V4, c4 = k4 - tap * V4;
        int vec[MM];
#pragma unroll
        for (int q = 0; q < MM / 4; ++q) {
            const int4 t4 = reinterpret_cast<const int4 *>(tab + tap * MM)[q];
            vec[4 * q] = t4.x; vec[4 * q + 1] = t4.y; vec[4 * q + 2] = t4.z; vec[4 * q + 3] = t4.w;
        }
        float4 a[MM];
#pragma unroll
        for (int m = 0; m < MM; ++m) a[m] = reinterpret_cast<const float4 *>(act)[vec[m] * V4 + c4];
#pragma unroll
        for (int m = 0; m < MM; ++m)
#pragma unroll
            for (int nc = 0; nc < NC; ++nc) {
                const float4 w4 = wv[nc][j];
                acc[m][nc] += a[m].x * w4.x + a[m].y * w4.y + a[m].z * w4.z + a[m].w * w4.w;
            }
    }
    __syncthreads();                                       // everyone is done with act
    // 4. block reduction in a fixed order: red[o][tid], then 8 threads per output sum 32 interleaved values each
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
        if (part == 0 && ewrite) {
            const float v = sum * e_sc + e_sh;
            const int hw = p.Hs * p.Ws, b = em / hw, r = em - b * hw, y = r / p.Ws, x = r - y * p.Ws;
            if (p.up) {
                p.out[(((size_t)b * 2 * p.Hs + 2 * y + py) * (2 * p.Ws) + 2 * x + px) * p.Cout + en] = p.relu ? fmaxf(v, 0.f) : v;
            } else if (p.s2d_out) {
                p.relu_out[(size_t)em * p.Cout + en] = fmaxf(v, 0.f);
                const size_t opix = ((size_t)b * (p.Hs >> 1) + (y >> 1)) * (p.Ws >> 1) + (x >> 1);
                p.s2d_out[opix * (size_t)(4 * p.Cout) + ((y & 1) * 2 + (x & 1)) * p.Cout + en] = v > 0.f ? v : p.slope * v;
            } else {
                p.out[(size_t)em * p.Cout + en] = p.relu ? fmaxf(v, 0.f) : v;
            }
        }
    }
}

static size_t unet_tiny_lds(const UnetTinyParams &p)
{
    const size_t V = p.up ? p.C0 + p.C1 : p.C0 / 4, nvec = (size_t)p.B * p.Hs * p.Ws * (p.up ? 1 : 4);
    const size_t act = (nvec + 1) * V * 4, red = (size_t)16 * 2 * 264 * 4;
    return 256 * 4 + (act > red ? act : red);
}

static bool unet_tiny_supported(const UnetTinyParams &p)
{
    const int V = p.up ? p.C0 + p.C1 : p.C0 / 4, ntap = p.up ? 4 : 16;
    if (p.M < 1 || p.M > 16 || p.M != p.B * p.Hs * p.Ws || V % 256 || ntap * (V / 4) > 8 * 256 || p.Cout % 2) return false;
    if (p.up ? (p.C0 % 256 || p.C1 % 256) : (p.C1 != 0 || p.C0 % 16)) return false;      // (the staging copies move 1-KB pieces of either source)
    if (p.s2d_out && (p.up || !p.relu_out || (p.Hs & 1) || (p.Ws & 1))) return false;
    return unet_tiny_lds(p) <= 150 * 1024;
}

template <int NJ, int MM>
static hipError_t launch_unet_tiny_t(const UnetTinyParams &p, hipStream_t s)
{
    static AttrMask attr_mask;
    if (attr_needed_on_this_device(attr_mask)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&unet_tiny<NJ, MM>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return e;
        attr_done_on_this_device(attr_mask);
    }
    const unsigned grid = (unsigned)((p.up ? 4 : 1) * (p.Cout / 2));
    hipLaunchKernelGGL((unet_tiny<NJ, MM>), dim3(grid), dim3(256), unet_tiny_lds(p), s, p);
    return hipGetLastError();
}

static hipError_t launch_unet_tiny(const UnetTinyParams &p, hipStream_t s)
{
    if (!unet_tiny_supported(p)) return hipErrorInvalidValue;
    const int V = p.up ? p.C0 + p.C1 : p.C0 / 4, nj = ((p.up ? 4 : 16) * (V / 4) + 255) / 256;
    const bool four = p.M <= 4;
    if (nj <= 2) return four ? launch_unet_tiny_t<2, 4>(p, s) : launch_unet_tiny_t<2, 16>(p, s);
    if (nj <= 4) return four ? launch_unet_tiny_t<4, 4>(p, s) : launch_unet_tiny_t<4, 16>(p, s);
    return four ? launch_unet_tiny_t<8, 4>(p, s) : launch_unet_tiny_t<8, 16>(p, s);
}

// ---- the plan -----------------------------------------------------------------------------------------------------------------------------------
struct UnetParam {
    std::string key;
    std::vector<int64_t> dims;
    std::vector<float> data;
    bool set = false;
    size_t numel() const { size_t n = 1; for (auto d : dims) n *= (size_t)d; return n; }
};

struct UnetLevel {
    int cin = 0, cout = 0;         // down-conv k: cin -> cout (= chans[k]) at stride 2
    int s2d = 0;                   // channels of its space-to-depth input: 4 cin_p (block 0 in its dense form: 4 cin padded to a K-tile)
    int cin_p = 0;                 // channels per quarter of that image: cin (block 0 in its live form: cin rounded up to a K-tile, the tail zero)
    bool live = true;              // weights = the 16 live (tap, quarter) pairs [co][16][cin_p]; false (block 0, dense form): the 3x3 rows [co][9][s2d]
    int up_cin = 0, up_cout = 0;   // transposed conv of block k: up_cin (cout, or 2 cout with the skip) -> up_cout
    std::string dc, dbn, uc, ubn;  // state-dict keys (dbn / ubn "" = none)
    int64_t down_w = -1, down_scale = -1, down_shift = -1, up_w = -1, up_scale = -1, up_shift = -1;   // byte offsets in the blob
    int64_t up_rl = -1;                  // block 0, 16-bit plans with two 64-channel sources: its GEMM rows in the fragment order of rowlast128 (rowconv.hip)
    int64_t up_sub = -1, up_bias = -1;   // block 0 only: its transposed conv in sub-pixel form [4][co][2][2][ci] + the conv bias, for the direct last-layer kernel
    size_t down_w_bytes = 0, up_w_bytes = 0;
};

// one launch (or launch + reduce) of the forward
struct UnetLaunch {
    std::string name, kernel;
    int kind = 0;                  // 0 input pass, 1 conv, 2 pixel shuffle, 3 unet_prepare (fused_prepare = 0), 4 the direct last-layer kernel (last_direct = 1), 5 rowlast128 (16-bit plans)
    int level = 0;
    bool down = false, last = false;
    int bm = 0, bn = 0, splits = 1, group = 1;
    bool fused_combine = false;
    bool tiny = false;             // the weight-streaming kernel of the <= 16-position levels (unet_tiny) instead of the implicit GEMM + reduce
};

static const unsigned kUnetCounters = 16384;

struct UnetPlan {
    int input_nc = 23, feat_nc = 23, output_nc = 3, ngf = 64, nd = 8, size = 512;
    int dtype = 0;                             // 0: fp32; 2: fp16 storage of activations and conv weights, fp32 accumulate / epilogue (the reference's opt.fp16: autocast around netG,
                                               // models/feature2face_G.py:28-30) -- the same launches on v_mfma_f32_32x32x16_f16 minus the fp32-only ones (unet_tiny, the A-B arms)
    size_t elt() const { return dtype ? 2 : 4; }
    int ktc() const { return dtype ? 64 : 32; }   // channels per K-tile (128 B)
    bool fused_prepare = true, input_pass = true;
    bool fused_splitk = false;                 // 2..8 K splits combined by the last-arriving workgroup instead of a reduce launch: measured SLOWER here (0.903 vs 0.889 ms at one frame,
                                               // equal at eight; profiles/r05_unet_small_native.txt) -- the split layers of this plan are short launches -- and its 6-split sum runs in another order than splitk_reduce's
    bool dense0 = false;                       // tune key `dense0`: block 0 (23 input channels) as dense 3x3 rows on 4 x 23 -> 96 channels, 27 K-tiles of which 15 multiply zeros (the host-sequenced
                                               // form's block 0); default: its quarters padded to 32 channels each and only the 16 live (tap, quarter) pairs walked (16 K-tiles).
                                               // 16-bit plans keep the dense form (a K-tile is 64 channels: 16 x 64 = 1024 of K against 9 x 128 = 1152, on a larger input)
    bool use_tiny = true;                      // tune key `tiny`
    bool last_direct = true;                   // the outermost transposed conv + tanh (+ tensor2im) on the direct sub-pixel kernel of the other variants' last layer (edge_layers.hip) instead of
                                               // a 3x3 GEMM with N = 12 of 32 columns live + a pixel-shuffle pass
    int last_bm = 0, last_bn = 0;              // tune: tile of the last GEMM (0 = the planner's rule)
    std::vector<int> chans;
    std::vector<UnetLevel> L;
    std::vector<UnetParam> params;
    std::map<std::string, int> index;
    size_t blob_bytes = 0;

    // per-batch state: the launch list and the workspace layout
    struct Batch {
        int B = 0;
        std::vector<UnetLaunch> launches;
        size_t y_off[2] = {0, 0}, u_off[2] = {0, 0}, g_off = 0, dtmp_off = 0, part_off = 0, cnt_off = 0, total = 0;
        std::vector<size_t> r_off;             // R_k, k < nd - 1; r_off[nd - 1] = relu(d_{nd-1}) of the innermost block
        size_t part_bytes = 0;
    };
    Batch cur;

    void add_param(const std::string &k, std::vector<int64_t> d)
    {
        index[k] = (int)params.size();
        UnetParam p; p.key = k; p.dims = std::move(d);
        params.push_back(std::move(p));
    }
    const float *data(const std::string &k) const { return params[index.at(k)].data.data(); }

    std::string build()
    {
        if (ngf < 32 || ngf % 32) return "ngf must be a multiple of 32";
        if (nd < 5 || nd > 12) return "num_downs must be in 5..12";
        if (output_nc < 1 || output_nc > 4) return "output_nc must be in 1..4";
        if (input_nc < 1 || input_nc > 48) return "input_nc must be in 1..48";
        if (feat_nc < 1 || feat_nc > input_nc) return "feat_nc must be in 1..input_nc";
        if (size < (1 << nd) || size % (1 << nd)) return "frame size must be a multiple of 2**num_downs";
        if (dtype != 0 && dtype != 2) return "dtype must be 0 (fp32) or 2 (fp16 storage)";
        if (dtype && ngf % 64) return "fp16 storage needs ngf % 64 == 0";
        if (dtype) { fused_prepare = true; input_pass = true; last_direct = true; use_tiny = false; fused_splitk = false; dense0 = true; }     // (the other arms are fp32 launches)
        if (!input_pass) dense0 = true;             // (lspf2f_unet_prepare writes the dense quarter layout)
        chans.clear();
        for (int i = 0; i < nd; ++i) chans.push_back(ngf * std::min(1 << i, 8));
        // state-dict keys: the nesting of nn.Sequential indices in UnetSkipConnectionBlock (models/networks.py:737-767)
        std::string pfx = "model.model";
        L.assign(nd, UnetLevel());
        for (int k = 0; k < nd; ++k) {
            UnetLevel &l = L[k];
            l.cin = k == 0 ? input_nc : chans[k - 1];
            l.cout = chans[k];
            l.live = k > 0 || !dense0;
            l.cin_p = l.live ? (l.cin + ktc() - 1) / ktc() * ktc() : l.cin;
            l.s2d = l.live ? 4 * l.cin_p : (4 * l.cin + ktc() - 1) / ktc() * ktc();
            l.up_cin = k == nd - 1 ? l.cout : 2 * l.cout;
            l.up_cout = k == 0 ? output_nc : chans[k - 1];
            if (k == 0) { l.dc = pfx + ".0"; l.uc = pfx + ".3"; pfx += ".1.model"; }
            else if (k == nd - 1) { l.dc = pfx + ".1"; l.uc = pfx + ".3"; l.ubn = pfx + ".4"; }
            else { l.dc = pfx + ".1"; l.dbn = pfx + ".2"; l.uc = pfx + ".5"; l.ubn = pfx + ".6"; pfx += ".3.model"; }
            add_param(l.dc + ".weight", {l.cout, l.cin, 4, 4});
            if (!l.dbn.empty())
                for (const char *t : {".weight", ".bias", ".running_mean", ".running_var"}) add_param(l.dbn + t, {l.cout});
            add_param(l.uc + ".weight", {l.up_cin, l.up_cout, 4, 4});
            if (k == 0) add_param(l.uc + ".bias", {l.up_cout});
            if (!l.ubn.empty())
                for (const char *t : {".weight", ".bias", ".running_mean", ".running_var"}) add_param(l.ubn + t, {l.up_cout});
        }
        // blob layout, 256-byte aligned pieces
        size_t off = 0;
        auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return (int64_t)o; };
        for (int k = 0; k < nd; ++k) {
            UnetLevel &l = L[k];
            l.down_w_bytes = (l.live ? (size_t)l.cout * 16 * l.cin_p : (size_t)l.cout * 9 * l.s2d) * elt();
            l.down_w = take(l.down_w_bytes);
            if (!l.dbn.empty()) { l.down_scale = take((size_t)l.cout * 4); l.down_shift = take((size_t)l.cout * 4); }
            const int n_up = k == 0 ? 4 * l.up_cout : l.up_cout;
            l.up_w_bytes = k == 0 ? (size_t)n_up * 9 * l.up_cin * 4 : (size_t)4 * l.up_cout * 4 * l.up_cin * elt();      // (block 0: the fp32 GEMM rows of the last_direct = 0 arm)
            l.up_w = take(l.up_w_bytes);
            l.up_scale = take((size_t)n_up * 4);
            l.up_shift = take((size_t)n_up * 4);
            if (k == 0) { l.up_sub = take((size_t)4 * l.up_cout * 4 * l.up_cin * 4); l.up_bias = take((size_t)l.up_cout * 4); }
            if (k == 0 && rowlast_ok()) l.up_rl = take((size_t)9 * 4 * 64 * 8 * 2);
        }
        blob_bytes = off;
        cur = Batch();
        return "";
    }

    // models/networks.py BatchNorm2d in eval mode, folded in double (eps 1e-5)
    void fold_bn(const std::string &key, int n, float *scale, float *shift) const
    {
        const float *g = data(key + ".weight"), *b = data(key + ".bias"), *m = data(key + ".running_mean"), *v = data(key + ".running_var");
        for (int i = 0; i < n; ++i) {
            const double s = (double)g[i] / std::sqrt((double)v[i] + 1e-5);
            scale[i] = (float)s;
            shift[i] = (float)((double)b[i] - (double)m[i] * s);
        }
    }

    // fp32 -> IEEE binary16, round to nearest even (the storage type of dtype 2)
    static void narrow(const std::vector<float> &src, char *dst)
    {
        for (size_t i = 0; i < src.size(); ++i) {
            const _Float16 h = (_Float16)src[i];
            std::memcpy(dst + 2 * i, &h, 2);
        }
    }

    std::string pack(void *blob, size_t bytes) const
    {
        if (bytes < blob_bytes) return "packed weight arena too small";
        for (const auto &p : params)
            if (!p.set) return "missing state-dict tensor: " + p.key;
        char *B = static_cast<char *>(blob);
        std::memset(B, 0, blob_bytes);
        // 4x4 / s2 tap ky -> (3x3 tap row ty, space-to-depth sub-row dy): input row 2y - 1 + ky = 2 (y + ty - 1) + dy
        static const int TY[4] = {0, 1, 1, 2}, DY[4] = {1, 0, 1, 0};
        // transposed conv: kernel row of output parity py, sub-pixel tap a (source row y + py + a - 1)
        static const int KT[2][2] = {{3, 1}, {2, 0}};
        for (int k = 0; k < nd; ++k) {
            const UnetLevel &l = L[k];
            const float *w = data(l.dc + ".weight");                 // [cout][cin][4][4]
            std::vector<float> stage;                                   // 16-bit plans: conv weights are packed in fp32 here, then narrowed (RNE) into the blob
            float *dw = reinterpret_cast<float *>(B + l.down_w);
            if (dtype) { stage.assign(l.down_w_bytes / 2, 0.f); dw = stage.data(); }
            if (!l.live) {
                // dense rows [co][ty][tx][s2d] of the 3x3 conv on the space-to-depth image (20 of the 36 (tap, quarter) blocks stay zero; the quarters of 23 channels are
                // not K-tile aligned, so block 0 cannot drop them)
                for (int co = 0; co < l.cout; ++co)
                    for (int ci = 0; ci < l.cin; ++ci)
                        for (int ky = 0; ky < 4; ++ky)
                            for (int kx = 0; kx < 4; ++kx)
                                dw[((size_t)(co * 3 + TY[ky]) * 3 + TY[kx]) * l.s2d + (DY[ky] * 2 + DY[kx]) * l.cin + ci] = w[(((size_t)co * l.cin + ci) * 4 + ky) * 4 + kx];
            } else {
                // [co][16 live (tap, quarter) pairs, tap-major / quarter-minor][ci]: the order the masked K cursor walks them
                int slot[4][4], n = 0;
                for (int ty = 0; ty < 3; ++ty)
                    for (int tx = 0; tx < 3; ++tx)
                        for (int dy = 0; dy < 2; ++dy)
                            for (int dx = 0; dx < 2; ++dx)
                                for (int ky = 0; ky < 4; ++ky)
                                    for (int kx = 0; kx < 4; ++kx)
                                        if (TY[ky] == ty && DY[ky] == dy && TY[kx] == tx && DY[kx] == dx) slot[ky][kx] = n++;
                for (int co = 0; co < l.cout; ++co)
                    for (int ci = 0; ci < l.cin; ++ci)
                        for (int ky = 0; ky < 4; ++ky)
                            for (int kx = 0; kx < 4; ++kx)
                                dw[((size_t)co * 16 + slot[ky][kx]) * l.cin_p + ci] = w[(((size_t)co * l.cin + ci) * 4 + ky) * 4 + kx];     // (block 0: channels cin .. cin_p - 1 stay zero)
            }
            if (dtype) narrow(stage, B + l.down_w);
            if (!l.dbn.empty()) fold_bn(l.dbn, l.cout, reinterpret_cast<float *>(B + l.down_scale), reinterpret_cast<float *>(B + l.down_shift));
            const float *wt = data(l.uc + ".weight");                // [up_cin][up_cout][4][4]
            float *uw = reinterpret_cast<float *>(B + l.up_w);
            float *usc = reinterpret_cast<float *>(B + l.up_scale), *ush = reinterpret_cast<float *>(B + l.up_shift);
            const int ci_n = l.up_cin, co_n = l.up_cout;
            if (k == 0) {
                // GEMM rows [par * co_n + co][3][3][ci]: tap (py + a, px + b) of the 3x3 window on the low-res source carries sub-pixel tap (a, b) of parity (py, px)
                for (int py = 0; py < 2; ++py)
                    for (int px = 0; px < 2; ++px)
                        for (int a = 0; a < 2; ++a)
                            for (int b = 0; b < 2; ++b)
                                for (int co = 0; co < co_n; ++co)
                                    for (int ci = 0; ci < ci_n; ++ci)
                                        uw[((((size_t)(py * 2 + px) * co_n + co) * 3 + py + a) * 3 + px + b) * ci_n + ci] =
                                            wt[(((size_t)ci * co_n + co) * 4 + KT[py][a]) * 4 + KT[px][b]];
                const float *bias = data(l.uc + ".bias");
                for (int i = 0; i < 4 * co_n; ++i) { usc[i] = 1.f; ush[i] = bias[i % co_n]; }
                std::memcpy(B + l.up_bias, bias, (size_t)co_n * 4);
                if (l.up_rl >= 0) {
                    std::vector<unsigned short> g16((size_t)4 * co_n * 9 * ci_n);
                    for (size_t i = 0; i < g16.size(); ++i) { const _Float16 hv = (_Float16)uw[i]; std::memcpy(&g16[i], &hv, 2); }
                    pack_rowlast_weights(g16.data(), reinterpret_cast<unsigned short *>(B + l.up_rl), 4 * co_n);
                }
            }
            {
                // sub-pixel form [par][co][a][b][ci] (block 0: a second form of its weights, for the direct last-layer kernel)
                float *sw = k == 0 ? reinterpret_cast<float *>(B + l.up_sub) : uw;
                if (dtype && k > 0) { stage.assign(l.up_w_bytes / 2, 0.f); sw = stage.data(); }
                for (int py = 0; py < 2; ++py)
                    for (int px = 0; px < 2; ++px)
                        for (int co = 0; co < co_n; ++co)
                            for (int a = 0; a < 2; ++a)
                                for (int b = 0; b < 2; ++b)
                                    for (int ci = 0; ci < ci_n; ++ci)
                                        sw[((((size_t)(py * 2 + px) * co_n + co) * 2 + a) * 2 + b) * ci_n + ci] =
                                            wt[(((size_t)ci * co_n + co) * 4 + KT[py][a]) * 4 + KT[px][b]];
                if (dtype && k > 0) narrow(stage, B + l.up_w);
                if (k > 0) fold_bn(l.ubn, co_n, usc, ush);
            }
        }
        return "";
    }

    // unet_tiny_supported() for a level of this plan (host mirror): <= 16 positions, vectors of a multiple of 256 floats, <= 8 float4 of K per thread, the input in 150 KB of LDS
    bool tiny_ok(int B, int h, int V, int ntap, int nvec_per_pix, int cout) const
    {
        const long M = (long)B * h * h;
        if (!use_tiny || M > 16 || V % 256 || (ntap == 4 && V >= 512 && (V / 2) % 256) || ntap * (V / 4) > 8 * 256 || cout % 2) return false;
        return 256 * 4 + ((size_t)M * nvec_per_pix + 1) * V * 4 <= 150 * 1024;
    }

    // 16-bit plans whose last layer reads two 64-channel sources: the row kernel of the other variants' 16-bit last conv (rowlast128, rowconv.hip) on the GEMM rows
    // + the pixel-shuffle pass (which adds the bias): 62 / 294 us -> see profiles/r05_unet_small_native.txt for the direct kernel's vector-ALU route it replaces
    // (rowlast128 is hard-wired to 12 GEMM columns = 4 parities x 3 channels per pixel: 48-byte records, lanes g4 < 3 -- other output_nc keep the direct kernel)
    bool rowlast_ok() const { return dtype != 0 && ngf == 64 && output_nc == 3 && (size / 2) % 64 == 0; }

    // spatial extent of d_k (= of Y_{k+1}'s source, of R_k)
    int hd(int k) const { return size >> (k + 1); }

    void tile_for(int M, int N, int ktiles, int par, UnetLaunch *u, bool allow_fused, size_t mout_x_cout) const
    {
        choose_tiling(M, N, ktiles, par, false, dtype, &u->bm, &u->bn, &u->splits, &u->group);
        const int per = (ktiles + u->splits - 1) / u->splits;
        u->splits = (ktiles + per - 1) / per;
        const long tiles = (long)par * ((M + u->bm - 1) / u->bm) * ((N + u->bn - 1) / u->bn);
        u->fused_combine = allow_fused && fused_splitk && u->splits >= 2 && u->splits <= 8 && tiles <= (long)kUnetCounters &&
                           (size_t)u->splits * mout_x_cout * sizeof(float) < (size_t)0x7fffffff;
    }

    void plan_batch(int B) { if (cur.B != B) cur = layout(B); }
    size_t workspace_bytes(int B) const { return layout(B).total; }

    Batch layout(int B) const
    {
        Batch bp;
        bp.B = B;
        std::vector<UnetLaunch> &launches = bp.launches;
        size_t &part_bytes = bp.part_bytes;
        size_t (&y_off)[2] = bp.y_off, (&u_off)[2] = bp.u_off;
        size_t &g_off = bp.g_off, &dtmp_off = bp.dtmp_off, &part_off = bp.part_off, &cnt_off = bp.cnt_off, &total = bp.total;
        std::vector<size_t> &r_off = bp.r_off;
        auto note_partial = [&](const UnetLaunch &u, size_t mout_x_cout) { if (u.splits > 1) part_bytes = std::max(part_bytes, (size_t)u.splits * mout_x_cout * sizeof(float)); };
        {
            UnetLaunch u; u.kind = input_pass ? 0 : 3; u.name = "input"; u.kernel = input_pass ? "unet_input_s2d" : "unet_prepare"; u.level = -1;
            launches.push_back(u);
        }
        for (int k = 0; k < nd; ++k) {
            const UnetLevel &l = L[k];
            const int h = hd(k);
            UnetLaunch u; u.kind = 1; u.level = k; u.down = true; u.name = "L" + std::to_string(k) + ".down";
            const bool km = l.live || (fused_prepare && k < nd - 1);
            const int ktiles = l.live ? 16 * l.cin_p / ktc() : 9 * l.s2d / ktc();
            tile_for(B * h * h, l.cout, ktiles, 1, &u, true, (size_t)B * h * h * l.cout);
            if (l.live && (fused_prepare || k == nd - 1) && tiny_ok(B, h, l.cin_p, 16, 4, l.cout)) {
                u.tiny = true; u.bm = u.bn = 1; u.splits = 1; u.group = 1; u.fused_combine = false;
                u.kernel = std::string("unet_tiny (down)") + (k < nd - 1 ? " -> lrelu s2d + relu" : "");
            } else
            u.kernel = std::string(km ? "igemm3x3<km>" : "igemm3x3") + (u.splits > 1 ? (u.fused_combine ? " (split-K combined in the launch)" : "+splitk_reduce") : "") +
                       (fused_prepare && k < nd - 1 ? " -> lrelu s2d + relu" : "");
            note_partial(u, (size_t)B * h * h * l.cout);
            launches.push_back(u);
            if (!fused_prepare && k < nd - 1) {
                UnetLaunch q; q.kind = 3; q.level = k; q.name = "L" + std::to_string(k) + ".prepare"; q.kernel = "unet_prepare";
                launches.push_back(q);
            }
        }
        for (int k = nd - 1; k >= 1; --k) {
            const UnetLevel &l = L[k];
            const int h = hd(k);                                       // source extent; writes 2h
            UnetLaunch u; u.kind = 1; u.level = k; u.name = "L" + std::to_string(k) + ".up";
            tile_for(B * h * h, l.up_cout, 4 * l.up_cin / ktc(), 4, &u, true, (size_t)B * 4 * h * h * l.up_cout);
            if (tiny_ok(B, h, l.up_cin, 4, 1, l.up_cout)) {
                u.tiny = true; u.bm = u.bn = 1; u.splits = 1; u.group = 1; u.fused_combine = false;
                u.kernel = "unet_tiny (sub-pixel up)";
            } else
            u.kernel = std::string("igemm3x3 (sub-pixel)") + (u.splits > 1 ? (u.fused_combine ? " (split-K combined in the launch)" : "+splitk_reduce") : "");
            note_partial(u, (size_t)B * 4 * h * h * l.up_cout);
            launches.push_back(u);
        }
        {
            const UnetLevel &l = L[0];
            const int h = hd(0);
            UnetLaunch u; u.kind = 1; u.level = 0; u.last = true; u.name = "L0.up";
            if (L[0].up_rl >= 0) {
                u.kind = 5; u.kernel = "rowlast128 (GEMM form on the row kernel, fp32 [B][H][W][4 x output_nc] out)";
                launches.push_back(u);
                UnetLaunch q; q.kind = 2; q.level = 0; q.name = "L0.shuffle"; q.kernel = "pixel_shuffle_tanh (+ bias)";
                launches.push_back(q);
            } else if (last_direct) {
                u.kind = 4; u.kernel = "last_conv (direct sub-pixel kernel: transposed conv + bias + tanh + tensor2im)";
                launches.push_back(u);
            } else {
            tile_for(B * h * h, 4 * l.up_cout, 9 * l.up_cin / 32, 1, &u, false, (size_t)B * h * h * 4 * l.up_cout);
            // N = 12: the narrowest tile the implicit GEMM has (128 x 32; 64 x 64 by the general rule: 4.47 -> 4.20 ms at eight frames)
            if (last_bm >= 0) { u.bm = 128; u.bn = 32; u.splits = 1; u.group = 1; }      // last_tile = -1: the general rule above, as the host-sequenced form runs it
            if (last_bm > 0 && last_bn > 0) { u.bm = last_bm; u.bn = last_bn; }
            u.kernel = "igemm3x3 (GEMM form, N = 4 parities x output_nc)";
            note_partial(u, (size_t)B * h * h * 4 * l.up_cout);
            launches.push_back(u);
            UnetLaunch q; q.kind = 2; q.level = 0; q.name = "L0.shuffle"; q.kernel = "pixel_shuffle_tanh";
            launches.push_back(q);
            }
        }
        // workspace: [arrival counters][Y ping-pong][R_k ...][U ping-pong][G][d scratch (fused_prepare = 0)][split-K slabs]
        size_t off = 0;
        auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
        cnt_off = take(kUnetCounters * sizeof(unsigned));
        size_t ymax[2] = {0, 0}, umax[2] = {0, 0}, dmax = 0;
        for (int k = 0; k < nd; ++k) {
            const size_t hy = (size_t)hd(k);                           // Y_k is [B][hd(k)][hd(k)][s2d_k]
            ymax[k & 1] = std::max(ymax[k & 1], (size_t)B * hy * hy * L[k].s2d * elt());
            dmax = std::max(dmax, (size_t)B * hy * hy * L[k].cout * 4);
        }
        for (int k = nd - 1; k >= 1; --k) {
            const size_t ho = 2 * (size_t)hd(k);
            umax[k & 1] = std::max(umax[k & 1], (size_t)B * ho * ho * L[k].up_cout * elt());
        }
        y_off[0] = take(ymax[0]); y_off[1] = take(ymax[1]);
        r_off.assign(nd, 0);
        for (int k = 0; k < nd; ++k) { const size_t h = (size_t)hd(k); r_off[k] = take((size_t)B * h * h * L[k].cout * elt()); }
        u_off[0] = take(umax[0]); u_off[1] = take(umax[1]);
        g_off = take((size_t)B * hd(0) * hd(0) * 4 * output_nc * 4);
        dtmp_off = fused_prepare ? off : take(dmax);
        part_off = take(part_bytes);
        total = off;
        return bp;
    }
};

}  // namespace lspf2f

using namespace lspf2f;

namespace {

struct UnetGraphKey {
    const void *feat, *cand, *out, *out_u8, *ws, *blob;
    int cand_batch, batch;
    bool operator==(const UnetGraphKey &o) const
    {
        return feat == o.feat && cand == o.cand && out == o.out && out_u8 == o.out_u8 && ws == o.ws && blob == o.blob && cand_batch == o.cand_batch && batch == o.batch;
    }
};
struct UnetGraph {
    UnetGraphKey key{};
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    void reset()
    {
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        exec = nullptr; graph = nullptr;
    }
};
const size_t kMaxUnetGraphs = 8;

thread_local std::string g_uerr;
int ufail(int code, const std::string &msg) { g_uerr = msg; return code; }
int uhipfail(hipError_t e, const std::string &what) { return ufail(LSPUNET_ERR_HIP, what + ": " + hipGetErrorString(e)); }

}  // namespace

struct lspunet_handle {
    UnetPlan plan;
    lspunet_config cfg{};
    const char *blob = nullptr;
    size_t blob_size = 0;
    char *ws = nullptr;
    size_t ws_size = 0;
    bool use_graph = true, counters_clean = false;
    hipStream_t cap_stream = nullptr;
    std::vector<UnetGraph> graphs;
    size_t next_victim = 0;
    void drop_graphs()
    {
        for (auto &g : graphs) g.reset();
        graphs.clear();
    }
    ~lspunet_handle()
    {
        drop_graphs();
        if (cap_stream) (void)hipStreamDestroy(cap_stream);
    }
};

// one launch of the planned forward
static int run_unet_launch(lspunet_handle *h, const UnetLaunch &u, const float *feat, const float *cand, int cand_batch, float *out, unsigned char *out_u8, int B, hipStream_t s)
{
    const UnetPlan &P = h->plan;
    auto wsf = [&](size_t off) { return reinterpret_cast<float *>(h->ws + off); };      // (typed by the kernels: the plan's storage type for Y / R / U, fp32 for G and the slabs)
    auto bl = [&](int64_t off) { return off < 0 ? nullptr : reinterpret_cast<const float *>(h->blob + off); };
    hipError_t e = hipSuccess;
    if (u.kind == 0) {
        UnetInputParams q{};
        q.feat = feat; q.cand = cand; q.out = wsf(P.cur.y_off[0]);
        q.B = B; q.S = P.size; q.C = P.input_nc; q.feat_nc = P.feat_nc; q.cand_bcast = cand_batch == 1 && B > 1; q.s2d_c = P.L[0].s2d; q.qc = P.L[0].cin_p;
        q.cols = 64;
        while ((P.size / 2) % q.cols) q.cols >>= 1;
        e = launch_unet_input(q, P.dtype, s);
    } else if (u.kind == 3) {
        PrepareParams q{};
        if (u.level < 0) {          // the input, already concatenated by the caller (input_pass = 0)
            q.src = feat; q.nchw = 1; q.B = B; q.H = P.size; q.W = P.size; q.C = P.input_nc; q.slope = 1.f;
            q.s2d = wsf(P.cur.y_off[0]); q.s2d_c = P.L[0].s2d; q.relu = nullptr;
        } else {
            const int k = u.level, hh = P.hd(k);
            q.src = wsf(P.cur.dtmp_off); q.nchw = 0; q.B = B; q.H = hh; q.W = hh; q.C = P.L[k].cout; q.slope = 0.2f;
            q.s2d = wsf(P.cur.y_off[(k + 1) & 1]); q.s2d_c = 4 * P.L[k].cout; q.relu = wsf(P.cur.r_off[k]);
        }
        e = launch_unet_prepare(q, s);
    } else if (u.kind == 4) {
        const UnetLevel &l = P.L[0];
        LastConvParams q{};
        q.src0 = wsf(P.cur.r_off[0]); q.src1 = wsf(P.cur.u_off[1]); q.dtype = P.dtype; q.w = bl(l.up_sub); q.bias = bl(l.up_bias);
        q.out = out; q.out_u8 = out_u8; q.B = B; q.Hs = P.hd(0); q.Ws = P.hd(0); q.C0 = l.cout; q.C1 = l.cout; q.Cout = l.up_cout; q.apply_tanh = 1;
        e = launch_last_conv(q, s);
    } else if (u.kind == 5) {
        const UnetLevel &l = P.L[0];
        RowLastParams q{};
        q.src0 = wsf(P.cur.r_off[0]); q.src1 = wsf(P.cur.u_off[1]); q.w = h->blob + l.up_rl; q.out = wsf(P.cur.g_off);
        q.B = B; q.H = P.hd(0); q.W = P.hd(0); q.R = rowlast_rows(B, P.hd(0), P.hd(0)); q.dtype = P.dtype;
        e = launch_rowlast(q, s);
    } else if (u.kind == 2) {
        ShuffleParams q{};
        q.g = wsf(P.cur.g_off); q.out = out; q.out_u8 = out_u8; q.B = B; q.Hs = P.hd(0); q.Ws = P.hd(0); q.Cout = P.output_nc; q.apply_tanh = 1;
        if (P.L[0].up_rl >= 0) { q.paired = 1; q.bias = bl(P.L[0].up_bias); }      // rowlast128's column order; its GEMM has no epilogue: the bias is added here
        e = launch_pixel_shuffle(q, s);
    } else {
        const int k = u.level;
        const UnetLevel &l = P.L[k];
        const int hh = P.hd(k);
        if (u.tiny) {
            UnetTinyParams q{};
            q.B = B; q.Hs = q.Ws = hh; q.M = B * hh * hh; q.slope = 0.2f;
            if (u.down) {
                q.src0 = wsf(P.cur.y_off[k & 1]); q.C0 = l.s2d; q.Cout = l.cout; q.w = bl(l.down_w); q.scale = bl(l.down_scale); q.shift = bl(l.down_shift);
                if (k == P.nd - 1) { q.out = wsf(P.cur.r_off[k]); q.relu = 1; }
                else { q.s2d_out = wsf(P.cur.y_off[(k + 1) & 1]); q.relu_out = wsf(P.cur.r_off[k]); }
            } else {
                q.up = 1; q.src0 = wsf(P.cur.r_off[k]); q.C0 = l.cout;
                if (k < P.nd - 1) { q.src1 = wsf(P.cur.u_off[(k + 1) & 1]); q.C1 = l.cout; }
                q.Cout = l.up_cout; q.w = bl(l.up_w); q.scale = bl(l.up_scale); q.shift = bl(l.up_shift); q.relu = 1;
                q.out = wsf(P.cur.u_off[k & 1]);
            }
            const hipError_t e2 = launch_unet_tiny(q, s);
            if (e2 != hipSuccess) return uhipfail(e2, "launch " + u.name);
            return LSPUNET_OK;
        }
        IgemmParams p{};
        p.dtype = P.dtype; p.B = B; p.stride = 1;
        const int ktc = P.ktc();
        p.partial = wsf(P.cur.part_off);
        if (u.down) {
            // Conv2d(k4, s2, p1) as the 3x3 conv on the space-to-depth image Y_k: [B][hh][hh][s2d]
            p.src0 = wsf(P.cur.y_off[k & 1]); p.src1 = nullptr; p.w = bl(l.down_w); p.scale = bl(l.down_scale); p.shift = bl(l.down_shift);
            p.Hs = p.Ws = p.Ho = p.Wo = hh; p.C0 = p.Cin = l.s2d; p.C1 = 0; p.Cout = l.cout;
            p.Mout = p.M = B * hh * hh;
            const bool inner = k == P.nd - 1;
            const bool km = l.live || (P.fused_prepare && !inner);
            if (l.live) {
                static const unsigned sub[3] = {2u, 3u, 1u};          // live sub-rows of tap row 0, 1, 2 as a bit set over dy
                for (int ty = 0; ty < 3; ++ty)
                    for (int tx = 0; tx < 3; ++tx)
                        for (int dy = 0; dy < 2; ++dy)
                            for (int dx = 0; dx < 2; ++dx)
                                if (((sub[ty] >> dy) & 1u) && ((sub[tx] >> dx) & 1u)) p.kmask |= 1ull << ((ty * 3 + tx) * 4 + dy * 2 + dx);
                p.kblk = l.cin_p;
                p.ktiles_total = 16 * l.cin_p / ktc;
            } else {
                p.ktiles_total = 9 * l.s2d / ktc;
                if (km) {            // block 0 on a masked-K instance with every K-tile live (dense rows): what carries the dual store
                    p.kblk = ktc;
                    for (int t = 0; t < 9; ++t)
                        for (int j = 0; j < l.s2d / ktc; ++j) p.kmask |= 1ull << (t * 4 + j);
                }
            }
            if (inner) { p.out = wsf(P.cur.r_off[k]); p.relu = 1; }        // only the up-conv reads it, through its ReLU
            else if (P.fused_prepare) { p.s2d_out = wsf(P.cur.y_off[(k + 1) & 1]); p.relu_out = wsf(P.cur.r_off[k]); p.slope = 0.2f; p.out = nullptr; }
            else p.out = wsf(P.cur.dtmp_off);
        } else if (!u.last) {
            // ConvTranspose2d(k4, s2, p1) over cat(R_k, U_{k+1}) in sub-pixel form; stored ReLU'd (read only through the parent's uprelu)
            p.src0 = wsf(P.cur.r_off[k]); p.C0 = l.cout;
            if (k < P.nd - 1) { p.src1 = wsf(P.cur.u_off[(k + 1) & 1]); p.C1 = l.cout; }
            p.Cin = p.C0 + p.C1; p.Cout = l.up_cout; p.w = bl(l.up_w); p.scale = bl(l.up_scale); p.shift = bl(l.up_shift);
            p.Hs = p.Ws = hh; p.Ho = p.Wo = 2 * hh; p.up4 = 1; p.relu = 1;
            p.Mout = B * 4 * hh * hh; p.M = B * hh * hh;
            p.ktiles_total = 4 * p.Cin / ktc;
            p.out = wsf(P.cur.u_off[k & 1]);
        } else {
            // the outermost transposed conv as a 3x3 GEMM on the low-res source, N = 4 parities x output_nc, + bias; tanh in the shuffle pass
            p.src0 = wsf(P.cur.r_off[0]); p.C0 = l.cout; p.src1 = wsf(P.cur.u_off[1]); p.C1 = l.cout;
            p.Cin = p.C0 + p.C1; p.Cout = 4 * l.up_cout; p.w = bl(l.up_w); p.scale = bl(l.up_scale); p.shift = bl(l.up_shift);
            p.Hs = p.Ws = p.Ho = p.Wo = hh;
            p.Mout = p.M = B * hh * hh;
            p.ktiles_total = 9 * p.Cin / 32;
            p.out = wsf(P.cur.g_off);
        }
        p.splits = u.splits;
        p.ktiles_per_split = (p.ktiles_total + u.splits - 1) / u.splits;
        if (u.fused_combine) {
            p.tile_cnt = reinterpret_cast<unsigned *>(h->ws + P.cur.cnt_off);
            p.slab_bytes = (size_t)u.splits * p.Mout * p.Cout * sizeof(float);
        }
        e = launch_igemm(p, u.bm, u.bn, u.group, s);
        if (e == hipSuccess && u.splits > 1 && !u.fused_combine) e = launch_splitk_reduce(p, s);
    }
    if (e != hipSuccess) return uhipfail(e, "launch " + u.name);
    return LSPUNET_OK;
}

static int check_unet_forward(lspunet_handle *h, const float *feat, const float *cand, int cand_batch, float *out, unsigned char *out_u8, int B)
{
    if (!h || !feat) return ufail(LSPUNET_ERR_INVALID_ARGUMENT, "null argument");
    if (!out && !out_u8) return ufail(LSPUNET_ERR_INVALID_ARGUMENT, "at least one of out_dev / out_u8_dev is required");
    if (B < 1 || B > h->cfg.max_batch) return ufail(LSPUNET_ERR_INVALID_ARGUMENT, "batch out of range (max_batch)");
    const int cand_nc = h->plan.input_nc - h->plan.feat_nc;
    if (cand_nc > 0 && !cand) return ufail(LSPUNET_ERR_INVALID_ARGUMENT, "cand_image is required (input_nc > feat_nc)");
    if (cand_nc > 0 && cand_batch != 1 && cand_batch != B) return ufail(LSPUNET_ERR_SHAPE, "cand_batch must be 1 (broadcast) or equal to batch");
    if (cand_nc > 0 && !h->plan.input_pass) return ufail(LSPUNET_ERR_UNSUPPORTED, "input_pass=0 takes one concatenated tensor (feat_nc == input_nc)");
    if (!h->blob) return ufail(LSPUNET_ERR_STATE, "weights not bound (lspunet_bind_weights)");
    if (!h->ws) return ufail(LSPUNET_ERR_STATE, "workspace not bound (lspunet_bind_workspace)");
    h->plan.plan_batch(B);
    if (h->ws_size < h->plan.cur.total) return ufail(LSPUNET_ERR_STATE, "workspace too small for this batch (lspunet_workspace_bytes)");
    return LSPUNET_OK;
}

extern "C" {

const char *lspunet_last_error(void) { return g_uerr.c_str(); }
int lspunet_abi_version(void) { return LSPUNET_ABI_VERSION; }

int lspunet_create(const lspunet_config *cfg, const char *tune, lspunet_handle **out)
{
    if (!cfg || !out) return ufail(LSPUNET_ERR_INVALID_ARGUMENT, "null argument");
    if (cfg->abi_version != LSPUNET_ABI_VERSION) return ufail(LSPUNET_ERR_INVALID_ARGUMENT, "ABI version mismatch");
    if (cfg->max_batch < 1) return ufail(LSPUNET_ERR_INVALID_ARGUMENT, "max_batch must be >= 1");
    lspunet_handle *h = new (std::nothrow) lspunet_handle();
    if (!h) return ufail(LSPUNET_ERR_INVALID_ARGUMENT, "out of host memory");
    h->cfg = *cfg;
    h->use_graph = (cfg->flags & LSPUNET_FLAG_NO_GRAPH) == 0;
    UnetPlan &P = h->plan;
    P.dtype = cfg->dtype;
    P.input_nc = cfg->input_nc; P.feat_nc = cfg->feat_nc; P.output_nc = cfg->output_nc; P.ngf = cfg->ngf; P.nd = cfg->num_downs; P.size = cfg->size;
    // "key=value,key=value": applied once, before the plan is built; the library reads no environment
    std::string t = tune ? tune : "";
    size_t i = 0;
    while (i < t.size()) {
        size_t j = t.find(',', i);
        if (j == std::string::npos) j = t.size();
        std::string tok = t.substr(i, j - i);
        i = j + 1;
        const size_t a = tok.find_first_not_of(" \t"), b = tok.find_last_not_of(" \t");
        if (a == std::string::npos) continue;
        tok = tok.substr(a, b - a + 1);
        const size_t eq = tok.find('=');
        char *endp = nullptr;
        const long v = eq == std::string::npos || eq == 0 || eq + 1 >= tok.size() ? 0 : std::strtol(tok.c_str() + eq + 1, &endp, 10);
        if (!endp || *endp) { delete h; return ufail(LSPUNET_ERR_INVALID_ARGUMENT, "tune: expected key=integer, got '" + tok + "'"); }
        const std::string k = tok.substr(0, eq);
        if (k == "graph") h->use_graph = h->use_graph && v != 0;
        else if (k == "fused_prepare") P.fused_prepare = v != 0;
        else if (k == "input_pass") P.input_pass = v != 0;
        else if (k == "fused_splitk") P.fused_splitk = v != 0;
        else if (k == "last_direct") P.last_direct = v != 0;
        else if (k == "tiny") P.use_tiny = v != 0;
        else if (k == "dense0") P.dense0 = v != 0;
        else if (k == "last_tile") { P.last_bm = v < 0 ? -1 : (int)(v / 1000); P.last_bn = v < 0 ? -1 : (int)(v % 1000); }      // e.g. 128032 = 128 x 32; -1 = the general tiling rule
        else { delete h; return ufail(LSPUNET_ERR_INVALID_ARGUMENT, "tune: unknown key '" + k + "'"); }
    }
    if (P.last_bm > 0 && !igemm_group_supported(P.last_bm, P.last_bn, 1, false)) { delete h; return ufail(LSPUNET_ERR_INVALID_ARGUMENT, "tune: last_tile is not an implicit-GEMM tile"); }
    const std::string e = P.build();
    if (!e.empty()) { delete h; return ufail(LSPUNET_ERR_UNSUPPORTED, e); }
    P.plan_batch(cfg->max_batch);
    *out = h;
    return LSPUNET_OK;
}

int lspunet_destroy(lspunet_handle *h)
{
    delete h;
    return LSPUNET_OK;
}

int lspunet_num_tensors(const lspunet_handle *h) { return h ? (int)h->plan.params.size() : ufail(LSPUNET_ERR_INVALID_ARGUMENT, "null handle"); }

int lspunet_tensor_info(const lspunet_handle *h, int i, const char **key, int64_t dims[4], int *ndim)
{
    if (!h || i < 0 || i >= (int)h->plan.params.size()) return ufail(LSPUNET_ERR_INVALID_ARGUMENT, "bad tensor index");
    const UnetParam &p = h->plan.params[i];
    if (key) *key = p.key.c_str();
    if (ndim) *ndim = (int)p.dims.size();
    if (dims) for (size_t d = 0; d < p.dims.size() && d < 4; ++d) dims[d] = p.dims[d];
    return LSPUNET_OK;
}

int lspunet_set_tensor(lspunet_handle *h, const char *key, const float *host, size_t numel)
{
    if (!h || !key || !host) return ufail(LSPUNET_ERR_INVALID_ARGUMENT, "null argument");
    auto it = h->plan.index.find(key);
    if (it == h->plan.index.end()) return ufail(LSPUNET_ERR_INVALID_ARGUMENT, std::string("unexpected state-dict key: ") + key);
    UnetParam &p = h->plan.params[it->second];
    if (numel != p.numel())
        return ufail(LSPUNET_ERR_SHAPE, std::string("size mismatch for ") + key + ": got " + std::to_string(numel) + ", expected " + std::to_string(p.numel()));
    p.data.assign(host, host + numel);
    p.set = true;
    return LSPUNET_OK;
}

size_t lspunet_packed_bytes(const lspunet_handle *h) { return h ? h->plan.blob_bytes : 0; }

int lspunet_pack_weights(lspunet_handle *h, void *host_blob, size_t bytes)
{
    if (!h || !host_blob) return ufail(LSPUNET_ERR_INVALID_ARGUMENT, "null argument");
    const std::string e = h->plan.pack(host_blob, bytes);
    if (!e.empty()) return ufail(e.rfind("missing", 0) == 0 ? LSPUNET_ERR_MISSING_TENSOR : LSPUNET_ERR_INVALID_ARGUMENT, e);
    for (auto &p : h->plan.params) { std::vector<float>().swap(p.data); p.set = false; }     // the fp32 copies are no longer needed
    return LSPUNET_OK;
}

int lspunet_bind_weights(lspunet_handle *h, const void *dev_blob, size_t bytes)
{
    if (!h || !dev_blob) return ufail(LSPUNET_ERR_INVALID_ARGUMENT, "null argument");
    if (bytes < h->plan.blob_bytes) return ufail(LSPUNET_ERR_SHAPE, "packed weight arena too small");
    if ((uintptr_t)dev_blob % 256) return ufail(LSPUNET_ERR_INVALID_ARGUMENT, "weight arena must be 256-byte aligned");
    h->drop_graphs();
    h->blob = static_cast<const char *>(dev_blob);
    h->blob_size = bytes;
    return LSPUNET_OK;
}

size_t lspunet_workspace_bytes(const lspunet_handle *h, int batch)
{
    if (!h || batch < 1) return 0;
    size_t need = 0;
    for (int b = 1; b <= batch; ++b) need = std::max(need, h->plan.workspace_bytes(b));     // split-K slabs come and go with the batch: not monotonic
    return need;
}

int lspunet_bind_workspace(lspunet_handle *h, void *dev_workspace, size_t bytes)
{
    if (!h || !dev_workspace) return ufail(LSPUNET_ERR_INVALID_ARGUMENT, "null argument");
    if ((uintptr_t)dev_workspace % 256) return ufail(LSPUNET_ERR_INVALID_ARGUMENT, "workspace must be 256-byte aligned");
    h->drop_graphs();
    h->counters_clean = false;
    h->ws = static_cast<char *>(dev_workspace);
    h->ws_size = bytes;
    return LSPUNET_OK;
}

int lspunet_forward(lspunet_handle *h, const float *feat_dev, const float *cand_dev, int cand_batch, float *out_dev, unsigned char *out_u8_dev, int batch,
                    void *hip_stream)
{
    int rc = check_unet_forward(h, feat_dev, cand_dev, cand_batch, out_dev, out_u8_dev, batch);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    if (!h->counters_clean) {      // arrival counters of the in-launch split-K combines: zero once per binding, every last arriver resets its own
        const hipError_t e = hipMemsetAsync(h->ws + h->plan.cur.cnt_off, 0, kUnetCounters * sizeof(unsigned), s);
        if (e != hipSuccess) return uhipfail(e, "hipMemsetAsync (split-K arrival counters)");
        h->counters_clean = true;
    }
    bool eager = !h->use_graph;
    if (!eager) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (s != nullptr && hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone) eager = true;
    }
    if (eager) {
        for (const auto &u : h->plan.cur.launches) {
            rc = run_unet_launch(h, u, feat_dev, cand_dev, cand_batch, out_dev, out_u8_dev, batch, s);
            if (rc) return rc;
        }
        return LSPUNET_OK;
    }
    UnetGraphKey key{feat_dev, cand_dev, out_dev, out_u8_dev, h->ws, h->blob, cand_batch, batch};
    UnetGraph *g = nullptr;
    for (auto &c : h->graphs)
        if (c.exec && c.key == key) { g = &c; break; }
    if (!g) {
        if (!h->cap_stream) {
            const hipError_t e = hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking);
            if (e != hipSuccess) return uhipfail(e, "hipStreamCreateWithFlags");
        }
        if (h->graphs.size() < kMaxUnetGraphs) h->graphs.emplace_back();
        g = &h->graphs[h->next_victim++ % h->graphs.size()];
        g->reset();
        hipError_t e = hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal);
        if (e != hipSuccess) return uhipfail(e, "hipStreamBeginCapture");
        for (const auto &u : h->plan.cur.launches) {
            rc = run_unet_launch(h, u, feat_dev, cand_dev, cand_batch, out_dev, out_u8_dev, batch, h->cap_stream);
            if (rc) break;
        }
        e = hipStreamEndCapture(h->cap_stream, &g->graph);
        if (rc) { g->reset(); return rc; }
        if (e != hipSuccess) { g->reset(); return uhipfail(e, "hipStreamEndCapture"); }
        e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
        if (e != hipSuccess) { g->reset(); return uhipfail(e, "hipGraphInstantiate"); }
        g->key = key;
    }
    const hipError_t e = hipGraphLaunch(g->exec, s);
    if (e != hipSuccess) return uhipfail(e, "hipGraphLaunch");
    return LSPUNET_OK;
}

int lspunet_forward_timed(lspunet_handle *h, const float *feat_dev, const float *cand_dev, int cand_batch, float *out_dev, unsigned char *out_u8_dev, int batch,
                          void *hip_stream, float *ms_per_launch)
{
    int rc = check_unet_forward(h, feat_dev, cand_dev, cand_batch, out_dev, out_u8_dev, batch);
    if (rc) return rc;
    if (!ms_per_launch) return ufail(LSPUNET_ERR_INVALID_ARGUMENT, "null argument");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    if (!h->counters_clean) {
        const hipError_t e = hipMemsetAsync(h->ws + h->plan.cur.cnt_off, 0, kUnetCounters * sizeof(unsigned), s);
        if (e != hipSuccess) return uhipfail(e, "hipMemsetAsync (split-K arrival counters)");
        h->counters_clean = true;
    }
    const size_t n = h->plan.cur.launches.size();
    std::vector<hipEvent_t> ev(n + 1);
    for (auto &e : ev)
        if (hipEventCreate(&e) != hipSuccess) return ufail(LSPUNET_ERR_HIP, "hipEventCreate");
    (void)hipEventRecord(ev[0], s);
    for (size_t i = 0; i < n && !rc; ++i) {
        rc = run_unet_launch(h, h->plan.cur.launches[i], feat_dev, cand_dev, cand_batch, out_dev, out_u8_dev, batch, s);
        (void)hipEventRecord(ev[i + 1], s);
    }
    const hipError_t e = hipStreamSynchronize(s);
    if (!rc && e != hipSuccess) rc = uhipfail(e, "hipStreamSynchronize");
    for (size_t i = 0; i < n && !rc; ++i) (void)hipEventElapsedTime(&ms_per_launch[i], ev[i], ev[i + 1]);
    for (auto &x : ev) (void)hipEventDestroy(x);
    return rc;
}

int lspunet_num_launches(lspunet_handle *h, int batch)
{
    if (!h || batch < 1 || batch > h->cfg.max_batch) return ufail(LSPUNET_ERR_INVALID_ARGUMENT, "batch out of range");
    h->plan.plan_batch(batch);
    return (int)h->plan.cur.launches.size();
}

int lspunet_launch_info(lspunet_handle *h, int batch, int index, const char **name, const char **kernel, int *tile_m, int *tile_n, int *split_k)
{
    if (!h || batch < 1 || batch > h->cfg.max_batch) return ufail(LSPUNET_ERR_INVALID_ARGUMENT, "batch out of range");
    h->plan.plan_batch(batch);
    if (index < 0 || index >= (int)h->plan.cur.launches.size()) return ufail(LSPUNET_ERR_INVALID_ARGUMENT, "bad launch index");
    const UnetLaunch &u = h->plan.cur.launches[index];
    if (name) *name = u.name.c_str();
    if (kernel) *kernel = u.kernel.c_str();
    if (tile_m) *tile_m = u.bm;
    if (tile_n) *tile_n = u.bn;
    if (split_k) *split_k = u.splits;
    return LSPUNET_OK;
}

}  // extern "C"
