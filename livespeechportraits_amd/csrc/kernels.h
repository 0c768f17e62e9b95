// Host-visible launch interface of the gfx950 kernels (igemm.hip, small_layers.hip, edge_layers.hip).  Internal to liblspf2f.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <atomic>

namespace lspf2f {

// Exact unsigned 32-bit division by a launch-invariant divisor (Granlund-Montgomery round-up
// method): q = (t + ((n - t) >> s1)) >> s2 with t = mulhi(n, m).  ~5 VALU ops instead of the ~40 of
// a hardware-emulated integer divide in every workgroup's prologue.
struct FastDiv {
    unsigned m = 0, s1 = 0, s2 = 0;
    static FastDiv make(unsigned d)
    {
        FastDiv f;
        unsigned l = 0;
        while ((1ull << l) < d) ++l;                       // ceil(log2 d)
        f.m = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
        f.s1 = l < 1 ? l : 1;
        f.s2 = l > 0 ? l - 1 : 0;
        return f;
    }
#ifdef __HIPCC__
    __device__ __forceinline__ unsigned div(unsigned n) const
    {
        const unsigned t = __umulhi(n, m);
        return (t + ((n - t) >> s1)) >> s2;
    }
#endif
};

// One fused 3x3 convolution as an implicit GEMM:  M = B*Ho*Wo output pixels, N = Cout,
// K = 9*Cin ordered (ky, kx, ci) with ci running over src0's channels then src1's.
struct IgemmParams {
    // activations / weights are fp32 (dtype 0) or bf16 (dtype 1: bf16 storage, fp32 accumulate / epilogue)
    const void *src0, *src1;    // NHWC [B][Hs][Ws][C0|C1]; src1 == nullptr when C1 == 0
    const void *w;              // [Cout][9*Cin]
    const float *scale, *shift; // [Cout] folded BatchNorm, or nullptr (identity)
    const void *residual;       // NHWC [M][Cout] or nullptr
    void *out;                  // NHWC [M][Cout]
    float *partial;             // split-K scratch [splits][M][Cout] (splits > 1), always fp32
    int dtype;                  // 0 = fp32, 1 = bf16, 2 = fp16 (16-bit storage, fp32 accumulate / epilogue)
    int B, Hs, Ws, Ho, Wo;
    int C0, C1, Cin, Cout;
    int stride;                 // 1 | 2
    int up;                     // nearest x2 upsample in front of the conv (Ho = 2*Hs), 9-tap gather form
    int up4;                    // same op in sub-pixel form: 4 parities x 2x2 taps, w = [4][Cout][4*Cin]
    int relu;
    int M;                      // GEMM rows per launch slice (up4: B*Hs*Ws per parity, else B*Ho*Wo)
    int Mout;                   // output pixels B*Ho*Wo
    int ktiles_total;           // taps*Cin/(32|64): a K-tile is 128 B of channels (taps = 9, or 4 for up4)
    int ktiles_per_split;
    int splits;
    FastDiv div_rhw, div_rw;    // dividers by the M-space extents (filled by launch_igemm)
    FastDiv div_gx, div_tile, div_cin;   // by gridDim.x, by the tile count of the fast-running tile index (ntn, or ntm when xcd == 2), by Cin:
    int ntm, ntn, gx, gy;       //   the per-block scalar divisions of the prologue as multiplies; gx, gy = the grid (filled by the launcher)
    float *psum, *psq, *pshift; // InstanceNorm plans, fused route: per-wave sums of (x - c), (x - c)^2 and the shift c (= the group's
    int in_groups;              //   first row), [B][in_groups][Cout] each (in_groups = wave row-groups per frame); nullptr otherwise
    size_t slab_bytes;          // bytes of the split-K scratch (buffer-descriptor range of the fused combine)
    unsigned *tile_cnt;         // fused split-K: one arrival counter per (parity, M-tile, N-tile), zero between launches; nullptr = the
                                //   partial slabs are combined by a separate splitk_reduce launch
    int out_f32;                // write fp32 output whatever the storage type (GEMM form of the last conv)
    int xcd;                    // block->tile order: 0 dispatch order, 1 per-XCD chunks m-major, 2 per-XCD chunks n-major (filled by launch_igemm)
    int xcd_force;              // 0 = by rule; 1 + mode forces it (tools)
    unsigned long long *stamps; // -DLSPF2F_IGEMM_STAMPS builds: [blocks][4 waves][16] cycle counters (tools/time_conv.py)
    unsigned long long kmask;   // 0 = every (tap, channel) K-tile exists.  Else bit (tap * 4 + j) says whether channel quarter j of tap `tap` does: the space-to-depth form of a
    int kblk;                   //   4x4 / stride-2 conv (lspf2f_conv3x3, k_group -4) has 16 live (tap, quarter) pairs of 36; kblk = channels per quarter; w holds only the live ones, in order
    int dbg;                    // ablation bits, honoured only by builds with -DLSPF2F_ABLATE (tools/ablate.sh): 1 no refetch,
                                // 4 no barrier, 8 no buffer flip, 16 no epilogue, 32 no K loop
    // `small` U-Net plans (masked-K instances and their reduce launches only; last so that every other field keeps its kernarg offset): instead of `out`, write
    void *s2d_out;              //   leaky_relu(v, slope) into the space-to-depth image [B][Ho/2][Wo/2][4 * Cout] (channel (dy * 2 + dx) * Cout + c) the next Conv2d(k4, s2, p1) reads
    void *relu_out;             //   and relu(v) into the NHWC skip tensor [B][Ho][Wo][Cout] the up-conv reads (both in the storage type); nullptr / nullptr = the plain `out`
    float slope;
};

// Kernel attributes (dynamic-LDS cap) are per device: an AttrMask has one bit per HIP device that already has the attribute of one
// kernel instantiation.  attr_needed_on_this_device() only LOOKS; the caller marks the device with attr_done_on_this_device() after
// hipFuncSetAttribute has succeeded, so a failed call is retried by the next launch, and a second host thread either sees the bit
// (the attribute is set) or sets the same attribute again (idempotent) -- it can never launch ahead of it.
struct AttrMask { std::atomic<unsigned long long> bits{0}; };
inline int attr_device_bit()
{
    int dev = 0;
    return (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev <= 63) ? dev : -1;
}
inline bool attr_needed_on_this_device(const AttrMask &mask)
{
    const int dev = attr_device_bit();
    return dev < 0 || !(mask.bits.load(std::memory_order_acquire) >> dev & 1ull);
}
inline void attr_done_on_this_device(AttrMask &mask)
{
    const int dev = attr_device_bit();
    if (dev >= 0) mask.bits.fetch_or(1ull << dev, std::memory_order_release);
}

struct TileConfig { int bm, bn; };
// tile shapes the igemm kernel is instantiated for
static const TileConfig kTileConfigs[] = {{128, 32}, {128, 128}, {128, 64}, {64, 128}, {64, 64}, {32, 128}, {32, 64}};
static const int kNumTileConfigs = sizeof(kTileConfigs) / sizeof(kTileConfigs[0]);

bool igemm_tile_supported(int bm, int bn);
// g = K-tiles per pipeline step: 1 (all shapes), 2 (128x64, 64x64), 4 (64x64, 32x64)
bool igemm_group_supported(int bm, int bn, int g, bool up);
hipError_t launch_igemm(const IgemmParams &p, int bm, int bn, int g, hipStream_t s);
hipError_t launch_splitk_reduce(const IgemmParams &p, hipStream_t s);

// Winograd F(2x2, 3x3) form of the stride-1 3x3 conv (wino.hip): fp32, one source, H % 8 == 0, W % 16 == 0, C % 8 == 0,
// N % (32 * nb) == 0 with nb = 32-channel blocks per wave (1 | 2).  A workgroup owns 4 x 8 Winograd tiles (8 x 16 output pixels) x 32 nb
// channels; K (the input channels) may be split 2..8 ways, combined inside the launch by the last-arriving workgroup of a tile.
struct WinoParams {
    const float *src;             // NHWC [B][H][W][C]
    const float *u;               // G g G^T in the fragment order of pack_wino_weights()
    const float *scale, *shift;   // [N] folded BatchNorm, or nullptr
    const float *residual;        // NHWC [B][H][W][N] or nullptr
    float *out;                   // NHWC [B][H][W][N]
    float *partial;               // splits > 1: fp32 slabs [splits][B*H*W][N]
    unsigned *tile_cnt;           // splits > 1: one arrival counter per (tile-block, channel group), zero between launches
    int B, H, W, C, N, relu, splits;
    // A-B switches of tools (all 0 = the shipped kernel): epilogue operands fetched in the epilogue instead of up front; block order 1 + mode
    // (0 dispatch order, 1 tile-block-major, 2 channel-group-major) instead of by operand size; copies as one block ahead of the MFMAs
    // instead of between them; the four MFMAs of an accumulator back to back instead of rotating over the accumulators
    int nopre, xcd_force, no_il, no_rot;
    int ureg;                   // one channel block per wave: U fragments by plain loads into registers instead of LDS-DMA + ds_read (wino.hip, UR form); 2: four register sets
    int out_wt;                 // tune key `out_wt`: the output leaves through write-through (sc1) stores (wino.hip WT instances; A-B runs)
    int prio;                   // tune key `wino_prio`: wave priority by K-loop progress (ProgressPrio, wino_common.h): 1..3 = scheme on the register form only, 4..6 = scheme 1..3 on every Winograd loop
    float *psum, *psq, *pshift; // InstanceNorm plans: per (frame, tile-block, channel) sums of (x - c), (x - c)^2 and the shift c (the tile-block's first pixel) of the
                                // 128 output pixels a workgroup writes, [B][tile-blocks per frame][N]; nullptr = no statistics (see instnorm.hip)
    // filled by launch_wino
    int steps_per_split, ntb, nng, tby, tbx, nmajor, xcd;
    size_t slab_bytes;
    FastDiv div_plane, div_fast, div_tbf, div_tbx;
    unsigned long long *stamps;   // -DLSPF2F_WINO_STAMPS builds: [blocks][4 waves][8] cycle counters (tools/wino_stamps.py)
};
bool wino_supported(const WinoParams &p, int nb);
hipError_t launch_wino(const WinoParams &p, int nb, hipStream_t s);

// Round 6: 2..4 consecutive wino3x3<1> layers of ONE shape (the convs of one or two ResidualBlocks, models/networks.py:650-675) as ONE launch of
// nlayers x (workgroups of a layer) workgroups: blockIdx / wgs = the layer.  A workgroup of layer k > 0 requests its weights, then waits on the arrival
// counters of the <= 9 tile-blocks of layer k - 1 its raw patch reads (wino.hip, wino3x3_chain).  Same arithmetic in the same order as nlayers launches
// of wino3x3<1>: bit-identical.  The layers must not alias each other's outputs (a buffer is written once per launch).
static const int kWinoChainMax = 4;
static const int kWinoArriveStride = 32;      // 32-bit words between two gate counters (one per 128-byte line)
struct WinoChainLayer {
    const float *src, *u, *scale, *shift, *residual;
    float *out;
    int relu;
    int res_fresh;                // the residual was written by an earlier layer of THIS launch (read behind the gate, past the L1)
};
struct WinoChainParams {
    WinoParams c;                 // the shared shape: B, H, W, C, N, splits, partial, tile_cnt, prio (src / u / scale / shift / residual / out / relu unused)
    WinoChainLayer L[kWinoChainMax];
    int nlayers;
    unsigned *arrive;             // [nlayers - 1][ntb][kWinoArriveStride] arrival counters (word 0 of each line), zero between launches: += 1 per finished (tile-block, channel group) of layer k, then += 1 per
                                  // workgroup of layer k + 1 that has read it; the last reader resets it
    unsigned *fail;               // one word, |= 1 when a gate gave up (spin_limit polls): the launch then finishes with garbage instead of hanging
    unsigned spin_limit;
    int sc1_loads;                // A-B arm: read what the previous layer wrote with sc1 loads (past the L1) instead of one acquire at the gate + plain loads
    // filled by launch_wino_chain
    int wgs;                      // workgroups per layer
    FastDiv div_wgs;
};
bool wino_chain_supported(const WinoChainParams &p);
hipError_t launch_wino_chain(const WinoChainParams &p, hipStream_t s);
void pack_wino_weights(const float *oihw, int cin, int cout, float *out);   // host: OIHW [cout][cin][3][3] -> [cout/32][4][cin/8][4][64][4]

// Winograd F(4x4, 3x3) form of the same conv (wino4.hip): H % 16 == 0, W % 32 == 0, C % 8 == 0, N % 32 == 0.  A workgroup owns 4 x 8 tiles of
// 4 x 4 output pixels (16 x 32 pixels) x 32 channels, one workgroup per CU; K may be split 2..8 ways like wino3x3.  Same parameter block.
bool wino4_supported(const WinoParams &p);
hipError_t launch_wino4(const WinoParams &p, hipStream_t s);
void pack_wino4_weights(const float *oihw, int cin, int cout, float *out);  // host: OIHW [cout][cin][3][3] -> [cout/32][wave 4][cin/8][9][64][4]

// Upsample(x2, nearest) + conv3x3 over the concat of two equally wide sources (or one) as a 9-multiply Winograd form (winoup.hip): fp32,
// Hs % 4 == 0, Ws % 8 == 0, C0 % 8 == 0, C1 in {0, C0}, N % (32 * nb) == 0.  A workgroup (3 waves) owns 4 x 8 source pixels (8 x 16 output pixels) x
// 32 nb channels; K may be split 2..8 ways, combined inside the launch.
struct WinoUpParams {
    const float *src0, *src1;     // NHWC [B][Hs][Ws][C0|C1]; src1 == nullptr when C1 == 0
    const float *u;               // fragment order of pack_winoup_weights()
    const float *scale, *shift;   // [N] folded BatchNorm, or nullptr
    float *out;                   // NHWC [B][2Hs][2Ws][N]
    float *partial;               // splits > 1: fp32 slabs [splits][B*4*Hs*Ws][N]
    unsigned *tile_cnt;           // splits > 1: arrival counters, zero between launches
    int B, Hs, Ws, C0, C1, N, relu, splits;
    int out_wt;                   // tune key `out_wt`: write-through output stores (winoup3x3<NB, true>)
    int prio;                     // tune key `wino_prio` >= 4: wave priority by K-loop progress here too (wino_common.h)
    float *psum, *psq, *pshift;   // InstanceNorm plans: per (frame, tile-block, channel) sums over the 128 output pixels a workgroup writes, as in WinoParams; nullptr = none
    // filled by launch_winoup
    int steps_per_split, ntb, nng, tby, tbx, nmajor;
    size_t slab_bytes;
    FastDiv div_plane, div_fast, div_tbf, div_tbx;
};
bool winoup_supported(const WinoUpParams &p, int nb);
hipError_t launch_winoup(const WinoUpParams &p, int nb, hipStream_t s);
void pack_winoup_weights(const float *oihw, int cin, int cout, float *out);   // host: OIHW -> [cout/32][3][cin/8][3][64][4]

// Tiny-M single-source conv (M <= 16 output pixels, whole input <= 64 KB): one launch, no split-K.
struct SmallMParams {
    const void *src;             // NHWC [B][Hs][Ws][Cin]            (fp32 or bf16, see dtype)
    const void *w;               // [Cout][9][Cin]
    const float *scale, *shift;  // [Cout] or nullptr
    const void *residual;        // [M][Cout] or nullptr
    void *out;                   // [M][Cout]
    int dtype;                   // 0 = fp32, 1 = bf16, 2 = fp16
    int B, Hs, Ws, Ho, Wo, Cin, Cout;
    int stride, up, relu, M;
    // optional: bytes [0, pf_bytes) at pf are the weights the NEXT launch of the forward streams (another weight-streaming layer).  A fifth wave
    // per workgroup pulls this workgroup's 1/gridDim share of them towards the chip while the other four work (small_layers.hip).
    const void *pf;
    unsigned pf_bytes;
    unsigned pf_dump;            // filled by launch_smallm: LDS byte offset of the dump slot
    int in_fused;                // InstanceNorm plans: InstanceNorm2d(affine=False, eps=1e-5) over each frame's pixels in the epilogue, then residual / ReLU (instead of an in_small launch)
    int stage_regs;              // 1: stage the input tensor through registers (the form of rounds 2-4: pairs of loads, each pair waited for before the next -- four dependent round trips
                                 // for a 4x4x512 tensor); 0 (default): LDS-DMA pieces, all in flight at once (fp32 storage; 16-bit inputs are widened on the way and keep the registers)
};
bool smallm_supported(const SmallMParams &p);
hipError_t launch_smallm(const SmallMParams &p, hipStream_t s);

// Full-K single-launch conv for the 16x16 / 8x8 levels at small batch (fullk.hip): fp32, stride 1, optional 9-tap nearest x2
// upsample and concat; a workgroup owns 16*pb pixels x 16 channels over the whole K, no split-K, no reduce launch.
struct FullKParams {
    const float *src0, *src1;    // NHWC [B][Hs][Ws][C0|C1]; src1 == nullptr when C1 == 0
    const float *w;              // [Cout][9][C0 + C1]
    const float *scale, *shift;  // [Cout] or nullptr
    const float *residual;       // [B][Ho][Wo][Cout] or nullptr
    float *out;                  // [B][Ho][Wo][Cout]
    int B, Hs, Ws, Ho, Wo, C0, C1, Cout;
    int up, relu;
    int ntm, ntn, tiles_per_img, wo_log2; // filled by launch_fullk
    int wtile;                   // weights in the tile-blocked layout of pack_fullk_weights() (the shipped path) instead of [Cout][9][Cin]
    unsigned long long *stamps;  // -DLSPF2F_FULLK_STAMPS builds: [blocks][4 waves][16] cycle counters
    // K split in two (8x8 outputs at batch 1: 128 tiles on 256 CUs): workgroup (tile, z) takes input-channel half z -- source z of a concat input, or
    // the lower / upper half of a single source, whose weights are then packed as pack_fullk_weights(rows, C0 / 2, 2, ...) -- and the second arriver
    // of a tile adds the two partial tiles in z order and runs the epilogue.  partial: [2][tiles][pb][256] floats; tile_cnt: zero on entry.
    int split;                   // 0 | 1 = off
    float *partial;
    unsigned *tile_cnt;
    int stride;                  // 0 | 1 = stride 1; 2 = stride-2 conv (Hs == 2 Ho, no upsample): the stride-2 convs of the small levels, whose 5-row bands
                                 // fit LDS once the K split halves the channels a workgroup stages
};
bool fullk_supported(const FullKParams &p, int pb);
hipError_t launch_fullk(const FullKParams &p, int pb, hipStream_t s);
void pack_fullk_weights(const float *rows, int c0, int nch, int cout, float *out);   // host: [Cout][9][nch * c0] -> tile-blocked

// The same single-launch full-K structure for the 8x8 / 4x4 / 2x2 levels of the 16-bit plans from 2 frames up (fullk16.hip): bf16 | fp16 storage, one source of
// 256 | 512 channels or two equal ones, stride 1 | 2 (one source) | nearest x2 upsample in front, Cout % 128 == 0.
struct FullK16Params {
    const void *src0, *src1;     // NHWC 16-bit [B][Hs][Ws][C0|C1]; src1 == nullptr when C1 == 0
    const void *w;               // 16-bit: tile-blocked (pack_fullk16_weights, wtile = 1) or rows [Cout][9][C0 + C1]
    const float *scale, *shift;  // [Cout] or nullptr
    const void *residual;        // 16-bit [B][Ho][Wo][Cout] or nullptr
    void *out;                   // 16-bit [B][Ho][Wo][Cout]
    int B, Hs, Ws, Ho, Wo, C0, C1, Cout;
    int up, relu, stride;
    int dtype;                   // 1 = bf16, 2 = fp16
    int wtile;
    int ntm, ntn, tiles_per_img, wo_log2;   // filled by launch_fullk16
};
bool fullk16_supported(const FullK16Params &p, int pb);
hipError_t launch_fullk16(const FullK16Params &p, int pb, hipStream_t s);
void pack_fullk16_weights(const uint16_t *rows, int c0, int nch, int cout, uint16_t *out);   // host: 16-bit [Cout][9][nch * c0] -> tile-blocked

// Patch-staged implicit GEMM for the stride-1 single-source convs of the 64x64 / 32x32 levels in 16-bit storage (patch16.hip): a workgroup owns
// 256 output pixels (4 rows x 64 or 8 rows x 32) x bn channels; the halo patch of each 64-channel block is staged once and read for all nine taps.
// H % (256 / tw) == 0, W % tw == 0, C % 64 == 0, Cout % bn == 0; weights = the implicit GEMM's rows [Cout][9][C].
struct PatchConvParams {
    const void *src, *w;          // NHWC 16-bit [B][H][W][C]; weights 16-bit [Cout][9][C]
    const float *scale, *shift;   // [Cout] or nullptr
    const void *residual;         // NHWC 16-bit [B][H][W][Cout] or nullptr
    void *out;                    // NHWC 16-bit [B][H][W][Cout]
    int B, H, W, C, Cout, relu;
    // sub-pixel up-conv form only (conv3x3_patchup16): second source of the concat (C1 = 0 | C channels), H x W = the LOW-res extent, out = [B][2H][2W][Cout], residual likewise,
    // weights = the implicit GEMM's up4 operand [4 parities][Cout][2][2][C + C1]
    const void *src1;
    int C1;
    int deep;                     // 64 channels per workgroup: 1 = the deep-ring form (conv3x3_patch16d: copies between the MFMAs, weights four K-tiles ahead), 0 = the first form
    int dtype;                    // 1 = bf16, 2 = fp16
    int dbg;                      // -DLSPF2F_ABLATE builds: 1 no copies in the K loop, 2 no fragment reads, 4 no MFMAs, 16 no epilogue
    unsigned long long *stamps;   // -DLSPF2F_PATCH_STAMPS builds: [blocks][8 waves][8] cycle sums (tools/probes/patch16_stamps.py)
    int tiles_x, tiles_per_img, ntm, ntn;   // filled by launch_patch16
    FastDiv div_tpi, div_tx, div_ntn;
};
bool patch16_supported(const PatchConvParams &p, int tw, int bn);
hipError_t launch_patch16(const PatchConvParams &p, int tw, int bn, hipStream_t s);
bool patchup16_supported(const PatchConvParams &p, int tw, int bn);
hipError_t launch_patchup16(const PatchConvParams &p, int tw, int bn, hipStream_t s);

// Weights-stationary conv for the 64 -> 64 and 128 -> 128 channel layers in bf16 storage (rowconv.hip): stride 1, one source,
// W % 64 == 0 (64 channels) / W % 32 == 0 (128 channels).
struct RowConvParams {
    const void *src, *w;          // NHWC bf16 [B][H][W][C]; weights bf16 [C][9][C]
    const float *scale, *shift;   // [64] or nullptr
    const void *residual;         // NHWC bf16 or nullptr
    void *out;                    // NHWC bf16
    int B, H, W, C;               // C = 64 | 128 channels in and out
    int R;                        // output rows per workgroup strip (rowconv_rows())
    int relu;
    int wfrag;                    // weights in the fragment order of pack_rowconv_weights() (the shipped path) instead of [64][9][64]
    int dtype;                    // 1 = bf16 (default when 0), 2 = fp16: the 16-bit storage type of src / w / residual / out
    int nsx, nsy, nblocks;        // filled by launch_rowconv
    FastDiv div_sx, div_sy;
};
bool rowconv_supported(const RowConvParams &p);
int rowconv_rows(int batch, int h, int w, int c);
hipError_t launch_rowconv(const RowConvParams &p, hipStream_t s);
void pack_rowconv_weights(const unsigned short *rows, unsigned short *out, int c);   // host: bf16 [c][9][c] -> [nb c/32][tap 9][kc c/16][lane 64][8]

// Activation-stationary conv for the 512 -> Cout layers of the 16x16 / 8x8 levels in bf16 storage (bandconv.hip): stride 1, one source of
// 512 channels, square frames of width 16 or 8, Cout % 32 == 0; weights in the fragment order of pack_bandconv_weights().
struct BandConvParams {
    const void *src, *w;          // NHWC bf16 [B][W][W][512]; weights bf16 [Cout/32][4][9][8][64][8]
    const float *scale, *shift;   // [Cout] or nullptr
    const void *residual;         // NHWC bf16 or nullptr
    void *out;                    // NHWC bf16 [B][W][W][Cout]
    int B, W, Cout, relu;
    int dtype;                    // 1 = bf16 (default when 0), 2 = fp16
    int ntiles, nblocks;          // filled by launch_bandconv
    FastDiv div_tiles;
};
bool bandconv_supported(const BandConvParams &p);
hipError_t launch_bandconv(const BandConvParams &p, hipStream_t s);
void pack_bandconv_weights(const unsigned short *rows, unsigned short *out, int cout);   // host: bf16 [cout][9][512] -> fragment order

// The last conv of bf16 plans as a row kernel (rowconv.hip): 3x3 conv on the low-res concat of two 64-channel sources with N = 4 parities x
// cout <= 16 outputs (the GEMM form of DESIGN.md 4.5), fp32 result [B][H][W][12] for pixel_shuffle_tanh.
struct RowLastParams {
    const void *src0, *src1;      // NHWC bf16 [B][H][W][64] each
    const void *w;                // bf16, fragment order of pack_rowlast_weights()
    float *out;                   // fp32 [B][H][W][4 * cout], columns in the PAIRED order n' = ((py * cout + co) * 2 + px) of pack_rowlast_weights (-> pixel_shuffle_tanh with paired = 1)
    float *out_nchw;              // != nullptr: the fused form -- shuffle + tanh in the epilogue, NCHW fp32 [B][cout][2H][2W] written directly (`out` unused)
    int cout, apply_tanh;         // fused form only
    int B, H, W, R;
    int dtype;                    // 1 = bf16 (default when 0), 2 = fp16
    int nsx, nsy, nblocks;        // filled by launch_rowlast
    FastDiv div_sx, div_sy;
};
bool rowlast_supported(const RowLastParams &p);
int rowlast_rows(int batch, int h, int w);
hipError_t launch_rowlast(const RowLastParams &p, hipStream_t s);
void pack_rowlast_weights(const unsigned short *rows, unsigned short *out, int nout);   // host: bf16 [nout][9][128] -> [tap 9][kc 4][lane 64][8]

// Upsample x2 + conv3x3 over the concat of two 128-channel sources -> 64 channels in sub-pixel form as a row kernel (rowconv.hip; L1.up of
// the bf16 plans).  H, W are the LOW-res extents; the output is [B][2H][2W][64].
struct RowUpParams {
    const void *src0, *src1;      // NHWC bf16 [B][H][W][128] each
    const void *w;                // bf16, fragment order of pack_rowup_weights()
    const float *scale, *shift;   // [64] or nullptr
    void *out;                    // NHWC bf16 [B][2H][2W][64]
    int B, H, W, R, relu;         // R = low-res rows per strip, even
    int dtype;                    // 1 = bf16 (default when 0), 2 = fp16
    int nsx, nsy, nblocks;        // filled by launch_rowup
    FastDiv div_sx, div_sy;
};
bool rowup_supported(const RowUpParams &p);
int rowup_rows(int batch, int h, int w);
hipError_t launch_rowup(const RowUpParams &p, hipStream_t s);
void pack_rowup_weights(const unsigned short *rows, unsigned short *out);   // host: bf16 [4][64][2][2][256] -> [nb 2][par 4][tap 4][kc 16][lane 64][8]

// First layer: cat([feature_map, cand_image]) -> Conv 3x3 s2 p1 -> ReLU, NCHW in, NHWC out.
struct FirstConvParams {
    const float *feat;   // [B][feat_nc][H][W]
    const float *cand;   // [cand_batch][cand_nc][H][W] or nullptr
    const float *w;      // [(ci*9 + ky*3 + kx)][Cout]
    void *out;           // NHWC [B][H/2][W/2][Cout], fp32 or bf16 (dtype)
    int dtype;
    int B, H, W, feat_nc, cand_nc, cand_batch, Cout;
    int ci_begin, ci_end;   // input-channel range of this pass ([0, feat_nc+cand_nc) = the whole layer)
    const float *base;      // optional pre-activation partial sums [1][H/2][W/2][Cout] to start from
    int relu;
    const float *bias;      // [Cout] conv bias (InstanceNorm plans: use_bias, networks.py:590) or nullptr; final pass only
    int force_direct;       // tests / A-B runs: 1 = the vector-ALU kernel, 2 = the register-staged matrix-core kernel (0 = by shape)
    int dbg;                // -DLSPF2F_ABLATE builds (first_conv_dma): 1 no MFMAs, 2 no stores, 4 no window copies, 8 no weight copies
};
hipError_t launch_first_conv(const FirstConvParams &p, hipStream_t s);

// Last layer: Upsample x2 -> Conv 3x3 s1 p1 over cat([src0, src1]) -> tanh, NHWC in, NCHW out.
struct LastConvParams {
    const void *src0, *src1;   // NHWC [B][Hs][Ws][C0|C1], fp32 or bf16 (dtype); weights stay fp32
    int dtype;
    const float *w;            // sub-pixel form [4 parities][Cout][2][2][Cin] (taps pre-summed)
    float *out;                // NCHW [B][Cout][2Hs][2Ws]
    int B, Hs, Ws, C0, C1, Cout;
    int apply_tanh;
    unsigned char *out_u8;     // optional HWC uint8 frame [B][2Hs][2Ws][Cout] = tensor2im(out); out may then be nullptr
    int route;                 // 0 = kernel chosen by shape (the matrix-core kernel, eight waves, where it applies); 1 strip, 2 rows, 3 generic, 4 the matrix-core kernel in
                               // its four-wave form of round 3 or fail, 5 the vector-ALU kernels by size (forced per handle: tests, A-B runs)
    const float *bias;         // [Cout] conv bias added before tanh (InstanceNorm plans) or nullptr
};
hipError_t launch_last_conv(const LastConvParams &p, hipStream_t s);

// second pass of the GEMM form of the last conv: [B][Hs][Ws][4*Cout] fp32 (channel = parity*Cout + co) ->
// tanh -> NCHW fp32 [B][Cout][2Hs][2Ws] and/or HWC uint8 (tensor2im)
struct ShuffleParams {
    const float *g;
    float *out;
    unsigned char *out_u8;
    int B, Hs, Ws, Cout, apply_tanh;
    int paired;        // columns of g in the order n' = ((py * Cout + co) * 2 + px) (rowlast128's operand order) instead of (py * 2 + px) * Cout + co
    const float *bias; // [Cout] added before the tanh, or nullptr (the `small` U-Net's outermost ConvTranspose2d carries a bias; the other variants' last conv has none)
};
hipError_t launch_pixel_shuffle(const ShuffleParams &p, hipStream_t s);
// one wave: {shader cycles, 100-MHz ticks} seen while spinning for duration_us of the constant counter (the clock the chip holds under load)
hipError_t launch_clock_probe(unsigned long long *out, unsigned duration_us, hipStream_t s);
// dst[0 .. bytes) = src[0 .. bytes) by a kernel (16-byte aligned pointers and size); either side may be pinned host memory
hipError_t launch_copy16(void *dst, const void *src, size_t bytes, hipStream_t s);
// read [src, src + bytes) with `blocks` workgroups and drop it (policy 1 plain loads, 2 non-temporal): the graph side branch of tune key `tail_prefetch`
hipError_t launch_touch_range(const void *src, size_t bytes, int policy, int blocks, unsigned *sink, hipStream_t s);

// InstanceNorm2d(affine=False, eps=1e-5) after a conv, fp32 NHWC (instnorm.hip).  `x` holds the raw conv output (bias included)
// and is normalised in place: x = relu?((x - mean[b][c]) * rstd[b][c] + residual?).
struct InstNormParams {
    float *x;                  // [B][hw][C]
    const float *partial;      // split-K partials [splits][B*hw][C] to fold into x first (splits > 1), else nullptr
    int splits;
    const float *bias;         // [C] added while folding the partials (the igemm epilogue adds it itself when splits == 1)
    const float *residual;     // [B][hw][C] or nullptr
    int relu;
    float *psum, *psq, *pshift;   // per group: sums of (x - c), (x - c)^2 and the shift c (the group's first row), [B][groups][C]
    float *mean, *rstd;        // [B][C]
    int B, hw, C, groups;
    int rows_per_group;        // rows behind one group's sums (the last group of a frame may hold fewer: hw - g * rows_per_group)
    int three_pass;            // in_small: 1 = the three-pass form of rounds 2-4 (tune key in_small_regs = 0; A-B runs), 0 = rows resident in registers
};
hipError_t launch_in_reduce_stats(const InstNormParams &p, hipStream_t s);   // fold partials, write raw x, 64-row partial sums
hipError_t launch_in_finalize(const InstNormParams &p, hipStream_t s);       // partial sums -> mean, rstd (double)
hipError_t launch_in_apply(const InstNormParams &p, hipStream_t s);          // streaming normalise (+ residual) (+ ReLU)
hipError_t launch_in_small(const InstNormParams &p, hipStream_t s);          // hw <= 1024: everything in one launch

// elementwise pass of the 'small' U-Net: space-to-depth (+ LeakyReLU) for the next 4x4 s2 conv, ReLU copy for the skip
struct PrepareParams {
    const float *src;
    int nchw, B, H, W, C;
    float slope;
    float *s2d; int s2d_c;
    float *relu;
};
hipError_t launch_unet_prepare(const PrepareParams &p, hipStream_t s);

}  // namespace lspf2f
