// Static execution plan of the feature2face generator: the residual U-Net unrolled into a flat
// list of fused conv launches, the expected state-dict tensors, the packed-weight blob layout
// and the liveness-based workspace layout.  Host-only C++ (no HIP types).
//
// Reference structure restated: models/networks.py:554-572 (large) / 458-476 (normal) build the
// nest; :592-640 (and :496-544) order each level's nn.Sequential; :650-675 ResidualBlock.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace lspf2f {

enum LayerKind { kFirstConv = 0, kIgemm = 1, kLastConv = 2 };
// InstanceNorm plans, per layer and batch: kInFused = sums in the igemm epilogue (wave shuffles), kInReduce = a streaming pass
// that also folds the split-K partials, kInSmall = one workgroup per (frame, 32 channels) does statistics + normalisation
enum InRoute { kInNone = 0, kInFused = 1, kInReduce = 2, kInSmall = 3, kInWino = 4 };   // kInWino: sums in the Winograd kernel's epilogue (per tile-block of 128 pixels)

struct TensorDesc {        // an activation tensor in the workspace (NHWC)
    std::string name;
    int c = 0, h = 0;      // channels, spatial extent (square)
    int def = -1;          // index of the producing layer
    int last_use = -1;     // index of the last consuming layer
    size_t offset = 0;     // byte offset in the workspace for the planned batch
};

struct LayerDesc {
    std::string name;
    LayerKind kind = kIgemm;
    int src0 = -1, src1 = -1, res = -1, out = -1;   // TensorDesc ids (-1: none / API tensor)
    int cin = 0, c0 = 0, c1 = 0, cout = 0;
    int hs = 0;            // spatial extent of the tensor(s) read
    int ho = 0;            // spatial extent written
    int stride = 1;
    bool up = false, relu = false, tanh_out = false, concat = false, residual = false;
    bool up4 = false;      // upsample conv executed in sub-pixel form (4 parities x 2x2 taps)
    std::string wkey;      // state-dict key of the OIHW weight
    std::string bnkey;     // state-dict prefix of the following BatchNorm2d ("" = none)
    std::string biaskey;   // state-dict key of the conv bias ("" = none; InstanceNorm plans: the level convs, networks.py:590)
    bool inorm = false;    // an InstanceNorm2d (affine=False, eps 1e-5) follows the conv: per-(frame, channel) statistics at run time
    int in_route = 0;      // per-batch: how those statistics are gathered (kInFused / kInReduce / kInSmall)
    int64_t w_off = -1, scale_off = -1, shift_off = -1;   // byte offsets in the packed blob
    int64_t wfk_off = -1;    // fp32 plans, 16x16 / 8x8 stride-1 layers: a second copy of the weights in the tile-blocked layout of the full-K kernel
    int64_t wbc_off = -1;    // bf16 plans, 512 -> Cout stride-1 layers at 16x16 / 8x8: a copy of the weights in the fragment order of bandconv.hip
    int64_t wrc_off = -1;    // bf16 plans, 64 -> 64 stride-1 layers: a copy of the weights in the fragment order of the weights-stationary kernel (rowconv.hip)
    int64_t wru_off = -1;    // bf16 plans, sub-pixel up-conv over two 128-channel sources -> 64 channels (L1.up): weights in the fragment order of rowup256
    int64_t wfk2_off = -1;   // fp32 plans, single-source 8x8 layers: the full-K kernel's weights packed as two half-sources (its K-split form at batch 1)
    int64_t wrl_off = -1;    // bf16 plans, last conv over two 64-channel sources: the GEMM-form weights in the fragment order of rowlast128 (rowconv.hip)
    int64_t wwg_off = -1;    // fp32 plans, stride-1 single-source convs at >= 32x32: G g G^T in the fragment order of the Winograd kernel (wino.hip)
    int64_t ww4_off = -1;    // fp32 plans, stride-1 single-source convs at >= 32x32 (extent % 32 == 0): the 6x6 G g G^T in the order of the F(4x4,3x3) kernel (wino4.hip)
    int64_t wwu_off = -1;    // fp32 plans, sub-pixel up-convs over two equally wide sources: the 9 transformed taps in the fragment order of winoup.hip
    int64_t wgemm_off = -1;  // bf16 plans, last conv only: the same sub-pixel weights as a 9-tap [4*cout][3][3][cin] bf16 GEMM operand
    // per-batch tiling decision
    int bm = 0, bn = 0, splits = 1, group = 1;   // group = K-tiles per pipeline step
    bool smallm = false;   // executed by the single-launch tiny-M kernel (M <= 16) instead of the igemm
    bool fused_splitk = false;   // fp32 plans: 2..8 K-splits combined inside the igemm launch by the last-arriving workgroup (no splitk_reduce
                                 // launch): +1.0 % at fp32 batch 1, none at batch 8, -2.4 % on bf16 batch 8 (A-B-A-B, one session; restricted
                                 // to the <= 16x16 / <= 8x8 levels of a bf16 plan it still loses 0.7-0.9 %)
    bool bandconv = false;  // executed by the activation-stationary kernel of the 16x16 / 8x8 levels (bandconv.hip, bf16 plans)
    int rowup = 0;         // > 0: executed by rowup256 (rowconv.hip) with this many low-res rows per strip
    int rowconv = 0;       // > 0: executed by the weights-stationary 64 -> 64 bf16 kernel (rowconv.hip) with this many output rows per strip
    int wino = 0;          // > 0: executed by the Winograd F(2x2,3x3) kernel (wino.hip) with this many 32-channel blocks per wave (1 | 2); `splits` = its K splits
    int wino4 = 0;         // 1: executed by the Winograd F(4x4,3x3) kernel (wino4.hip); `splits` = its K splits
    int winoup = 0;        // > 0: executed by the up-conv Winograd kernel (winoup.hip) with this many 32-channel blocks per wave; `splits` = its K splits
    int patch16 = 0;       // > 0: executed by the patch-staged 16-bit kernel (patch16.hip) with this tile width (64 | 32 pixels; 4 | 8 rows); bn = channels per workgroup
    int fullk = 0;         // > 0: executed by the full-K single-launch kernel (fullk.hip) with this many 16-pixel blocks per tile
                           // (splits == 2 with it: K in two halves over twice the workgroups, combined in the launch)
};

struct ParamDesc {         // an expected state-dict entry
    std::string key;
    std::vector<int64_t> dims;
    std::vector<float> data;
    bool set = false;
    size_t numel() const { size_t n = 1; for (auto d : dims) n *= (size_t)d; return n; }
};

struct Plan {
    int variant = 1, nres = 2, input_nc = 13, feat_nc = 1, output_nc = 3, ngf = 64, num_downs = 8, size = 512;
    bool keep_intermediates = false;
    int dtype = 0;             // 0: fp32 activations + weights; 1: bf16 storage (fp32 accumulate), first/last-layer weights fp32; 2: fp16 storage, likewise
                               // (the reference's opt.fp16 / autocast configuration; the row / band / up-conv kernels of the 16-bit plans are templated on
                               // the storage type, so an fp16 plan takes the same kernel per layer as the bf16 plan)
    int norm = 0;              // 0: BatchNorm2d in eval mode (folded, the shipped checkpoints); 1: InstanceNorm2d (norm_layer argument
                               // of the reference constructors, networks.py:555 / :459): conv biases on, statistics at run time, fp32 only
    int fullk16_levels = 3;    // 16-bit plans: which small levels run on the full-K kernel (fullk16.hip; tune key `fullk16`, bit mask as in fullk16_choice(); 0 = none)
    int fullk16_min_frames = 2; // ... from this many frames up (tune key `fullk16_min_frames`)
    bool use_bandconv = true;  // bf16 plans: tune key `bandconv=0` puts the 16x16 / 8x8 layers back on the implicit GEMM (A-B runs)
    int bandconv_min_blocks = 128;  // (16x16 / 8x8 levels; the 4x4 / 2x2 levels, one tile per 2 / 8 frames, have their own bound below)
    int bandconv_min_frames_small = 1 << 30;   // 4x4 / 2x2 levels (a tile = 2 / 8 whole frames): never by default -- at 8 frames the 64 / 16
                                               // workgroups of such a launch lose to the igemm (normal 4460 -> 4388, large 2881 -> 2840 frames/s,
                                               // A-B-A-B); tune key `bandconv_min_frames` lowers it for measurements at larger batches   // ... and they only leave it when the launch has at least this many workgroups
    bool use_patch16 = true;   // 16-bit plans: tune key `patch16=0` keeps the stride-1 convs of the 64x64 / 32x32 levels on the implicit GEMM (A-B runs)
    bool use_patchup16 = true; // ... and `patchup16=0` the sub-pixel up-convs over 32x32 / 64x64 sources (conv3x3_patchup16)
    int patch16_deep = 1;      // ... and `patch16_deep=0` its 64-channel tiles in the first form (3-slot ring, copies in the load segment) instead of conv3x3_patch16d (A-B runs)
    int patch16_min_blocks = 192;   // ... which they only leave when the launch has at least this many workgroups (tune key `patch16_min_blocks`)
    bool use_rowup = true;     // bf16 plans: tune key `rowup=0` keeps L1.up on the implicit GEMM (A-B runs)
    bool rowlast_fused = true; // bf16 plans: rowlast128 applies pixel shuffle + tanh in its epilogue when only fp32 frames are wanted (tune key `rowlast_fused=0`: the two-launch form, A-B runs)
    bool use_rowlast = true;   // bf16 plans: tune key `rowlast=0` keeps the GEMM-form last conv on the implicit-GEMM kernel (A-B runs)
    int fullk_split_max_tiles = 128;   // tune key `fullk_split_tiles` (tools): 256 also splits the 16x16 layers at batch 1
    int use_fullk_s2 = 0;          // tune key `fullk_s2`: the stride-2 convs of the small levels at batch 1 on the K-split full-K kernel instead of the implicit
                                   // GEMM + split-K reduce: 1 = L4 / L5 / L6.down (16x16, 8x8, 4x4 outputs), 2 = only those writing <= 8x8 (L5 / L6.down).
                                   // Round 3 measured 1 as SLOWER for the whole forward although two of its three launches are shorter; round 4 found why
                                   // (tools/s2_layer_delta.py): L4.down itself loses 8 us and two far-away up-convs lose 9 us each
    bool use_fullk_split = true;   // the 8x8 layers at batch 1 run the full-K kernel with K in two halves over twice the workgroups (tune key `fullk_split=0` at
                                   // create: unsplit, A-B runs)
    bool use_wino = true;      // fp32 plans: stride-1 convs at >= 32x32 on the Winograd kernel (tune key `wino=0`: the implicit GEMM, A-B runs)
    bool use_wino4 = false;    // fp32 plans: ... and of those the layers wino4_choice() takes on the F(4x4,3x3) kernel (LSPF2F_FLAG_WINO4; measured slower at batch 1 and
                               // equal at batch 8, DESIGN.md 4.11, so off by default); decides whether the blob carries the 6x6 transformed weights
    int in_small_max_hw = 1024;  // `in_small_max_hw`: InstanceNorm plans take the one-launch route (in_small: one workgroup per (frame, 32 channels)) up to this many pixels per frame; above it
                                 // in_reduce_stats + in_finalize + in_apply spread the frame over the chip
    bool in_smallm_fused = true; // `in_smallm_fused`: InstanceNorm plans normalise in conv3x3_smallm's epilogue (a workgroup holds every pixel of its channels) instead of an in_small launch behind it
    bool in_wino_stats = true;   // `in_wino_stats`: InstanceNorm plans take a wino3x3 layer's statistics from its epilogue instead of a pass over its output
    int smallm_kb = 64;          // `smallm_kb`: largest input tensor (KB, fp32 in LDS) the tiny-M kernel takes.  64 until round 5; with LDS-DMA staging a 128-KB tensor (one workgroup per CU) is
                                 // one round of copies: L6.down (512 -> 512 at stride 2, 8x8 -> 4x4) leaves the 36-way split-K implicit GEMM + reduce launch at one frame
    int out_wt = 1;              // `out_wt`: wino3x3 / winoup3x3 write their output through (sc1 stores) instead of leaving it dirty in L2 for the end-of-kernel write-back: bit-identical, +0.6 % at batch 1, +0.3 % at batch 8, +1.0 % `normal` batch 1 (A-B-A-B x3, profiles/r05_outwt_ab.txt); 0 = plain stores
    bool fused_splitk16 = false; // `fused_splitk16`: 16-bit plans combine 2..8 K-splits inside the igemm launch like the fp32 plans do (off until measured: round 5)
    int wino_prio = 1;           // `wino_prio`: wino3x3<1>'s register form sets its wave priority by K-loop progress, the workgroup that is BEHIND leads (ProgressPrio, wino_common.h): bit-identical,
                                 // +0.7-0.85 % at batch 1 (two boxes, A-B-A-B x3 each), neutral at batch 8; 3 / 4..6 = the variants measured within 0.2 % of it (profiles/r05_wino_prio_ab.txt); 0 = off
    int wino_ureg = 1;           // `wino_ureg`: wino3x3<1> keeps its U fragments in registers (wino.hip UR form: 1 = three register sets, two steps ahead; 2 = four sets, A-B arm)
    bool wino_pre = true, wino_il = true, wino_rot = true;   // tools (`wino_pre` / `wino_il` / `wino_rot` of lspf2f_create_tuned): A-B switches of wino3x3
    int wino_xcd = -1, igemm_xcd = -1;                       // tools: forced block orders (-1 = by operand size)
    int winoup_nb = 0, winoup_target = 1024;   // tools (tune keys `winoup_nb` / `winoup_target`): force the channel blocks per wave / the workgroup count aimed at
    bool use_winoup = true;    // fp32 plans: sub-pixel up-convs on the up-conv Winograd kernel (tune key `winoup=0`: the implicit GEMM, A-B runs)
    bool use_rowconv = true;   // bf16 plans: 64 -> 64 layers on the weights-stationary kernel (tune key `rowconv=0`: the igemm, A-B runs)
    size_t elt() const { return dtype ? 2 : 4; }
    int ktile_channels() const { return dtype ? 64 : 32; }   // a K-tile is 128 B of channels
    bool layer_weights_typed(const LayerDesc &l) const { return l.kind == kIgemm; }   // else fp32
    // bf16: the last conv runs as an implicit GEMM on the low-res source (N = 4 parities x cout) + a pixel-shuffle/tanh pass
    bool last_as_gemm(const LayerDesc &l) const { return dtype != 0 && l.kind == kLastConv && l.cin % 64 == 0; }
    std::vector<LayerDesc> layers;
    std::vector<TensorDesc> tensors;
    std::vector<ParamDesc> params;
    std::map<std::string, int> param_index;
    size_t blob_bytes = 0;

    // per-batch state
    int planned_batch = 0;
    size_t act_bytes = 0;       // activation arena
    size_t partial_bytes = 0;   // split-K scratch
    size_t partial_offset = 0;
    size_t stats_bytes = 0;     // InstanceNorm plans: per-group sums and shifts [3][B][groups][C] + the finalised (mean, rstd) [2][B][C]
    size_t stats_offset = 0;
    int stats_groups_max = 0;
    // persistent region at workspace offset 0: pre-activation contribution of the candidate channels to the
    // first conv, [H/2][W/2][ngf] fp32 (constant per person: demo.py:89-95 builds img_candidates once)
    size_t cand_cache_bytes() const { return ((size_t)(size / 2) * (size / 2) * ngf * sizeof(float) + 255) / 256 * 256; }
    // head of the workspace: slot 0 = that per-person cache (lspf2f_set_candidates), slot 1 = the same quantity for a
    // candidate stack broadcast over ONE forward's batch (never aliases slot 0)
    // then kTileCounters arrival counters of the in-launch split-K combine (zero between launches; zeroed once per workspace binding)
    static const size_t kTileCounters = 16384;
    size_t counters_offset() const { return 2 * cand_cache_bytes(); }
    size_t persistent_bytes() const { return 2 * cand_cache_bytes() + kTileCounters * sizeof(unsigned); }

    // max_batch_forms > 0: the blob carries only the weight forms the plans of batch 1 .. max_batch_forms read (0: every form)
    std::string build(int variant, int input_nc, int feat_nc, int output_nc, int ngf, int num_downs,
                      int size, bool keep, int dtype = 0, int norm = 0, int max_batch_forms = 0);   // returns "" or an error message
    // weight forms a layer can be packed in (bit mask)
    enum : unsigned { kFormRows = 1, kFormFullK = 2, kFormFullK2 = 4, kFormWino = 8, kFormWino4 = 16, kFormWinoUp = 32, kFormRowUp = 64,
                      kFormBand = 128, kFormRow = 256, kFormGemmLast = 512 };
    static unsigned forms_used(const LayerDesc &tiled, const Plan &p);   // which form the kernel chosen for a (batch-planned) layer reads
    void assign_offsets(const std::vector<unsigned> *used);               // lays the blob out with the forms of `used` (nullptr: every form)
    int blob_pad_kb = 0;           // tune key `blob_pad_kb` (tools): empty KB in front of the first layer's weights
    bool keep_all_forms = false;   // tune key `all_forms=1`: every form whatever the batch range (tests that look at forms other batches would use)
    void plan_batch(int batch);
    size_t workspace_bytes(int batch) const;   // without mutating the current plan
    std::string pack(void *blob, size_t bytes) const;   // "" or error
    int64_t layer_flops(const LayerDesc &l) const;
    int64_t layer_act_bytes(const LayerDesc &l) const;
};

// tile / split-K heuristic shared by the planner and lspf2f_conv3x3
void choose_tiling(int M, int N, int ktiles, int par, bool up9, int dtype, int *bm, int *bn, int *splits, int *group);
// tiny-M kernel eligibility (mirrors smallm_supported() in small_layers.hip)
inline bool smallm_eligible(int M, int cin, int c1, int cout, size_t in_bytes, int max_kb = 64)
{
    return M <= 16 && c1 == 0 && cin % 256 == 0 && 9 * (cin / 4) <= 5 * 256 && in_bytes <= (size_t)max_kb * 1024 && cout % 2 == 0;
}
// weights-stationary kernel eligibility (mirrors rowconv_supported() in rowconv.hip): bf16 storage, one source of 64 or 128 channels,
// as many out, stride 1, no upsample, BatchNorm (folded) plans
inline bool rowconv_layer(int ho, int c0, int c1, int cout, int stride, bool up, bool up4, int dtype, bool inorm)
{
    if (dtype == 0 || c1 != 0 || cout != c0 || stride != 1 || up || up4 || inorm) return false;      // 16-bit storage: bf16 (1) or fp16 (2)
    return (c0 == 64 && ho % 64 == 0) || (c0 == 128 && ho % 32 == 0);
}
// row kernel of the sub-pixel up-conv (mirrors rowup_supported() in rowconv.hip)
inline bool rowup_layer(int hs, int c0, int c1, int cout, bool up4, int dtype, bool inorm)
{
    return dtype != 0 && up4 && c0 == 128 && c1 == 128 && cout == 64 && !inorm && hs % 32 == 0;
}
// activation-stationary kernel eligibility (mirrors bandconv_supported() in bandconv.hip)
inline bool bandconv_layer(int ho, int c0, int c1, int cout, int stride, bool up, bool up4, int dtype, bool inorm)
{
    return dtype != 0 && c0 == 512 && c1 == 0 && cout % 32 == 0 && stride == 1 && !up && !up4 && !inorm && (ho == 16 || ho == 8 || ho == 4 || ho == 2);
}
// patch-staged 16-bit kernel (mirrors patch16_supported() in patch16.hip): stride-1 single-source convs of >= 128 channels at 64x64 (tiles of 4 rows x 64 pixels) and
// 32x32 (8 rows x 32); returns the tile width (0 = keep the implicit GEMM) and the channels per workgroup: 128 when that still fills the chip, else 64
inline int patch16_choice(int batch, int ho, int c0, int c1, int cout, int stride, bool up, bool up4, int dtype, bool inorm, int min_blocks, int *bn)
{
    if (dtype == 0 || c1 != 0 || stride != 1 || up || up4 || inorm) return 0;
    if ((ho != 64 && ho != 32) || c0 % 64 || c0 < 128 || cout % 64) return 0;
    const long mtiles = (long)batch * ho * ho / 256;
    if (cout % 128 == 0 && mtiles * (cout / 128) >= min_blocks) { *bn = 128; return ho == 64 ? 64 : 32; }
    if (mtiles * (cout / 64) >= min_blocks) { *bn = 64; return ho == 64 ? 64 : 32; }
    return 0;
}
// its sub-pixel up-conv form (mirrors patchup16_supported()): up4 layers over one source or two equally wide ones at a 64x64 / 32x32 LOW-res extent
inline int patchup16_choice(int batch, int hs, int c0, int c1, int cout, bool up4, int dtype, bool inorm, int min_blocks, int *bn)
{
    if (dtype == 0 || !up4 || inorm || (c1 != 0 && c1 != c0)) return 0;
    if ((hs != 64 && hs != 32 && hs != 16) || c0 % 64 || c0 + c1 < 128 || cout % 64) return 0;
    const long mtiles = (long)batch * hs * hs / 256 * 4;
    if (hs >= 32 && cout % 128 == 0 && mtiles * (cout / 128) >= min_blocks) { *bn = 128; return hs == 64 ? 64 : 32; }
    if (mtiles * (cout / 64) >= min_blocks) { *bn = 64; return hs == 64 ? 64 : hs; }      // (16x16 sources: a tile = one whole low-res frame, 64 channels per workgroup only)
    return 0;
}
// full-K kernel eligibility (mirrors fullk_supported() in fullk.hip); returns the pixel blocks per tile (1 | 2) or 0
// which layers get the tile-blocked weight copy at pack time (independent of the batch: the blob layout must not depend on it)
inline bool fullk_layer(int hs, int ho, int c0, int c1, int cout, int stride, bool up, bool up4, int dtype)
{
    if (dtype != 0 || stride != 1 || up4) return false;
    if (ho != 2 && ho != 4 && ho != 8 && ho != 16) return false;     // (4x4 / 2x2 belong to the tiny-M kernel while the batch has <= 16 output pixels: fullk_choice)
    if (up ? 2 * hs != ho : hs != ho) return false;
    return (c0 == 128 || c0 == 256 || c0 == 512) && (c1 == 0 || c1 == c0) && cout % 128 == 0;
}
// the stride-2 convs of the small levels (outputs 16x16 / 8x8 / 4x4 from a single source of 256 | 512 channels) on the K-split full-K kernel: half the
// channels of their 5-row band fit LDS.  Batch-independent part (who gets the half-source weight copy) and the per-batch choice (pixel blocks per tile or 0).
inline bool fullk_s2_layer(int hs, int ho, int c0, int c1, int cout, int stride, bool up, bool up4, int dtype, bool inorm)
{
    return dtype == 0 && stride == 2 && !up && !up4 && !inorm && c1 == 0 && (c0 == 256 || c0 == 512) && cout % 128 == 0 &&
           (ho == 16 || ho == 8 || ho == 4) && hs == 2 * ho;
}
inline int fullk_s2_choice(int batch, int hs, int ho, int c0, int cout, int level = 1)
{
    if (level == 2 && ho > 8) return 0;
    if (batch != 1) return 0;                                 // measured at batch 1 only; from 2 frames up the implicit GEMM has rows enough
    const int nr = 16 / ho;                                   // one 16-pixel block per tile
    const long tiles = (long)batch * ((ho + nr - 1) / nr) * (cout / 16);
    const int rows = std::min(2 * (nr - 1) + 3, hs);
    if (((size_t)rows * hs + 1) * (c0 / 2 + 4) * sizeof(float) > 150 * 1024) return 0;
    return tiles <= 512 ? 1 : 0;                              // 2 x tiles workgroups, one per CU at a time: up to four rounds
}
// K split of the full-K kernel: when its tiles fill at most half the chip (8x8 outputs at batch 1: 4 x cout / 16 = 128 tiles on 256 CUs) and the input
// has two sources or one of >= 256 channels to halve
inline bool fullk_split(int batch, int ho, int c0, int c1, int cout, int pb, int max_tiles = 128)
{
    if (ho != 8 && ho != 16) return false;
    const int nr = pb * (16 / ho);
    const long tiles = (long)batch * (ho / nr) * (cout / 16);
    return tiles <= max_tiles && (c1 == c0 || (c1 == 0 && c0 >= 256));
}
inline int fullk_choice(int batch, int hs, int ho, int c0, int c1, int cout, int stride, bool up, bool up4, int dtype)
{
    if (!fullk_layer(hs, ho, c0, c1, cout, stride, up, up4, dtype)) return 0;
    if (up ? 2 * hs != ho : hs != ho) return 0;
    if ((c0 != 128 && c0 != 256 && c0 != 512) || (c1 != 0 && c1 != c0) || cout % 128) return 0;
    // whole tiles must fit one dispatch wave of the chip with room to spare: <= 512 workgroups (2 per CU on 256 CUs)
    const int ntn = cout / 16;
    // 4x4 / 2x2 frames (a tile = one whole frame, 16 / 4 of its 16 rows used): from the batch the tiny-M kernel stops taking (> 16 output pixels) up.  Measured per layer,
    // 512 -> 512 (tools/time_conv.py): 4x4 at 2 / 4 / 8 frames 9.1 / 8.9 / 9.3 us against 13.5 / 15.1 / 20.1 for igemm + reduce; 2x2 at 8 frames 8.7 against 13.7
    if (ho <= 4 && (long)batch * ho * ho <= 16) return 0;
    for (int pb = 1; pb <= (ho == 16 ? 2 : 1); ++pb) {
        const int nr = pb * (16 / ho);
        const long tiles = (long)batch * ((ho + nr - 1) / nr) * ntn;
        // 8x8 frames: up to four rounds of workgroups still beat the split-K implicit GEMM (8 frames: 28.1 against 35.6 us)
        if (ho == 8 && tiles > 512 && tiles <= 1024) return 1;
        // the band of source rows behind a tile must fit the 150 KB of LDS fullk_supported() (fullk.hip) grants: (rows * Ws + 1 zero pixel) x (C0 + 4 pad)
        // floats per source -- 133 KB for the widest shape the generators build (4 rows x 16 px x 512 ch)
        const int rows = std::min(up ? nr / 2 + 2 : nr + 2, hs);
        if ((size_t)(c1 ? 2 : 1) * ((size_t)rows * hs + 1) * (c0 + 4) * sizeof(float) > 150 * 1024) continue;
        if (tiles <= 256 || (pb == (ho == 16 ? 2 : 1) && tiles <= 512)) return pb;
    }
    return 0;
}
// The 16-bit twin (fullk16.hip): the 8x8 / 4x4 / 2x2 levels of the bf16 / fp16 plans from 2 frames up -- stride 1, stride 2 (one source) or nearest x2 upsample in
// front, one source of 256 | 512 channels or two equal ones.  Batch-independent part (who gets the tile-blocked 16-bit weight copy) and the per-batch choice
// (pixel blocks per tile, or 0).  `levels` (tune key `fullk16`): bit 0 = the 4x4 / 2x2 levels, bit 1 = the stride-2 / upsampling convs that WRITE 8x8
// (igemm + splitk_reduce otherwise), bit 2 = the stride-1 single-source 8x8 layers (bandconv512 otherwise).
inline bool fullk16_layer(int hs, int ho, int c0, int c1, int cout, int stride, bool up, bool up4, int dtype, bool inorm)
{
    if (dtype == 0 || up4 || inorm) return false;
    if (ho != 2 && ho != 4 && ho != 8) return false;
    if (up) { if (stride != 1 || 2 * hs != ho) return false; }
    else if (stride == 2) { if (hs != 2 * ho || c1 != 0) return false; }
    else if (stride != 1 || hs != ho) return false;
    return (c0 == 256 || c0 == 512) && (c1 == 0 || c1 == c0) && cout % 128 == 0;
}
inline int fullk16_choice(int batch, int hs, int ho, int c0, int c1, int cout, int stride, bool up, bool up4, int dtype, int levels = 3, int min_frames = 2)
{
    if (!fullk16_layer(hs, ho, c0, c1, cout, stride, up, up4, dtype, false) || batch < min_frames) return 0;
    const int bit = ho <= 4 ? 1 : (up || stride == 2) ? 2 : 4;
    if (!(levels & bit)) return 0;
    const int nr = 16 / ho > 0 ? 16 / ho : 1;                 // output rows per 16-pixel block (a 4x4 / 2x2 frame is one block)
    const long tiles = (long)batch * ((ho + nr - 1) / nr) * (cout / 16);
    const int S = (!up && stride == 2) ? 2 : 1;
    const int rows = std::min(up ? nr / 2 + 2 : S * (nr - 1) + 3, hs);
    if ((size_t)(c1 ? 2 : 1) * ((size_t)rows * hs + 1) * (c0 * 2 + 16) > 150 * 1024) return 0;
    return tiles <= 1024 ? 1 : 0;                             // up to four rounds of workgroups (8x8 at 8 frames), like the fp32 kernel
}
// Winograd kernel eligibility, batch-independent part (which layers get the G g G^T copy at pack time); the per-batch choice asks
// wino_supported() itself (kernels.h) through wino_choice() in plan.cpp
static const int kWinoMinExtent = 16;     // 16x16 only from 4 frames up (wino_choice): below that the full-K kernel is as fast
inline bool wino_layer(int hs, int ho, int c0, int c1, int cout, int stride, bool up, bool up4, int dtype, bool inorm)
{
    // InstanceNorm plans too (inorm): the kernel then writes the raw conv output (+ bias) and the statistics come from the separate passes
    (void)inorm;
    return dtype == 0 && stride == 1 && !up && !up4 && c1 == 0 && hs == ho && ho >= kWinoMinExtent && ho % 16 == 0 &&
           c0 % 8 == 0 && cout % 32 == 0;
}
// per batch: 32-channel blocks per wave (0 = keep the implicit GEMM) and K splits
int wino_choice(int batch, int ho, int cin, int cout, int *splits);
// the F(4x4, 3x3) kernel (wino4.hip): tile-blocks of 16 x 32 output pixels -> extents that are multiples of 32
inline bool wino4_layer(int hs, int ho, int c0, int c1, int cout, int stride, bool up, bool up4, int dtype, bool inorm)
{
    return wino_layer(hs, ho, c0, c1, cout, stride, up, up4, dtype, inorm) && ho >= 32 && ho % 32 == 0;
}
// per batch: 1 (and the K splits) when the layer runs on it, 0 = keep wino_choice()'s answer
int wino4_choice(int batch, int ho, int cin, int cout, int *splits);
// the up-conv form (winoup.hip): sub-pixel up-convs of fp32 plans whose two sources are equally wide
// (any Upsample + Conv3x3 layer: those below kUp4MinExtent keep their 9-tap rows for the full-K kernel / implicit GEMM and take this kernel from the batch where it wins)
inline bool winoup_layer(int hs, int c0, int c1, int cout, bool up_any, int dtype, bool inorm)
{
    (void)inorm;
    return dtype == 0 && up_any && (c1 == c0 || c1 == 0) && c0 % 8 == 0 && cout % 32 == 0 && hs % 8 == 0;
}
int winoup_choice(int batch, int hs, int cin, int cout, int *splits, int force_nb = 0, int target = 1024, bool small_level = false);
static const int kUp4MinExtent = 32;   // up-convs writing >= 32x32 use the sub-pixel form

}  // namespace lspf2f
