/*
 * lspf2f.h -- C ABI of the MI355X-native feature2face renderer (liblspf2f.so).
 *
 * The reference (YuanxunLu/LiveSpeechPortraits) has no FFI: the hot path sits behind plain
 * Python classes.  Each entry point below names the reference interface it stands in for
 * (file:line under the reference tree).  INTEGRATION.md shows the ctypes stub a maintainer
 * of the reference would add.
 *
 * Conventions
 *   - every function returns LSPF2F_OK (0) or a negative lspf2f_status; nothing throws across
 *     the ABI; lspf2f_last_error() returns a thread-local message for the last failure.
 *   - plain pointers and sizes only.  "dev" pointers are HIP device pointers (e.g.
 *     torch.Tensor.data_ptr() of a ROCm tensor); "host" pointers are ordinary memory.
 *   - the library never allocates device memory: the caller provides the packed-weight arena
 *     and the workspace (sizes are queried), so PyTorch's allocator stays the only owner of HBM.
 *   - lspf2f_forward() enqueues kernels on the given hipStream_t and returns; it never
 *     synchronises the device and is graph-capturable.
 *   - a handle is not thread-safe; distinct handles are independent.
 *   - there is no CPU fallback anywhere behind this ABI.
 */
#ifndef LSPF2F_H
#define LSPF2F_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* the library is built with -fvisibility=hidden: exactly the functions declared below are exported */
#pragma GCC visibility push(default)

#define LSPF2F_ABI_VERSION 1

typedef enum lspf2f_status {
    LSPF2F_OK = 0,
    LSPF2F_ERR_INVALID_ARGUMENT = -1,
    LSPF2F_ERR_UNSUPPORTED = -2,     /* e.g. ngf % 32 != 0 (size == 'small', networks.py:680-769, has its own handle: lspunet.h) */
    LSPF2F_ERR_MISSING_TENSOR = -3,  /* a state-dict key the network needs was never supplied */
    LSPF2F_ERR_SHAPE = -4,
    LSPF2F_ERR_STATE = -5,           /* call order violated (e.g. forward before bind) */
    LSPF2F_ERR_HIP = -6,             /* a HIP runtime call failed */
    LSPF2F_ERR_NO_DEVICE = -7
} lspf2f_status;

/* Generator variant = opt.size of the reference (models/feature2face_G.py:16-21,
 * config/May.yaml:23 'large', config/Obama1.yaml:23 'normal'). */
typedef enum lspf2f_variant {
    LSPF2F_VARIANT_NORMAL = 0,  /* Feature2FaceGenerator_normal, networks.py:458-483: 1 ResidualBlock/side */
    LSPF2F_VARIANT_LARGE = 1    /* Feature2FaceGenerator_large,  networks.py:554-579: 2 ResidualBlocks/side */
} lspf2f_variant;

typedef enum lspf2f_dtype {
    LSPF2F_DTYPE_F32 = 0,       /* fp32 storage, fp32 MFMA (v_mfma_f32_32x32x2_f32): the parity configuration */
    LSPF2F_DTYPE_BF16 = 1,      /* bf16 activations + conv weights in HBM, v_mfma_f32_32x32x16_bf16, fp32 accumulate and
                                   epilogue; API tensors stay fp32.  The reference has no bf16 path (only fp16 autocast,
                                   feature2face_G.py:28-30): compared against the fp32 oracle with a declared tolerance */
    LSPF2F_DTYPE_F16 = 2        /* fp16 (IEEE binary16) activations + conv weights in HBM, v_mfma_f32_32x32x16_f16, fp32 accumulate
                                   and epilogue; API tensors stay fp32.  Replaces the reference's opt.fp16 branch -- torch.cuda.amp.autocast
                                   around netG (models/feature2face_G.py:28-30, feature2face_model.py:232-236) -- and is pinned on it:
                                   the oracle under torch.autocast(float16) is the comparison (tests/test_gpu_plans.py) */
} lspf2f_dtype;

/* flags for lspf2f_config.flags */
#define LSPF2F_FLAG_KEEP_INTERMEDIATES 1u  /* give every layer output its own workspace region
                                              (debug / per-layer parity tests); default is
                                              liveness-based reuse */
#define LSPF2F_FLAG_NO_GRAPH 2u            /* launch every kernel eagerly instead of replaying a
                                              cached hipGraph (also: env LSP_HIP_GRAPH=0) */
#define LSPF2F_FLAG_INSTANCE_NORM 4u       /* the norm_layer=nn.InstanceNorm2d variant of the generators (constructor argument,
                                              models/networks.py:459 / :555): every norm is InstanceNorm2d(affine=False,
                                              eps=1e-5) computed at run time per (frame, channel); the level convs carry a
                                              bias (use_bias, :494 / :590) and the state dict holds `<conv>.bias` instead of
                                              the BatchNorm tensors.  fp32 only. */

#define LSPF2F_FLAG_WINO4 8u               /* fp32 plans: the stride-1 ResidualBlock convs of the >= 32x32 levels (models/networks.py:650-675) run as
                                              Winograd F(4x4,3x3) (csrc/wino4.hip) where the planner finds the shape eligible, instead of
                                              F(2x2,3x3).  Parity: 2.0e-6 on the `large` golden (F(2x2): 1.7e-6).  Off by default: measured
                                              slower at batch 1 and equal at batch 8.  The packed blob then carries the
                                              6x6 transformed weights of those layers (4x their 9-tap bytes). */

/* Mirrors the option fields the reference reads on this path
 * (options/base_options_feature2face.py:49-50 ngf / n_downsample_G, :40 loadSize; the
 * constructor arguments of feature2face_G.py:19-21). */
typedef struct lspf2f_config {
    int32_t abi_version;   /* LSPF2F_ABI_VERSION */
    int32_t variant;       /* lspf2f_variant */
    int32_t input_nc;      /* 13 = 1 feature-map channel + 12 candidate channels */
    int32_t feat_nc;       /* channels that come from feature_map (1); the rest from cand_image */
    int32_t output_nc;     /* 3 */
    int32_t ngf;           /* 64 */
    int32_t num_downs;     /* 8 */
    int32_t height;        /* 512 (square frames: height == width) */
    int32_t width;
    int32_t max_batch;     /* largest batch lspf2f_forward() will be given */
    int32_t dtype;         /* lspf2f_dtype */
    uint32_t flags;
} lspf2f_config;

typedef struct lspf2f_handle lspf2f_handle;

/* ---- lifecycle ------------------------------------------------------------------------- */

/* Replaces: Feature2Face_G.__init__ (models/feature2face_G.py:9-24) + the module construction
 * of networks.py:554-572 / 458-476.  Builds the static execution plan; touches no device. */
int lspf2f_create(const lspf2f_config *cfg, lspf2f_handle **out);
/* The same with the A-B switches of tools, tests and measurements: `tune` = "key=value,key=value" (integers; NULL or "" = none; an unknown
 * key is LSPF2F_ERR_INVALID_ARGUMENT).  The library never reads the process environment -- every switch arrives here, once per handle,
 * before the plan is built.  Keys (default): graph (1) | wino (1), wino4 (flag), winoup (1): the Winograd kernels | wino_ureg (1:
 * U fragments of wino3x3<1> in registers; 2: four register sets), in_wino_stats (1: InstanceNorm statistics from the wino3x3 epilogue), in_small_regs (1: the one-launch InstanceNorm pass keeps its rows in registers -- one read of the slab instead of three), in_smallm_fused (1: the tiny-M kernel normalises in its own epilogue under InstanceNorm plans), wino_prio (wave priority by K-loop progress: 1 = the workgroup that is behind leads, 2 = the one ahead, 3 = 1 with the older half of the grid kept at level 1 through its last quarter -- on wino3x3<1>'s register form; 4..6 = the same three schemes on every Winograd loop; 0 = off; default 1), wino_pre (1), wino_il (1), wino_rot (1), wino_xcd (-1), igemm_xcd (-1), winoup_nb (0), winoup_target (1024): their tiling / issue-order variants |
 * bandconv (1), bandconv_min_blocks (128), bandconv_min_frames, patch16 (1: the stride-1 convs of the 64x64 / 32x32 levels on the patch-staged kernel), patch16_deep (1: its 64-channel tiles run the deep-ring form conv3x3_patch16d), patchup16 (1: and the sub-pixel up-convs over 32x32 / 64x64 sources on its up-conv form), patch16_min_blocks (192), rowup (1), rowlast (1), rowlast_fused (1: rowlast128 shuffles + applies tanh in its epilogue when only fp32 frames are wanted), rowconv (1): kernels of the 16-bit plans |
 * fullk_split (1), fullk_split_tiles (128), fullk_s2 (0): the full-K kernel's K split | fused_splitk (1), fused_splitk16 (0: the 16-bit plans combine 2..8 K-splits in the launch too), out_wt (1: the Winograd kernels write their output through to memory, sc1 stores; 0: plain stores, left dirty in L2), prefetch (1), smallm_dma (1: the tiny-M kernel stages its input tensor by LDS-DMA, every piece in flight at once; 0: through registers), smallm_kb (128: largest input tensor, in KB, the tiny-M kernel takes; 64 = rounds 2-4) |
 * tail_prefetch (0; 1 | 2: a side branch of the forward's graph reads the weights of the <= 16x16 levels with plain | non-temporal loads while the levels above compute), tail_prefetch_at (-1: layer index the branch forks in front of), tail_prefetch_wgs (32), tail_prefetch_mb (0 = the whole range): measured, off (profiles/r06_tail_prefetch_ab.txt) |
 * fullk16 (3: which small levels of a 16-bit plan run on conv3x3_fullk16 -- bit 0 the 4x4 / 2x2 levels, bit 1 the stride-2 / upsampling convs that write 8x8, bit 2 the stride-1 8x8 layers; 0 = none), fullk16_min_frames (2) |
 * in_small_max_hw (1024: InstanceNorm plans take the one-launch statistics route up to this many pixels per frame), all_forms (0; 1: the packed blob carries every weight form whatever the
 * handle's batch range -- tests that look at forms other batches would use), blob_pad_kb (0: empty KB in front of the first layer's weights, placement experiments) |
 * lastconv (0 = by shape; 1..5 force a last-conv kernel), lastconv_direct (0), firstconv (0 = by shape; 1, 2 force a first-conv kernel). */
int lspf2f_create_tuned(const lspf2f_config *cfg, const char *tune, lspf2f_handle **out);
int lspf2f_destroy(lspf2f_handle *h);
const char *lspf2f_last_error(void);
int lspf2f_abi_version(void);

/* ---- weights: state dict in, packed arena out -------------------------------------------- */

/* Replaces: net.load_state_dict(state_dict, strict=False) (models/base_model.py:219).
 * The expected tensors are exactly the reference's state-dict entries (minus
 * num_batches_tracked), keys WITHOUT the DataParallel 'module.' prefix,
 * e.g. "netG.model.model.0.weight". */
int lspf2f_num_tensors(const lspf2f_handle *h);
/* name/shape of expected tensor i; dims[] receives up to 4 extents, *ndim their count */
int lspf2f_tensor_info(const lspf2f_handle *h, int i, const char **name, int64_t dims[4], int *ndim);
/* Supply one tensor (host pointer, fp32, contiguous, OIHW for conv weights).  The data is
 * copied.  Unknown keys are an error (the reference's strict=False would silently ignore). */
int lspf2f_set_tensor(lspf2f_handle *h, const char *key, const float *host, size_t numel);

/* Size of the packed arena, and the host-side pack: checks every expected tensor was supplied
 * (LSPF2F_ERR_MISSING_TENSOR otherwise -- the reference is silent here), folds eval-mode
 * BatchNorm (networks.py:606-607, 664, 667; eps 1e-5) into per-channel scale/shift and
 * reorders conv weights OIHW -> [Cout][ky][kx][Cin].  Pure CPU; the blob is what rank 0
 * broadcasts over RCCL to the other GPUs. */
size_t lspf2f_packed_bytes(const lspf2f_handle *h);
int lspf2f_pack_weights(lspf2f_handle *h, void *host_blob, size_t bytes);
/* Attach a device copy of the packed blob (caller-owned, must outlive the handle's use). */
int lspf2f_bind_weights(lspf2f_handle *h, const void *dev_blob, size_t bytes);

/* ---- workspace --------------------------------------------------------------------------- */

/* Bytes a workspace must have to run ANY batch of 1 .. `batch` frames (the per-batch plans differ in kernels and split-K scratch, so the need of a
 * smaller batch can exceed that of a larger one; the returned value covers them all). */
size_t lspf2f_workspace_bytes(const lspf2f_handle *h, int batch);
int lspf2f_bind_workspace(lspf2f_handle *h, void *dev_workspace, size_t bytes);

/* ---- the hot path ------------------------------------------------------------------------ */

/* Replaces: Feature2FaceModel.inference(feature_map, cand_image)
 * (models/feature2face_model.py:225-237) -> Feature2Face_G.forward (feature2face_G.py:27-34)
 * -> Feature2FaceGenerator_{large,normal}.forward (networks.py:575-579 / 479-483).
 *   feat_dev  [batch][feat_nc][H][W]            fp32 NCHW
 *   cand_dev  [cand_batch][input_nc-feat_nc][H][W]  fp32 NCHW; cand_batch is 1 (broadcast to
 *             every frame, as demo.py:266 passes it) or == batch; NULL iff feat_nc == input_nc
 *             (the reference's `cand_image == None` branch)
 *   out_dev   [batch][output_nc][H][W]          fp32 NCHW, values in [-1, 1] (tanh)
 * The torch.cat of the reference is never materialised. */
int lspf2f_forward(lspf2f_handle *h, const float *feat_dev, const float *cand_dev, int cand_batch,
                   float *out_dev, int batch, void *hip_stream);

/* Optional: pre-compute the candidate stack's contribution to the first convolution.
 * demo.py:89-95 builds img_candidates once per person and passes the same tensor to every
 * inference() call (demo.py:266); 12 of the first layer's 13 input channels are therefore constant.
 * After this call, lspf2f_forward*() may be given cand_dev == NULL: the first layer then only
 * convolves the feature-map channel and adds the cached partial sums (kept in the first
 * (H/2)*(W/2)*ngf*4 bytes of the workspace, so re-binding the workspace or the weights drops it).
 * cand_dev: [1][input_nc-feat_nc][H][W]; NULL clears the cache.  Enqueued on hip_stream. */
int lspf2f_set_candidates(lspf2f_handle *h, const float *cand_dev, void *hip_stream);

/* Same forward with the reference's frame post-processing fused into the last kernel.
 * Replaces: util.tensor2im(pred_fake[0]) (util/util.py:19-42, called at demo.py:268), i.e.
 * (x + 1) / 2 * 255 -> clip [0,255] -> uint8, CHW -> HWC -- computed on the device, so the per-frame
 * D2H copy shrinks from 3 MiB fp32 to 0.75 MiB and the numpy pass disappears.
 *   out_u8_dev [batch][H][W][output_nc] uint8 (may be NULL); out_dev as in lspf2f_forward (may be
 *   NULL when out_u8_dev is given). */
int lspf2f_forward_ex(lspf2f_handle *h, const float *feat_dev, const float *cand_dev, int cand_batch,
                      float *out_dev, unsigned char *out_u8_dev, int batch, void *hip_stream);

/* ---- introspection (tests, bench, profiling) ---------------------------------------------- */

typedef struct lspf2f_layer_info {
    const char *name;        /* "L0.down", "L3.d.res1.b", "L0.up", ... */
    const char *kernel;      /* which kernel family executes it */
    int32_t cin, cout, h_in, h_out, stride;
    int32_t upsample, concat, residual, relu, tanh_out;
    int32_t tile_m, tile_n, split_k, k_group;
    int64_t flops_per_frame;          /* algorithmic: 2*Cout*Cin*9*Hout*Wout */
    int64_t exec_flops_per_frame;     /* what the kernel issues (4/9 of it for sub-pixel up-convs) */
    int64_t act_bytes_per_frame;      /* algorithmic activation bytes */
    int64_t weight_bytes;
    int64_t w_offset;                 /* byte offsets into the packed blob (-1: none) */
    int64_t scale_offset;
    int64_t shift_offset;
    int64_t out_offset;               /* byte offset of the output in the workspace for the
                                         batch last bound via lspf2f_plan_batch(); NHWC */
} lspf2f_layer_info;

int lspf2f_num_layers(const lspf2f_handle *h);
/* (re)plans buffers for `batch` frames; forward() does this implicitly */
int lspf2f_plan_batch(lspf2f_handle *h, int batch);
int lspf2f_layer_info_get(const lspf2f_handle *h, int i, lspf2f_layer_info *out);
/* forward with a hipEvent pair around every layer; ms_per_layer[num_layers] receives the
 * durations.  Synchronises the stream (profiling aid, not the hot path). */
int lspf2f_forward_timed(lspf2f_handle *h, const float *feat_dev, const float *cand_dev,
                         int cand_batch, float *out_dev, int batch, void *hip_stream,
                         float *ms_per_layer);
/* Duration of a SUBSET of the forward's launches without host gaps (profiling aid): the selected layers are captured in network
 * order into one graph, replayed `reps` times between two events on hip_stream; *ms_per_replay = elapsed / reps.  part[num_layers]:
 * 0 skip, 1 the layer's main kernel(s) only (a split-K layer without its reduce launch), 2 only its split-K reduce launch, 3 all of
 * it.  Run a forward first (the layers read what it left in the workspace).  This is how bench.py times one kernel class inside
 * otherwise unchanged launches: ROCm 7.2 cannot time events recorded by graph nodes, so a kernel cannot be bracketed inside the
 * replay of the whole forward.  Synchronises the stream. */
int lspf2f_subset_timed(lspf2f_handle *h, const float *feat_dev, const float *cand_dev, int cand_batch, float *out_dev,
                        int batch, void *hip_stream, const int *part, int reps, float *ms_per_replay,
                        int *launches_per_replay);

/* Single fused 3x3 convolution, the unit the generator is made of, exposed for per-kernel
 * parity tests against torch.nn.functional.conv2d (+ batch_norm eval + relu).
 *   src0/src1 NHWC [batch][hs][ws][c0|c1] (src1 may be NULL, c1 = 0; both are read as
 *   cat([src0, src1], channel)); w_packed [cout][3][3][c0+c1]; scale/shift [cout] or NULL;
 *   residual NHWC [batch][ho][wo][cout] or NULL; out NHWC [batch][ho][wo][cout].
 *   stride in {1,2}; upsample: 0 none; 1 nearest x2 before the conv (9-tap gather form, stride 1);
 *   2 the same op in sub-pixel form: w_packed is [4 parities][cout][2][2][c0+c1] with the
 *   aliasing 3x3 taps pre-summed (see plan.cpp pack()).
 *   dtype: lspf2f_dtype of src0/src1/w_packed/residual/out (scale/shift are always fp32; bf16 needs
 *   c0, c1 % 64 == 0).
 *   tile_m/tile_n/split_k/k_group = 0 selects the planner's choice (k_group = K-tiles fetched
 *   per pipeline step: 1, 2 or 4); scratch is needed when split_k != 1
 *   (size from lspf2f_conv3x3_scratch_bytes).  Special tiles: 1 x 1 = the tiny-M kernel (<= 16 output pixels);
 *   16 x 16 / 32 x 16 = the full-K single-launch kernel of the 16x16 / 8x8 levels (fp32, stride 1, extents 2..16,
 *   c0 in {128, 256, 512}, c1 in {0, c0}, cout % 128 == 0; no scratch).  With k_group == -1 the full-K kernel takes w_packed in
 *   its own tile-blocked layout, [cout/16][source][tap][4 waves][g][64 lanes][4] with lane (li, kq) of block (nt, T, w, g) holding
 *   channels 4 * (kq * 4G + w * G + g) .. + 3 (G = c0 / 64) of output row 16 nt + li -- the copy the packer adds for those layers.
 *   4001 / 4002 (with k_group == -1) = the Winograd F(2x2,3x3) kernel with 1 / 2 blocks of 32 output channels per wave (fp32, one source,
 *   stride 1, hs % 16 == 0, c0 % 8 == 0, cout % 32 == 0 / % 64 == 0): w_packed holds G g G^T in the fragment order
 *   [cout/32][xi-row 4][c0/8][j 4][64 lanes][4] (lane l: output channel 32 nblock + (l & 31), input channels 8 s + 4 (l >> 5) .. + 3, the copy
 *   the packer adds for those layers); split_k = K slices (1..8, 0 = 1) combined inside the launch -- the scratch then holds the slabs
 *   followed by one arrival counter per (tile-block, channel group), which must be ZERO on entry and is left zero.  4003 = 4001 with the wave's U
 *   fragments loaded straight into registers instead of through LDS (the form the plans take; same operands, same results bit for bit); 4004 = 4003 with four
 *   register sets instead of three (operands requested three K-steps ahead; tune key wino_ureg=2).
 *   k_group == -4 (fp32, any tile, stride 1, one source of 4 ci channels, ci a multiple of 32): the 3x3 conv on a SPACE-TO-DEPTH image that equals Conv2d(k4, s2, p1)
 *   (channel (dy * 2 + dx) * ci + c of pixel (y, x) = channel c of pixel (2y + dy, 2x + dx)); only 16 of the 36 (tap, quarter) pairs carry weights, and w_packed holds
 *   just those: [cout][live pairs in tap-major, quarter-minor order][ci] -- 16/36 of the matrix work of the dense form (the `small` U-Net's down-convs).
 *   6001 (with k_group == -1) = the Winograd F(4x4,3x3) kernel (fp32, one source, stride 1, hs % 16 == 0, ws % 32 == 0, c0 % 8 == 0,
 *   cout % 32 == 0): w_packed holds the 6x6 G g G^T in the order [cout/32][wave (a, b) 4][c0/8][f 9][64 lanes][4] -- position
 *   (3a + f / 3, 3b + f % 3), lane as above; split_k and scratch as for 4001 with tile-blocks of 16 x 32 pixels x 32 channels.
 *   7064 / 7032 with tile_n = 128 | 64 = the patch-staged kernel of the 16-bit plans (bf16 | fp16, one source of c0 % 64 == 0 >= 128 channels, stride 1, no upsample):
 *   a workgroup owns 4 rows x 64 / 8 rows x 32 pixels x tile_n channels (hs % 4 == 0, ws % 64 == 0 / hs % 8 == 0, ws % 32 == 0, cout % tile_n == 0); w_packed in the
 *   default layout [cout][3][3][c0]; no scratch.  7164 / 7132 / 7116 (tile_n 64 only) = its sub-pixel up-conv form (upsample == 2; hs, ws the LOW-res extent, multiples of the tile's 4 x 64 / 8 x 32 / 16 x 16 pixels; c1 in {0, c0}; w_packed = [4][cout][2][2][c0 + c1]). */
size_t lspf2f_conv3x3_scratch_bytes(int batch, int hs, int ws, int c0, int c1, int cout, int stride,
                                    int upsample, int tile_m, int tile_n, int split_k, int k_group, int dtype);
int lspf2f_conv3x3(const void *src0, const void *src1, const void *w_packed, const float *scale,
                   const float *shift, const void *residual, void *out, int batch, int hs, int ws,
                   int c0, int c1, int cout, int stride, int upsample, int relu, int tile_m,
                   int tile_n, int split_k, int k_group, int dtype, void *scratch, size_t scratch_bytes,
                   void *hip_stream);

/* Round 6 (tools/probes/wino_pair_probe.py, tests/test_gpu_conv.py): `nlayers` (1..4) consecutive Winograd convs of ONE shape (batch x hs x hs x c -> c, fp32: the convs of one
 * or two ResidualBlocks, models/networks.py:650-675), layer k reading src[k] (= out[k - 1] for k > 0), weights u_packed[k] in the fragment order of tile 4001, optional
 * scale[k] / shift[k] / residual[k] (NULL entries allowed), relu[k].  mode 0: one launch of wino3x3<1> per layer, exactly as the plans run them; mode 1: ONE launch of
 * nlayers x (workgroups of a layer) workgroups, a layer's workgroups gated on the arrival counters of the tile-blocks they read (wino.hip, wino3x3_chain) -- same
 * arithmetic in the same order, bit-identical; mode 2: the chain kernel launched once per layer (what its own prologue costs).  The outputs must be distinct buffers
 * that are no input of an earlier layer.  scratch: lspf2f_wino_chain_scratch_bytes() bytes, ZERO on entry, left zero -- except its LAST 32-bit word, which a launch
 * sets to 1 if a gate gave up waiting (the results are then garbage; it never happens while the dispatcher hands out workgroups in order). */
size_t lspf2f_wino_chain_scratch_bytes(int nlayers, int batch, int hs, int c, int split_k);
int lspf2f_wino_chain(int nlayers, const float *const *src, const float *const *u_packed, const float *const *scale, const float *const *shift,
                      const float *const *residual, float *const *out, const int *relu, int batch, int hs, int c, int split_k, int mode,
                      void *scratch, size_t scratch_bytes, void *hip_stream);

/* ---- building blocks of the `size == 'small'` generator (Feature2FaceGenerator_Unet, models/networks.py:680-769) ----
 * That U-Net is made of 4x4 stride-2 convs and 4x4 stride-2 transposed convs.  Both map onto lspf2f_conv3x3:
 *   Conv2d(k4, s2, p1) on X  ==  a 3x3 stride-1 conv on the space-to-depth image Y[i][j][(dy*2+dx)*C + c] = X[2i+dy][2j+dx][c]
 *                                with W3[co][ty][tx][(dy,dx,c)] = W[co][c][ky][kx], (ty,dy) = {0:(0,1), 1:(1,0), 2:(1,1), 3:(2,0)}[ky]
 *                                (likewise x; the other 20 of 36 (tap, dy, dx) blocks are zero);
 *   ConvTranspose2d(k4, s2, p1)  ==  the sub-pixel form (upsample = 2) with W[par][co][a*2+b][ci] = Wt[ci][co][{{3,1},{2,0}}[py][a]][..[px][b]].
 * lspf2f_unet_prepare is the elementwise pass between them; the reference's in-place LeakyReLU/ReLU mean a stored skip
 * tensor d is read as leaky_relu(d) by the next down-conv and as relu(d) by the up-conv (networks.py:737-741, 751-767). */

/* src: fp32, NHWC [batch][h][w][c] (src_nchw == 0) or NCHW [batch][c][h][w] (src_nchw != 0); h, w even.
 *   s2d_out   NHWC [batch][h/2][w/2][s2d_channels] or NULL: channel (dy*2+dx)*c + k = act(src[2i+dy][2j+dx][k]) with
 *             act(v) = v > 0 ? v : slope * v (slope 1 = identity); channels >= 4c are written as 0 (s2d_channels >= 4c, % 4 == 0)
 *   relu_out  NHWC [batch][h][w][c] or NULL: max(src, 0) */
int lspf2f_unet_prepare(const float *src_dev, int src_nchw, int batch, int h, int w, int c, float slope,
                        float *s2d_out_dev, int s2d_channels, float *relu_out_dev, void *hip_stream);

/* g NHWC [batch][hs][ws][4*cout] fp32 (channel = parity*cout + co, parity = py*2+px) -> optional tanh ->
 *   out_f32 NCHW [batch][cout][2hs][2ws] and/or out_u8 HWC uint8 [batch][2hs][2ws][cout] (util.tensor2im); cout <= 4 */
int lspf2f_pixel_shuffle(const float *g_dev, int batch, int hs, int ws, int cout, int apply_tanh,
                         float *out_f32_dev, unsigned char *out_u8_dev, void *hip_stream);

/* Where a layer's weights sit in the packed blob, per weight FORM (tests; -1: this handle's blob does not carry that form -- it carries only
 * the forms the plans of batch 1 .. max_batch read).  Forms: 0 rows ([cout][tap][cin], sub-pixel rows [4][cout][2][2][cin], first conv
 * [cin][tap][cout]; = lspf2f_layer_info.w_offset), 1 full-K tile-blocked, 2 full-K as two half-sources, 3 Winograd F(2x2,3x3) fragments,
 * 4 Winograd F(4x4,3x3) fragments, 5 up-conv Winograd fragments, 6 rowup256 fragments, 7 bandconv512 fragments, 8 rowconv fragments,
 * 9 GEMM form of the last conv, 10 rowlast128 fragments. */
int64_t lspf2f_layer_form_offset(const lspf2f_handle *h, int layer, int form);

/* Hazard-test aid (tests/test_gpu_hazards.py; SURVEY.md section 5 "race detection"): every byte of the bound workspace a forward may only use as
 * SCRATCH -- the activation arena, the split-K partial slabs, the InstanceNorm statistics, the per-forward candidate slot, and the per-person
 * candidate cache too when none is set -- is overwritten with `byte` (0xFF: every fp32 / bf16 / fp16 word becomes a NaN) on `hip_stream`.  A
 * forward that follows must produce the same bits as before: a kernel that reads a workspace word before this forward's own producer wrote it, or
 * that relies on what the previous forward left behind, turns NaN.  The arrival counters of the in-launch split-K combines are persistent state
 * (each last arriver resets its own; zero between launches is the protocol's invariant), so they are NOT poisoned but READ BACK:
 * *nonzero_counters = how many of them a finished forward left non-zero (0 is the only healthy answer).  Synchronises `hip_stream`.  Graphs stay cached:
 * the next forward replays the same hipGraph on the poisoned workspace. */
int lspf2f_debug_poison(lspf2f_handle *h, unsigned char byte, void *hip_stream, unsigned *nonzero_counters);

/* A copy executed by a KERNEL on `hip_stream` (16-byte aligned pointers and size).  Replaces, in the frame loop, the transfers either side of the generator -- the feature map going up
 * (`feature_map.to(device)`, demo.py:262-265) and the frame coming down (`util.tensor2im(...)` reads it on the host, demo.py:268) -- when the host side is PINNED memory: a pinned host
 * pointer is an ordinary address to a kernel, and a kernel is the next packet of the stream's own queue, whereas hipMemcpyAsync between two launches hands the stream to the copy engine
 * and back (two cross-queue waits the runtime resolves from a host thread: milliseconds on a busy host).  Either pointer may be device or pinned host memory. */
int lspf2f_memcpy(void *dst, const void *src, size_t bytes, void *hip_stream);

/* Measurement aid (bench.py `roofline.clock_ghz_observed`): ONE wave spins for about `duration_us` microseconds of the constant 100 MHz
 * counter (s_memrealtime) and writes {shader cycles (s_memtime), 100-MHz ticks} it saw to out_dev[0..1].  Launched on a side stream while
 * the timed region runs, cycles / ticks x 0.1 is the shader clock in GHz the chip held under that load.  duration_us <= 2 000 000. */
int lspf2f_clock_probe(unsigned long long *out_dev, unsigned duration_us, void *hip_stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* LSPF2F_H */
