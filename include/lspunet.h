/*
 * lspunet.h -- C ABI of the `size == 'small'` feature2face generator on the MI355X (gfx950): the pix2pix U-Net
 * Feature2FaceGenerator_Unet behind Feature2FaceModel.inference() when opt.size == 'small' (SURVEY.md 8a row a13).
 * Exported by the same shared library as lspf2f.h (livespeechportraits_amd/liblspf2f.so).
 *
 * Reference path replaced (file:line under the reference tree):
 *   models/feature2face_G.py:16-17       Feature2Face_G picks Feature2FaceGenerator_Unet(input_nc=23, output_nc=3, ...)
 *   models/networks.py:680-697           Feature2FaceGenerator_Unet: the nest of UnetSkipConnectionBlocks
 *   models/networks.py:702-769           UnetSkipConnectionBlock: Conv2d(k4, s2, p1) / ConvTranspose2d(k4, s2, p1), BatchNorm2d (eval),
 *                                        in-place LeakyReLU(0.2) / ReLU, torch.cat skip, Tanh
 *   models/feature2face_model.py:225-237 inference(): torch.cat([feature_map, cand_image], 1) -> netG
 *
 * What the handle plans: per level ONE launch for the down-conv -- a 3x3 implicit GEMM over the space-to-depth image that walks only
 * the 16 live (tap, quarter) K blocks of the 36 and writes, from its epilogue, the two tensors the reference's in-place activations make of its output
 * (leaky_relu(d) in space-to-depth form for the next down-conv, relu(d) for the skip) -- ONE for the transposed conv in sub-pixel form (4 parities x 2x2
 * taps, the skip concatenation as a second base pointer), one space-to-depth pass for the two input tensors (the torch.cat is never materialised) and the
 * pixel shuffle + tanh (+ util.tensor2im) behind the last transposed conv; the whole forward is replayed from a hipGraph per (pointers, batch).
 *
 * Conventions as lspf2f.h: 0 / negative status (the LSPF2F_* values), lspunet_last_error() for the message, nothing throws, the library allocates no
 * device memory (weights arena and workspace are the caller's, sizes are queried), lspunet_forward() enqueues on the given hipStream_t and returns.
 * fp32, or fp16 storage for the reference's opt.fp16 (lspunet_config.dtype).  There is no CPU path.
 */
#ifndef LSPUNET_H
#define LSPUNET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* the library is built with -fvisibility=hidden: exactly the functions declared below are exported */
#pragma GCC visibility push(default)

#define LSPUNET_ABI_VERSION 1

#define LSPUNET_OK 0
#define LSPUNET_ERR_INVALID_ARGUMENT (-1)
#define LSPUNET_ERR_UNSUPPORTED (-2)
#define LSPUNET_ERR_MISSING_TENSOR (-3) /* a state-dict key the network needs was never supplied (the reference's strict=False is silent) */
#define LSPUNET_ERR_SHAPE (-4)
#define LSPUNET_ERR_STATE (-5)
#define LSPUNET_ERR_HIP (-6)

#define LSPUNET_FLAG_NO_GRAPH 1u /* launch eagerly instead of replaying a cached hipGraph */

/* Constructor arguments of Feature2FaceGenerator_Unet (models/networks.py:681) + options/base_options_feature2face.py:40, 49-50. */
typedef struct lspunet_config {
    int32_t abi_version; /* LSPUNET_ABI_VERSION */
    int32_t input_nc;    /* 23 (feature2face_G.py:17) */
    int32_t feat_nc;     /* how many of them come from feature_map; the rest from cand_image.  feat_nc == input_nc: one already concatenated tensor */
    int32_t output_nc;   /* 3 (1..4) */
    int32_t ngf;         /* 64 (a multiple of 32) */
    int32_t num_downs;   /* 8 (>= 5) */
    int32_t size;        /* 512: square frames, a multiple of 2^num_downs */
    int32_t max_batch;
    int32_t dtype;       /* 0: fp32 (the parity configuration); 2: fp16 storage of activations and conv weights, fp32 accumulate and epilogue, API tensors fp32 -- the
                            reference's opt.fp16, torch.cuda.amp.autocast around netG (models/feature2face_G.py:28-30); values as lspf2f_dtype (bf16 is not offered here) */
    uint32_t flags;      /* LSPUNET_FLAG_* */
} lspunet_config;

typedef struct lspunet_handle lspunet_handle;

/* Feature2FaceGenerator_Unet.__init__ (models/networks.py:681-692).  Builds the static plan; touches no device.
 * `tune` = "key=value,..." (integers; NULL / "" = none; an unknown key is an error) -- the A-B switches of tests and measurements:
 *   graph (1) | fused_splitk (0; 1: 2..8 K splits combined inside the launch by the last-arriving workgroup instead of a reduce launch) | last_tile (0 = by rule; bm * 1000 + bn forces the tile of the
 *   last GEMM, e.g. 128032; -1 = the general tiling rule) | last_direct (1: the outermost transposed conv + tanh + tensor2im on the direct sub-pixel kernel of the other variants'
 *   last layer; 0: a 3x3 GEMM with N = 4 x output_nc + a pixel-shuffle pass) | tiny (1: the <= 16-position levels on the weight-streaming kernel unet_tiny) | dense0 (0: block 0's space-to-depth quarters are padded
 *   to a K-tile and only its 16 live (tap, quarter) pairs walked; 1: the dense 3x3 rows on 4 x 23 -> 96 channels, the host-sequenced form's block 0) | fused_prepare (1: the
 *   down-convs write leaky_relu / relu copies themselves; 0: a separate elementwise launch per level, the round-3 form) | input_pass (1: the two-source input kernel;
 *   0: lspf2f_unet_prepare on a concatenated tensor, feat_nc == input_nc only).  With fp16 storage only `graph` has an effect (the other arms are fp32 launches). */
int lspunet_create(const lspunet_config *cfg, const char *tune, lspunet_handle **out);
int lspunet_destroy(lspunet_handle *h);
const char *lspunet_last_error(void);
int lspunet_abi_version(void);

/* Weight ingress -- net.load_state_dict (models/base_model.py:212-219).  Keys are the reference's own, relative to netG
 * ("model.model.0.weight", "model.model.1.model.2.running_var", ...; no num_batches_tracked), host fp32, contiguous. */
int lspunet_num_tensors(const lspunet_handle *h);
int lspunet_tensor_info(const lspunet_handle *h, int index, const char **key, int64_t dims[4], int *ndim);
int lspunet_set_tensor(lspunet_handle *h, const char *key, const float *host_data, size_t numel);
/* Host-side pack: folds eval-mode BatchNorm (eps 1e-5) into per-channel scale / shift in double, Conv2d(k4, s2, p1) weights -> [co][16 live (tap, quarter)
 * pairs][ci] (block 0, 23 input channels: the dense space-to-depth rows), ConvTranspose2d weights -> [4 parities][co][2][2][ci], the outermost one -> the
 * 3x3 GEMM rows [4 * output_nc][3][3][ci].  The blob is what rank 0 would broadcast. */
size_t lspunet_packed_bytes(const lspunet_handle *h);
int lspunet_pack_weights(lspunet_handle *h, void *host_blob, size_t bytes);
int lspunet_bind_weights(lspunet_handle *h, const void *dev_blob, size_t bytes);       /* 256-byte aligned, caller-owned */
/* bytes a workspace needs for any batch of 1 .. `batch` frames */
size_t lspunet_workspace_bytes(const lspunet_handle *h, int batch);
int lspunet_bind_workspace(lspunet_handle *h, void *dev_workspace, size_t bytes);     /* 256-byte aligned, caller-owned */

/* Feature2FaceModel.inference(feature_map, cand_image) (models/feature2face_model.py:225-237) -> Feature2FaceGenerator_Unet.forward (networks.py:694-697):
 *   feat_dev [batch][feat_nc][S][S] fp32 NCHW; cand_dev [cand_batch][input_nc - feat_nc][S][S] fp32 NCHW, cand_batch 1 (broadcast, as demo.py:266 passes
 *   it) or == batch; NULL iff feat_nc == input_nc.
 *   out_dev  [batch][output_nc][S][S] fp32 NCHW in [-1, 1] (tanh), or NULL;
 *   out_u8_dev [batch][S][S][output_nc] uint8 = util.tensor2im (util/util.py:19-42), or NULL -- at least one of the two. */
int lspunet_forward(lspunet_handle *h, const float *feat_dev, const float *cand_dev, int cand_batch, float *out_dev, unsigned char *out_u8_dev,
                    int batch, void *hip_stream);

/* The same forward launched eagerly with a hipEvent pair around every launch of the plan; ms_per_launch[lspunet_num_launches(h, batch)] receives the
 * durations.  Synchronises the stream (profiling aid, not the hot path). */
int lspunet_forward_timed(lspunet_handle *h, const float *feat_dev, const float *cand_dev, int cand_batch, float *out_dev, unsigned char *out_u8_dev,
                          int batch, void *hip_stream, float *ms_per_launch);

/* Introspection (tests, bench): launches of one forward at this batch; name / kernel of launch i ("L3.down", "igemm3x3<km>+splitk_reduce", ...). */
int lspunet_num_launches(lspunet_handle *h, int batch);
int lspunet_launch_info(lspunet_handle *h, int batch, int index, const char **name, const char **kernel, int *tile_m, int *tile_n, int *split_k);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* LSPUNET_H */
