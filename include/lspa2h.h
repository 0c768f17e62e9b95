/*
 * lspa2h.h -- C ABI of the MI355X (gfx950) head-pose generator: the autoregressive WaveNet behind
 * LiveSpeechPortraits' Audio2HeadposeModel.generate_sequences().
 * Exported by the same shared library as lspf2f.h (livespeechportraits_amd/liblspf2f.so).
 *
 * Reference path replaced (file:line under the reference tree):
 *   models/audio2headpose_model.py:133-187   generate_sequences(): per-frame sliding-window loop
 *   models/audio2headpose.py:8-52            Audio2Headpose: audio_downsample MLP + WaveNet
 *   models/networks.py:74-214, 217-326       WaveNet / residual_block (dilated k=2 convs, gated units)
 *   models/losses.py:68-112                  Sample_GMM (softmax -> multinomial -> randn * sigma + mu)
 *
 * The reference re-runs the whole receptive field (255 positions) for every frame; this library evaluates
 * the same function incrementally (per-layer dilation queues), which is exact: the last output of a window
 * whose length equals the receptive field never touches the zero padding.
 *
 * Conventions as lspf2f.h: 0 / negative status, lspa2h_last_error() for the message, no exceptions, no
 * device allocation, no implicit synchronisation; every device pointer is caller-owned fp32.
 * Randomness: the library draws nothing.  The caller passes the N(0,1) draws (and, for ncenter > 1, the
 * Exp(1) draws torch.multinomial consumes) so that a seeded host generator reproduces the reference stream.
 */
#ifndef LSPA2H_H
#define LSPA2H_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* the library is built with -fvisibility=hidden: exactly the functions declared below are exported */
#pragma GCC visibility push(default)

#define LSPA2H_ABI_VERSION 1

#define LSPA2H_OK 0
#define LSPA2H_ERR_INVALID_ARGUMENT (-1)
#define LSPA2H_ERR_UNSUPPORTED (-2)
#define LSPA2H_ERR_STATE (-3)
#define LSPA2H_ERR_HIP (-4)
#define LSPA2H_ERR_SHAPE (-5)

/* Run the whole loop in ONE workgroup that streams the weights every step: the first implementation, kept as an
 * independent second route to the same bits.  Measured on MI355X, default network: 84 us per frame + 20 ms to fill
 * the first receptive field, against 31.5 us + 0.55 ms for the default layer-pipelined kernel. */
#define LSPA2H_FLAG_SINGLE_WORKGROUP 1u
#define LSPA2H_FLAG_CONSECUTIVE_BLOCKS 2u   /* tools: the pipeline's workgroups on consecutive block ids (spread over the XCDs) instead of every 8th */

#define LSPA2H_LOSS_GMM 0 /* opt.loss == 'GMM': output (2*ndim+1)*ncenter, sampled */
#define LSPA2H_LOSS_L2 1  /* opt.loss == 'L2' : output ndim, used as is (audio2headpose_model.py:182-183) */

/* Field names follow options/base_options_audio2headpose.py:60-80. */
typedef struct lspa2h_config {
    int32_t abi_version;       /* LSPA2H_ABI_VERSION */
    int32_t residual_layers;   /* A2H_wavenet_residual_layers (7): dilations 1,2,..,2^(layers-1) per block */
    int32_t residual_blocks;   /* A2H_wavenet_residual_blocks (2) */
    int32_t residual_channels; /* A2H_wavenet_residual_channels: must be 128 */
    int32_t dilation_channels; /* A2H_wavenet_dilation_channels: must be 128 */
    int32_t skip_channels;     /* A2H_wavenet_skip_channels: must be 256 */
    int32_t kernel_size;       /* A2H_wavenet_kernel_size: must be 2 */
    int32_t input_channels;    /* A2H_wavenet_input_channels (12); must equal ndim (the samples are fed back) */
    int32_t cond_channels;     /* A2H_wavenet_cond_channels (512); must equal hidden_size */
    int32_t hidden_size;       /* APC_hidden_size (512); audio feature rows have 2*hidden_size values */
    int32_t ncenter;           /* A2H_GMM_ncenter (1), <= 8 */
    int32_t ndim;              /* A2H_GMM_ndim (12), <= 16 */
    int32_t loss;              /* LSPA2H_LOSS_* */
    int32_t max_audio_frames;  /* largest n_audio lspa2h_generate will see (sizes the workspace) */
    uint32_t flags;            /* LSPA2H_FLAG_* */
} lspa2h_config;

typedef struct lspa2h_handle lspa2h_handle;

/* Audio2Headpose.__init__ (models/audio2headpose.py:8-37) */
int lspa2h_create(const lspa2h_config *cfg, lspa2h_handle **out);
int lspa2h_destroy(lspa2h_handle *h);
const char *lspa2h_last_error(void);
int lspa2h_abi_version(void);
/* WaveNet.receptive_field (models/networks.py:166): 1 + blocks * (2^layers - 1) */
int lspa2h_receptive_field(const lspa2h_handle *h);

/* Weight ingress -- net.load_state_dict (models/base_model.py:212-219).  Keys are the reference's own
 * ("audio_downsample.0.weight", "WaveNet.residual_blocks.3.filter_conv.weight", ...), host fp32 data. */
int lspa2h_num_tensors(const lspa2h_handle *h);
int lspa2h_tensor_info(const lspa2h_handle *h, int index, const char **key, size_t *numel);
int lspa2h_set_tensor(lspa2h_handle *h, const char *key, const float *host_data, size_t numel);
size_t lspa2h_packed_bytes(const lspa2h_handle *h);
int lspa2h_pack_weights(lspa2h_handle *h, void *host_dst, size_t bytes); /* eval-BN folded, kernel layouts */
int lspa2h_bind_weights(lspa2h_handle *h, const void *packed_dev, size_t bytes);
size_t lspa2h_workspace_bytes(const lspa2h_handle *h);
int lspa2h_bind_workspace(lspa2h_handle *h, void *workspace_dev, size_t bytes);

/* Audio2HeadposeModel.generate_sequences(audio_feats, pre_headpose, fill_zero=True, sigma_scale, opt)
 * (models/audio2headpose_model.py:133-187):
 *   audio_dev    [n_audio][2*hidden_size]  APC features (the reference's audio_feats.reshape(-1, 512*2))
 *   pre_dev      [ndim]                    initial head pose, repeated over the first receptive field
 *   noise_dev    [nframe][ndim]            the torch.randn(1, ndim) draw of every frame, or NULL (treated as 0)
 *   expq_dev     [nframe][ncenter]         the Exp(1) draws of torch.multinomial; required iff ncenter > 1 (GMM)
 *   out_dev      [nframe][ndim]            pred_headpose
 * nframe must equal n_audio - frame_future.  Asynchronous on `stream` (hipStream_t). */
int lspa2h_generate(lspa2h_handle *h, const float *audio_dev, int n_audio, const float *pre_dev, const float *noise_dev,
                    const float *expq_dev, float sigma_scale, int frame_future, float *out_dev, int nframe, void *stream);

/* Waits for `stream` and reports whether the last lspa2h_generate completed: *code == 0, or the identifier of the
 * first inter-workgroup hand-off that timed out (the kernel never spins unbounded; output rows are then undefined). */
int lspa2h_status(lspa2h_handle *h, void *stream, uint32_t *code);

/* Sample_GMM (models/losses.py:68-112) for `rows` independent rows at once -- what the LSTM decoder branch of
 * generate_sequences does in one call (audio2headpose_model.py:196-199).  Stateless.
 *   params_dev [rows][(2*ndim+1)*ncenter]  network output: ncenter logits, ncenter*ndim means, ncenter*ndim -log(sigma)
 *   noise_dev  [rows][ndim]     torch.randn(rows, ndim), or NULL (= 0)
 *   expq_dev   [rows][ncenter]  the Exp(1) draws of torch.multinomial(prob, 1); required iff ncenter > 1
 *   out_dev    [rows][ndim]     noise * exp(-nls) * sigma_scale + mu of the selected centre */
int lspa2h_sample_gmm(const float *params_dev, int rows, int ncenter, int ndim, const float *noise_dev,
                      const float *expq_dev, float sigma_scale, float *out_dev, void *stream);

/* Test / profiling hooks (no reference counterpart). */
/* down_audio_feats of Audio2Headpose.forward (audio2headpose.py:47): [n_audio][hidden_size], valid after generate */
int lspa2h_debug_cond(const lspa2h_handle *h, const float **cond_dev, int *rows, int *cols);
/* Wall-clock split of the last lspa2h_generate_timed call, measured with events on `stream` (synchronises). */
int lspa2h_generate_timed(lspa2h_handle *h, const float *audio_dev, int n_audio, const float *pre_dev, const float *noise_dev,
                          const float *expq_dev, float sigma_scale, int frame_future, float *out_dev, int nframe, void *stream,
                          float *precompute_ms, float *loop_ms);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* LSPA2H_H */
