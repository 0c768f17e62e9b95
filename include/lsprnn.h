/*
 * lsprnn.h -- C ABI of the MI355X (gfx950) recurrent stacks of the audio front-end:
 * unidirectional multi-layer GRU / LSTM inference over one sequence, plus the dense layers around them.
 * Exported by livespeechportraits_amd/liblspf2f.so.
 *
 * Reference modules replaced (file:line under the reference tree):
 *   models/networks.py:19-66      APC_encoder: ModuleList of 3 nn.GRU(…, hidden 512), batch 1 (demo.py:186-191)
 *   models/audio2feature.py:41-46 nn.LSTM(input 512, hidden 256, num_layers 3) of Audio2Feature
 *   models/audio2feature.py:35-40, 47-54, 66-69   the Linear/BatchNorm1d/LeakyReLU stacks before and after it
 * The arithmetic of nn.GRU / nn.LSTM lives in PyTorch (torch==1.7.1 pinned by cog.yaml:9; 2.10 in this image):
 *   GRU : r = s(W_ir x + b_ir + W_hr h + b_hr), z likewise, n = tanh(W_in x + b_in + r * (W_hn h + b_hn)),
 *         h' = (1 - z) * n + z * h
 *   LSTM: i, f, g, o = W_i x + b_i + W_h h + b_h; c' = s(f) c + s(i) tanh(g); h' = s(o) tanh(c')
 * Zero initial state, batch 1, fp32.  Conventions as lspf2f.h (0 / negative status, lsprnn_last_error(), no device
 * allocation, asynchronous on `stream`).
 */
#ifndef LSPRNN_H
#define LSPRNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* the library is built with -fvisibility=hidden: exactly the functions declared below are exported */
#pragma GCC visibility push(default)

#define LSPRNN_ABI_VERSION 1

#define LSPRNN_OK 0
#define LSPRNN_ERR_INVALID_ARGUMENT (-1)
#define LSPRNN_ERR_UNSUPPORTED (-2)
#define LSPRNN_ERR_STATE (-3)
#define LSPRNN_ERR_HIP (-4)
#define LSPRNN_ERR_SHAPE (-5)

#define LSPRNN_CELL_GRU 0
#define LSPRNN_CELL_LSTM 1

/* One launch per layer instead of the wavefront kernel (all layers in one launch).  The library takes this route by itself when the
 * wavefront's L * Gw mutually polling workgroups cannot all be resident on the device (a partitioned or very small device); the per-layer
 * kernels still need their own G = hidden_size / P (16..64) workgroups resident at once and fail with LSPRNN_ERR_UNSUPPORTED below that. */
#define LSPRNN_FLAG_PER_LAYER 1u

typedef struct lsprnn_config {
    int32_t abi_version; /* LSPRNN_ABI_VERSION */
    int32_t cell;        /* LSPRNN_CELL_* */
    int32_t num_layers;  /* 1..8 */
    int32_t input_size;  /* layer 0 input width, a multiple of 4 */
    int32_t hidden_size; /* 256 or 512 (the sizes the reference uses) */
    int32_t max_steps;   /* longest sequence lsprnn_forward will see (sizes the workspace) */
    uint32_t flags;      /* LSPRNN_FLAG_* */
} lsprnn_config;

typedef struct lsprnn_handle lsprnn_handle;

int lsprnn_create(const lsprnn_config *cfg, lsprnn_handle **out);
int lsprnn_destroy(lsprnn_handle *h);
const char *lsprnn_last_error(void);
int lsprnn_abi_version(void);

/* Weight ingress: torch.nn.GRU / nn.LSTM parameter names "weight_ih_l<k>", "weight_hh_l<k>", "bias_ih_l<k>",
 * "bias_hh_l<k>" (gate order r,z,n / i,f,g,o as in PyTorch), host fp32. */
int lsprnn_num_tensors(const lsprnn_handle *h);
int lsprnn_tensor_info(const lsprnn_handle *h, int index, const char **key, size_t *numel);
int lsprnn_set_tensor(lsprnn_handle *h, const char *key, const float *host_data, size_t numel);
size_t lsprnn_packed_bytes(const lsprnn_handle *h);
int lsprnn_pack_weights(lsprnn_handle *h, void *host_dst, size_t bytes);
int lsprnn_bind_weights(lsprnn_handle *h, const void *packed_dev, size_t bytes);
size_t lsprnn_workspace_bytes(const lsprnn_handle *h);
int lsprnn_bind_workspace(lsprnn_handle *h, void *workspace_dev, size_t bytes);

/* output, _ = rnn(x) for x [1][T][input_size]: out_dev [T][hidden_size] = the last layer's hidden states. */
int lsprnn_forward(lsprnn_handle *h, const float *x_dev, int T, float *out_dev, void *stream);

/* Waits for `stream`; *code == 0 if the last forward completed, else the hand-off that timed out (bounded polls). */
int lsprnn_status(lsprnn_handle *h, void *stream, uint32_t *code);

/* y[M][N] = act((x[M][K] W[N][K]^T) * scale[n] + shift[n]): nn.Linear (+ eval BatchNorm1d folded into scale/shift,
 * + LeakyReLU(0.2) when leaky != 0).  scale_dev may be NULL (= 1).  K % 4 == 0.  Stateless. */
int lsprnn_linear(const float *x_dev, const float *w_dev, const float *scale_dev, const float *shift_dev, float *y_dev,
                  int M, int N, int K, int leaky, void *stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* LSPRNN_H */
