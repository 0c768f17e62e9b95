/* lspmel.h -- C ABI of the mel front-end of the audio path (first item).
 * Exported by livespeechportraits_amd/liblspf2f.so; gfx950 only, no CPU path.
 *
 * Replaces (reference file:line):
 *   funcs/utils.py:61-83        compute_mel_one_sequence: a Python loop of 2 * nframe single-window calls (266 samples every
 *                               133.33 samples) of
 *   funcs/audio_funcs.py:20-75  Audio2Mel(n_fft 512, hop 133, win 266, 80 mels, 90..7600 Hz).forward: reflect pad, ONE stft
 *                               frame, magnitude, mel filterbank, log(clamp(1e-5)), normalisation to [0, 1]
 *   demo.py:185                 the call site (1374 windows for the 687-frame clip)
 * Here all windows of an utterance are one batch: gather (reflect pad folded into the index map), a windowed 512-point DFT as a
 * [nwin x 268] x [268 x 514] fp32 MFMA GEMM (only 266 taps of the frame are non-zero), magnitude, the [80 x 257] filterbank as
 * a second GEMM, log / normalise.
 * The filterbank is librosa.filters.mel of librosa 0.7.0 (requirements.txt; absent from the build image) restated from the
 * published algorithm: Slaney scale, area-normalised triangles -- parity-unpinned; the rest of the path is pinned on the
 * reference's own functions (oracle/make_golden_mel.py).
 */
#ifndef LSPMEL_H
#define LSPMEL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* the library is built with -fvisibility=hidden: exactly the functions declared below are exported */
#pragma GCC visibility push(default)

#define LSPMEL_OK 0
#define LSPMEL_ERR_INVALID_ARGUMENT (-1)
#define LSPMEL_ERR_SHAPE (-2)
#define LSPMEL_ERR_HIP (-3)

#define LSPMEL_SAMPLE_RATE 16000
#define LSPMEL_N_FFT 512
#define LSPMEL_WIN 266          /* int(16000 / 60) */
#define LSPMEL_N_MELS 80

/* 2 * int(nsamples / 16000 * 60): the number of mel rows compute_mel_one_sequence produces (utils.py:68-69) */
int lspmel_num_windows(int64_t nsamples);

/* Host side, once: the two GEMM operands as one float blob -- [514][268] windowed DFT rows (cos rows 0..256, -sin rows
 * 257..513; torch.hann_window(266) folded in, frame offset 123) followed by [80][260] filterbank rows (zero padded). */
size_t lspmel_basis_floats(void);
int lspmel_make_basis(float *host_blob, size_t nfloats);

size_t lspmel_workspace_bytes(int nwindows);

/* audio_dev float32 [nsamples] (what librosa.load(sr=16000) returns, demo.py:178) -> mel_dev float32 [nwindows][80] with
 * nwindows == lspmel_num_windows(nsamples); basis_dev = the uploaded blob; workspace >= lspmel_workspace_bytes(nwindows). */
int lspmel_compute(const float *audio_dev, int64_t nsamples, const float *basis_dev, int nwindows, float *mel_dev,
                   void *workspace_dev, size_t workspace_bytes, void *hip_stream);

const char *lspmel_last_error(void);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
