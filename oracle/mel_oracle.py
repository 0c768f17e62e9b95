"""ORACLE (test infrastructure) -- the mel front-end of the audio path on the CPU.

Restates, with the same torch / numpy operations in the same order:
  funcs/utils.py:61-83        compute_mel_one_sequence: 2 * nframe windows of 266 samples every 133.33 samples, each through
  funcs/audio_funcs.py:20-75  Audio2Mel(n_fft=512, hop=133, win=266, 80 mels, 90..7600 Hz).forward: reflect pad 189, ONE stft frame
                              (center=False; the 266-tap hann window sits at samples 123..388 of the 512-point frame), magnitude,
                              mel filterbank, log(clamp(1e-5)), (x - log 1e-5) / -log 1e-5
  librosa.filters.mel         (third-party, requirements.txt: librosa==0.7.0, ABSENT from this image): Slaney mel scale
                              (htk=False), triangular filters, norm=1 (area normalisation 2 / (f[i+2] - f[i])), float32 --
                              restated here from the published algorithm.
Pinning: oracle/make_golden_mel.py executes the REFERENCE's own two functions with exactly two shims -- torch.stft gets
return_complex=False (what torch 1.7.1 did implicitly; torch 2.10 refuses the implicit form) and librosa.filters.mel is this
file's `slaney_mel_filterbank` -- and asserts this restatement bit-identical.  So the windowing / stft / log path is pinned on
the reference's code; the FILTERBANK VALUES are parity-unpinned (no librosa to compare with).
Only tests/ and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

SR, N_FFT, HOP, WIN, N_MELS, FMIN, FMAX = 16000, 512, int(16000 / 120), int(16000 / 60), 80, 90, 7600.0
MIN_MEL = math.log(1e-5)


def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    if f.ndim:
        m = f >= min_log_hz
        mels[m] = min_log_mel + np.log(f[m] / min_log_hz) / logstep
    elif f >= min_log_hz:
        mels = min_log_mel + np.log(f / min_log_hz) / logstep
    return mels


def _mel_to_hz(mels):
    mels = np.asanyarray(mels, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * mels
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    m = mels >= min_log_mel
    freqs[m] = min_log_hz * np.exp(logstep * (mels[m] - min_log_mel))
    return freqs


def slaney_mel_filterbank(sr=SR, n_fft=N_FFT, n_mels=N_MELS, fmin=FMIN, fmax=FMAX) -> np.ndarray:
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) of librosa 0.7.0 with its defaults htk=False, norm=1, dtype=float32"""
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float32)
    fftfreqs = np.linspace(0, float(sr) / 2, 1 + n_fft // 2, endpoint=True)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


class Audio2Mel:
    """audio_funcs.py:20-75 for the parameters compute_mel_one_sequence passes"""

    def __init__(self):
        self.window = torch.hann_window(WIN).float()
        self.mel_basis = torch.from_numpy(slaney_mel_filterbank()).float()

    def forward(self, audio: torch.Tensor) -> torch.Tensor:        # [B, 1, T] -> [B, 80, T']
        p = (N_FFT - HOP) // 2
        audio = F.pad(audio, (p, p), "reflect").squeeze(1)
        fft = torch.view_as_real(torch.stft(audio, n_fft=N_FFT, hop_length=HOP, win_length=WIN, window=self.window, center=False,
                                            return_complex=True))
        real_part, imag_part = fft.unbind(-1)
        magnitude = torch.sqrt(real_part ** 2 + imag_part ** 2)
        mel_output = torch.matmul(self.mel_basis, magnitude)
        log_mel_spec = torch.log(torch.clamp(mel_output, min=1e-5))
        return (log_mel_spec - MIN_MEL) / -MIN_MEL


def compute_mel_one_sequence(audio: np.ndarray, winlen=1 / 60, winstep=0.5 / 60, sr=16000) -> np.ndarray:
    """utils.py:61-83: float64 [2 * nframe, 80]"""
    a2m = Audio2Mel()
    nframe = int(audio.shape[0] / 16000 * 60)
    mel_nframe = 2 * nframe
    mel_frame_len = int(sr * winlen)
    mel_frame_step = sr * winstep
    mel80s = np.zeros([mel_nframe, 80])
    for i in range(mel_nframe):
        st = int(i * mel_frame_step)
        clip = audio[st: st + mel_frame_len]
        if len(clip) < mel_frame_len:
            clip = np.concatenate([clip, np.zeros([mel_frame_len - len(clip)])])
        mel80s[i] = a2m.forward(torch.from_numpy(clip).unsqueeze(0).unsqueeze(0).float()).numpy()[0].T
    return mel80s
