"""Mel front-end of the audio path on the device -- drop-in for ``funcs/utils.compute_mel_one_sequence`` (utils.py:61-83, called
at demo.py:185), which loops over 2 * nframe windows and pushes each through ``Audio2Mel`` (funcs/audio_funcs.py:20-75) alone.
Here the utterance's windows are one batch through ``lspmel_compute`` (include/lspmel.h).  No CPU path."""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np
import torch

from . import _native as N

_basis = {}          # device -> uploaded GEMM operands (windowed DFT rows + filterbank)


def _basis_on(device: torch.device) -> torch.Tensor:
    key = str(device)
    if key not in _basis:
        lib = N.load()
        n = int(lib.lspmel_basis_floats())
        host = torch.empty(n, dtype=torch.float32)
        N.check_mel(lib.lspmel_make_basis(host.data_ptr(), n))
        _basis[key] = host.to(device)
    return _basis[key]


def compute_mel(audio: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """float32 device tensor [nsamples] (16 kHz) -> float32 device tensor [2 * int(nsamples / 16000 * 60), 80]"""
    if audio.device.type != "cuda":
        raise RuntimeError("the mel front-end runs on the MI355X only (no CPU path); the reference's host path is funcs/utils.py:61-83")
    if audio.dim() != 1 or audio.dtype != torch.float32:
        raise ValueError("audio must be a 1-d float32 tensor (librosa.load(..., sr=16000) gives float32)")
    lib = N.load()
    audio = audio.contiguous()
    nwin = int(lib.lspmel_num_windows(audio.shape[0]))
    if nwin < 1:
        raise ValueError("audio shorter than one video frame (%d samples)" % audio.shape[0])
    dev = audio.device
    mel = out if out is not None else torch.empty((nwin, 80), dtype=torch.float32, device=dev)
    if tuple(mel.shape) != (nwin, 80) or not mel.is_contiguous() or mel.device != dev:
        raise ValueError("out must be a contiguous [%d, 80] tensor on %s" % (nwin, dev))
    ws = torch.empty(int(lib.lspmel_workspace_bytes(nwin)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        N.check_mel(lib.lspmel_compute(audio.data_ptr(), audio.shape[0], _basis_on(dev).data_ptr(), nwin, mel.data_ptr(), ws.data_ptr(),
                                       ws.numel(), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return mel


def compute_mel_one_sequence(audio, hop_length=int(16000 / 120), winlen=1 / 60, winstep=0.5 / 60, sr=16000, fps=60, device="cuda:0"):
    """Same signature and return type as the reference function: numpy float64 [mel_nframe, 80].  Only the parameter set the
    reference itself uses is supported (it hard-codes them into Audio2Mel as well)."""
    if (hop_length, sr, fps) != (int(16000 / 120), 16000, 60) or abs(winlen - 1 / 60) > 1e-12 or abs(winstep - 0.5 / 60) > 1e-12:
        raise NotImplementedError("only the reference's own settings (16 kHz, 60 fps, winlen 1/60, winstep 0.5/60) are supported")
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("the mel front-end runs on the MI355X only (no CPU path)")
    a = torch.from_numpy(np.ascontiguousarray(audio, dtype=np.float32)).to(dev)      # `.float()` of utils.py:78
    return compute_mel(a).cpu().numpy().astype(np.float64)
