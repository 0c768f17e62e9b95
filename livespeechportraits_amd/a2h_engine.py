"""Host-side handle of the head-pose generator (include/lspa2h.h): weight ingress, packing, buffers.

torch is plumbing here (device memory, stream); all arithmetic is in csrc/a2h.hip.  No CPU path."""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _native as N


class HeadposeEngine:
    """One Audio2Headpose network on one device.

    Mirrors what ``Audio2Headpose.__init__`` reads from ``opt`` (reference models/audio2headpose.py:8-37)."""

    def __init__(self, residual_layers: int = 7, residual_blocks: int = 2, residual_channels: int = 128,
                 dilation_channels: int = 128, skip_channels: int = 256, kernel_size: int = 2,
                 input_channels: int = 12, cond_channels: int = 512, hidden_size: int = 512,
                 ncenter: int = 1, ndim: int = 12, loss: str = "GMM", max_audio_frames: int = 4096,
                 single_workgroup: bool = False):
        if loss not in N.A2H_LOSS_IDS:
            raise ValueError("loss must be 'GMM' or 'L2', got %r" % (loss,))
        self.lib = N.load()
        self.cfg = N.A2HConfig(N.A2H_ABI_VERSION, residual_layers, residual_blocks, residual_channels, dilation_channels,
                               skip_channels, kernel_size, input_channels, cond_channels, hidden_size, ncenter, ndim,
                               N.A2H_LOSS_IDS[loss], max_audio_frames,
                               (N.A2H_FLAG_SINGLE_WORKGROUP if single_workgroup or os.environ.get("LSP_A2H_KERNEL") == "stream" else 0) |
                               (N.A2H_FLAG_CONSECUTIVE_BLOCKS if os.environ.get("LSP_A2H_SPREAD", "0") not in ("", "0") else 0))   # (env: tools only)
        self.h = ctypes.c_void_p()
        N.check_a2h(self.lib.lspa2h_create(ctypes.byref(self.cfg), ctypes.byref(self.h)))
        self.ndim, self.ncenter, self.loss = ndim, ncenter, loss
        self.hidden_size = hidden_size
        self.max_audio_frames = max_audio_frames
        self.receptive_field = self.lib.lspa2h_receptive_field(self.h)
        self.blob: Optional[torch.Tensor] = None
        self.ws: Optional[torch.Tensor] = None

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            self.lib.lspa2h_destroy(h)

    # ---- weights ---------------------------------------------------------------------------
    def tensor_keys(self) -> Dict[str, int]:
        out = {}
        key, numel = ctypes.c_char_p(), ctypes.c_size_t()
        for i in range(self.lib.lspa2h_num_tensors(self.h)):
            N.check_a2h(self.lib.lspa2h_tensor_info(self.h, i, ctypes.byref(key), ctypes.byref(numel)))
            out[key.value.decode()] = numel.value
        return out

    def load_state_dict(self, sd) -> None:
        """sd: reference keys (optionally 'module.'-prefixed) -> tensor / ndarray.  Missing keys are an error."""
        want = self.tensor_keys()
        have = {}
        for k, v in sd.items():
            k = k[7:] if k.startswith("module.") else k
            if k.startswith("Audio2Headpose."):
                k = k[len("Audio2Headpose."):]
            have[k] = v
        missing = [k for k in want if k not in have]
        if missing:
            raise KeyError("state dict lacks %d tensors, e.g. %s" % (len(missing), sorted(missing)[:3]))
        for k, numel in want.items():
            v = have[k]
            a = v.detach().float().cpu().contiguous().numpy() if isinstance(v, torch.Tensor) else np.ascontiguousarray(v, np.float32)
            if a.size != numel:
                raise ValueError("%s: expected %d values, got shape %s" % (k, numel, a.shape))
            N.check_a2h(self.lib.lspa2h_set_tensor(self.h, k.encode(), a.ctypes.data_as(ctypes.c_void_p), a.size))

    def bind(self, device) -> None:
        nbytes = self.lib.lspa2h_packed_bytes(self.h)
        host = torch.empty(nbytes, dtype=torch.uint8)
        N.check_a2h(self.lib.lspa2h_pack_weights(self.h, ctypes.c_void_p(host.data_ptr()), nbytes))
        self.blob = host.to(device)
        self.ws = torch.empty(self.lib.lspa2h_workspace_bytes(self.h), dtype=torch.uint8, device=device)
        N.check_a2h(self.lib.lspa2h_bind_weights(self.h, ctypes.c_void_p(self.blob.data_ptr()), self.blob.numel()))
        N.check_a2h(self.lib.lspa2h_bind_workspace(self.h, ctypes.c_void_p(self.ws.data_ptr()), self.ws.numel()))

    # ---- generate --------------------------------------------------------------------------
    def _args(self, audio, pre, noise, expq, sigma_scale, frame_future):
        if self.blob is None:
            raise RuntimeError("HeadposeEngine.bind(device) first")
        for name, t in (("audio", audio), ("pre", pre), ("noise", noise), ("expq", expq)):
            if t is not None and (not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous()):
                raise ValueError("%s must be a contiguous float32 device tensor (there is no CPU path)" % name)
        n_audio = audio.shape[0]
        if audio.dim() != 2 or audio.shape[1] != 2 * self.hidden_size:
            raise ValueError("audio must be [n_audio, %d]" % (2 * self.hidden_size))
        nframe = n_audio - frame_future
        if noise is not None and tuple(noise.shape) != (nframe, self.ndim):
            raise ValueError("noise must be [%d, %d]" % (nframe, self.ndim))
        if expq is not None and tuple(expq.shape) != (nframe, self.ncenter):
            raise ValueError("expq must be [%d, %d]" % (nframe, self.ncenter))
        out = torch.empty(max(nframe, 0), self.ndim, dtype=torch.float32, device=audio.device)
        ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        stream = ctypes.c_void_p(torch.cuda.current_stream(audio.device).cuda_stream)
        return out, [self.h, ptr(audio), n_audio, ptr(pre), ptr(noise), ptr(expq), ctypes.c_float(sigma_scale),
                     frame_future, ptr(out), nframe, stream]

    def generate(self, audio: torch.Tensor, pre: torch.Tensor, noise: Optional[torch.Tensor], expq: Optional[torch.Tensor],
                 sigma_scale: float, frame_future: int) -> torch.Tensor:
        """[n_audio - frame_future, ndim] head poses; asynchronous on the current stream."""
        out, args = self._args(audio, pre, noise, expq, sigma_scale, frame_future)
        with torch.cuda.device(audio.device):
            N.check_a2h(self.lib.lspa2h_generate(*args))
        return out

    def generate_checked(self, audio, pre, noise, expq, sigma_scale: float, frame_future: int, retries: int = 2) -> torch.Tensor:
        """generate() + the status word, synchronously.  The pipeline's workgroups hand activations to each other through polled mailboxes;
        every poll is bounded, so a workgroup that could not get onto the chip (another stream holding the CUs for the whole time-out) turns
        into a status code, never a hang.  The results of such a call are invalid: wait for the device to drain and run it again -- the
        draws are inputs (``noise`` / ``expq``), so the retry computes exactly the same poses -- and only raise if it keeps happening."""
        code = 0
        for attempt in range(retries + 1):
            out = self.generate(audio, pre, noise, expq, sigma_scale, frame_future)
            code = self.status(audio.device)
            if code == 0:
                return out
            torch.cuda.synchronize(audio.device)          # whatever occupied the CUs finishes; the next attempt starts on a quiet device
        raise RuntimeError("head-pose kernel: inter-workgroup hand-off 0x%x timed out %d times in a row" % (code, retries + 1))

    def generate_timed(self, audio, pre, noise, expq, sigma_scale, frame_future) -> Tuple[torch.Tensor, float, float]:
        out, args = self._args(audio, pre, noise, expq, sigma_scale, frame_future)
        pre_ms, loop_ms = ctypes.c_float(), ctypes.c_float()
        with torch.cuda.device(audio.device):
            N.check_a2h(self.lib.lspa2h_generate_timed(*args, ctypes.byref(pre_ms), ctypes.byref(loop_ms)))
        return out, pre_ms.value, loop_ms.value

    def status(self, device=None) -> int:
        """Synchronises the current stream; 0 if the last generate() completed, else the code of the hand-off that timed out."""
        code = ctypes.c_uint32()
        dev = device if device is not None else self.blob.device
        with torch.cuda.device(dev):
            N.check_a2h(self.lib.lspa2h_status(self.h, ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream), ctypes.byref(code)))
        return code.value

    def debug_cond(self) -> torch.Tensor:
        """down_audio_feats of the last generate() call (copied out of the workspace)."""
        p, rows, cols = ctypes.c_void_p(), ctypes.c_int(), ctypes.c_int()
        N.check_a2h(self.lib.lspa2h_debug_cond(self.h, ctypes.byref(p), ctypes.byref(rows), ctypes.byref(cols)))
        off = p.value - self.ws.data_ptr()
        return self.ws[off: off + rows.value * cols.value * 4].view(torch.float32).view(rows.value, cols.value).clone()
