"""Host-side handles of the recurrent stacks and dense layers of the audio front-end (include/lsprnn.h).
torch is plumbing (device memory, stream); the arithmetic is in csrc/rnn.hip and csrc/gemm_f32.h.  No CPU path."""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional

import numpy as np
import torch

from . import _native as N


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _require(name: str, t: torch.Tensor) -> None:
    if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError("%s must be a contiguous float32 device tensor (there is no CPU path)" % name)


class RecurrentEngine:
    """torch.nn.GRU / nn.LSTM (batch_first, unidirectional, zero initial state) for ONE sequence."""

    def __init__(self, cell: str, num_layers: int, input_size: int, hidden_size: int, max_steps: int = 8192, per_layer: Optional[bool] = None):
        if cell not in N.RNN_CELL_IDS:
            raise ValueError("cell must be 'GRU' or 'LSTM'")
        self.lib = N.load()
        if per_layer is None:      # tools / tests: LSP_RNN_KERNEL=layers forces one launch per layer (read HERE; the library reads no environment)
            per_layer = os.environ.get("LSP_RNN_KERNEL") == "layers"
        self.cfg = N.RNNConfig(N.RNN_ABI_VERSION, N.RNN_CELL_IDS[cell], num_layers, input_size, hidden_size, max_steps,
                               N.RNN_FLAG_PER_LAYER if per_layer else 0)
        self.h = ctypes.c_void_p()
        N.check_rnn(self.lib.lsprnn_create(ctypes.byref(self.cfg), ctypes.byref(self.h)))
        self.cell, self.num_layers, self.input_size, self.hidden_size, self.max_steps = cell, num_layers, input_size, hidden_size, max_steps
        self.blob: Optional[torch.Tensor] = None
        self.ws: Optional[torch.Tensor] = None

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            self.lib.lsprnn_destroy(h)

    def tensor_keys(self) -> Dict[str, int]:
        out, key, numel = {}, ctypes.c_char_p(), ctypes.c_size_t()
        for i in range(self.lib.lsprnn_num_tensors(self.h)):
            N.check_rnn(self.lib.lsprnn_tensor_info(self.h, i, ctypes.byref(key), ctypes.byref(numel)))
            out[key.value.decode()] = numel.value
        return out

    def load_state_dict(self, sd) -> None:
        """Keys 'weight_ih_l0' ... as in nn.GRU / nn.LSTM.state_dict(); missing keys are an error."""
        want = self.tensor_keys()
        missing = [k for k in want if k not in sd]
        if missing:
            raise KeyError("state dict lacks %s" % sorted(missing)[:3])
        for k, numel in want.items():
            v = sd[k]
            a = v.detach().float().cpu().contiguous().numpy() if isinstance(v, torch.Tensor) else np.ascontiguousarray(v, np.float32)
            if a.size != numel:
                raise ValueError("%s: expected %d values, got shape %s" % (k, numel, a.shape))
            N.check_rnn(self.lib.lsprnn_set_tensor(self.h, k.encode(), a.ctypes.data_as(ctypes.c_void_p), a.size))

    def bind(self, device) -> None:
        nbytes = self.lib.lsprnn_packed_bytes(self.h)
        host = torch.empty(nbytes, dtype=torch.uint8)
        N.check_rnn(self.lib.lsprnn_pack_weights(self.h, ctypes.c_void_p(host.data_ptr()), nbytes))
        self.blob = host.to(device)
        self.ws = torch.empty(self.lib.lsprnn_workspace_bytes(self.h), dtype=torch.uint8, device=device)
        N.check_rnn(self.lib.lsprnn_bind_weights(self.h, ctypes.c_void_p(self.blob.data_ptr()), self.blob.numel()))
        N.check_rnn(self.lib.lsprnn_bind_workspace(self.h, ctypes.c_void_p(self.ws.data_ptr()), self.ws.numel()))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [T, input_size] -> [T, hidden_size] (last layer's hidden states); asynchronous on the current stream."""
        if self.blob is None:
            raise RuntimeError("RecurrentEngine.bind(device) first")
        _require("x", x)
        if x.dim() != 2 or x.shape[1] != self.input_size:
            raise ValueError("x must be [T, %d]" % self.input_size)
        out = torch.empty((x.shape[0], self.hidden_size), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            N.check_rnn(self.lib.lsprnn_forward(self.h, ctypes.c_void_p(x.data_ptr()), x.shape[0], ctypes.c_void_p(out.data_ptr()), _stream(x.device)))
        return out

    def forward_checked(self, x: torch.Tensor, retries: int = 2) -> torch.Tensor:
        """forward() + the status word, synchronously; a lost inter-workgroup hand-off (bounded polls: another stream held the CUs) is retried
        after the device has drained -- the recurrence is deterministic, the retry computes the same numbers -- and raised if it persists."""
        code = 0
        for attempt in range(retries + 1):
            out = self.forward(x)
            code = self.status()
            if code == 0:
                return out
            torch.cuda.synchronize(x.device)
        raise RuntimeError("recurrent kernel: inter-workgroup hand-off timed out %d times in a row (status %d)" % (retries + 1, code))

    def status(self) -> int:
        code = ctypes.c_uint32()
        dev = self.blob.device
        with torch.cuda.device(dev):
            N.check_rnn(self.lib.lsprnn_status(self.h, _stream(dev), ctypes.byref(code)))
        return code.value


class Linear:
    """nn.Linear, optionally followed by eval-mode BatchNorm1d and LeakyReLU(0.2), as one fused GEMM.
    The BatchNorm fold (scale = gamma / sqrt(var + eps), shift = (bias - mean) * scale + beta) is done once, in float64."""

    def __init__(self, weight, bias, bn=None, leaky: bool = False, device=None):
        w = np.ascontiguousarray(np.asarray(weight, np.float32))
        b = np.asarray(bias, np.float64)
        if bn is not None:
            gamma, beta, mean, var = (np.asarray(t, np.float64) for t in bn)
            scale = gamma / np.sqrt(var + 1e-5)
            shift = (b - mean) * scale + beta
            self.scale = torch.from_numpy(scale.astype(np.float32)).to(device)
        else:
            shift, self.scale = b, None
        self.w = torch.from_numpy(w).to(device)
        self.shift = torch.from_numpy(shift.astype(np.float32)).to(device)
        self.leaky = bool(leaky)
        self.out_features, self.in_features = w.shape

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        _require("x", x)
        if x.dim() != 2 or x.shape[1] != self.in_features:
            raise ValueError("x must be [M, %d]" % self.in_features)
        y = torch.empty((x.shape[0], self.out_features), dtype=torch.float32, device=x.device)
        lib = N.load()
        with torch.cuda.device(x.device):
            N.check_rnn(lib.lsprnn_linear(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(self.w.data_ptr()),
                                          ctypes.c_void_p(self.scale.data_ptr()) if self.scale is not None else None,
                                          ctypes.c_void_p(self.shift.data_ptr()), ctypes.c_void_p(y.data_ptr()),
                                          x.shape[0], self.out_features, self.in_features, int(self.leaky), _stream(x.device)))
        return y
