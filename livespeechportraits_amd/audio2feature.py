"""Parameter container + device evaluation of the reference's ``Audio2Feature`` in its default LSTM form
(models/audio2feature.py:8-72): downsample MLP (Linear 1024->512, BatchNorm1d, LeakyReLU, Linear 512->512), 3-layer
LSTM(512 -> 256), fc MLP (256->512 BN LReLU ->512 BN LReLU -> output).  Same state-dict keys as the reference."""
from __future__ import annotations

import torch
import torch.nn as nn

from .rnn_engine import Linear, RecurrentEngine


class Audio2Feature(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        if opt.feature_decoder != "LSTM":
            raise NotImplementedError("Audio2Feature: only the (default) LSTM decoder has a HIP path")
        H = opt.APC_hidden_size
        out = (2 * opt.A2L_GMM_ndim + 1) * opt.A2L_GMM_ncenter if opt.loss == "GMM" else opt.A2L_GMM_ndim * opt.predict_length
        self.downsample = nn.Sequential(nn.Linear(2 * H, H), nn.BatchNorm1d(H), nn.LeakyReLU(0.2), nn.Linear(H, H))
        self.LSTM = nn.LSTM(input_size=H, hidden_size=256, num_layers=3, dropout=0, bidirectional=False, batch_first=True)
        self.fc = nn.Sequential(nn.Linear(256, 512), nn.BatchNorm1d(512), nn.LeakyReLU(0.2),
                                nn.Linear(512, 512), nn.BatchNorm1d(512), nn.LeakyReLU(0.2), nn.Linear(512, out))
        self._packed = None
        self._version = None

    def mark_dirty(self):
        self._packed = None

    def _bn(self, m):
        return (m.weight.detach().cpu().numpy(), m.bias.detach().cpu().numpy(), m.running_mean.cpu().numpy(), m.running_var.cpu().numpy())

    def _pack(self, device, T):
        version = tuple(p._version for p in self.parameters()) + tuple(b._version for b in self.buffers())
        pk = self._packed
        if pk is None or self._version != version or pk["lstm"].max_steps < T or pk["lstm"].blob.device != device:
            lin = lambda m, bn=None, leaky=False: Linear(m.weight.detach().cpu().numpy(), m.bias.detach().cpu().numpy(),
                                                         self._bn(bn) if bn is not None else None, leaky, device)
            d, f = self.downsample, self.fc
            lstm = RecurrentEngine("LSTM", 3, self.opt.APC_hidden_size, 256, max_steps=max(T, 4096))
            lstm.load_state_dict({k: v for k, v in self.LSTM.state_dict().items()})
            lstm.bind(device)
            pk = {"d0": lin(d[0], d[1], True), "d3": lin(d[3]), "lstm": lstm,
                  "f0": lin(f[0], f[1], True), "f3": lin(f[3], f[4], True), "f6": lin(f[6])}
            self._packed, self._version = pk, version
        return pk

    def forward(self, audio_features):
        """[1, item_len, APC_hidden] -> [1, item_len // 2, output]   (audio2feature.py:62-69)"""
        if audio_features.dim() != 3 or audio_features.shape[0] != 1:
            raise ValueError("audio_features must be [1, T, ndim] (generate_sequences passes one utterance)")
        if not audio_features.is_cuda:
            raise RuntimeError("Audio2Feature here is the MI355X path: device tensors only (no CPU path)")
        bs, item_len, ndim = audio_features.shape
        x = audio_features.float().contiguous().reshape(-1, ndim * 2)          # pairs of APC frames -> one video frame
        pk = self._pack(audio_features.device, x.shape[0])
        x = pk["d3"](pk["d0"](x))
        x = pk["lstm"].forward_checked(x)        # status word checked; a lost hand-off is retried on a drained device (rnn_engine.py)
        x = pk["f6"](pk["f3"](pk["f0"](x)))
        return x.reshape(bs, item_len // 2, -1)

    def status(self) -> int:
        return self._packed["lstm"].status() if self._packed else 0
