"""Parameter containers with the reference's state-dict keys for Audio2Headpose
(reference models/audio2headpose.py:8-37, models/networks.py:74-185, 217-288).

These modules hold weights only -- unmodified ``Audio2Headpose.pkl`` checkpoints load into them -- and
have no forward(): the arithmetic lives in csrc/a2h.hip behind include/lspa2h.h.
Key layout (185 tensors with the default options):
  audio_downsample.{0,3}.{weight,bias}, audio_downsample.1.{weight,bias,running_mean,running_var,num_batches_tracked}
  WaveNet.start_conv{1,2}.*, WaveNet.residual_blocks.<i>.{filter,gate,residual,skip,cond_filter,cond_gate}_conv.*,
  WaveNet.end_conv_{1,2}.*
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .a2h_engine import HeadposeEngine


def _c1(cin: int, cout: int, k: int = 1, dilation: int = 1, bias: bool = True) -> nn.Conv1d:
    return nn.Conv1d(cin, cout, k, dilation=dilation, bias=bias)


class _GatedBlock(nn.Module):
    def __init__(self, dilation, res, dil, skip, k, bias, cond):
        super().__init__()
        self.filter_conv = _c1(res, dil, k, dilation, bias)
        self.gate_conv = _c1(res, dil, k, dilation, bias)
        self.residual_conv = _c1(dil, res, 1, 1, bias)
        self.skip_conv = _c1(dil, skip, 1, 1, bias)
        self.cond_filter_conv = _c1(cond, dil)
        self.cond_gate_conv = _c1(cond, dil)


class _WaveNetParams(nn.Module):
    def __init__(self, layers, blocks, res, dil, skip, k, bias, in_ch, out_ch, cond):
        super().__init__()
        self.start_conv1 = _c1(in_ch, res)
        self.start_conv2 = _c1(res, res)
        self.residual_blocks = nn.ModuleList(
            _GatedBlock(2 ** i, res, dil, skip, k, bias, cond) for _ in range(blocks) for i in range(layers))
        self.end_conv_1 = _c1(skip, out_ch)
        self.end_conv_2 = _c1(out_ch, out_ch)
        self.receptive_field = 1 + blocks * (k - 1) * (2 ** layers - 1)


class Audio2Headpose(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        if not getattr(opt, "A2H_wavenet_use_bias", True):
            raise NotImplementedError("A2H_wavenet_use_bias=False: the kernels expect the (default) biased convs")
        nd, nc, H = opt.A2H_GMM_ndim, opt.A2H_GMM_ncenter, opt.APC_hidden_size
        out_ch = (2 * nd + 1) * nc if opt.loss == "GMM" else nd
        self.audio_downsample = nn.Sequential(nn.Linear(2 * H, H), nn.BatchNorm1d(H), nn.LeakyReLU(0.2), nn.Linear(H, H))
        # the reference passes residual_channels / dilation_channels in swapped positions (audio2headpose.py:25-26
        # vs networks.py:95-96); both are 128 in every shipped configuration, which the kernels require anyway
        self.WaveNet = _WaveNetParams(opt.A2H_wavenet_residual_layers, opt.A2H_wavenet_residual_blocks,
                                      opt.A2H_wavenet_dilation_channels, opt.A2H_wavenet_residual_channels,
                                      opt.A2H_wavenet_skip_channels, opt.A2H_wavenet_kernel_size, True,
                                      opt.A2H_wavenet_input_channels, out_ch, opt.A2H_wavenet_cond_channels)
        self.item_length = self.WaveNet.receptive_field + opt.time_frame_length - 1
        self._engine = None
        self._engine_version = None

    def mark_dirty(self):
        self._engine = None

    def load_state_dict(self, *a, **kw):
        self._engine = None
        return super().load_state_dict(*a, **kw)

    def engine(self, device: torch.device, n_audio: int) -> HeadposeEngine:
        """Pack the current parameters for the device (cached until the weights change)."""
        version = tuple(p._version for p in self.parameters()) + tuple(b._version for b in self.buffers())
        e = self._engine
        if e is None or self._engine_version != version or e.max_audio_frames < n_audio or e.blob.device != device:
            o = self.opt
            e = HeadposeEngine(o.A2H_wavenet_residual_layers, o.A2H_wavenet_residual_blocks, o.A2H_wavenet_residual_channels,
                               o.A2H_wavenet_dilation_channels, o.A2H_wavenet_skip_channels, o.A2H_wavenet_kernel_size,
                               o.A2H_wavenet_input_channels, o.A2H_wavenet_cond_channels, o.APC_hidden_size,
                               o.A2H_GMM_ncenter, o.A2H_GMM_ndim, o.loss, max_audio_frames=max(n_audio, 2048))
            e.load_state_dict({k: v for k, v in self.state_dict().items() if not k.endswith("num_batches_tracked")})
            e.bind(device)
            self._engine, self._engine_version = e, version
        return e

    def forward(self, *a, **kw):
        raise RuntimeError("Audio2Headpose has no per-window forward here: use Audio2HeadposeModel.generate_sequences "
                           "(the HIP path evaluates the WaveNet incrementally; there is no CPU path)")


class Audio2Headpose_LSTM(nn.Module):
    """``feature_decoder == 'LSTM'`` (reference models/audio2headpose.py:56-100): Linear-BN-LeakyReLU-Linear on the 1024-d
    audio rows, LSTM(512 -> 256, 3 layers), Linear-BN-LReLU-Linear-BN-LReLU-Linear to the GMM parameters.  Same keys."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        nd, nc, H = opt.A2H_GMM_ndim, opt.A2H_GMM_ncenter, opt.APC_hidden_size
        out = (2 * nd + 1) * nc if opt.loss == "GMM" else nd
        self.audio_downsample = nn.Sequential(nn.Linear(2 * H, H), nn.BatchNorm1d(H), nn.LeakyReLU(0.2), nn.Linear(H, H))
        self.LSTM = nn.LSTM(input_size=H, hidden_size=256, num_layers=3, dropout=0, bidirectional=False, batch_first=True)
        self.fc = nn.Sequential(nn.Linear(256, 512), nn.BatchNorm1d(512), nn.LeakyReLU(0.2),
                                nn.Linear(512, 512), nn.BatchNorm1d(512), nn.LeakyReLU(0.2), nn.Linear(512, out))
        self._packed = None
        self._version = None

    def mark_dirty(self):
        self._packed = None

    def _pack(self, device, T):
        from .rnn_engine import Linear, RecurrentEngine
        version = tuple(p._version for p in self.parameters()) + tuple(b._version for b in self.buffers())
        pk = self._packed
        if pk is None or self._version != version or pk["lstm"].max_steps < T or pk["lstm"].blob.device != device:
            bn = lambda m: (m.weight.detach().cpu().numpy(), m.bias.detach().cpu().numpy(), m.running_mean.cpu().numpy(), m.running_var.cpu().numpy())
            lin = lambda m, b=None, leaky=False: Linear(m.weight.detach().cpu().numpy(), m.bias.detach().cpu().numpy(),
                                                        bn(b) if b is not None else None, leaky, device)
            d, f = self.audio_downsample, self.fc
            lstm = RecurrentEngine("LSTM", 3, self.opt.APC_hidden_size, 256, max_steps=max(T, 4096))
            lstm.load_state_dict(dict(self.LSTM.state_dict()))
            lstm.bind(device)
            pk = {"d0": lin(d[0], d[1], True), "d3": lin(d[3]), "lstm": lstm,
                  "f0": lin(f[0], f[1], True), "f3": lin(f[3], f[4], True), "f6": lin(f[6])}
            self._packed, self._version = pk, version
        return pk

    def forward(self, audio_features):
        """[1, T, 2*APC_hidden] -> [1, T, output]   (audio2headpose.py:88-99)"""
        if audio_features.dim() != 3 or audio_features.shape[0] != 1:
            raise ValueError("audio_features must be [1, T, ndim]")
        if not audio_features.is_cuda:
            raise RuntimeError("Audio2Headpose_LSTM here is the MI355X path: device tensors only (no CPU path)")
        x = audio_features[0].float().contiguous()
        pk = self._pack(audio_features.device, x.shape[0])
        x = pk["lstm"].forward(pk["d3"](pk["d0"](x)))
        return pk["f6"](pk["f3"](pk["f0"](x))).unsqueeze(0)

    def status(self) -> int:
        return self._packed["lstm"].status() if self._packed else 0
