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
