"""Drop-in for the reference's ``models/audio2headpose_model.py`` (inference side).

Same class name, constructor ``(opt)``, ``model_names``, ``setup``/``eval`` (BaseModel) and
``generate_sequences(audio_feats, pre_headpose, fill_zero=True, sigma_scale=0.0, opt=[])`` ->
``np.ndarray [nframe, A2H_GMM_ndim]`` (reference :133-187).  The per-frame Python loop, its 255-wide
re-evaluation and the ``.cpu()`` GMM sampling become one device call (csrc/a2h.hip).

Random numbers: the reference draws, per frame and on the CPU default generator, ``torch.multinomial``
(one sample) and ``torch.randn(1, ndim)`` (losses.py:87, 95).  The same draws are made here, in the same
order, before the device call, so a run under ``torch.manual_seed(s)`` reproduces the reference's.
"""
from __future__ import annotations

import numpy as np
import torch

from . import networks
from .audio2headpose import Audio2Headpose, Audio2Headpose_LSTM
from .base_model import BaseModel


def draw_gmm_noise(nframe: int, ncenter: int, ndim: int):
    """The CPU RNG stream of ``nframe`` Sample_GMM calls (losses.py:68-112 with b*T == 1):
    multinomial(prob, 1, replacement=True) == argmax(prob / Exp(1)) consumes ``ncenter`` exponential draws,
    then ``torch.randn(1, ndim)``.  Returns (noise [nframe, ndim], expq [nframe, ncenter])."""
    noise = torch.empty(nframe, ndim)
    expq = torch.empty(nframe, ncenter)
    for i in range(nframe):
        expq[i] = torch.empty(1, ncenter).exponential_(1)[0]
        noise[i] = torch.randn(1, ndim).float()[0]
    return noise, expq


class Audio2HeadposeModel(BaseModel):
    def __init__(self, opt):
        BaseModel.__init__(self, opt)
        self.model_names = ["Audio2Headpose"]
        if opt.feature_decoder not in ("WaveNet", "LSTM"):
            raise NotImplementedError("feature_decoder=%r: the reference knows 'WaveNet' and 'LSTM'" % (opt.feature_decoder,))
        if not self.gpu_ids:
            raise RuntimeError("Audio2HeadposeModel here is the MI355X path: gpu_ids must name a device (no CPU path)")
        net = Audio2Headpose(opt) if opt.feature_decoder == "WaveNet" else Audio2Headpose_LSTM(opt)
        self.Audio2Headpose = networks.init_net(net, init_type="normal", init_gain=0.02, gpu_ids=opt.gpu_ids)

    def _net(self) -> Audio2Headpose:
        n = self.Audio2Headpose
        return n.module if hasattr(n, "module") else n

    def generate_sequences(self, audio_feats, pre_headpose, fill_zero=True, sigma_scale=0.0, opt=[]):
        opt = opt if opt != [] else self.opt
        net = self._net()
        H = self.opt.APC_hidden_size
        audio = np.asarray(audio_feats, dtype=np.float32).reshape(-1, 2 * H)
        frame_future = opt.frame_future
        nframe = audio.shape[0] - frame_future
        if self.opt.feature_decoder == "LSTM":
            return self._generate_lstm(net, audio, sigma_scale)
        if not fill_zero:
            return None                                   # reference :166-167
        if nframe < 1:
            return np.zeros([max(nframe, 0), opt.A2H_GMM_ndim])
        nd, nc = self.opt.A2H_GMM_ndim, self.opt.A2H_GMM_ncenter
        dev = self.device
        noise = expq = None
        if self.opt.loss == "GMM":
            noise_h, expq_h = draw_gmm_noise(nframe, nc, nd)
            noise = noise_h.to(dev)
            expq = expq_h.to(dev) if nc > 1 else None
        eng = net.engine(dev, audio.shape[0])
        a = torch.from_numpy(audio).to(dev)
        pre = torch.from_numpy(np.asarray(pre_headpose, dtype=np.float32).reshape(-1)).to(dev)
        out = eng.generate_checked(a, pre, noise, expq, float(sigma_scale), int(frame_future))     # status checked, retried on a lost hand-off
        return out.cpu().numpy().astype(np.float64)       # the reference fills an np.zeros (float64) array, :149

    def _generate_lstm(self, net, audio, sigma_scale):
        """reference :189-202: one forward over ALL audio rows, one Sample_GMM over all of them; returns [rows, ndim] float32.
        RNG order of that single call: torch.multinomial draws rows x ncenter exponentials, then torch.randn(rows, ndim)."""
        import ctypes
        from . import _native as N
        nd, nc = self.opt.A2H_GMM_ndim, self.opt.A2H_GMM_ncenter
        rows = audio.shape[0]
        dev = self.device
        preds = net.forward(torch.from_numpy(np.ascontiguousarray(audio)).unsqueeze(0).to(dev))[0].contiguous()
        if net.status():
            raise RuntimeError("LSTM kernel: inter-workgroup hand-off timed out")
        if self.opt.loss != "GMM":
            return preds.cpu().numpy()
        expq = torch.empty(rows, nc).exponential_(1)
        noise = torch.randn(rows, nd).float()
        out = torch.empty((rows, nd), dtype=torch.float32, device=dev)
        nz, eq = noise.to(dev), (expq.to(dev) if nc > 1 else None)
        with torch.cuda.device(dev):
            N.check_a2h(N.load().lspa2h_sample_gmm(ctypes.c_void_p(preds.data_ptr()), rows, nc, nd, ctypes.c_void_p(nz.data_ptr()),
                                                   ctypes.c_void_p(eq.data_ptr()) if eq is not None else None, ctypes.c_float(float(sigma_scale)),
                                                   ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return out.cpu().numpy()
