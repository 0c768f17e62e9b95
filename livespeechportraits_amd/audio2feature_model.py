"""Drop-in for the reference's ``models/audio2feature_model.py`` (inference side): same class name, ``(opt)``
constructor, ``model_names``, BaseModel ``setup``/``eval`` and
``generate_sequences(audio_feats, sample_rate=16000, fps=60, fill_zero=True, opt=[])`` -> ``np.ndarray [nframe, ndim]``
(reference :98-137).  LSTM decoder only (the default)."""
from __future__ import annotations

import numpy as np
import torch

from . import networks
from .audio2feature import Audio2Feature
from .base_model import BaseModel


class Audio2FeatureModel(BaseModel):
    def __init__(self, opt):
        BaseModel.__init__(self, opt)
        self.model_names = ["Audio2Feature"]
        if not self.gpu_ids:
            raise RuntimeError("Audio2FeatureModel here is the MI355X path: gpu_ids must name a device (no CPU path)")
        self.Audio2Feature = networks.init_net(Audio2Feature(opt), init_type="normal", init_gain=0.02, gpu_ids=opt.gpu_ids)

    def generate_sequences(self, audio_feats, sample_rate=16000, fps=60, fill_zero=True, opt=[]):
        opt = opt if opt != [] else self.opt
        frame_future = opt.frame_future
        audio_feats = np.asarray(audio_feats, np.float32)
        nframe = int(audio_feats.shape[0] / 2)
        if not frame_future == 0:                                  # reference :117-119: repeat the last row 2*frame_future times
            tail = np.repeat(audio_feats[-1], 2 * frame_future).reshape(-1, 2 * frame_future).T
            audio_feats = np.concatenate([audio_feats, tail])
        net = self.Audio2Feature.module if hasattr(self.Audio2Feature, "module") else self.Audio2Feature
        x = torch.from_numpy(np.ascontiguousarray(audio_feats)).unsqueeze(0).float().to(self.device)
        preds = net.forward(x)
        code = net.status()
        if code:
            raise RuntimeError("LSTM kernel: inter-workgroup hand-off 0x%x timed out" % code)
        preds = preds[0, frame_future:].cpu().numpy() if not frame_future == 0 else preds[0, :].cpu().numpy()
        assert preds.shape[0] == nframe
        return preds
