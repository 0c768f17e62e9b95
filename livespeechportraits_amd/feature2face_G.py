"""Drop-in for the reference's ``models/feature2face_G.py`` (Feature2Face_G, :8-34).

Same constructor argument (``opt``), same attribute name ``netG`` (it prefixes every
checkpoint key), same variant selection on ``opt.size`` -- but ``forward`` runs the
hand-written gfx950 kernels instead of torch.nn.
"""
from __future__ import annotations

import warnings

import torch.nn as nn

from .networks import Feature2FaceGenerator


class Feature2Face_G(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.isTrain = getattr(opt, "isTrain", False)
        size = getattr(opt, "size", None)
        if size == "small":
            raise NotImplementedError(
                "opt.size == 'small' (pix2pix U-Net, networks.py:680-769, 23-channel input) is not "
                "selected by any shipped config and is not implemented by the HIP renderer")
        if size not in ("normal", "large"):
            raise ValueError("opt.size must be 'normal' or 'large' (config/*.yaml:23), got %r" % (size,))
        # feature2face_G.py:19-21 hard-codes input_nc=13, output_nc=3
        self.netG = Feature2FaceGenerator(size, input_nc=13, output_nc=3,
                                          num_downs=opt.n_downsample_G, ngf=opt.ngf, feat_nc=1)
        if getattr(opt, "fp16", 0):
            warnings.warn("opt.fp16 is ignored: the HIP renderer computes in fp32 "
                          "(>= the precision of the reference's autocast branch)")

    def forward(self, input):
        return self.netG(input)

    def render(self, feature_map, cand_image):
        """inference() fast path: the two API tensors go to the first kernel as they are;
        the reference's torch.cat (feature2face_model.py:231) is never materialised."""
        return self.netG.render(feature_map, cand_image)
