"""Drop-in for the reference's ``models/feature2face_G.py`` (Feature2Face_G, :8-34).

Same constructor argument (``opt``), same attribute name ``netG`` (it prefixes every
checkpoint key), same variant selection on ``opt.size`` -- but ``forward`` runs the
hand-written gfx950 kernels instead of torch.nn.
"""
from __future__ import annotations

import warnings

import torch.nn as nn

from .networks import Feature2FaceGenerator
from .unet_small import Feature2FaceGenerator_Unet


class Feature2Face_G(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.isTrain = getattr(opt, "isTrain", False)
        size = getattr(opt, "size", None)
        if size not in ("small", "normal", "large"):
            raise ValueError("opt.size must be 'small', 'normal' or 'large' (feature2face_G.py:16-21), got %r" % (size,))
        if size == "small":
            # feature2face_G.py:17: the pix2pix U-Net on a 23-channel input; no shipped config selects it
            # opt.fp16 (feature2face_G.py:28-30 wraps whichever netG in autocast): the fp16 storage plan of include/lspunet.h
            fp16 = bool(getattr(opt, "fp16", 0))
            if fp16 and opt.ngf % 64 != 0:
                warnings.warn("opt.fp16 needs ngf %% 64 == 0 (got %d): running fp32" % opt.ngf)
                fp16 = False
            self.netG = Feature2FaceGenerator_Unet(input_nc=23, output_nc=3, num_downs=opt.n_downsample_G, ngf=opt.ngf, dtype="f16" if fp16 else "f32")
        else:
            # feature2face_G.py:19-21 hard-codes input_nc=13, output_nc=3
            # opt.fp16 (base_options_feature2face.py:59; feature2face_G.py:28-30 wraps netG in torch.cuda.amp.autocast): fp16 storage path of
            # the HIP renderer -- fp16 activations and conv weights, fp32 accumulate and epilogue, a float16 tensor out like autocast returns
            fp16 = bool(getattr(opt, "fp16", 0))
            if fp16 and opt.ngf % 64 != 0:
                warnings.warn("opt.fp16 needs ngf %% 64 == 0 (got %d): running fp32" % opt.ngf)
                fp16 = False
            self.netG = Feature2FaceGenerator(size, input_nc=13, output_nc=3,
                                              num_downs=opt.n_downsample_G, ngf=opt.ngf, feat_nc=1, dtype="f16" if fp16 else "f32")

    def forward(self, input):
        return self.netG(input)

    def render(self, feature_map, cand_image):
        """inference() fast path: the two API tensors go to the first kernel as they are;
        the reference's torch.cat (feature2face_model.py:231) is never materialised."""
        return self.netG.render(feature_map, cand_image)
