"""Drop-in for the reference's ``models/feature2face_model.py`` -- inference side.

Same class name (so ``create_model`` finds it by the lowercase-name rule,
models/__init__.py:29-49), same constructor ``(opt)``, ``model_names``, ``setup``,
``eval`` and the hot entry point

    inference(feature_map [B,1,H,W], cand_image [B|1,12,H,W] | None) -> [B,3,H,W] in [-1,1]

(models/feature2face_model.py:225-237), executed by liblspf2f on an MI355X.  ``cand_image``
may have batch 1 while ``feature_map`` has batch B: the candidate stack is constant per
person (demo.py:89-95), so it is broadcast inside the first kernel.
"""
from __future__ import annotations

import torch

from . import networks
from .base_model import BaseModel
from .feature2face_G import Feature2Face_G
from .unet_small import Feature2FaceGenerator_Unet


class Feature2FaceModel(BaseModel):
    def __init__(self, opt):
        BaseModel.__init__(self, opt)
        if self.isTrain:
            raise NotImplementedError("Feature2FaceModel training (optimize_parameters, discriminator, "
                                      "VGG/GAN losses) is out of scope of the HIP renderer")
        self.model_names = ["Feature2Face_G"]
        self.Feature2Face_G = networks.init_net(Feature2Face_G(opt), init_type="normal", init_gain=0.02,
                                                gpu_ids=self.gpu_ids)

    def _g(self) -> Feature2Face_G:
        net = self.Feature2Face_G
        return net.module if isinstance(net, (networks.SingleDeviceParallel, networks.MultiDeviceParallel)) else net

    def inference(self, feature_map, cand_image):
        with torch.no_grad():
            net = self.Feature2Face_G
            if isinstance(net, networks.MultiDeviceParallel):      # opt.gpu_ids with several ids: the batch is sliced over all of them
                return net.render(feature_map, cand_image)
            return self._g().render(feature_map, cand_image)

    def inference_image(self, feature_map, cand_image, replica: int = 0, out=None):
        """inference() followed by util.tensor2im, fused on the device: uint8 [B,H,W,3] frames
        (``util.tensor2im(pred_fake[0])`` of demo.py:268 is ``inference_image(...)[0].cpu().numpy()``).
        ``replica`` > 0 renders through a further handle on the same device and packed weights (single-device normal / large generators): calls with
        different replicas on different streams overlap (render_loop.render_frames(streams=2)).  ``out``: a caller-owned uint8 [B,H,W,3] device tensor for the frames (the library replays one cached hipGraph per set of pointers and keeps eight; a miss re-captures, which
        costs the host ~nothing next to the forward -- measured equal with 12 buffer pairs in rotation, tools/graph_recapture_probe.py -- so this is about not allocating per call);
        render_loop.render_frames passes its own."""
        with torch.no_grad():
            g = self._g().netG
            if feature_map.device.type != "cuda":
                raise RuntimeError("the feature2face HIP renderer needs ROCm tensors; there is no CPU fallback")
            if isinstance(g, Feature2FaceGenerator_Unet):        # size == 'small': its own native plan (include/lspunet.h), same fused uint8 output
                return g.render(feature_map, cand_image, out_u8=True) if out is None else g._get_engine(feature_map.device).render(
                    feature_map.float(), None if cand_image is None else cand_image.float(), True, out=out)
            net = self.Feature2Face_G
            if isinstance(net, networks.MultiDeviceParallel):    # several gpu_ids: sliced over all of them like inference(), uint8 fused on every device
                if out is not None:                              # the slices land on their own devices and are gathered into a new tensor: a caller-owned buffer cannot be honoured
                    raise ValueError("inference_image(out=...) is not supported with several gpu_ids (the frames are gathered from the devices into a new tensor)")
                return net.render_image(feature_map, cand_image)
            e = g._engine_for(feature_map.shape[-1], feature_map.shape[0], feature_map.device)
            if replica:
                e = g._twin_engine(replica, e)
            return e.forward_image(feature_map.float(), cand_image.float() if cand_image is not None else None, out_u8=out)

    def supports_replicas(self) -> bool:
        """whether inference_image(replica=k) means another handle (else the argument is ignored: the small U-Net, several gpu_ids)"""
        return not isinstance(self._g().netG, Feature2FaceGenerator_Unet) and not isinstance(self.Feature2Face_G, networks.MultiDeviceParallel)

    # the reference's abstract training hooks (base_model.py:70-86); kept so callers that probe
    # for them get a clear message instead of an AttributeError
    def set_input(self, data=None, data_info=None):
        raise NotImplementedError("training path is out of scope")

    def forward(self):
        raise NotImplementedError("training path is out of scope")

    def optimize_parameters(self):
        raise NotImplementedError("training path is out of scope")
