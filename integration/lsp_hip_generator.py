"""Reference-side binding of liblspf2f.so -- the file a LiveSpeechPortraits maintainer would drop next to
``models/feature2face_G.py`` to keep the reference's OWN ``Feature2FaceModel`` / ``Feature2Face_G`` classes (construction,
checkpoint loading, option handling all unchanged) and run the generator forward on an MI355X.

Self-contained on purpose: ctypes + torch only, nothing imported from livespeechportraits_amd, only the C ABI of
include/lspf2f.h.  (SURVEY.md 8b option 1: "keep the reference class and swap only its Feature2Face_G attribute".)

    model = create_model(opt); model.setup(opt); model.eval()            # the reference's own code (demo.py:168-172)
    install(model, device="cuda:0")                                        # <- the one added line
    pred = model.inference(feature_map, cand_image)                        # demo.py:266, unchanged

Entry points used and the reference interface each replaces:
  lspf2f_create                         Feature2Face_G.__init__ (models/feature2face_G.py:9-24), networks.py:554-572 / 458-476
  lspf2f_set_tensor / _pack_weights     net.load_state_dict(...) (models/base_model.py:212-219): same key names
  lspf2f_bind_weights / _bind_workspace nn.DataParallel's replicate (networks.py:400) / ATen's implicit intermediates
  lspf2f_forward                        Feature2Face_G.forward (feature2face_G.py:27-34) -> generator forward (networks.py:575-579)
"""
import ctypes
import os

import torch

ABI_VERSION = 1
_VARIANTS = {"normal": 0, "large": 1}


class _Cfg(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("abi_version", "variant", "input_nc", "feat_nc", "output_nc", "ngf", "num_downs",
                                              "height", "width", "max_batch", "dtype")] + [("flags", ctypes.c_uint32)]


def _load(path=None):
    path = path or os.environ.get("LSPF2F_LIBRARY", "liblspf2f.so")
    lib = ctypes.CDLL(path)
    lib.lspf2f_last_error.restype = ctypes.c_char_p
    lib.lspf2f_packed_bytes.restype = ctypes.c_size_t
    lib.lspf2f_packed_bytes.argtypes = [ctypes.c_void_p]
    lib.lspf2f_workspace_bytes.restype = ctypes.c_size_t
    lib.lspf2f_workspace_bytes.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.lspf2f_create.argtypes = [ctypes.POINTER(_Cfg), ctypes.POINTER(ctypes.c_void_p)]
    lib.lspf2f_destroy.argtypes = [ctypes.c_void_p]
    lib.lspf2f_set_tensor.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t]
    lib.lspf2f_pack_weights.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    lib.lspf2f_bind_weights.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    lib.lspf2f_bind_workspace.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    lib.lspf2f_forward.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                   ctypes.c_void_p]
    return lib


class HipGenerator:
    """Stands where ``Feature2FaceModel.Feature2Face_G`` stood: callable on the concatenated ``[B, 13, H, W]`` tensor exactly as
    the reference's ``inference`` calls it (models/feature2face_model.py:229-236), weights taken from the reference module."""

    def __init__(self, net, opt, size=512, max_batch=8, library=None):
        """``net``: the reference's Feature2Face_G (possibly DataParallel-wrapped), already holding its checkpoint."""
        self.lib = _load(library)
        self.size, self.max_batch = size, max_batch
        cfg = _Cfg(ABI_VERSION, _VARIANTS[opt.size], 13, 1, 3, opt.ngf, opt.n_downsample_G, size, size, max_batch, 0, 0)
        self.h = ctypes.c_void_p()
        self._ok(self.lib.lspf2f_create(ctypes.byref(cfg), ctypes.byref(self.h)))
        for k, v in net.state_dict().items():
            if k.endswith("num_batches_tracked"):
                continue
            k = k[7:] if k.startswith("module.") else k               # DataParallel prefix (base_model.py:213-215)
            v = v.detach().to("cpu", torch.float32).contiguous()
            self._ok(self.lib.lspf2f_set_tensor(self.h, k.encode(), v.data_ptr(), v.numel()))
        self.blob = torch.empty(self.lib.lspf2f_packed_bytes(self.h), dtype=torch.uint8)
        self._ok(self.lib.lspf2f_pack_weights(self.h, self.blob.data_ptr(), self.blob.numel()))   # fails if a tensor is missing
        self.device = None

    def _ok(self, rc):
        if rc:
            raise RuntimeError("liblspf2f: %s" % self.lib.lspf2f_last_error().decode())

    def to(self, device):
        self.device = torch.device(device)
        self.blob = self.blob.to(self.device)
        self.ws = torch.empty(self.lib.lspf2f_workspace_bytes(self.h, self.max_batch), dtype=torch.uint8, device=self.device)
        self._ok(self.lib.lspf2f_bind_weights(self.h, self.blob.data_ptr(), self.blob.numel()))
        self._ok(self.lib.lspf2f_bind_workspace(self.h, self.ws.data_ptr(), self.ws.numel()))
        return self

    def eval(self):
        return self

    def render(self, feature_map, cand_image):
        """[B,1,H,W] + [B|1,12,H,W] -> [B,3,H,W]; the two tensors go to the first kernel as they are (no torch.cat)."""
        f, c = feature_map.float().contiguous(), cand_image.float().contiguous()
        out = torch.empty(f.shape[0], 3, self.size, self.size, device=f.device)
        with torch.cuda.device(self.device):
            self._ok(self.lib.lspf2f_forward(self.h, f.data_ptr(), c.data_ptr(), c.shape[0], out.data_ptr(), f.shape[0],
                                             torch.cuda.current_stream(self.device).cuda_stream))
        return out

    def __call__(self, x):
        """x = cat([feature_map, cand_image], 1) as Feature2FaceModel.inference builds it."""
        return self.render(x[:, :1], x[:, 1:])

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.lspf2f_destroy(self.h)
            self.h = None


def install(model, device="cuda:0", size=512, max_batch=8, library=None):
    """Swap the generator of a constructed reference ``Feature2FaceModel`` for the HIP one; everything else of the reference
    object (options, inference(), fp16 flag handling, ...) keeps running the reference's code."""
    gen = HipGenerator(model.Feature2Face_G, model.opt, size, max_batch, library).to(device)
    model.Feature2Face_G = gen
    return gen
