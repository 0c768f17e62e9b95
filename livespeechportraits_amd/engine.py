"""Host-side owner of one liblspf2f handle: state dict in, frames out.

PyTorch is plumbing here: it owns the device allocations (packed-weight arena,
workspace, inputs/outputs) and the stream; every FLOP of the generator runs in
the hand-written gfx950 kernels behind the C ABI (include/lspf2f.h).

Replaces, behind ``Feature2FaceModel.inference`` (models/feature2face_model.py:225-237):
``Feature2Face_G.forward`` (models/feature2face_G.py:27-34) and
``Feature2FaceGenerator_{large,normal}.forward`` (models/networks.py:575-579, 479-483).
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Mapping, Optional

import numpy as np
import torch

from . import _native as N


class Engine:
    def __init__(self, variant: str = "large", input_nc: int = 13, feat_nc: int = 1,
                 output_nc: int = 3, ngf: int = 64, num_downs: int = 8, size: int = 512,
                 max_batch: int = 1, keep_intermediates: bool = False, dtype: str = "f32", norm: str = "batch",
                 wino4: bool = False, tune=None):
        if variant not in N.VARIANT_IDS:
            raise ValueError("opt.size must be 'normal' or 'large' for the HIP renderer "
                             "(got %r; the 'small' U-Net is not on the shipped path)" % (variant,))
        if dtype not in N.DTYPE_IDS:
            raise ValueError("dtype must be 'f32', 'bf16' or 'f16'")
        if norm not in N.NORM_IDS:
            raise ValueError("norm must be 'batch' (BatchNorm2d, the shipped checkpoints) or 'instance' (norm_layer=nn.InstanceNorm2d)")
        self.dtype, self.norm = dtype, norm
        self.lib = N.load()
        self.variant, self.input_nc, self.feat_nc, self.output_nc = variant, input_nc, feat_nc, output_nc
        self.ngf, self.num_downs, self.size, self.max_batch = ngf, num_downs, size, max_batch
        cfg = N.Config(N.ABI_VERSION, N.VARIANT_IDS[variant], input_nc, feat_nc, output_nc, ngf,
                       num_downs, size, size, max_batch, N.DTYPE_IDS[dtype],
                       (N.FLAG_KEEP_INTERMEDIATES if keep_intermediates else 0) | (N.FLAG_INSTANCE_NORM if norm == "instance" else 0) |
                       (N.FLAG_WINO4 if wino4 else 0))
        h = ctypes.c_void_p()
        # `wino4`: the stride-1 convs of the >= 32x32 levels on the Winograd F(4x4,3x3) kernel (off by default: DESIGN.md 4.11).  `tune`: the
        # A-B switches of tools and tests (a dict or "k=v,..." string, merged over LSP_HIP_* environment variables by N.tune_string)
        N.check(self.lib.lspf2f_create_tuned(ctypes.byref(cfg), N.tune_string(tune), ctypes.byref(h)))
        self._h = h
        # candidate-stack cache (lspf2f_set_candidates): off by default so that a bare Engine never reuses
        # work across calls; Feature2FaceModel turns it on (the stack is constant per person)
        self.auto_cand_cache = False
        self._cand_key = None
        self._blob_dev: Optional[torch.Tensor] = None
        self._ws: Optional[torch.Tensor] = None
        self.device: Optional[torch.device] = None

    # ---- lifecycle -------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self.lib.lspf2f_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights ----------------------------------------------------------------------
    def expected_tensors(self) -> Dict[str, tuple]:
        out = {}
        name = ctypes.c_char_p()
        dims = (ctypes.c_int64 * 4)()
        nd = ctypes.c_int()
        for i in range(self.lib.lspf2f_num_tensors(self._h)):
            N.check(self.lib.lspf2f_tensor_info(self._h, i, ctypes.byref(name), ctypes.byref(dims),
                                                ctypes.byref(nd)))
            out[name.value.decode()] = tuple(int(dims[d]) for d in range(nd.value))
        return out

    def load_state_dict(self, sd: Mapping[str, object], strict: bool = True) -> List[str]:
        """Feed every expected tensor to the library.  Keys may carry the DataParallel
        'module.' prefix (models/base_model.py:213-215).  Unlike the reference's
        strict=False, a missing key is an error; extra keys (num_batches_tracked, or a
        discriminator's) are returned."""
        norm = {}
        for k, v in sd.items():
            norm[k[7:] if k.startswith("module.") else k] = v
        expected = self.expected_tensors()
        missing = [k for k in expected if k not in norm]
        if missing and strict:
            raise KeyError("checkpoint is missing %d generator tensors, e.g. %s" % (len(missing), missing[:3]))
        for k, shape in expected.items():
            if k not in norm:
                continue
            v = norm[k]
            if isinstance(v, torch.Tensor):
                v = v.detach().to("cpu", torch.float32).contiguous().numpy()
            v = np.ascontiguousarray(v, dtype=np.float32)
            if tuple(v.shape) != shape:
                raise ValueError("shape mismatch for %s: checkpoint %s, network %s" % (k, v.shape, shape))
            N.check(self.lib.lspf2f_set_tensor(self._h, k.encode(), v.ctypes.data, v.size))
        return [k for k in norm if k not in expected]

    def packed_bytes(self) -> int:
        return int(self.lib.lspf2f_packed_bytes(self._h))

    def pack(self) -> torch.Tensor:
        """BN fold + layout reorder on the host (C++); returns the blob as a CPU uint8 tensor."""
        blob = torch.empty(self.packed_bytes(), dtype=torch.uint8)
        N.check(self.lib.lspf2f_pack_weights(self._h, blob.data_ptr(), blob.numel()))
        return blob

    def bind(self, blob: torch.Tensor, device: Optional[torch.device] = None) -> None:
        """Attach a packed blob.  A CPU blob is uploaded to ``device``; a device blob (e.g.
        the receive buffer of the RCCL broadcast) is used in place."""
        if blob.device.type != "cuda":
            if device is None:
                raise ValueError("device required to upload a host blob")
            if not torch.cuda.is_available():
                raise N.NativeLibraryError("no ROCm device visible: the feature2face HIP renderer has no CPU path")
            blob = blob.to(device)
        if blob.dtype != torch.uint8 or blob.numel() != self.packed_bytes():
            # exact size: a blob packed for another frame size has a different layout (and a different size)
            raise ValueError("bad packed blob: %d bytes, this plan packs to %d" % (blob.numel(), self.packed_bytes()))
        self.device = blob.device
        self._blob_dev = blob
        N.check(self.lib.lspf2f_bind_weights(self._h, blob.data_ptr(), blob.numel()))
        self._cand_key = None
        self._ensure_workspace(self.max_batch)

    def _ensure_workspace(self, batch: int) -> None:
        need = int(self.lib.lspf2f_workspace_bytes(self._h, batch))
        if self._ws is None or self._ws.numel() < need:
            self._cand_key = None
            self._ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
            N.check(self.lib.lspf2f_bind_workspace(self._h, self._ws.data_ptr(), self._ws.numel()))

    # ---- hot path -----------------------------------------------------------------------
    def _check_inputs(self, feat: torch.Tensor, cand: Optional[torch.Tensor]):
        if self._blob_dev is None:
            raise RuntimeError("weights not bound: call load_state_dict() / pack() / bind() first")
        s, cand_nc = self.size, self.input_nc - self.feat_nc
        if feat.dim() != 4 or tuple(feat.shape[1:]) != (self.feat_nc, s, s):
            raise ValueError("feature_map must be [B,%d,%d,%d], got %s" % (self.feat_nc, s, s, tuple(feat.shape)))
        b = feat.shape[0]
        if b < 1 or b > self.max_batch:
            raise ValueError("batch %d outside [1, max_batch=%d]" % (b, self.max_batch))
        for t in (feat, cand):
            if t is None:
                continue
            if t.device != self.device:
                raise ValueError("input on %s, engine on %s" % (t.device, self.device))
            if t.dtype != torch.float32:
                raise TypeError("inputs must be float32")
        if cand_nc:
            if cand is None:
                raise ValueError("cand_image is required for input_nc=%d" % self.input_nc)
            if cand.dim() != 4 or tuple(cand.shape[1:]) != (cand_nc, s, s) or cand.shape[0] not in (1, b):
                raise ValueError("cand_image must be [1 or B,%d,%d,%d], got %s" % (cand_nc, s, s, tuple(cand.shape)))
        return b

    def set_candidates(self, cand: Optional[torch.Tensor]) -> None:
        """Pre-compute the candidate stack's contribution to the first conv (12 of its 13 input
        channels are constant per person -- demo.py:89-95).  ``forward(feat, cand)`` with the SAME,
        unmodified tensor then skips that work; ``None`` clears the cache."""
        if cand is None:
            N.check(self.lib.lspf2f_set_candidates(self._h, None, None))
            self._cand_key = None
            return
        s, cand_nc = self.size, self.input_nc - self.feat_nc
        if tuple(cand.shape) != (1, cand_nc, s, s) or cand.dtype != torch.float32 or not cand.is_contiguous():
            raise ValueError("set_candidates wants a contiguous float32 [1,%d,%d,%d] tensor" % (cand_nc, s, s))
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):       # launches (and graph capture) must happen with the engine's device current
            N.check(self.lib.lspf2f_set_candidates(self._h, cand.data_ptr(), ctypes.c_void_p(stream)))
        self._cand_key = (id(cand), cand.data_ptr(), cand._version)
        self._cand_ref = cand       # keep it alive: the key contains its id()

    def _cand_arg(self, cand: Optional[torch.Tensor]):
        """(pointer, batch) to hand to the library: NULL when the cached contribution applies."""
        if cand is None:
            return None, 0
        if self.auto_cand_cache and cand.shape[0] == 1 and cand.is_contiguous():
            key = (id(cand), cand.data_ptr(), cand._version)
            if key != self._cand_key:
                self.set_candidates(cand)
            return None, 1
        if self._cand_key is not None and (id(cand), cand.data_ptr(), cand._version) == self._cand_key:
            return None, 1
        return cand.data_ptr(), cand.shape[0]

    def _check_out(self, out: torch.Tensor, shape, dtype, what: str) -> None:
        """a caller-owned output tensor goes to the library as a bare pointer: shape, dtype, device and layout are checked HERE (a short or strided buffer would be an
        out-of-bounds device write) -- the same rule SmallUnetEngine.render applies to its `out`"""
        if not isinstance(out, torch.Tensor) or tuple(out.shape) != tuple(shape) or out.dtype != dtype or out.device != self.device or not out.is_contiguous():
            raise ValueError("%s must be a contiguous %s tensor of shape %s on %s (got %s %s on %s, contiguous=%s)" % (
                what, dtype, tuple(shape), self.device, getattr(out, "dtype", type(out)), tuple(getattr(out, "shape", ())), getattr(out, "device", "?"),
                getattr(out, "is_contiguous", lambda: "?")()))

    def forward(self, feat: torch.Tensor, cand: Optional[torch.Tensor], out: Optional[torch.Tensor] = None) -> torch.Tensor:
        b = self._check_inputs(feat, cand)
        feat = feat.contiguous()
        cand = cand.contiguous() if cand is not None else None
        if out is None:
            out = torch.empty((b, self.output_nc, self.size, self.size), dtype=torch.float32, device=self.device)
        else:
            self._check_out(out, (b, self.output_nc, self.size, self.size), torch.float32, "out")
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            cptr, cb = self._cand_arg(cand)
            N.check(self.lib.lspf2f_forward(self._h, feat.data_ptr(), cptr, cb, out.data_ptr(), b, ctypes.c_void_p(stream)))
        return out

    def forward_image(self, feat: torch.Tensor, cand: Optional[torch.Tensor], out_u8: Optional[torch.Tensor] = None,
                      also_float: bool = False):
        """Forward + the reference's util.tensor2im (util/util.py:19-42) fused into the last kernel:
        returns uint8 frames [B,H,W,3] (HWC, what demo.py:268 hands to the JPEG writer), and the
        fp32 NCHW tensor as well when ``also_float``."""
        b = self._check_inputs(feat, cand)
        feat = feat.contiguous()
        cand = cand.contiguous() if cand is not None else None
        if out_u8 is None:
            out_u8 = torch.empty((b, self.size, self.size, self.output_nc), dtype=torch.uint8, device=self.device)
        else:
            self._check_out(out_u8, (b, self.size, self.size, self.output_nc), torch.uint8, "out_u8")
        out = torch.empty((b, self.output_nc, self.size, self.size), dtype=torch.float32, device=self.device) if also_float else None
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            cptr, cb = self._cand_arg(cand)
            N.check(self.lib.lspf2f_forward_ex(self._h, feat.data_ptr(), cptr, cb,
                                               out.data_ptr() if out is not None else None, out_u8.data_ptr(), b,
                                               ctypes.c_void_p(stream)))
        return (out_u8, out) if also_float else out_u8

    def debug_poison(self, byte: int = 0xFF) -> int:
        """Hazard-test aid (lspf2f_debug_poison): every scratch byte of the bound workspace becomes ``byte`` (0xFF = NaN in every float format); returns how
        many split-K arrival counters the finished forwards left non-zero (0 is the only healthy answer).  The cached graphs stay: the next forward replays
        on the poisoned workspace and must produce the same bits."""
        if self._ws is None:
            raise RuntimeError("no workspace bound")
        n = ctypes.c_uint32(0)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            N.check(self.lib.lspf2f_debug_poison(self._h, int(byte) & 0xFF, ctypes.c_void_p(stream), ctypes.byref(n)))
        return int(n.value)

    def forward_timed(self, feat, cand, out=None):
        b = self._check_inputs(feat, cand)
        if out is None:
            out = torch.empty((b, self.output_nc, self.size, self.size), dtype=torch.float32, device=self.device)
        n = self.lib.lspf2f_num_layers(self._h)
        ms = (ctypes.c_float * n)()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            N.check(self.lib.lspf2f_forward_timed(self._h, feat.data_ptr(), cand.data_ptr() if cand is not None else None,
                                                  cand.shape[0] if cand is not None else 0, out.data_ptr(), b,
                                                  ctypes.c_void_p(stream), ms))
        return out, [float(x) for x in ms]

    def subset_timed(self, feat, cand, part, out=None, reps: int = 10) -> float:
        """milliseconds per replay of a graph holding only the selected layers' launches (part[i]: 0 skip, 1 main kernels, 2 split-K
        reduce only, 3 all) -- no host gaps; run forward() first"""
        b = self._check_inputs(feat, cand)
        if out is None:
            out = torch.empty((b, self.output_nc, self.size, self.size), dtype=torch.float32, device=self.device)
        n = self.lib.lspf2f_num_layers(self._h)
        if len(part) != n:
            raise ValueError("part needs one entry per layer (%d)" % n)
        arr = (ctypes.c_int * n)(*[int(x) for x in part])
        ms, cnt = ctypes.c_float(), ctypes.c_int()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            N.check(self.lib.lspf2f_subset_timed(self._h, feat.data_ptr(), cand.data_ptr() if cand is not None else None,
                                                 cand.shape[0] if cand is not None else 0, out.data_ptr(), b,
                                                 ctypes.c_void_p(stream), arr, reps, ctypes.byref(ms), ctypes.byref(cnt)))
        return float(ms.value)

    # ---- introspection -------------------------------------------------------------------
    def layers(self, batch: Optional[int] = None) -> List[dict]:
        if batch is not None:
            N.check(self.lib.lspf2f_plan_batch(self._h, batch))
        info = N.LayerInfo()
        out = []
        for i in range(self.lib.lspf2f_num_layers(self._h)):
            N.check(self.lib.lspf2f_layer_info_get(self._h, i, ctypes.byref(info)))
            d = {f: getattr(info, f) for f, _ in N.LayerInfo._fields_}
            d["name"] = d["name"].decode()
            d["kernel"] = d["kernel"].decode()
            out.append(d)
        return out

    def form_offset(self, layer: int, form: str) -> int:
        """byte offset of a layer's weights in the packed blob in the given form (N.FORM_IDS), -1 when this handle's blob does not carry it"""
        return int(self.lib.lspf2f_layer_form_offset(self._h, layer, N.FORM_IDS[form]))

    def workspace_bytes(self, batch: int) -> int:
        return int(self.lib.lspf2f_workspace_bytes(self._h, batch))

    def intermediate(self, name: str, batch: int) -> torch.Tensor:
        """NHWC view [B,H,W,C] of a layer output inside the workspace (meaningful after a
        forward of that batch on an engine built with keep_intermediates=True)."""
        for l in self.layers(batch):
            if l["name"] == name and l["out_offset"] >= 0:
                n = batch * l["h_out"] * l["h_out"] * l["cout"]
                tdt, eb = {"bf16": (torch.bfloat16, 2), "f16": (torch.float16, 2)}.get(self.dtype, (torch.float32, 4))
                raw = self._ws[l["out_offset"]: l["out_offset"] + eb * n]
                return raw.view(tdt).view(batch, l["h_out"], l["h_out"], l["cout"])
        raise KeyError(name)
