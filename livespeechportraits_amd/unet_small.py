"""The reference's `size == 'small'` generator (Feature2FaceGenerator_Unet, models/networks.py:680-769; selected at
models/feature2face_G.py:16-17 with a 23-channel input) on the MI355X.

SmallUnetEngine is the thin ctypes host of the NATIVE plan behind include/lspunet.h (csrc/unet.hip): state dict in, packed blob,
caller-owned workspace, one launch per convolution -- the down-convs write leaky_relu / relu copies of their output from their own epilogue, the torch.cat
of feature map and candidates is two base pointers of the input pass -- replayed from a hipGraph.

HostSequencedUnetEngine is the round-3/4 form the native plan replaced, kept as the independent second route the tests compare it with bit for bit: one C
call per launch from Python, out of the library's building blocks,

  Conv2d(k4, s2, p1)           -> lspf2f_unet_prepare (space-to-depth + the in-place LeakyReLU) + lspf2f_conv3x3 on 4x the
                                  channels with the 16 real taps scattered into a 3x3 pattern; 20 of its 36 (tap, channel quarter)
                                  blocks are zero, and k_group = -4 makes the kernel's K cursor walk only the 16 live ones
                                  (pack_down_live; block 0 with its 23 padded input channels keeps the dense pack_down)
  ConvTranspose2d(k4, s2, p1)  -> lspf2f_conv3x3 in sub-pixel form (upsample = 2): 4 output parities x 2x2 taps
  last ConvTranspose + Tanh    -> 3x3 GEMM with N = 4 parities x 3 on the low-res source + lspf2f_pixel_shuffle (tanh)

(graph=True captures those ~40 launches through torch.cuda.graphs; bit-identical and not faster: the device is the limiter, tools/unet_small_time.py.)
All arithmetic is in the HIP kernels (no CPU path).  The mappings are documented in include/lspf2f.h next to lspf2f_unet_prepare."""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _native as N

_KY = {0: (0, 1), 1: (1, 0), 2: (1, 1), 3: (2, 0)}      # 4x4 s2 tap ky -> (3x3 tap row ty, space-to-depth row dy)
_KT = ((3, 1), (2, 0))                                   # transposed conv: ky of output parity py, sub-pixel tap a


def _fold_bn(sd, key):
    g, b = sd[key + ".weight"].astype(np.float64), sd[key + ".bias"].astype(np.float64)
    m, v = sd[key + ".running_mean"].astype(np.float64), sd[key + ".running_var"].astype(np.float64)
    s = g / np.sqrt(v + 1e-5)
    return s.astype(np.float32), (b - m * s).astype(np.float32)


def pack_down(w: np.ndarray, s2d_channels: int) -> np.ndarray:
    """[co][ci][4][4] -> [co][3][3][s2d_channels] for the 3x3 conv on the space-to-depth image."""
    co, ci = w.shape[:2]
    out = np.zeros((co, 3, 3, s2d_channels), np.float32)
    for ky in range(4):
        ty, dy = _KY[ky]
        for kx in range(4):
            tx, dx = _KY[kx]
            out[:, ty, tx, (dy * 2 + dx) * ci:(dy * 2 + dx + 1) * ci] = w[:, :, ky, kx]
    return out


_LIVE = {0: (1,), 1: (0, 1), 2: (0,)}                   # 3x3 tap row (column) -> the space-to-depth sub-rows (sub-columns) that carry a 4x4 tap


def pack_down_live(w: np.ndarray) -> np.ndarray:
    """[co][ci][4][4] -> [co][16 live (tap, quarter) pairs][ci]: the 3x3 / space-to-depth form WITHOUT its 20 zero blocks, in the order the kernel's K cursor
    walks them (tap-major, quarter (dy * 2 + dx) minor) -- lspf2f_conv3x3 with k_group = -4 (include/lspf2f.h)."""
    co, ci = w.shape[:2]
    inv = {v: k for k, v in _KY.items()}                 # (ty, dy) -> ky
    blocks = []
    for ty in range(3):
        for tx in range(3):
            for dy in range(2):
                for dx in range(2):
                    if dy in _LIVE[ty] and dx in _LIVE[tx]:
                        blocks.append(w[:, :, inv[(ty, dy)], inv[(tx, dx)]])
    assert len(blocks) == 16
    return np.ascontiguousarray(np.stack(blocks, axis=1))          # [co][16][ci]


def pack_up(wt: np.ndarray) -> np.ndarray:
    """ConvTranspose2d weight [ci][co][4][4] -> sub-pixel form [4 parities][co][2][2][ci]."""
    ci, co = wt.shape[:2]
    out = np.zeros((4, co, 2, 2, ci), np.float32)
    for py in range(2):
        for px in range(2):
            for a in range(2):
                for b in range(2):
                    out[py * 2 + px, :, a, b, :] = wt[:, :, _KT[py][a], _KT[px][b]].T
    return out


def pack_last(wt: np.ndarray) -> np.ndarray:
    """[ci][co][4][4] -> [4*co][3][3][ci]: the sub-pixel taps as one 3x3 conv on the low-res source (channel par*co + c)."""
    sub = pack_up(wt)
    ci, co = wt.shape[:2]
    out = np.zeros((4 * co, 3, 3, ci), np.float32)
    for py in range(2):
        for px in range(2):
            for a in range(2):
                for b in range(2):
                    out[(py * 2 + px) * co:(py * 2 + px + 1) * co, py + a, px + b, :] = sub[py * 2 + px, :, a, b, :]
    return out


def block_keys(num_downs: int, prefix: str = "model"):
    out, pfx = [], prefix + ".model"
    for depth in range(num_downs):
        if depth == 0:
            out.append((pfx + ".0", None, pfx + ".3", None)); pfx += ".1.model"
        elif depth == num_downs - 1:
            out.append((pfx + ".1", None, pfx + ".3", pfx + ".4"))
        else:
            out.append((pfx + ".1", pfx + ".2", pfx + ".5", pfx + ".6")); pfx += ".3.model"
    return out


class HostSequencedUnetEngine:
    def __init__(self, input_nc: int = 23, output_nc: int = 3, num_downs: int = 8, ngf: int = 64, graph: bool = False, live_taps: bool = True):
        if ngf % 32 or num_downs < 5 or not (1 <= output_nc <= 4):
            raise ValueError("ngf must be a multiple of 32, num_downs >= 5, output_nc <= 4")
        self.lib = N.load()
        self.use_graph = graph
        self.live_taps = live_taps                # down-convs issue only the 16 live (tap, quarter) K blocks of the 36 (round 4); False: the dense form (A-B, tests)
        self._graphs: Dict[tuple, tuple] = {}     # (B, S, out_u8) -> (CUDAGraph, static input, static output, scratch)
        self.input_nc, self.output_nc, self.num_downs, self.ngf = input_nc, output_nc, num_downs, ngf
        self.chans = [ngf * min(2 ** i, 8) for i in range(num_downs)]
        self.s2d0 = (4 * input_nc + 31) // 32 * 32
        self.layers: Optional[List[dict]] = None
        self.device = None
        self._scratch = None

    def load_state_dict(self, sd: Dict[str, np.ndarray], prefix: str = "model", device="cuda:0") -> None:
        self._graphs = {}                         # captured launches point at the previous weights
        sd = {k: (v.detach().float().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v, np.float32)) for k, v in sd.items()}
        keys = block_keys(self.num_downs, prefix)
        need = [k + ".weight" for blk in keys for k in (blk[0], blk[2])]
        missing = [k for k in need if k not in sd]
        if missing:
            raise KeyError("state dict lacks %s" % missing[:3])
        dev = torch.device(device)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        L = []
        for k, (dc, dbn, uc, ubn) in enumerate(keys):
            cin = self.input_nc if k == 0 else self.chans[k - 1]
            s2d = self.s2d0 if k == 0 else 4 * cin
            live = self.live_taps and k > 0 and cin % 32 == 0              # (block 0: 23 input channels, padded to 96 -- its quarters are not K-tile aligned)
            e = {"down_w": up(pack_down_live(sd[dc + ".weight"]) if live else pack_down(sd[dc + ".weight"], s2d)), "down_live": live, "s2d": s2d, "cin": cin, "cout": self.chans[k]}
            e["down_scale"], e["down_shift"] = (up(t) for t in _fold_bn(sd, dbn)) if dbn else (None, None)
            wt = sd[uc + ".weight"]
            if k == 0:
                e["up_w"] = up(pack_last(wt))
                bias = sd.get(uc + ".bias")
                e["up_shift"] = up(np.tile(bias if bias is not None else np.zeros(self.output_nc, np.float32), 4).astype(np.float32))
                e["up_scale"] = up(np.ones(4 * self.output_nc, np.float32))        # the entry point takes scale and shift together
            else:
                e["up_w"] = up(pack_up(wt))
                e["up_scale"], e["up_shift"] = (up(t) for t in _fold_bn(sd, ubn))
            e["up_cout"] = wt.shape[1]
            L.append(e)
        self.layers, self.device = L, dev

    # ---- launches ---------------------------------------------------------------------------------
    def _conv(self, src0, src1, w, scale, shift, out, stride, upsample, relu, k_group=0):
        b, hs, ws, c0 = src0.shape
        c1 = src1.shape[3] if src1 is not None else 0
        cout = out.shape[3]
        sb = self.lib.lspf2f_conv3x3_scratch_bytes(b, hs, ws, c0, c1, cout, stride, upsample, 0, 0, 0, k_group, 0)
        if self._scratch is None or self._scratch.numel() < sb:
            self._scratch = torch.empty(max(sb, 256), dtype=torch.uint8, device=self.device)
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        N.check(self.lib.lspf2f_conv3x3(p(src0), p(src1), p(w), p(scale), p(shift), None, p(out), b, hs, ws, c0, c1, cout,
                                        stride, upsample, int(relu), 0, 0, 0, k_group, 0, p(self._scratch), self._scratch.numel(),
                                        ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def _prepare(self, src, nchw, b, h, w, c, slope, s2d, s2d_c, relu):
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        N.check(self.lib.lspf2f_unet_prepare(p(src), int(nchw), b, h, w, c, ctypes.c_float(slope), p(s2d), s2d_c, p(relu),
                                             ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def forward(self, x: torch.Tensor, out_u8: bool = False) -> torch.Tensor:
        """x [B, input_nc, S, S] fp32 on the device -> [B, output_nc, S, S] fp32 (or uint8 HWC frames)."""
        if self.layers is None:
            raise RuntimeError("HostSequencedUnetEngine.load_state_dict first")
        if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != self.input_nc:
            raise ValueError("x must be a float32 device tensor [B, %d, S, S] (there is no CPU path)" % self.input_nc)
        x = x.contiguous()
        B, _, S, _ = x.shape
        if S % (1 << self.num_downs):
            raise ValueError("frame size must be a multiple of 2**num_downs")
        if not self.use_graph or torch.cuda.is_current_stream_capturing():
            return self._run(x, out_u8)
        key = (B, S, bool(out_u8))
        if key not in self._graphs:
            with torch.cuda.device(self.device):
                xs = x.clone()
                side = torch.cuda.Stream(self.device)
                side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(side):              # warm-up off the default stream: sizes the scratch, sets kernel attributes
                    self._run(xs, out_u8)
                torch.cuda.current_stream(self.device).wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    outs = self._run(xs, out_u8)
            self._graphs[key] = (g, xs, outs, self._scratch)      # the scratch the captured launches point at stays alive with the graph
        g, xs, outs, _ = self._graphs[key]
        xs.copy_(x)
        g.replay()
        return outs.clone()

    def _run(self, x: torch.Tensor, out_u8: bool) -> torch.Tensor:
        B, _, S, _ = x.shape
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=self.device)
        nd, L = self.num_downs, self.layers
        with torch.cuda.device(self.device):
            # ---- down path
            h = S
            y = new(B, h // 2, h // 2, L[0]["s2d"])
            self._prepare(x, True, B, h, h, self.input_nc, 1.0, y, L[0]["s2d"], None)          # no activation in front of block 0
            skips = []
            d = None
            for k in range(nd):
                h //= 2
                inner = k == nd - 1
                d = new(B, h, h, L[k]["cout"])
                self._conv(y, None, L[k]["down_w"], L[k]["down_scale"], L[k]["down_shift"], d, 1, 0, relu=inner, k_group=-4 if L[k]["down_live"] else 0)
                if inner:
                    break                                                                     # stored as relu(d): only the up-conv reads it
                y = new(B, h // 2, h // 2, 4 * L[k]["cout"])
                r = new(B, h, h, L[k]["cout"])
                self._prepare(d, False, B, h, h, L[k]["cout"], 0.2, y, 4 * L[k]["cout"], r)      # lrelu for the next conv, relu for the skip
                skips.append(r)
            # ---- up path
            u = None
            for k in range(nd - 1, 0, -1):
                src0 = d if u is None else skips[k]
                o = new(B, 2 * h, 2 * h, L[k]["up_cout"])
                self._conv(src0, u, L[k]["up_w"], L[k]["up_scale"], L[k]["up_shift"], o, 1, 2, relu=True)   # relu: read only through the parent's uprelu
                u, h = o, 2 * h
            g = new(B, h, h, 4 * self.output_nc)
            self._conv(skips[0], u, L[0]["up_w"], L[0]["up_scale"], L[0]["up_shift"], g, 1, 0, relu=False)
            p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
            if out_u8:
                out = torch.empty((B, 2 * h, 2 * h, self.output_nc), dtype=torch.uint8, device=self.device)
                N.check(self.lib.lspf2f_pixel_shuffle(p(g), B, h, h, self.output_nc, 1, None, p(out),
                                                      ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
            else:
                out = new(B, self.output_nc, 2 * h, 2 * h)
                N.check(self.lib.lspf2f_pixel_shuffle(p(g), B, h, h, self.output_nc, 1, p(out), None,
                                                      ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        return out


class SmallUnetEngine:
    """Host of one native plan per (frame size, feat_nc): include/lspunet.h.  forward(x) takes the concatenated [B, input_nc, S, S] tensor
    (Feature2FaceGenerator_Unet.forward, models/networks.py:694-697); render(feat, cand) takes the two tensors of Feature2FaceModel.inference()
    (models/feature2face_model.py:225-237) and never concatenates them."""

    def __init__(self, input_nc: int = 23, output_nc: int = 3, num_downs: int = 8, ngf: int = 64, max_batch: int = 8, graph: bool = True, tune=None, dtype: str = "f32"):
        if ngf % 32 or num_downs < 5 or not (1 <= output_nc <= 4):
            raise ValueError("ngf must be a multiple of 32, num_downs >= 5, output_nc <= 4")
        if dtype not in ("f32", "f16") or (dtype == "f16" and ngf % 64):
            raise ValueError("dtype must be 'f32' or 'f16' (fp16 storage -- the reference's opt.fp16 -- needs ngf % 64 == 0)")
        self.dtype = dtype
        self.lib = N.load()
        self.input_nc, self.output_nc, self.num_downs, self.ngf = input_nc, output_nc, num_downs, ngf
        self.max_batch, self.use_graph = max_batch, graph
        self.tune = tune if isinstance(tune, (str, type(None))) else ",".join("%s=%d" % kv for kv in sorted(tune.items()))
        self.sd: Optional[Dict[str, np.ndarray]] = None
        self.device = None
        self._plans: Dict[tuple, dict] = {}        # (S, feat_nc) -> {handle, blob, ws, max_batch}

    def load_state_dict(self, sd: Dict[str, np.ndarray], prefix: str = "model", device="cuda:0") -> None:
        self.close()
        cut = len(prefix) - len("model")            # keys reach the library relative to netG: "model.model.0.weight", ...
        self.sd = {k[cut:]: np.ascontiguousarray(v.detach().float().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v, np.float32))
                   for k, v in sd.items() if k.startswith(prefix + ".") and not k.endswith("num_batches_tracked")}
        self.device = torch.device(device)

    def close(self) -> None:
        for p in self._plans.values():
            self.lib.lspunet_destroy(p["handle"])
        self._plans = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _plan(self, S: int, feat_nc: int, batch: int) -> dict:
        if self.sd is None:
            raise RuntimeError("SmallUnetEngine.load_state_dict first")
        key = (S, feat_nc)
        p = self._plans.get(key)
        if p is not None and batch <= p["max_batch"]:
            return p
        if p is not None:
            self.lib.lspunet_destroy(p["handle"])
            del self._plans[key]
        mb = max(batch, self.max_batch)
        cfg = N.UnetConfig(N.UNET_ABI_VERSION, self.input_nc, feat_nc, self.output_nc, self.ngf, self.num_downs, S, mb, N.DTYPE_IDS[self.dtype],
                           0 if self.use_graph else N.UNET_FLAG_NO_GRAPH)
        h = ctypes.c_void_p()
        N.check_unet(self.lib.lspunet_create(ctypes.byref(cfg), self.tune.encode() if self.tune else None, ctypes.byref(h)))
        try:
            name, dims, nd = ctypes.c_char_p(), (ctypes.c_int64 * 4)(), ctypes.c_int()
            for i in range(self.lib.lspunet_num_tensors(h)):
                N.check_unet(self.lib.lspunet_tensor_info(h, i, ctypes.byref(name), ctypes.byref(dims), ctypes.byref(nd)))
                k = name.value.decode()
                if k not in self.sd:
                    raise KeyError("state dict lacks %s" % k)
                a = self.sd[k]
                if list(a.shape) != [dims[j] for j in range(nd.value)]:
                    raise ValueError("%s: shape %s, expected %s" % (k, list(a.shape), [dims[j] for j in range(nd.value)]))
                N.check_unet(self.lib.lspunet_set_tensor(h, name.value, a.ctypes.data_as(ctypes.c_void_p), a.size))
            nb = self.lib.lspunet_packed_bytes(h)
            host = torch.empty(nb, dtype=torch.uint8)
            N.check_unet(self.lib.lspunet_pack_weights(h, ctypes.c_void_p(host.data_ptr()), nb))
            blob = host.to(self.device)
            N.check_unet(self.lib.lspunet_bind_weights(h, ctypes.c_void_p(blob.data_ptr()), nb))
            wb = self.lib.lspunet_workspace_bytes(h, mb)
            ws = torch.empty(wb, dtype=torch.uint8, device=self.device)
            N.check_unet(self.lib.lspunet_bind_workspace(h, ctypes.c_void_p(ws.data_ptr()), wb))
        except Exception:
            self.lib.lspunet_destroy(h)
            raise
        p = {"handle": h, "blob": blob, "ws": ws, "max_batch": mb}
        self._plans[key] = p
        return p

    def _check(self, t: torch.Tensor, c: int, what: str) -> torch.Tensor:
        if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 4 or t.shape[1] != c or t.shape[2] != t.shape[3]:
            raise ValueError("%s must be a float32 device tensor [B, %d, S, S] (there is no CPU path)" % (what, c))
        return t.contiguous()

    def render(self, feat: torch.Tensor, cand: Optional[torch.Tensor], out_u8: bool = False, timed: Optional[list] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """feat [B, feat_nc, S, S], cand [1 | B, input_nc - feat_nc, S, S] (None: feat carries every channel) -> [B, output_nc, S, S] fp32 or uint8 HWC frames.
        `out`: a caller-owned result tensor (the library caches one hipGraph per set of pointers: a render loop that reuses its buffers replays, one that hands in fresh
        ones re-captures -- eight graphs are kept)"""
        feat_nc = self.input_nc if cand is None else self.input_nc - cand.shape[1]
        feat = self._check(feat, feat_nc, "feature_map")
        B, _, S, _ = feat.shape
        if cand is not None:
            cand = self._check(cand, self.input_nc - feat_nc, "cand_image")
            if cand.shape[2] != S or cand.shape[0] not in (1, B):
                raise ValueError("cand_image must be [1 | B, %d, %d, %d]" % (self.input_nc - feat_nc, S, S))
        if S % (1 << self.num_downs):
            raise ValueError("frame size must be a multiple of 2**num_downs")
        p = self._plan(S, feat_nc, B)
        with torch.cuda.device(self.device):
            want = ((B, S, S, self.output_nc), torch.uint8) if out_u8 else ((B, self.output_nc, S, S), torch.float32)
            if out is None:
                out = torch.empty(want[0], dtype=want[1], device=self.device)
            elif tuple(out.shape) != want[0] or out.dtype != want[1] or out.device != self.device or not out.is_contiguous():
                raise ValueError("out must be a contiguous %s tensor of shape %s on %s" % (want[1], want[0], self.device))
            ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
            args = (p["handle"], ptr(feat), ptr(cand), 0 if cand is None else cand.shape[0], None if out_u8 else ptr(out), ptr(out) if out_u8 else None, B,
                    ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
            if timed is None:
                N.check_unet(self.lib.lspunet_forward(*args))
            else:
                ms = (ctypes.c_float * self.lib.lspunet_num_launches(p["handle"], B))()
                N.check_unet(self.lib.lspunet_forward_timed(*args, ms))
                timed[:] = list(ms)
        return out

    def forward(self, x: torch.Tensor, out_u8: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x [B, input_nc, S, S] fp32 on the device -> [B, output_nc, S, S] fp32 (or uint8 HWC frames)."""
        return self.render(x, None, out_u8, out=out)

    def launches(self, S: int, batch: int, feat_nc: Optional[int] = None) -> List[dict]:
        p = self._plan(S, self.input_nc if feat_nc is None else feat_nc, batch)
        out, name, kern, tm, tn, sk = [], ctypes.c_char_p(), ctypes.c_char_p(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        for i in range(self.lib.lspunet_num_launches(p["handle"], batch)):
            N.check_unet(self.lib.lspunet_launch_info(p["handle"], batch, i, ctypes.byref(name), ctypes.byref(kern), ctypes.byref(tm), ctypes.byref(tn), ctypes.byref(sk)))
            out.append({"name": name.value.decode(), "kernel": kern.value.decode(), "tile": (tm.value, tn.value), "split_k": sk.value})
        return out


# ---- parameter container with the reference's keys ---------------------------------------------------
class _Slot(nn.Identity):
    """occupies an index of the reference's nn.Sequential that holds no parameters (activations, Tanh)"""


class UnetSkipConnectionBlock(nn.Module):
    def __init__(self, outer_nc, inner_nc, input_nc=None, submodule=None, outermost=False, innermost=False):
        super().__init__()
        input_nc = outer_nc if input_nc is None else input_nc
        down = nn.Conv2d(input_nc, inner_nc, 4, 2, 1, bias=False)
        if outermost:
            mods = [down, submodule, _Slot(), nn.ConvTranspose2d(inner_nc * 2, outer_nc, 4, 2, 1), _Slot()]
        elif innermost:
            mods = [_Slot(), down, _Slot(), nn.ConvTranspose2d(inner_nc, outer_nc, 4, 2, 1, bias=False), nn.BatchNorm2d(outer_nc)]
        else:
            mods = [_Slot(), down, nn.BatchNorm2d(inner_nc), submodule, _Slot(),
                    nn.ConvTranspose2d(inner_nc * 2, outer_nc, 4, 2, 1, bias=False), nn.BatchNorm2d(outer_nc)]
        self.model = nn.Sequential(*mods)


class Feature2FaceGenerator_Unet(nn.Module):
    """Weights container + device evaluation; same constructor and state-dict keys as the reference class."""

    def __init__(self, input_nc=4, output_nc=3, num_downs=8, ngf=64, dtype="f32"):
        super().__init__()
        self.dtype = dtype                        # 'f32', or 'f16' for the reference's opt.fp16 (autocast around netG, models/feature2face_G.py:28-30)
        blk = UnetSkipConnectionBlock(ngf * 8, ngf * 8, innermost=True)
        for _ in range(num_downs - 5):
            blk = UnetSkipConnectionBlock(ngf * 8, ngf * 8, submodule=blk)
        blk = UnetSkipConnectionBlock(ngf * 4, ngf * 8, submodule=blk)
        blk = UnetSkipConnectionBlock(ngf * 2, ngf * 4, submodule=blk)
        blk = UnetSkipConnectionBlock(ngf, ngf * 2, submodule=blk)
        self.model = UnetSkipConnectionBlock(output_nc, ngf, input_nc=input_nc, submodule=blk, outermost=True)
        self.input_nc, self.output_nc, self.num_downs, self.ngf = input_nc, output_nc, num_downs, ngf
        self._engine = None
        self._version = None

    def mark_dirty(self):
        self._engine = None

    def _get_engine(self, device) -> SmallUnetEngine:
        version = tuple(p._version for p in self.parameters()) + tuple(b._version for b in self.buffers())
        if self._engine is None or self._version != version or self._engine.device != device:
            if self._engine is not None:
                self._engine.close()
            e = SmallUnetEngine(self.input_nc, self.output_nc, self.num_downs, self.ngf, dtype=self.dtype)
            e.load_state_dict({k: v for k, v in self.state_dict().items() if not k.endswith("num_batches_tracked")}, "model", device)
            self._engine, self._version = e, version
        return self._engine

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.device.type != "cuda":
            raise RuntimeError("the feature2face HIP renderer needs ROCm tensors; there is no CPU fallback")
        out = self._get_engine(x.device).forward(x.float())
        return out.half() if self.dtype == "f16" else out      # under autocast the reference's generator returns a float16 tensor

    def render(self, feat: torch.Tensor, cand: Optional[torch.Tensor], out_u8: bool = False) -> torch.Tensor:
        """feature2face_model.py:229-231: cat([feature_map, cand_image], 1) unless cand_image is None -- here two base pointers of the input pass; a shared
        candidate stack (batch 1) is broadcast over a batch of feature maps"""
        if feat.device.type != "cuda":
            raise RuntimeError("the feature2face HIP renderer needs ROCm tensors; there is no CPU fallback")
        out = self._get_engine(feat.device).render(feat.float(), None if cand is None else cand.float(), out_u8)
        return out.half() if (self.dtype == "f16" and not out_u8) else out
