"""Parameter containers of the feature2face generator + the bridge to the HIP engine.

These nn.Modules exist for ONE reason: to own parameters/buffers under exactly the
names the reference's checkpoints use (``netG.model.model.<idx>...``), so that
``load_state_dict`` / ``state_dict`` / ``torch.save`` round-trip unmodified
``Feature2Face.pkl`` files.  Their ``forward`` never touches torch.nn compute: it
hands the tensors to liblspf2f (hand-written gfx950 kernels).  There is no CPU path.

Key layout restated from (not copied from) the reference:
  models/networks.py:592-640 / 496-544   per-level Sequential order
  models/networks.py:650-675             ResidualBlock.block indices 0,1,3,4
  models/networks.py:347-402             init_weights: conv N(0, 0.02), BN weight N(1, 0.02), bias 0
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.nn as nn

from .engine import Engine
from .topology import VARIANTS, level_channels


class _Slot(nn.Identity):
    """Parameter-free placeholder that keeps nn.Sequential indices aligned with the
    reference (stands where it has nn.ReLU / nn.Upsample)."""


def _conv(cin: int, cout: int, stride: int, bias: bool = False) -> nn.Conv2d:
    return nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=bias)


def _norm_kind(norm_layer) -> str:
    """The reference constructors take ``norm_layer`` (networks.py:459, :555): BatchNorm2d (default) or InstanceNorm2d."""
    if norm_layer in (None, nn.BatchNorm2d, "batch"):
        if norm_layer is None:
            raise NotImplementedError("norm_layer=None (no normalisation) is not supported by the HIP renderer")
        return "batch"
    if norm_layer in (nn.InstanceNorm2d, "instance"):
        return "instance"
    raise NotImplementedError("norm_layer must be nn.BatchNorm2d or nn.InstanceNorm2d, got %r" % (norm_layer,))


def _norm(kind: str, channels: int) -> nn.Module:
    # InstanceNorm2d(affine=False, track_running_stats=False) owns no tensors: a parameter-free slot keeps the index
    return nn.BatchNorm2d(channels) if kind == "batch" else _Slot()


class ResidualBlock(nn.Module):
    """Container for conv-norm-ReLU-conv-norm (+x, ReLU) -- keys block.{0,1,3,4}; the convs never have a bias."""

    def __init__(self, channels: int, norm: str = "batch"):
        super().__init__()
        self.block = nn.Sequential(_conv(channels, channels, 1), _norm(norm, channels), _Slot(),
                                   _conv(channels, channels, 1), _norm(norm, channels))


class ResUnetSkipConnectionBlock(nn.Module):
    """One nesting level; ``model`` mirrors the reference's Sequential index by index."""

    def __init__(self, depth: int, num_downs: int, nres: int, ngf: int, input_nc: int, output_nc: int, norm: str = "batch"):
        super().__init__()
        outer, innermost = depth == 0, depth == num_downs - 1
        cin, inner, cout = level_channels(depth, ngf, input_nc, output_nc)
        use_bias = norm == "instance"                         # networks.py:494 / :590
        seq = [_conv(cin, inner, 2, use_bias)]
        if not (outer or innermost):
            seq.append(_norm(norm, inner))
        seq.append(_Slot())                                   # ReLU
        seq += [ResidualBlock(inner, norm) for _ in range(nres)]
        if not innermost:
            seq.append(ResUnetSkipConnectionBlock(depth + 1, num_downs, nres, ngf, input_nc, output_nc, norm))
        seq.append(_Slot())                                   # Upsample
        seq.append(_conv(inner if innermost else 2 * inner, cout, 1, use_bias))
        if not outer:
            seq += [_norm(norm, cout), _Slot()]
            seq += [ResidualBlock(cout, norm) for _ in range(nres)]
        self.model = nn.Sequential(*seq)


class Feature2FaceGenerator(nn.Module):
    """Feature2FaceGenerator_{normal,large}: parameters here, arithmetic in liblspf2f."""

    def __init__(self, variant: str, input_nc: int = 13, output_nc: int = 3, num_downs: int = 8,
                 ngf: int = 64, feat_nc: int = 1, norm_layer=nn.BatchNorm2d, dtype: str = "f32"):
        super().__init__()
        self.dtype = dtype                       # storage type of activations / conv weights: 'f32', or 'f16' for the reference's opt.fp16
        self.variant, self.input_nc, self.output_nc = variant, input_nc, output_nc
        self.num_downs, self.ngf, self.feat_nc = num_downs, ngf, feat_nc
        self.norm = _norm_kind(norm_layer)
        self.model = ResUnetSkipConnectionBlock(0, num_downs, VARIANTS[variant], ngf, input_nc, output_nc, self.norm)
        self._engine: Optional[Engine] = None
        self._blob: Optional[torch.Tensor] = None      # packed weights on the device
        self._dirty = True
        self._adopted = False                          # the bound blob arrived packed (adopt_packed): this module's own parameters are NOT its source
        self._blob_version = 0                         # bumped whenever the packed blob is rebuilt (replicas on other devices copy it again)
        self._twins = {}                               # replica index -> (blob version, primary engine, Engine): further handles on the SAME device and blob (render_loop streams)
        self.register_load_state_dict_post_hook(lambda *_: self.mark_dirty())

    # -- weight ingress ------------------------------------------------------------
    def mark_dirty(self):
        self._dirty = True
        self._adopted = False                          # a state dict was loaded (or the weights re-initialised): the module's parameters are the source again

    def _key_prefix(self) -> str:
        return "netG.model"

    def _engine_for(self, size: int, batch: int, device: torch.device) -> Engine:
        e = self._engine
        if self._adopted and e is not None and (e.size != size or e.max_batch < batch or e.device != device):
            # the weights came packed from another rank (distributed.setup_engine); this rank's parameters are init_weights() noise, so a
            # re-pack for another frame size / batch range / device would silently bind garbage
            raise RuntimeError(
                "this generator renders from a packed blob adopted from another rank (adopt_packed): it serves %dx%d frames in batches of <= %d on %s; "
                "asked for %dx%d, batch %d on %s.  Build the adopted engine with the final max_batch (distributed.setup_engine(max_batch=...)) or "
                "load a state dict on this rank." % (e.size, e.size, e.max_batch, e.device, size, size, batch, device))
        if e is None or e.size != size or e.max_batch < batch:
            same_size = e is not None and e.size == size
            mb = max(batch, e.max_batch) if same_size else batch
            e = Engine(self.variant, self.input_nc, self.feat_nc, self.output_nc, self.ngf,
                       self.num_downs, size, mb, norm=self.norm, dtype=self.dtype)
            # The packed layout depends on the frame size (an up-conv switches to the 16-tap sub-pixel form once it writes
            # >= 32x32, plan.cpp) and on the batch range the handle plans for (the blob carries only the weight forms those plans
            # read): the blob is reused when just max_batch grew AND the wider range reads the same forms (same size in bytes).
            if same_size and not self._dirty and self._blob is not None and self._blob.device == device \
                    and self._blob.numel() == e.packed_bytes():
                e.bind(self._blob)
            else:
                self._dirty = True
            e.auto_cand_cache = os.environ.get("LSP_HIP_CAND_CACHE", "1") != "0"
            self._engine = e
        if self._dirty or e.device != device:
            sd = {"netG." + k: v for k, v in self.state_dict().items()}
            e.load_state_dict(sd)
            e.bind(e.pack(), device)
            self._blob = e._blob_dev
            self._dirty = False
            self._blob_version += 1
        return e

    def _twin_engine(self, index: int, primary: Engine) -> Engine:
        """A further handle on the primary's device AND packed blob (no second copy of the weights; its own workspace and hipGraphs): what lets two forwards be in flight at
        once on two streams -- a batch's kernel tails, prologues and boundaries are filled by the other's work (render_loop.render_frames(streams=2); profiles/r05_concurrent_streams.txt)."""
        if index == 0:
            return primary
        t = self._twins.get(index)
        if t is None or t[0] != self._blob_version or t[1] is not primary:
            if t is not None:
                t[2].close()
            e = Engine(self.variant, self.input_nc, self.feat_nc, self.output_nc, self.ngf, self.num_downs, primary.size, primary.max_batch, norm=self.norm, dtype=self.dtype)
            e.bind(primary._blob_dev)
            e.auto_cand_cache = primary.auto_cand_cache
            self._twins[index] = t = (self._blob_version, primary, e)
        return t[2]

    def adopt_packed(self, engine: Engine) -> None:
        """Use an engine whose weights were bound elsewhere (multi-GPU: the blob arrived by
        RCCL broadcast, this rank never saw a state dict)."""
        self._engine, self._blob, self._dirty = engine, engine._blob_dev, False
        self._adopted = True
        self._blob_version += 1

    # -- forward ---------------------------------------------------------------------
    def render(self, feat: torch.Tensor, cand: Optional[torch.Tensor]) -> torch.Tensor:
        if feat.device.type != "cuda":
            raise RuntimeError(
                "the feature2face HIP renderer needs ROCm tensors (got %s); there is no CPU fallback -- "
                "the reference's own CPU path is models/networks.py run under PyTorch" % feat.device)
        e = self._engine_for(feat.shape[-1], feat.shape[0], feat.device)
        out = e.forward(feat.float(), cand.float() if cand is not None else None)
        # under autocast the reference's generator returns a float16 tensor (tanh of a half tensor, feature2face_G.py:28-30)
        return out.half() if self.dtype == "f16" else out

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x = cat([feature_map, cand_image], 1) as the reference's G receives it."""
        feat = x[:, :self.feat_nc].contiguous()
        cand = x[:, self.feat_nc:].contiguous() if self.input_nc > self.feat_nc else None
        return self.render(feat, cand)


def Feature2FaceGenerator_normal(input_nc=13, output_nc=3, num_downs=8, ngf=64, norm_layer=nn.BatchNorm2d):
    """networks.py:458-483; ``norm_layer=nn.InstanceNorm2d`` selects the run-time normalisation variant"""
    return Feature2FaceGenerator("normal", input_nc, output_nc, num_downs, ngf, norm_layer=norm_layer)


def Feature2FaceGenerator_large(input_nc=13, output_nc=3, num_downs=8, ngf=64, norm_layer=nn.BatchNorm2d):
    """networks.py:554-579"""
    return Feature2FaceGenerator("large", input_nc, output_nc, num_downs, ngf, norm_layer=norm_layer)


def init_weights(net: nn.Module, init_type: str = "normal", init_gain: float = 0.02) -> None:
    """Same distributions as the reference's default ('normal') initialisation."""
    if init_type != "normal":
        raise NotImplementedError("only init_type='normal' (the one feature2face_model.py:27 uses)")
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, (nn.Conv2d, nn.Conv1d, nn.Linear)):   # reference: class name contains 'Conv' or 'Linear'
                m.weight.normal_(0.0, init_gain)
                if m.bias is not None:
                    m.bias.zero_()
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.normal_(1.0, init_gain)
                m.bias.zero_()
            if hasattr(m, "mark_dirty"):
                m.mark_dirty()


class SingleDeviceParallel(nn.Module):
    """Stands where the reference wraps G in nn.DataParallel (networks.py:400): same
    ``.module`` attribute, same 'module.' key prefix in state dicts, no replicate/scatter --
    multi-GPU here is one process per GPU (livespeechportraits_amd/distributed.py)."""

    def __init__(self, module: nn.Module):
        super().__init__()
        self.module = module

    def forward(self, *a, **kw):
        return self.module(*a, **kw)


class MultiDeviceParallel(nn.Module):
    """``nn.DataParallel(net, gpu_ids)`` of the reference (models/networks.py:392-401) with more than one id, in ONE process.

    DataParallel replicates the module and broadcasts all parameters (487 MB for 'large') on EVERY forward.  Here every listed device
    holds a replica of the PACKED blob -- one peer copy per device whenever the weights are (re)packed, never per frame -- and its own
    engine (one liblspf2f handle per device and stream, include/lspf2f.h).  ``render`` slices the batch over the devices with
    ``distributed.shard_range`` (contiguous, balanced: what DataParallel's scatter does), enqueues every slice on its device's stream from
    this one host thread (the launches are asynchronous, so the devices run concurrently) and gathers the frames on ``gpu_ids[0]``,
    DataParallel's output device.  No collective, no per-frame weight traffic.  Same ``.module`` attribute and 'module.' key prefix.
    The multi-process route (distributed.py, one rank per GPU over RCCL) remains the one bench.py scales with."""

    def __init__(self, module: nn.Module, device_ids):
        super().__init__()
        self.module = module
        self.device_ids = [int(d) for d in device_ids]
        if len(self.device_ids) < 1:
            raise ValueError("MultiDeviceParallel needs at least one device id")
        self._replicas = {}        # slot -> (key, Engine): engines of slots 1.. (slot 0 is the module's own)
        self._cands = {}           # slot -> (key, tensor): the shared candidate stack on that device

    # -- plumbing kept overridable for the CPU bookkeeping test (fake engines on fake devices) ------------------------------------
    def _device(self, slot: int) -> torch.device:
        return torch.device("cuda", self.device_ids[slot])

    def _generator(self):
        return getattr(self.module, "netG", self.module)

    def _primary_engine(self, g, size: int, batch: int, device: torch.device):
        return g._engine_for(size, batch, device)

    def _make_replica(self, g, primary, device: torch.device):
        # same configuration AND same max_batch as the primary: the blob layout depends on the batch range the handle plans for
        e = Engine(g.variant, g.input_nc, g.feat_nc, g.output_nc, g.ngf, g.num_downs, primary.size, primary.max_batch, norm=g.norm, dtype=g.dtype)
        e.bind(primary._blob_dev.to(device))                 # ONE peer copy of the packed blob (no re-pack, no state dict on this device)
        e.auto_cand_cache = primary.auto_cand_cache
        return e

    def _replica(self, slot: int, g, primary):
        key = (id(primary), g._blob_version)
        have = self._replicas.get(slot)
        if have is None or have[0] != key:
            if have is not None and hasattr(have[1], "close"):
                have[1].close()                               # the replaced replica's handle, blob copy and workspace go now, not at interpreter exit
            self._replicas[slot] = have = (key, self._make_replica(g, primary, self._device(slot)))
        return have[1]

    def _shared_cand(self, slot: int, cand: torch.Tensor) -> torch.Tensor:
        """the per-person candidate stack on device `slot`, as fp32: keyed on the CALLER's tensor (identity, storage, version), so a half / double stack
        is converted and peer-copied when it changes, not on every frame"""
        key = (id(cand), cand.data_ptr(), cand._version)
        have = self._cands.get(slot)
        if have is None or have[0] != key:
            self._cands[slot] = have = (key, cand.float().to(self._device(slot)).contiguous(), cand)    # (the source tensor is kept alive: the key holds its id())
        return have[1]

    @staticmethod
    def spans(batch: int, ndev: int):
        """[(slot, lo, hi)] -- the non-empty contiguous slices of a batch over min(ndev, batch) devices"""
        from .distributed import shard_range
        n = max(1, min(ndev, batch))
        return [(r,) + shard_range(batch, r, n) for r in range(n)]

    def render(self, feature_map: torch.Tensor, cand_image: Optional[torch.Tensor]) -> torch.Tensor:
        return self._render(feature_map, cand_image, image=False)

    def render_image(self, feature_map: torch.Tensor, cand_image: Optional[torch.Tensor]) -> torch.Tensor:
        """render() + util.tensor2im fused into every device's last kernel: uint8 [B,H,W,3] frames gathered on gpu_ids[0] (a quarter of the fp32 peer traffic)"""
        return self._render(feature_map, cand_image, image=True)

    def _render(self, feature_map: torch.Tensor, cand_image: Optional[torch.Tensor], image: bool) -> torch.Tensor:
        g = self._generator()
        b = feature_map.shape[0]
        spans = self.spans(b, len(self.device_ids))
        if not isinstance(g, Feature2FaceGenerator):
            # the 'small' U-Net (its own native plan, include/lspunet.h): the module's device does it all.  (The uint8 route of that generator is
            # Feature2FaceModel.inference_image's own branch; it never reaches this class.)
            if image:
                raise RuntimeError("render_image over several devices serves the normal / large generators")
            return self.module.render(feature_map, cand_image) if hasattr(self.module, "render") else g.render(feature_map, cand_image)
        dev0 = self._device(0)
        if feature_map.device != dev0:
            raise RuntimeError("inputs must live on gpu_ids[0] = %s like DataParallel's (got %s)" % (dev0, feature_map.device))
        per = max(hi - lo for _, lo, hi in spans)
        primary = self._primary_engine(g, feature_map.shape[-1], per, dev0)
        outs = []
        for slot, lo, hi in spans:
            e = primary if slot == 0 else self._replica(slot, g, primary)
            f = feature_map[lo:hi].float()
            c = None
            if cand_image is not None:
                if cand_image.shape[0] == 1:
                    c = self._shared_cand(slot, cand_image)                 # constant per person (demo.py:89-95): converted / copied when it changes, not per frame
                else:
                    c = cand_image[lo:hi].float()
                    c = c if slot == 0 else c.to(self._device(slot), non_blocking=True)
            if slot != 0:
                f = f.to(self._device(slot), non_blocking=True)
            run = e.forward_image if image else e.forward
            outs.append(run(f.contiguous(), c.contiguous() if c is not None else None))
        if len(spans) == 1:                                   # one frame (or one device): nothing to gather
            out = outs[0]
        else:
            out = torch.empty((b,) + tuple(outs[0].shape[1:]), dtype=outs[0].dtype, device=dev0)
            for (slot, lo, hi), o in zip(spans, outs):
                out[lo:hi].copy_(o, non_blocking=True)        # peer copies onto the output device; torch orders them behind the producing streams
        return out.half() if (g.dtype == "f16" and not image) else out

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x = cat([feature_map, cand_image], 1) as the reference's G receives it (scattered along the batch like DataParallel does)"""
        g = self._generator()
        if not isinstance(g, Feature2FaceGenerator):
            return self.module(x)
        feat = x[:, :g.feat_nc].contiguous()
        cand = x[:, g.feat_nc:].contiguous() if g.input_nc > g.feat_nc else None
        return self.render(feat, cand)


def init_net(net: nn.Module, init_type="normal", init_gain=0.02, gpu_ids=()):
    """networks.py:380-402 of the reference: initialise, move to gpu_ids[0], wrap.  One id -> SingleDeviceParallel; several ->
    MultiDeviceParallel (the batch is sliced over every listed device, like nn.DataParallel(net, gpu_ids))."""
    init_weights(net, init_type, init_gain)
    if len(gpu_ids) > 0:
        if not torch.cuda.is_available():
            raise RuntimeError("gpu_ids=%r but no ROCm device is visible" % (list(gpu_ids),))
        bad = [g for g in gpu_ids if not 0 <= int(g) < torch.cuda.device_count()]
        if bad:
            raise RuntimeError("gpu_ids=%r: no such device(s) %r (%d visible)" % (list(gpu_ids), bad, torch.cuda.device_count()))
        net = net.to("cuda:%d" % gpu_ids[0])
        net = SingleDeviceParallel(net) if len(gpu_ids) == 1 else MultiDeviceParallel(net, gpu_ids)
    return net
