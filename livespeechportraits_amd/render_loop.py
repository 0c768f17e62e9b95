"""Batched render loop -- the demo.py:260-272 frame loop restated for throughput.

The reference renders one frame per ``inference()`` call (batch 1) and converts each result on the CPU
(``util.tensor2im``, demo.py:268).  The generator is stateless across frames, so the same loop can feed B
feature maps per call, share the constant candidate stack (cand batch 1), take uint8 HWC frames straight
from the last kernel and overlap the D2H copy of batch i with the rendering of batch i+1.
"""
from __future__ import annotations

from typing import Callable, Iterable, Iterator, List, Optional

import numpy as np
import torch


def batched(it: Iterable, n: int) -> Iterator[List]:
    buf: List = []
    for x in it:
        buf.append(x)
        if len(buf) == n:
            yield buf
            buf = []
    if buf:
        yield buf


def render_frames(model, feature_maps: Iterable[torch.Tensor], cand_image: torch.Tensor, batch: int = 8,
                  device: Optional[torch.device] = None,
                  on_frame: Optional[Callable[[int, np.ndarray], None]] = None, streams: int = 1) -> List[np.ndarray]:
    """``feature_maps`` yields [1,H,W] (or [C,H,W]) CPU/GPU tensors as
    ``facedataset.dataset.get_data_test_mode`` does (demo.py:262); ``cand_image`` is demo.py's
    ``img_candidates`` ([1,12,H,W], already on the device).  Returns (or streams to ``on_frame``) uint8 HWC
    frames, i.e. exactly what ``util.tensor2im(pred_fake[0])`` produced per frame in the reference loop.
    ``model`` is a Feature2FaceModel (anything with ``inference_image``).
    ``streams`` > 1: that many batches in flight at once, each on its own HIP stream and its own handle on the same packed weights -- one batch's kernel tails and boundaries are
    filled by the next one's work (+5 % at 8 fp32 frames, +16 % on the 16-bit plans; same frames, bit for bit).  Ignored where the model cannot give a second handle."""
    device = device or cand_image.device
    frames: List[np.ndarray] = []
    pending: List = []                  # (first index, pinned host tensor, event), oldest first
    idx = 0
    nstream = max(1, int(streams)) if device.type == "cuda" and getattr(model, "supports_replicas", lambda: False)() else 1
    lanes = [torch.cuda.Stream(device) for _ in range(nstream)] if nstream > 1 else [None]
    if nstream > 1:
        cur = torch.cuda.current_stream(device)
        for s in lanes:
            s.wait_stream(cur)                                 # cand_image (and whatever produced the maps so far) was enqueued there

    def flush(keep: int):
        while len(pending) > keep:
            i0, host, ev = pending.pop(0)
            ev.synchronize()
            for k in range(host.shape[0]):
                arr = host[k].numpy().copy()
                if on_frame is not None:
                    on_frame(i0 + k, arr)
                else:
                    frames.append(arr)

    for n, chunk in enumerate(batched(feature_maps, batch)):
        lane = lanes[n % nstream]
        ctx = torch.cuda.stream(lane) if lane is not None else _null()
        with ctx:
            maps = torch.stack([m if m.dim() == 3 else m.unsqueeze(0) for m in chunk]).to(device, torch.float32, non_blocking=True)
            if lane is not None:
                u8 = model.inference_image(maps, cand_image, replica=n % nstream)
            else:
                u8 = model.inference_image(maps, cand_image)   # [b,H,W,3] uint8 on the device
            host = torch.empty(u8.shape, dtype=torch.uint8, pin_memory=device.type == "cuda")
            host.copy_(u8, non_blocking=True)
            ev = torch.cuda.Event() if device.type == "cuda" else None
            if ev is not None:
                ev.record()
        if ev is None:
            for k in range(host.shape[0]):
                (on_frame(idx + k, host[k].numpy().copy()) if on_frame else frames.append(host[k].numpy().copy()))
        else:
            pending.append((idx, host, ev))
            flush(nstream)                                      # the batches of the other lanes stay in flight; older ones are certainly done soon
        idx += len(chunk)
    flush(0)
    if nstream > 1:
        for s in lanes:
            torch.cuda.current_stream(device).wait_stream(s)
    return frames


class _null:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def render_frames_from_landmarks(model, landmarks: Iterable, shoulders: Iterable, cand_image: torch.Tensor,
                                 pad=None, load_size: int = 512, batch: int = 8,
                                 on_frame: Optional[Callable[[int, np.ndarray], None]] = None) -> List[np.ndarray]:
    """demo.py:260-272 with the edge map drawn on the device: per frame the loop moves the 73 landmarks and the shoulder
    points (~1.5 KB) instead of a host-rasterised 1 MiB feature map.  ``landmarks`` yields [73, 2] arrays (``pred_landmarks[i]``
    of demo.py:262), ``shoulders`` yields [n, 2] arrays (``pred_shoulders[i]``); ``pad`` as ``facedataset.dataset.image_pad``."""
    from .feature_map import FeatureMapRasteriser
    device = cand_image.device
    rast = None
    maps_buf = None

    def chunks():
        nonlocal rast, maps_buf
        for lm, sh in zip(batched(landmarks, batch), batched(shoulders, batch)):
            lm_a, sh_a = np.stack([np.asarray(x) for x in lm]), np.stack([np.asarray(x) for x in sh])
            if rast is None:
                rast = FeatureMapRasteriser(load_size, sh_a.shape[1], device)
                maps_buf = torch.empty((batch, 1, load_size, load_size), dtype=torch.float32, device=device)
            yield rast.rasterise(lm_a, sh_a, pad, out=maps_buf[:lm_a.shape[0]])

    frames: List[np.ndarray] = []
    pending = None
    idx = 0
    for maps in chunks():
        u8 = model.inference_image(maps, cand_image)
        host = torch.empty(u8.shape, dtype=torch.uint8, pin_memory=True)
        host.copy_(u8, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        if pending is not None:
            i0, h0, e0 = pending
            e0.synchronize()
            for k in range(h0.shape[0]):
                (on_frame(i0 + k, h0[k].numpy().copy()) if on_frame else frames.append(h0[k].numpy().copy()))
        pending = (idx, host, ev)
        idx += u8.shape[0]
    if pending is not None:
        i0, h0, e0 = pending
        e0.synchronize()
        for k in range(h0.shape[0]):
            (on_frame(i0 + k, h0[k].numpy().copy()) if on_frame else frames.append(h0[k].numpy().copy()))
    return frames
