"""Batched render loop -- the demo.py:260-272 frame loop restated for throughput.

The reference renders one frame per ``inference()`` call (batch 1) and converts each result on the CPU
(``util.tensor2im``, demo.py:268).  The generator is stateless across frames, so the same loop can feed B
feature maps per call, share the constant candidate stack (cand batch 1), take uint8 HWC frames straight
from the last kernel and overlap the D2H copy of batch i with the rendering of batch i+1.
"""
from __future__ import annotations

import inspect
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

    The loop owns ``streams + 1`` sets of buffers (a pinned staging tensor for the maps, their device tensor, a pinned tensor for the frames) and cycles through them: the maps of
    batch i + 1 are gathered into pinned memory and uploaded while batch i renders, and nothing is allocated per batch (a pageable ``torch.stack(...).to(device)`` and a fresh pinned
    result tensor per batch kept this loop at 160-290 frames/s on a generator that renders 1019: round 5).
    ``streams`` > 1: that many batches in flight at once, each on its own HIP stream and its own handle on the same packed weights -- one batch's kernel tails and boundaries are
    filled by the next one's work (generator alone: +5 % at 8 fp32 frames, +16 % on the 16-bit plans; same frames, bit for bit).  Ignored where the model cannot give a second handle."""
    device = device or cand_image.device
    frames: List[np.ndarray] = []
    pending: List = []                  # (first index, frames in the batch, buffer set, event), oldest first
    idx = 0
    on_gpu = device.type == "cuda"
    nstream = max(1, int(streams)) if on_gpu and getattr(model, "supports_replicas", lambda: False)() else 1
    lanes = [torch.cuda.Stream(device) for _ in range(nstream)] if nstream > 1 else [None]
    sets: List[dict] = []               # nstream + 1 buffer sets, made at the first batch (shapes come from the data)
    try:
        takes_out = "out" in inspect.signature(model.inference_image).parameters      # (stand-in models of the tests do not)
    except (TypeError, ValueError):
        takes_out = False

    def emit(i0, host, n):
        for k in range(n):
            arr = host[k].numpy().copy()
            if on_frame is not None:
                on_frame(i0 + k, arr)
            else:
                frames.append(arr)

    def flush(keep: int):
        while len(pending) > keep:
            i0, n, bs, ev = pending.pop(0)
            ev.synchronize()
            emit(i0, bs["host"], n)

    for n, chunk in enumerate(batched(feature_maps, batch)):
        chunk = [m if m.dim() == 3 else m.unsqueeze(0) for m in chunk]
        b = len(chunk)
        if not on_gpu:                                          # (stand-in models on the host: tests)
            emit(idx, model.inference_image(torch.stack(chunk).to(device, torch.float32), cand_image), b)
            idx += b
            continue
        flush(nstream)                                          # the batch that used this buffer set nstream + 1 batches ago is out of it
        if not sets:
            shape = (batch,) + tuple(chunk[0].shape)
            for _ in range(nstream + 1):
                sets.append({"stage": torch.empty(shape, dtype=torch.float32, pin_memory=True), "dev": torch.empty(shape, dtype=torch.float32, device=device), "host": None, "u8": None})
        bs = sets[n % (nstream + 1)]
        lane = lanes[n % nstream]
        cpu_rows = [k for k, m in enumerate(chunk) if m.device.type != "cuda"]
        for k in cpu_rows:
            bs["stage"][k].copy_(chunk[k])                      # host memcpy into pinned memory (overlaps the batches in flight)
        if lane is not None:
            lane.wait_stream(torch.cuda.current_stream(device))  # cand_image, and maps that live on the device, were produced there
        with (torch.cuda.stream(lane) if lane is not None else _null()):
            if len(cpu_rows) == b:
                bs["dev"][:b].copy_(bs["stage"][:b], non_blocking=True)
            else:
                for k, m in enumerate(chunk):
                    bs["dev"][k].copy_(bs["stage"][k] if m.device.type != "cuda" else m, non_blocking=True)
            # the frames land in this buffer set's own device tensor: stable pointers -> the handle replays its cached hipGraph (a fresh result tensor per call made it re-capture,
            # ~15 ms per batch).  A model that does not take `out` (stand-ins) allocates its own.
            kw = {"replica": n % nstream} if lane is not None else {}
            if takes_out:
                if bs["u8"] is None:
                    H = chunk[0].shape[-1]
                    bs["u8"] = torch.empty((batch, H, H, 3), dtype=torch.uint8, device=device)
                kw["out"] = bs["u8"][:b]
            u8 = model.inference_image(bs["dev"][:b], cand_image, **kw)
            if bs["host"] is None:
                bs["host"] = torch.empty((batch,) + tuple(u8.shape[1:]), dtype=torch.uint8, pin_memory=True)
            bs["host"][:b].copy_(u8, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        pending.append((idx, b, bs, ev))
        idx += b
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
    hosts: List[Optional[torch.Tensor]] = [None, None]         # two pinned result tensors, alternating (none allocated per batch)

    def drain():
        nonlocal pending
        if pending is not None:
            i0, n0, h0, e0 = pending
            e0.synchronize()
            for k in range(n0):
                (on_frame(i0 + k, h0[k].numpy().copy()) if on_frame else frames.append(h0[k].numpy().copy()))
            pending = None

    for n, maps in enumerate(chunks()):
        u8 = model.inference_image(maps, cand_image)
        if hosts[n & 1] is None:
            hosts[n & 1] = torch.empty((batch,) + tuple(u8.shape[1:]), dtype=torch.uint8, pin_memory=True)
        host = hosts[n & 1]                                     # (its previous user, batch n - 2, was drained before batch n - 1 was issued)
        host[:u8.shape[0]].copy_(u8, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        drain()
        pending = (idx, u8.shape[0], host, ev)
        idx += u8.shape[0]
    drain()
    return frames
