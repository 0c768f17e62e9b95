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
                  on_frame: Optional[Callable[[int, np.ndarray], None]] = None) -> List[np.ndarray]:
    """``feature_maps`` yields [1,H,W] (or [C,H,W]) CPU/GPU tensors as
    ``facedataset.dataset.get_data_test_mode`` does (demo.py:262); ``cand_image`` is demo.py's
    ``img_candidates`` ([1,12,H,W], already on the device).  Returns (or streams to ``on_frame``) uint8 HWC
    frames, i.e. exactly what ``util.tensor2im(pred_fake[0])`` produced per frame in the reference loop.
    ``model`` is a Feature2FaceModel (anything with ``inference_image``)."""
    device = device or cand_image.device
    frames: List[np.ndarray] = []
    pending = None                      # (first index, pinned host tensor, event)
    idx = 0

    def flush():
        nonlocal pending
        if pending is None:
            return
        i0, host, ev = pending
        ev.synchronize()
        for k in range(host.shape[0]):
            arr = host[k].numpy().copy()
            if on_frame is not None:
                on_frame(i0 + k, arr)
            else:
                frames.append(arr)
        pending = None

    for chunk in batched(feature_maps, batch):
        maps = torch.stack([m if m.dim() == 3 else m.unsqueeze(0) for m in chunk]).to(device, torch.float32,
                                                                                    non_blocking=True)
        u8 = model.inference_image(maps, cand_image)            # [b,H,W,3] uint8 on the device
        host = torch.empty(u8.shape, dtype=torch.uint8, pin_memory=device.type == "cuda")
        host.copy_(u8, non_blocking=True)
        ev = torch.cuda.Event() if device.type == "cuda" else None
        if ev is not None:
            ev.record()
        flush()                                                 # previous batch, now certainly done
        if ev is None:
            for k in range(host.shape[0]):
                (on_frame(idx + k, host[k].numpy().copy()) if on_frame else frames.append(host[k].numpy().copy()))
        else:
            pending = (idx, host, ev)
        idx += len(chunk)
    flush()
    return frames
