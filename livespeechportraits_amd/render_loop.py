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
                  on_frame: Optional[Callable[[int, np.ndarray], None]] = None, streams: int = 2) -> List[np.ndarray]:
    """``feature_maps`` yields [1,H,W] (or [C,H,W]) CPU/GPU tensors as
    ``facedataset.dataset.get_data_test_mode`` does (demo.py:262); ``cand_image`` is demo.py's
    ``img_candidates`` ([1,12,H,W], already on the device).  Returns (or streams to ``on_frame``) uint8 HWC
    frames, i.e. exactly what ``util.tensor2im(pred_fake[0])`` produced per frame in the reference loop.
    ``model`` is a Feature2FaceModel (anything with ``inference_image``).

    ``streams`` lanes, each a HIP stream, a handle on the same packed weights (``inference_image(replica=k)``) and its own buffers (pinned staging tensor for the maps, their
    device tensor, the frames' device tensor, a pinned tensor for the frames): batch n runs on lane n % streams; before a lane is reused the host waits for THAT STREAM to drain
    and hands out its frames.  Nothing is allocated per batch, the maps of the next batch are gathered
    into pinned memory while the lanes render, and two batches in flight fill each other's kernel tails (generator alone: +5 % at 8 fp32 frames, +16 % on the 16-bit plans).
    Measured, round 5 (tools/render_loop_profile.py, tools/host_probe.py): the loop of rounds 2-4 ran at 70-290 frames/s on a generator that renders 665-1019 -- because of the
    host-side copy described at the gather below, not because of how it waited or allocated (a hipGraph cache miss costs nothing next to a forward; the late event waits seen in the
    first measurements were not separated from that throttling).
    A model that cannot give a second handle (the `small` U-Net, several gpu_ids, stand-ins) gets one lane on the current stream: enqueue, wait, hand out."""
    device = device or cand_image.device
    frames: List[np.ndarray] = []
    idx = 0
    on_gpu = device.type == "cuda"
    nlane = max(1, int(streams)) if on_gpu and getattr(model, "supports_replicas", lambda: False)() else 1
    try:
        takes_out = "out" in inspect.signature(model.inference_image).parameters      # (stand-in models of the tests do not)
    except (TypeError, ValueError):
        takes_out = False
    lanes: List[dict] = []              # made at the first batch (shapes come from the data)

    def emit(i0, host, n):
        for k in range(n):
            arr = host[k].numpy().copy()
            if on_frame is not None:
                on_frame(i0 + k, arr)
            else:
                frames.append(arr)

    def drain(lane):
        if lane["busy"] is not None:
            i0, n = lane["busy"]
            (lane["stream"] if lane["stream"] is not None else torch.cuda.current_stream(device)).synchronize()
            emit(i0, lane["host"], n)
            lane["busy"] = None

    for n, chunk in enumerate(batched(feature_maps, batch)):
        chunk = [m if m.dim() == 3 else m.unsqueeze(0) for m in chunk]
        b = len(chunk)
        if not on_gpu:                                          # (stand-in models on the host: tests)
            emit(idx, model.inference_image(torch.stack(chunk).to(device, torch.float32), cand_image), b)
            idx += b
            continue
        if not lanes:
            shape = (batch,) + tuple(chunk[0].shape)
            cur = torch.cuda.current_stream(device)
            for k in range(nlane):
                st = torch.cuda.Stream(device) if nlane > 1 else None
                if st is not None:
                    st.wait_stream(cur)                          # cand_image was produced there
                lanes.append({"stream": st, "busy": None, "host": None, "u8": None,
                              "stage": torch.empty(shape, dtype=torch.float32, pin_memory=True), "dev": torch.empty(shape, dtype=torch.float32, device=device)})
                lanes[-1]["stage_np"] = lanes[-1]["stage"].numpy()
        lane = lanes[n % nlane]
        drain(lane)                                             # its previous batch: frames handed out, buffers free
        cpu_rows = [k for k, m in enumerate(chunk) if m.device.type != "cuda"]
        for k in cpu_rows:
            # host memcpy into pinned memory (the other lanes keep rendering).  Through numpy, i.e. on THIS thread: torch's CPU copy_ fans a 1-MiB tensor out over its
            # intra-op pool (128 threads on the GPU boxes, under a 16-CPU container quota), whose spinning workers get the whole process throttled for milliseconds at a time
            # -- measured as 4-6 ms per MiB "copied" and 8-13 ms waits for a 1.5-ms forward (tools/render_loop_profile.py, tools/host_probe.py)
            m = chunk[k]
            if m.dtype == torch.float32 and m.is_contiguous():
                np.copyto(lane["stage_np"][k], m.numpy())
            else:
                lane["stage"][k].copy_(m)
        if lane["stream"] is not None and len(cpu_rows) < b:
            lane["stream"].wait_stream(torch.cuda.current_stream(device))      # maps that live on the device were produced there
        with (torch.cuda.stream(lane["stream"]) if lane["stream"] is not None else _null()):
            if len(cpu_rows) == b:
                lane["dev"][:b].copy_(lane["stage"][:b], non_blocking=True)
            else:
                for k, m in enumerate(chunk):
                    lane["dev"][k].copy_(lane["stage"][k] if m.device.type != "cuda" else m, non_blocking=True)
                    if m.device.type == "cuda" and lane["stream"] is not None:
                        m.record_stream(lane["stream"])           # read on the lane's stream: the caching allocator must not hand its memory out again before that copy ran
            kw = {"replica": n % nlane} if nlane > 1 else {}
            if takes_out:
                if lane["u8"] is None:
                    H = chunk[0].shape[-1]
                    lane["u8"] = torch.empty((batch, H, H, 3), dtype=torch.uint8, device=device)
                kw["out"] = lane["u8"][:b]
            u8 = model.inference_image(lane["dev"][:b], cand_image, **kw)
            if lane["host"] is None:
                lane["host"] = torch.empty((batch,) + tuple(u8.shape[1:]), dtype=torch.uint8, pin_memory=True)
            lane["host"][:b].copy_(u8, non_blocking=True)
        lane["busy"] = (idx, b)
        idx += b
        if nlane == 1:
            drain(lane)                                         # one lane: nothing to overlap with -- wait now (prompt) rather than behind the next batch's enqueue
    # hand out what is still in flight, oldest first
    for lane in sorted((l for l in lanes if l["busy"] is not None), key=lambda l: l["busy"][0]):
        drain(lane)
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
    idx = 0
    host = None                                                 # one pinned result tensor, reused
    for maps in chunks():
        u8 = model.inference_image(maps, cand_image)
        if host is None:
            host = torch.empty((batch,) + tuple(u8.shape[1:]), dtype=torch.uint8, pin_memory=True)
        host[:u8.shape[0]].copy_(u8, non_blocking=True)
        # one batch at a time: the rasteriser's output tensor is reused, and the wait is on the stream's own tail
        torch.cuda.current_stream(device).synchronize()
        for k in range(u8.shape[0]):
            (on_frame(idx + k, host[k].numpy().copy()) if on_frame else frames.append(host[k].numpy().copy()))
        idx += u8.shape[0]
    return frames
