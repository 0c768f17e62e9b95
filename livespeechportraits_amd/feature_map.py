"""The landmark edge map of the render loop on the device -- the part of the reference's ``FaceDataset`` that demo.py's frame
loop uses (datasets/face_dataset.py:34-42 part_list, :276-323 get_data_test_mode / get_feature_image / draw_face_feature_maps /
draw_shoulder_points; called at demo.py:262-265).

The reference rasterises ~106 thick line segments per frame on the host with ``cv2.line(img, pt1, pt2, 255, 2)``, divides by
255 and copies 1 MiB to the GPU.  Here the ~1 KB of landmark coordinates goes to the device instead and one kernel
(``lspraster_edge_maps``, include/lspraster.h) writes the ``[B, 1, H, W]`` {0, 1} float tensor where ``inference()`` reads it.
There is no CPU path.  Parity with cv2 is unpinned (no OpenCV in the build image): the kernel is bit-exact to
oracle/raster_oracle.c, a restatement of OpenCV 4.4.0's published cv::line algorithm.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence

import numpy as np
import torch

from . import _native as N

# face_dataset.py:34-42: contour, right / left eyebrow, nose, right / left eye, mouth, tongue -- polylines over 73 landmarks
PART_LIST = [[list(range(0, 15))],
             [[15, 16, 17, 18, 18, 19, 20, 15]],
             [[21, 22, 23, 24, 24, 25, 26, 21]],
             [list(range(35, 44))],
             [[27, 65, 28, 68, 29], [29, 67, 30, 66, 27]],
             [[33, 69, 32, 72, 31], [31, 71, 34, 70, 33]],
             [list(range(46, 53)), [52, 53, 54, 55, 56, 57, 46]],
             [[46, 63, 62, 61, 52], [52, 60, 59, 58, 46]]]
N_LANDMARKS = 73


def edge_list(n_shoulder: int = 0, n_landmarks: int = N_LANDMARKS) -> np.ndarray:
    """int32 [nseg, 2] index pairs into cat([landmarks, shoulders]): the face polylines (face_dataset.py:314-320) followed
    by the two shoulder chains of n_shoulder / 2 points each (:297-305)."""
    seg = [(e[i], e[i + 1]) for part in PART_LIST for e in part for i in range(len(e) - 1)]
    num = n_shoulder // 2
    seg += [(n_landmarks + i * num + j, n_landmarks + i * num + j + 1) for i in range(2) for j in range(num - 1)]
    return np.asarray(seg, np.int32).reshape(-1, 2)


class FeatureMapRasteriser:
    """Device-side ``get_data_test_mode``.  One instance per (frame size, shoulder count, device); the edge list lives on the
    device, so a call moves only the points."""

    def __init__(self, load_size: int = 512, n_shoulder: int = 18, device="cuda:0", thickness: int = 2):
        self.lib = N.load()
        self.load_size, self.n_shoulder, self.thickness = int(load_size), int(n_shoulder), int(thickness)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the edge-map rasteriser runs on the MI355X only (no CPU path); the reference's host path is "
                               "datasets/face_dataset.py with OpenCV")
        self._seg = torch.from_numpy(edge_list(self.n_shoulder)).to(self.device)

    # -- batched entry point ---------------------------------------------------------------------------------------
    def rasterise(self, landmarks, shoulders=None, pad: Optional[Sequence[float]] = None, out: Optional[torch.Tensor] = None,
                  as_uint8: bool = False) -> torch.Tensor:
        """landmarks [B, 73, 2] (x, y), shoulders [B, n_shoulder, 2] or None (numpy or torch, any float type / int32, host or
        device) -> float32 [B, 1, H, W] in {0, 1} on the device (uint8 [B, H, W] in {0, 255} when ``as_uint8``).
        ``pad`` = (top, bottom, left, right) shifts the shoulders by (right - left, top - bottom) as get_feature_image does
        (face_dataset.py:287-292; applied to a copy, the reference shifts its argument in place)."""
        return self.rasterise_points(self._points(landmarks, shoulders, pad), out, as_uint8)

    def rasterise_points(self, pts: torch.Tensor, out: Optional[torch.Tensor] = None, as_uint8: bool = False) -> torch.Tensor:
        """The same with the points already laid out as the kernel wants them: ONE device tensor [B, 73 + n_shoulder, 2] (landmarks
        then shoulder points, pad shift applied), int32 / float32 / float64 -- a render loop that keeps such a buffer moves one
        ~1.5 KB H2D copy per frame and launches nothing else."""
        if pts.device != self.device or pts.dim() != 3 or pts.shape[1] != N_LANDMARKS + self.n_shoulder or pts.shape[2] != 2 \
                or not pts.is_contiguous() or pts.dtype not in (torch.int32, torch.float32, torch.float64):
            raise ValueError("points must be a contiguous [B, %d, 2] int32/float32/float64 tensor on %s" % (N_LANDMARKS + self.n_shoulder, self.device))
        b = pts.shape[0]
        s = self.load_size
        if as_uint8:
            res = out if out is not None else torch.empty((b, s, s), dtype=torch.uint8, device=self.device)
            f32, u8 = None, res
        else:
            res = out if out is not None else torch.empty((b, 1, s, s), dtype=torch.float32, device=self.device)
            f32, u8 = res, None
        want = ((b, s, s), torch.uint8) if as_uint8 else ((b, 1, s, s), torch.float32)
        if res.device != self.device or not res.is_contiguous() or tuple(res.shape) != want[0] or res.dtype != want[1]:
            raise ValueError("out must be a contiguous %s tensor of shape %s on %s" % (want[1], list(want[0]), self.device))
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        with torch.cuda.device(self.device):
            N.check_raster(self.lib.lspraster_edge_maps(p(pts), N.RASTER_POINT_DTYPES[str(pts.dtype).replace("torch.", "")], b,
                                                        pts.shape[1], p(self._seg), self._seg.shape[0], self.thickness, s, s,
                                                        p(f32), p(u8),
                                                        ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        return res

    def _points(self, landmarks, shoulders, pad) -> torch.Tensor:
        def to_t(a):
            t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
            if t.dtype not in (torch.int32, torch.float32, torch.float64):
                t = t.to(torch.float64)
            return t
        lm = to_t(landmarks)
        if lm.dim() == 2:
            lm = lm.unsqueeze(0)
        if lm.dim() != 3 or lm.shape[1] != N_LANDMARKS or lm.shape[2] < 2:
            raise ValueError("landmarks must be [B, %d, 2] (x, y)" % N_LANDMARKS)
        lm = lm[..., :2]
        if self.n_shoulder:
            if shoulders is None:
                raise ValueError("this rasteriser was built for %d shoulder points" % self.n_shoulder)
            sh = to_t(shoulders)
            if sh.dim() == 2:
                sh = sh.unsqueeze(0)
            if tuple(sh.shape) != (lm.shape[0], self.n_shoulder, 2):
                raise ValueError("shoulders must be [B, %d, 2]" % self.n_shoulder)
            if pad is not None:
                top, bottom, left, right = pad
                sh = sh + torch.tensor([right - left, top - bottom], dtype=sh.dtype, device=sh.device)
            if sh.dtype != lm.dtype:
                wide = torch.promote_types(sh.dtype, lm.dtype)
                sh, lm = sh.to(wide), lm.to(wide)
            pts = torch.cat([lm.to(self.device), sh.to(self.device)], 1)
        else:
            pts = lm.to(self.device)
        return pts.contiguous()

    # -- the reference's method names (face_dataset.py) ---------------------------------------------------------------
    def get_data_test_mode(self, landmarks, shoulder, pad=None) -> torch.Tensor:
        """face_dataset.py:276-281: one frame -> float32 [1, H, W] (on the device here)"""
        return self.rasterise(landmarks, shoulder, pad)[0]

    def get_feature_image(self, landmarks, size=None, shoulders=None, image_pad=None) -> torch.Tensor:
        """face_dataset.py:284-294: the uint8 edge image [H, W] (on the device here)"""
        if size is not None and tuple(size) != (self.load_size, self.load_size):
            raise ValueError("size must be (loadSize, loadSize) = (%d, %d)" % (self.load_size, self.load_size))
        return self.rasterise(landmarks, shoulders, image_pad, as_uint8=True)[0]
