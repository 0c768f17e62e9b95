"""ORACLE (test infrastructure) -- the landmark edge map of the render loop on the CPU.

Restates datasets/face_dataset.py:34-42 (part_list), :276-323 (get_data_test_mode, get_feature_image, draw_shoulder_points,
draw_face_feature_maps) around oracle/raster_oracle.c, which restates OpenCV 4.4.0's cv::line (requirements.txt:
opencv_python==4.4.0.40).  PARITY UNPINNED: OpenCV does not exist in this image (see the header of raster_oracle.c).
Only tests/ and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libraster_oracle.so")
_lib = None

# face_dataset.py:34-42 -- 8 facial parts, each a list of polylines over the 73 landmarks
PART_LIST = [[list(range(0, 15))],
             [[15, 16, 17, 18, 18, 19, 20, 15]],
             [[21, 22, 23, 24, 24, 25, 26, 21]],
             [list(range(35, 44))],
             [[27, 65, 28, 68, 29], [29, 67, 30, 66, 27]],
             [[33, 69, 32, 72, 31], [31, 71, 34, 70, 33]],
             [list(range(46, 53)), [52, 53, 54, 55, 56, 57, 46]],
             [[46, 63, 62, 61, 52], [52, 60, 59, 58, 46]]]


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "raster_oracle.c")
        if not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.raster_edge_map.restype = None
        _lib.raster_edge_map.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_void_p]
    return _lib


def _trunc(a) -> np.ndarray:
    """[int(flt) for flt in row] (face_dataset.py:301, :318): truncation toward zero"""
    return np.trunc(np.asarray(a, np.float64)).astype(np.int32)


def draw(points_i32: np.ndarray, segments: np.ndarray, size, thickness: int = 2) -> np.ndarray:
    """cv2.line(img, p[a], p[b], 255, thickness) for every (a, b) of `segments` on a zero uint8 image of size (w, h)"""
    w, h = size
    pts = np.ascontiguousarray(points_i32, np.int32)
    seg = np.ascontiguousarray(segments, np.int32)
    img = np.empty((h, w), np.uint8)
    lib().raster_edge_map(pts.ctypes.data, pts.shape[0], seg.ctypes.data, seg.shape[0], thickness, h, w, img.ctypes.data)
    return img


def draw_face_feature_maps(keypoints, size=(512, 512)) -> np.ndarray:
    """face_dataset.py:311-322"""
    seg = [(e[i], e[i + 1]) for part in PART_LIST for e in part for i in range(len(e) - 1)]
    return draw(_trunc(keypoints), np.array(seg, np.int32), size)


def draw_shoulder_points(img: np.ndarray, shoulder_points) -> np.ndarray:
    """face_dataset.py:297-305: two chains of num = n/2 points"""
    sp = _trunc(shoulder_points)
    num = sp.shape[0] // 2
    seg = [(i * num + j, i * num + j + 1) for i in range(2) for j in range(num - 1)]
    if not seg:
        return img
    extra = draw(sp, np.array(seg, np.int32), (img.shape[1], img.shape[0]))
    return np.maximum(img, extra)          # every primitive writes 255 on a 0/255 image: drawing onto `img` == the union


def get_feature_image(landmarks, size, shoulders=None, image_pad=None) -> np.ndarray:
    """face_dataset.py:284-294 (the in-place shift of `shoulders` is applied to a copy here)"""
    im = draw_face_feature_maps(landmarks, size)
    if shoulders is not None:
        sh = np.array(shoulders, dtype=np.asarray(shoulders).dtype, copy=True)
        if image_pad is not None:
            top, bottom, left, right = image_pad
            sh[:, 0] += right - left
            sh[:, 1] += top - bottom
        im = draw_shoulder_points(im, sh)
    return im


def get_data_test_mode(landmarks, shoulder, pad=None, load_size: int = 512) -> np.ndarray:
    """face_dataset.py:276-281: float32 [1, H, W] with values exactly {0, 1}"""
    return get_feature_image(landmarks, (load_size, load_size), shoulder, pad)[np.newaxis, :].astype(np.float32) / 255.
