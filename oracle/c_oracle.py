"""ORACLE (test infrastructure) -- ctypes binding of oracle/f2f_oracle.c.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libf2f_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "f2f_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "all"] if force
                              else ["make", "-C", _HERE, "-s", "all"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.f2f_generator_forward.restype = ctypes.c_long
        _lib.f2f_generator_forward.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 7 + \
            [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        _lib.f2f_conv3x3.restype = None
        _lib.f2f_conv3x3.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return _lib


def flatten_params(topo, sd: Dict[str, np.ndarray]) -> np.ndarray:
    """State dict -> walk-order float blob (num_batches_tracked dropped)."""
    parts = [np.ascontiguousarray(sd[k], np.float32).ravel() for k in topo.tensors
             if not k.endswith("num_batches_tracked")]
    return np.concatenate(parts)


def generator_forward(topo, sd, x: np.ndarray, apply_tanh: bool = True) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    b = x.shape[0]
    assert x.shape[1:] == (topo.input_nc, topo.size, topo.size)
    blob = flatten_params(topo, sd)
    out = np.empty((b, topo.output_nc, topo.size, topo.size), np.float32)
    used = lib().f2f_generator_forward(blob.ctypes.data, topo.nres, topo.input_nc, topo.output_nc,
                                       topo.ngf, topo.num_downs, topo.size, b, x.ctypes.data,
                                       out.ctypes.data, 1 if apply_tanh else 0)
    if used != blob.size:
        raise RuntimeError("C oracle consumed %d parameter floats, blob has %d" % (used, blob.size))
    return out


def conv3x3(x: np.ndarray, w: np.ndarray, stride: int) -> np.ndarray:
    """x [cin,h,w], w [cout,cin,3,3] -> [cout,ho,wo]"""
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    cin, h, wd = x.shape
    cout = w.shape[0]
    ho, wo = (h - 1) // stride + 1, (wd - 1) // stride + 1
    out = np.empty((cout, ho, wo), np.float32)
    lib().f2f_conv3x3(x.ctypes.data, cin, h, wd, w.ctypes.data, cout, stride, out.ctypes.data)
    return out
