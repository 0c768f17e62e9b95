"""Manifold projection of the APC features on the device (include/lsplle.h) behind the reference's function
names (funcs/utils.py:100-118, 171-179; called at demo.py:196-200):

    ind = KNN_with_torch(audio_feats, APC_feat_database, K=Knear)
    weights, feat_fuse = compute_LLE_projection_all_frame(audio_feats, APC_feat_database, ind, audio_feats.shape[0])
    audio_feats = audio_feats * (1 - LLE_percent) + feat_fuse * LLE_percent

Same arguments (numpy arrays in) and return types (int64 indices; float64 weights, float32 features).
``project()`` does all three steps with device tensors in and out.  No CPU path."""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import numpy as np
import torch

from . import _native as N


def _dev(device) -> torch.device:
    dev = torch.device(device if device is not None else "cuda:0")
    if dev.type != "cuda":
        raise RuntimeError("manifold projection runs on the GPU only (there is no CPU path)")
    return dev


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _check(name: str, t: torch.Tensor, dtype) -> None:
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise ValueError("%s must be a contiguous %s device tensor" % (name, dtype))


def knn(feats: torch.Tensor, db: torch.Tensor, K: int = 10) -> torch.Tensor:
    """int64 [n, K] neighbour rows, nearest first (device tensors in/out; asynchronous on the current stream)."""
    _check("feats", feats, torch.float32); _check("db", db, torch.float32)
    if feats.dim() != 2 or db.dim() != 2 or feats.shape[1] != db.shape[1]:
        raise ValueError("feats [n, d] and db [m, d] must share d")
    lib = N.load()
    n, d = feats.shape
    m = db.shape[0]
    ind = torch.empty((n, K), dtype=torch.int64, device=feats.device)
    ws = torch.empty(lib.lsplle_knn_workspace_bytes(n, m), dtype=torch.uint8, device=feats.device)
    with torch.cuda.device(feats.device):
        N.check_lle(lib.lsplle_knn(_ptr(feats), n, _ptr(db), m, d, K, _ptr(ind), _ptr(ws), ws.numel(),
                                   ctypes.c_void_p(torch.cuda.current_stream(feats.device).cuda_stream)))
    return ind


def lle(feats: torch.Tensor, db: torch.Tensor, ind: torch.Tensor, percent: Optional[float] = None
        ) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    """(weights float64 [n, K], feat_fuse float32 [n, d], blend or None)."""
    _check("feats", feats, torch.float32); _check("db", db, torch.float32); _check("ind", ind, torch.int64)
    lib = N.load()
    n, d = feats.shape
    K = ind.shape[1]
    if ind.shape[0] != n:
        raise ValueError("ind must have one row per frame")
    w = torch.empty((n, K), dtype=torch.float64, device=feats.device)
    fuse = torch.empty_like(feats)
    blend = torch.empty_like(feats) if percent is not None else None
    with torch.cuda.device(feats.device):
        N.check_lle(lib.lsplle_solve(_ptr(feats), n, _ptr(db), db.shape[0], d, _ptr(ind), K, _ptr(w), _ptr(fuse), _ptr(blend),
                                     ctypes.c_float(percent if percent is not None else 0.0),
                                     ctypes.c_void_p(torch.cuda.current_stream(feats.device).cuda_stream)))
    return w, fuse, blend


def project(feats: torch.Tensor, db: torch.Tensor, K: int = 10, percent: float = 1.0) -> torch.Tensor:
    """demo.py:196-200 in one call: feats * (1 - percent) + LLE reconstruction * percent."""
    return lle(feats, db, knn(feats, db, K), percent)[2]


# ---- the reference's function names (numpy in, numpy out) ------------------------------------------
def KNN_with_torch(feats, feat_database, K=10, device=None):
    dev = _dev(device)
    f = torch.from_numpy(np.ascontiguousarray(feats, np.float32)).to(dev)
    b = torch.from_numpy(np.ascontiguousarray(feat_database, np.float32)).to(dev)
    return knn(f, b, K).cpu().numpy()


def compute_LLE_projection_all_frame(feats, feat_database, ind, nframe=None, device=None):
    dev = _dev(device)
    f = torch.from_numpy(np.ascontiguousarray(feats, np.float32)).to(dev)
    b = torch.from_numpy(np.ascontiguousarray(feat_database, np.float32)).to(dev)
    i = torch.from_numpy(np.ascontiguousarray(ind, np.int64)).to(dev)
    w, fuse, _ = lle(f, b, i)
    return w.cpu().numpy(), fuse.cpu().numpy()
