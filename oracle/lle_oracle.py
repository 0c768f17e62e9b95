"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the reference's manifold projection:
  knn()        funcs/utils.py:100-118  KNN_with_torch
  lle_frame()  funcs/utils.py:121-158  solve_LLE_projection
  lle_all()    funcs/utils.py:171-179  compute_LLE_projection_all_frame
  blend()      demo.py:200
Same numpy / torch calls on the same dtypes (float32 normal equations through numpy.linalg.solve, float64 weights),
so it reproduces the reference bit for bit; oracle/make_golden_lle.py asserts that against the real functions."""
from __future__ import annotations

import numpy as np
import torch


def knn(feats, db, K=10):
    f, b = torch.from_numpy(np.asarray(feats)), torch.from_numpy(np.asarray(db))
    bn = (b ** 2).sum(-1)
    fn = (f ** 2).sum(-1)
    diss = fn.view(-1, 1) + bn.view(1, -1) - 2 * f @ b.t()
    return diss.topk(K, dim=1, largest=False).indices.cpu().numpy()


def lle_frame(feat, base):
    """feat [d], base [K, d] (nearest first) -> (w [K] float64 summing to 1, reconstruction [d])."""
    K = base.shape[0]
    if K == 1:
        return np.array([1]), base[0]
    w = np.zeros(K)
    rhs = feat - base[0]
    A = (base[1:] - base[0]).T
    w[1:] = np.linalg.solve(A.T.dot(A), A.T.dot(rhs))
    w[0] = 1 - w[1:].sum()
    return w, w.dot(base)


def lle_all(feats, db, ind):
    feats = np.asarray(feats)
    fuse = np.zeros_like(feats)
    w = np.zeros([feats.shape[0], ind.shape[1]])
    for i in range(feats.shape[0]):
        w[i], fuse[i] = lle_frame(feats[i], db[ind[i]])
    return w, fuse


def blend(feats, fuse, percent):
    return feats * (1 - percent) + fuse * percent
