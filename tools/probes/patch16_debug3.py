#!/usr/bin/env python3
"""Single K-tile (cb 0, tap 5): express the wrong outputs as a combination of the 8 k-quad partial sums of that K-tile, per output channel"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import test_gpu_conv as T

def main():
    b, cin, cout, h, tw, bn = 1, 128, 128, 32, 32, 128
    cb, tap = int(sys.argv[1]), int(sys.argv[2])
    dev = torch.device("cuda:0")
    x0 = T.bf16r(T.rnd(b, cin, h, h, seed=171)); w = T.rnd(cout, cin, 3, 3, seed=172) * 0.05
    wm = torch.zeros_like(w)
    wm[:, cb * 64:(cb + 1) * 64, tap // 3, tap % 3] = w[:, cb * 64:(cb + 1) * 64, tap // 3, tap % 3]
    for rep in range(4):
        got = T.run_conv(dev, x0, None, wm, None, None, None, 1, 0, False, (7000 + tw, bn), 0, 0, dtype=1)
        ref = T.ref_conv(x0, None, T.bf16r(wm), None, None, None, 1, False, False)
        bad = (got - ref).abs() > (ref.abs() * 2.0 ** -8 + 1e-3)
        if not bad.any():
            print("rep %d ok" % rep); continue
        idx = bad.nonzero()
        ys = idx[:, 2].unique().tolist(); chs = idx[:, 1].unique().tolist()
        print("rep %d: bad %d, ch %s, y %s" % (rep, bad.sum().item(), chs, ys))
        parts = []
        for s in range(8):
            ws_ = torch.zeros_like(wm)
            c0 = cb * 64 + s * 8
            ws_[:, c0:c0 + 8] = T.bf16r(wm)[:, c0:c0 + 8]
            parts.append(F.conv2d(x0.double(), ws_.double(), padding=1))
        P = torch.stack(parts, -1)     # [1][cout][h][h][8]
        for n in chs:
            A = P[0, n, ys[0]:ys[-1] + 1].reshape(-1, 8)
            y = got[0, n, ys[0]:ys[-1] + 1].reshape(-1, 1).double()
            sol = torch.linalg.lstsq(A, y).solution.flatten()
            res = (A @ sol.view(-1, 1) - y).abs().max().item()
            print("   ch %d: coefficients of k-quads 0..7 = %s   (residual %.1e)" % (n, " ".join("%5.2f" % v for v in sol.tolist()), res))

main()
