#!/usr/bin/env python3
"""Where does the Winograd kernel differ from the convolution?  (GPU; a diagnostic for kernel work, not a test.)
Runs small problems through lspf2f_conv3x3(tile 4001/4002) and groups the error by output parity, tile position inside the
tile-block, tile-block, output channel and input channel, so that a wrong index map shows its shape in one run."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_conv import run_wino   # noqa: E402


def report(tag, got, ref):
    err = (got.double() - ref).abs()[0].numpy()          # [N][H][W]
    n, h, w = err.shape
    scale = max(1e-30, float(ref.abs().max()))
    bad = err > 1e-4 * scale
    print("== %s: max-abs %.3e (range %.2f), %.2f %% of outputs off by > 1e-4 of the range" % (tag, err.max(), scale, 100.0 * bad.mean()))
    if not bad.any():
        return
    print("   by output parity (y&1, x&1):", [[round(float(bad[:, a::2, b::2].mean()), 3) for b in range(2)] for a in range(2)])
    ty = (np.arange(h)[:, None] % 8) // 2
    tx = (np.arange(w)[None, :] % 16) // 2
    print("   by tile inside the tile-block (rows ty 0..3, cols tx 0..7):")
    for a in range(4):
        print("     ", " ".join("%.2f" % bad[:, (ty == a) & (tx == b)].mean() for b in range(8)))
    print("   by tile-block (rows of 8, cols of 16):")
    for a in range(h // 8):
        print("     ", " ".join("%.2f" % bad[:, 8 * a:8 * a + 8, 16 * b:16 * b + 16].mean() for b in range(w // 16)))
    per_n = bad.reshape(n, -1).mean(1)
    print("   by output channel:", " ".join("%.1f" % v for v in per_n))
    print("   border rows/cols bad fraction: top %.2f bottom %.2f left %.2f right %.2f interior %.2f" % (
        bad[:, 0].mean(), bad[:, -1].mean(), bad[:, :, 0].mean(), bad[:, :, -1].mean(), bad[:, 1:-1, 1:-1].mean()))


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    for (c, n, h, nb, sp) in [(16, 32, 32, 1, 1), (16, 64, 32, 2, 1), (64, 32, 32, 1, 2)]:
        x = torch.randn(1, c, h, h, generator=g)
        w = torch.randn(n, c, 3, 3, generator=g) / (3 * c ** 0.5)
        ref = F.conv2d(x.double(), w.double(), None, 1, 1)
        report("random c%d n%d h%d nb%d split%d" % (c, n, h, nb, sp), run_wino(dev, x, w, None, None, None, False, nb, sp), ref)
        # centre tap only: out[co] = sum_ci w[co][ci] x[ci] -- no spatial mixing, isolates the channel / fragment order
        wc = torch.zeros_like(w)
        wc[:, :, 1, 1] = w[:, :, 1, 1]
        report("  centre tap only", run_wino(dev, x, wc, None, None, None, False, nb, sp), F.conv2d(x.double(), wc.double(), None, 1, 1))
        # which input channel does the kernel pair with weight channel ci?  x = one-hot channel planes of distinct constants
        if sp == 1:
            xc = torch.zeros(1, c, h, h)
            for ci in range(c):
                xc[0, ci] = float(ci + 1)
            wd = torch.zeros(n, c, 3, 3)
            for ci in range(c):
                wd[ci % n, ci, 1, 1] = 1.0            # out[ci] (ci < n) should read ci + 1 away from nothing
            got = run_wino(dev, xc, wd, None, None, None, False, nb, sp)[0, :, h // 2, h // 2]
            print("   channel pairing (expect 1..%d then sums for wrapped): %s" % (c, [round(float(v), 2) for v in got[:min(n, c)]]))
        # single taps: the spatial offset each of the 9 taps produces
        xi = torch.zeros(1, c, h, h)
        xi[0, 3, 12, 20] = 1.0
        for ky in range(3):
            for kx in range(3):
                wt = torch.zeros(n, c, 3, 3)
                wt[5, 3, ky, kx] = 1.0
                got = run_wino(dev, xi, wt, None, None, None, False, nb, sp)[0, 5]
                pos = (got.abs() > 1e-3).nonzero().tolist()
                exp = [[12 - (ky - 1), 20 - (kx - 1)]]
                if pos != exp:
                    print("   tap (%d,%d): non-zeros at %s values %s, expected %s" % (ky, kx, pos[:6], [round(float(got[a, b]), 3) for a, b in pos[:6]], exp))
    print("done")


main()
