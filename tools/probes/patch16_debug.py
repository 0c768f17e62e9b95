#!/usr/bin/env python3
"""Where does the patch-staged 16-bit kernel differ from the fp64 conv?  python tools/probes/patch16_debug.py b cin cout h tw bn"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import test_gpu_conv as T

def main():
    b, cin, cout, h, tw, bn = [int(x) for x in sys.argv[1:7]]
    dev = torch.device("cuda:0")
    x0 = T.bf16r(T.rnd(b, cin, h, h, seed=171)); w = T.rnd(cout, cin, 3, 3, seed=172) * 0.05
    got = T.run_conv(dev, x0, None, w, None, None, None, 1, 0, False, (7000 + tw, bn), 0, 0, dtype=1)
    ref = T.ref_conv(x0, None, T.bf16r(w), None, None, None, 1, False, False)
    bad = (got - ref).abs() > (ref.abs() * 2.0 ** -8 + 1e-3)
    print("b%d c%d o%d h%d tw%d bn%d: bad %d of %d" % (b, cin, cout, h, tw, bn, bad.sum().item(), bad.numel()))
    if bad.any():
        idx = bad.nonzero()
        for d, name in enumerate(("frame", "channel", "y", "x")):
            u, c = idx[:, d].unique(return_counts=True)
            print("  %s: %s" % (name, ", ".join("%d:%d" % (a, n) for a, n in zip(u.tolist()[:40], c.tolist()[:40]))))

main()
