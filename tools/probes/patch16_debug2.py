#!/usr/bin/env python3
"""Which K-tile of the patch-staged kernel goes wrong?  weights non-zero in one (tap, channel block) at a time"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import test_gpu_conv as T

def main():
    b, cin, cout, h, tw, bn = [int(x) for x in sys.argv[1:7]]
    dev = torch.device("cuda:0")
    x0 = T.bf16r(T.rnd(b, cin, h, h, seed=171)); w = T.rnd(cout, cin, 3, 3, seed=172) * 0.05
    def run(wm, tag):
        got = T.run_conv(dev, x0, None, wm, None, None, None, 1, 0, False, (7000 + tw, bn), 0, 0, dtype=1)
        ref = T.ref_conv(x0, None, T.bf16r(wm), None, None, None, 1, False, False)
        bad = (got - ref).abs() > (ref.abs() * 2.0 ** -8 + 1e-3)
        if bad.any():
            idx = bad.nonzero()
            desc = []
            for d, name in enumerate(("f", "ch", "y", "x")):
                u = idx[:, d].unique().tolist()
                desc.append("%s %d..%d (%d)" % (name, u[0], u[-1], len(u)))
            z = (got[bad] == 0).float().mean().item()
            print("%s: bad %d  %s  got==0 on %.2f of them" % (tag, bad.sum().item(), "  ".join(desc), z))
        else:
            print("%s: ok" % tag)
    for rep in range(3): run(w, "full #%d" % rep)
    for cb in range(cin // 64):
        for tap in range(9):
            wm = torch.zeros_like(w)
            wm[:, cb * 64:(cb + 1) * 64, tap // 3, tap % 3] = w[:, cb * 64:(cb + 1) * 64, tap // 3, tap % 3]
            run(wm, "cb %d tap %d" % (cb, tap))

main()
