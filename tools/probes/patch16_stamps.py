#!/usr/bin/env python3
"""Segment times of conv3x3_patch16's K loop (a -DLSPF2F_PATCH_STAMPS build): per wave group, shader cycles summed over the K-tiles.
  python tools/probes/patch16_stamps.py c cout h tw bn [batch]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from livespeechportraits_amd import _native as N

def main():
    c, cout, h, tw, bn = [int(x) for x in sys.argv[1:6]]
    b = int(sys.argv[6]) if len(sys.argv) > 6 else 8
    lib = N.load(); dev = torch.device("cuda:0")
    x = torch.randn(b, h, h, c, device=dev).to(torch.bfloat16)
    w = (torch.randn(cout, 3, 3, c, device=dev) * 0.02).to(torch.bfloat16)
    res = torch.randn(b, h, h, cout, device=dev).to(torch.bfloat16)
    out = torch.empty(b, h, h, cout, device=dev, dtype=torch.bfloat16)
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    scratch = torch.zeros(4096 * 8 * 8 * 8, dtype=torch.uint8, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        N.check(lib.lspf2f_conv3x3(p(x), None, p(w), p(sc), p(sh), p(res), p(out), b, h, h, c, 0, cout, 1, 0, 1, 7000 + tw, bn, 0, 0, 1, p(scratch), scratch.numel(), st))
    torch.cuda.synchronize()
    nblk = b * h * h // 256 * (cout // bn)
    s = scratch.view(torch.int64).view(4096, 8, 8)[:nblk].cpu().numpy().astype(np.float64)
    ktiles = 9 * c // 64
    print("c%d o%d h%d tw%d bn%d b%d: %d workgroups, %d K-tiles; shader cycles per K-tile (median over workgroups; waves 0-3 = first group, 4-7 = second)" % (c, cout, h, tw, bn, b, nblk, ktiles))
    names = ["load segment", "barrier after load", "MFMA segment", "barrier after MFMA", "whole loop"]
    for g in range(2):
        v = s[:, g * 4:(g + 1) * 4, :]
        print("  group %d: " % g + "   ".join("%s %.0f" % (n, np.median(v[:, :, i]) / ktiles) for i, n in enumerate(names)))
    t0 = s[:, :, 5]
    print("  first wave start -> last wave start: %.0f cycles" % (t0.max() - t0.min()))

main()
