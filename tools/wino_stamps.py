#!/usr/bin/env python3
"""Phase stamps of the Winograd kernel (GPU, library built with -DLSPF2F_WINO_STAMPS: tools/wino_stamps_job.sh):
  python tools/wino_stamps.py c h nb splits [batch]
Prints, per phase, the median / p90 over all waves of the shader cycles since the wave's kernel entry, and the wall time per launch."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livespeechportraits_amd import _native as N   # noqa: E402


def main():
    c, h, nb, sp = [int(a) for a in sys.argv[1:5]]
    b = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    tile = 4000 + nb                                         # nb = 3: tile 4003, one channel block per wave with the U fragments in registers
    if nb == 3:
        nb = 1
    lib, dev = N.load(), torch.device("cuda:0")
    x = torch.randn(b, h, h, c, device=dev)
    wu = torch.randn(16 * c * c, device=dev) * 0.02
    sc, sh = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    res = torch.randn(b, h, h, c, device=dev)
    out = torch.empty(b, h, h, c, device=dev)
    blocks = b * (h // 8) * (h // 16) * (c // (32 * nb)) * sp
    slab = sp * b * h * h * c * 4 if sp > 1 else 0
    ncnt = b * (h // 8) * (h // 16) * (c // (32 * nb))
    used = (slab + ncnt * 4 + 255) // 256 * 256            # where the library puts the stamps: behind slabs and arrival counters
    scr = torch.zeros(used + blocks * 4 * 8 * 8, dtype=torch.uint8, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    run = lambda: N.check(lib.lspf2f_conv3x3(p(x), None, p(wu), p(sc), p(sh), p(res), p(out), b, h, h, c, 0, c, 1, 0, 1, tile, 0, sp, -1, 0, p(scr), scr.numel(), st))
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    t = scr[used:].view(torch.int64).view(blocks, 4, 8).cpu().numpy().astype(np.float64)
    if not t.any():
        print("no stamps: the library was not built with -DLSPF2F_WINO_STAMPS"); return
    names = ["entry", "prologue done (descriptors, epilogue operands requested)", "first step landed + barrier", "K loop done", "Z patch written + barrier",
             "epilogue stores issued", "ticket taken (split-K)", "combine done (last arriver)"]
    print("c%d h%d nb%d%s splits %d batch %d: %d workgroups, %.1f us per launch (eager, weights warm)" % (c, h, nb, " (U in registers)" if tile == 4003 else "", sp, b, blocks, us))
    d = t - t[:, :, :1]
    for i, n_ in enumerate(names):
        col = d[:, :, i][t[:, :, i] != 0]
        if col.size:
            print("   %-62s %8.0f %8.0f   (%d waves)" % (n_, np.median(col), np.percentile(col, 90), col.size))
    k = t[:, :, 3] - t[:, :, 2]
    steps = (c // 8 + sp - 1) // sp
    print("   K loop: median %.0f cycles for %d steps = %.0f per step; MFMA issue per step and wave = %d cycles" % (np.median(k), steps, np.median(k) / steps, 64 * 16 * nb))
    for xcd in range(8):          # each XCD has its own counter
        sub = t[xcd::8]
        t0 = sub[:, :, 0].min()
        last = np.where(sub[:, :, 7] != 0, sub[:, :, 7], np.where(sub[:, :, 6] != 0, sub[:, :, 6], sub[:, :, 5])).max()
        print("   XCD %d: first wave start -> last wave end %d cycles; spread of starts %d" % (xcd, last - t0, sub[:, :, 0].max() - t0))


main()
