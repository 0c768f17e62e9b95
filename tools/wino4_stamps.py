#!/usr/bin/env python3
"""Phase stamps of the Winograd F(4x4,3x3) kernel (GPU, library built with -DLSPF2F_WINO_STAMPS: tools/sessions/wino4_stamps_job.sh):
  python tools/wino4_stamps.py c h splits [batch]
Prints, per phase, the median / p90 over all waves of the shader cycles since the wave's kernel entry, and the wall time per launch."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livespeechportraits_amd import _native as N   # noqa: E402


def main():
    c, h, sp = [int(a) for a in sys.argv[1:4]]
    b = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    lib, dev = N.load(), torch.device("cuda:0")
    x = torch.randn(b, h, h, c, device=dev)
    wu = torch.randn(36 * c * c, device=dev) * 0.02
    sc, sh = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    res = torch.randn(b, h, h, c, device=dev)
    out = torch.empty(b, h, h, c, device=dev)
    ncnt = b * (h // 16) * (h // 32) * (c // 32)
    blocks = ncnt * sp
    slab = sp * b * h * h * c * 4 if sp > 1 else 0
    used = (slab + ncnt * 4 + 255) // 256 * 256            # where the library puts the stamps: behind slabs and arrival counters
    scr = torch.zeros(used + blocks * 4 * 8 * 8, dtype=torch.uint8, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    run = lambda: N.check(lib.lspf2f_conv3x3(p(x), None, p(wu), p(sc), p(sh), p(res), p(out), b, h, h, c, 0, c, 1, 0, 1, 6001, 0, sp, -1, 0, p(scr), scr.numel(), st))
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
    steps = (c // 8 + sp - 1) // sp
    print("c%d h%d splits %d batch %d: %d workgroups, %d steps each, %.1f us per launch (eager, weights warm)" % (c, h, sp, b, blocks, steps, us))
    names = {2: "steps 0, 1 landed + barrier", 3: "operands of step 0 in registers + barrier", 4: "step 0 done (36 MFMAs + operands of step 1 + copies of step 2)",
             5: "step 2 done", 6: "K loop done", 7: "accumulators in the LDS patch + barrier"}
    d = t - t[:, :, :1]
    for i in sorted(names):
        col = d[:, :, i][t[:, :, i] != 0]
        if col.size:
            print("   %-66s %8.0f %8.0f   (%d waves)" % (names[i], np.median(col), np.percentile(col, 90), col.size))
    print("   %-66s %8.0f %8.0f" % ("prologue done -> epilogue stores issued (duration)", np.median(t[:, :, 1]), np.percentile(t[:, :, 1], 90)))
    k = t[:, :, 6] - t[:, :, 3]
    print("   K loop: median %.0f cycles for %d steps = %.0f per step (step 0 alone %.0f%s); MFMA issue per step and wave = %d cycles" % (
        np.median(k), steps, np.median(k) / steps, np.median(t[:, :, 4] - t[:, :, 3]),
        ", steps 1-2 %.0f each" % (np.median(t[:, :, 5] - t[:, :, 4]) / 2) if steps > 2 else "", 64 * 36))
    for xcd in range(8):          # each XCD has its own counter
        sub = t[xcd::8]
        t0 = sub[:, :, 0].min()
        print("   XCD %d: first wave start -> last patch barrier %d cycles; spread of starts %d" % (xcd, sub[:, :, 7].max() - t0, sub[:, :, 0].max() - t0))


main()
