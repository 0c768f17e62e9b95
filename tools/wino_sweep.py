#!/usr/bin/env python3
"""Time the Winograd kernel (csrc/wino.hip) against the implicit GEMM on the stride-1 layer shapes of the plans (GPU):
  python tools/wino_sweep.py [batch ...]
Per shape: wino3x3<nb> for nb in {1, 2} x K splits, and igemm3x3 with the planner's tile, each as the median of 20 hipEvent pairs
around 10 back-to-back eager launches (weights warm: an upper bound on what the graph replay sees for cold weights)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livespeechportraits_amd import _native as N   # noqa: E402

SHAPES = [(64, 256), (128, 128), (256, 64), (512, 32), (512, 16)]


def timed(run):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 100)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    batches = [int(a) for a in sys.argv[1:]] or [1]
    lib, dev = N.load(), torch.device("cuda:0")
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for b in batches:
        for c, h in SHAPES:
            x = torch.randn(b, h, h, c, device=dev)
            w9 = torch.randn(c, 3, 3, c, device=dev) * 0.02
            wu = torch.randn(16 * c * c, device=dev) * 0.02            # timing only: any values in the fragment order
            sc, sh = torch.ones(c, device=dev), torch.zeros(c, device=dev)
            res = torch.randn(b, h, h, c, device=dev)
            out = torch.empty(b, h, h, c, device=dev)
            gf = 2 * c * c * 9 * h * h * b / 1e6            # MFLOP, so that gf / us = TFLOP/s
            sb = lib.lspf2f_conv3x3_scratch_bytes(b, h, h, c, 0, c, 1, 0, 0, 0, 0, 0, 0)
            scr = torch.zeros(max(sb, 4), dtype=torch.uint8, device=dev)
            t = timed(lambda: N.check(lib.lspf2f_conv3x3(p(x), None, p(w9), p(sc), p(sh), p(res), p(out), b, h, h, c, 0, c, 1, 0, 1, 0, 0, 0, 0, 0, p(scr), scr.numel(), st)))
            print("b%d c%d h%d  igemm (planner tile)        %7.1f us  %6.1f TFLOP/s algorithmic" % (b, c, h, t, gf / t), flush=True)
            for nb in (1, 3, 4, 2):                                      # 3 = tile 4003: nb = 1 with the U fragments in registers; 4 = tile 4004: four register sets
                nbk = 1 if nb in (3, 4) else nb
                if c % (32 * nbk):
                    continue
                for sp in (1, 2, 4, 8):
                    wgs = b * (h // 8) * (h // 16) * (c // (32 * nbk)) * sp
                    if c // 8 // sp < 4 or wgs < 128 or (sp > 1 and wgs > 2048):
                        continue
                    sb = lib.lspf2f_conv3x3_scratch_bytes(b, h, h, c, 0, c, 1, 0, 4000 + nb, 0, sp, -1, 0)
                    scr = torch.zeros(max(sb, 4), dtype=torch.uint8, device=dev)
                    try:
                        t = timed(lambda: N.check(lib.lspf2f_conv3x3(p(x), None, p(wu), p(sc), p(sh), p(res), p(out), b, h, h, c, 0, c, 1, 0, 1, 4000 + nb, 0, sp, -1, 0,
                                                                     p(scr), scr.numel(), st)))
                    except N.Lspf2fError as ex:
                        print("b%d c%d h%d  wino<%d> split %d: %s" % (b, c, h, nb, sp, ex))
                        continue
                    print("b%d c%d h%d  wino<%s> split %d (%5d WGs) %7.1f us  %6.1f TFLOP/s algorithmic  %5.1f executed" % (b, c, h, "1r" if nb == 3 else "1r4" if nb == 4 else str(nb), sp, wgs, t, gf / t, gf * 4 / 9 / t), flush=True)


main()
