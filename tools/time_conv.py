#!/usr/bin/env python3
"""Time ONE conv shape through lspf2f_conv3x3 with a forced tile (GPU): median of hipEvent pairs, eager launches.
  python tools/time_conv.py c0 c1 cout hs up tile_m tile_n [batch [k_group [dtype(0|1) [residual(0|1) [split_k [stride]]]]]]"""
import ctypes, sys
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livespeechportraits_amd import _native as N

def main():
    c0, c1, cout, hs, up, tm, tn = [int(x) for x in sys.argv[1:8]]
    b = int(sys.argv[8]) if len(sys.argv) > 8 else 1
    kg = int(sys.argv[9]) if len(sys.argv) > 9 else 0        # -1: full-K tile-blocked weight layout (timing only: same bytes)
    dt = int(sys.argv[10]) if len(sys.argv) > 10 else 0
    with_res = int(sys.argv[11]) if len(sys.argv) > 11 else 0
    split = int(sys.argv[12]) if len(sys.argv) > 12 else 0   # 2 with tile 16 16: the K-split form of the full-K kernel (no stamps then: they share the scratch)
    stride = int(sys.argv[13]) if len(sys.argv) > 13 else 1
    tdt = torch.bfloat16 if dt else torch.float32
    lib = N.load(); dev = torch.device("cuda:0")
    ho = 2 * hs if up else hs // stride
    d0 = torch.randn(b, hs, hs, c0, device=dev).to(tdt); d1 = torch.randn(b, hs, hs, c1, device=dev).to(tdt) if c1 else None
    w = (torch.randn(cout, 4 if up == 2 else 3, 4 if up == 2 else 3, c0 + c1, device=dev) * 0.02).to(tdt)     # timing only: any k_group == -1 layout has the same bytes; up == 2: the sub-pixel operand [4][cout][2][2][cin]
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    out = torch.empty(b, ho, ho, cout, device=dev, dtype=tdt)
    res = torch.randn(b, ho, ho, cout, device=dev).to(tdt) if with_res else None
    sb = lib.lspf2f_conv3x3_scratch_bytes(b, hs, hs, c0, c1, cout, stride, up, tm, tn, split, kg, dt)
    scratch = torch.zeros(max(sb, 2048 * 4 * 16 * 8), dtype=torch.uint8, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    def run():
        N.check(lib.lspf2f_conv3x3(p(d0), p(d1), p(w), p(sc), p(sh), p(res), p(out), b, hs, hs, c0, c1, cout, stride, up, 1, tm, tn, split, kg, dt, p(scratch), scratch.numel(), st))
    for _ in range(5): run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 100)
    ts.sort()
    st64 = scratch[:2048 * 4 * 16 * 8].view(torch.int64).view(2048, 4, 16).cpu().numpy()
    if st64.any() and not split:
        import numpy as np
        nb = (st64[:, 0, 0] != 0).sum()
        d = (st64[:nb] - st64[:nb, :, :1]).astype(np.float64)
        names = ["start", "dma issued", "6 taps issued", "dma landed", "barrier", "tap0", "tap1", "tap2", "tap3", "tap4", "tap5", "tap6", "tap7", "tap8", "k done", "end"]
        if tm >= 64:
            names = ["start", "descriptors done", "first fetch issued", "first tile landed+barrier", "-", "-", "K loop done", "epilogue done", "tile ids done", "row descriptors done", "weight offsets done"] + ["-"] * 5
        print("stamps (shader cycles since kernel entry of the wave; median / p90 over %d blocks x 4 waves):" % nb)
        for i, n_ in enumerate(names):
            print("   %-14s %8.0f %8.0f" % (n_, np.median(d[:, :, i]), np.percentile(d[:, :, i], 90)))
        last = 7 if tm >= 64 else 15
        for x in range(8):      # each XCD has its own counter: dispatch order deals block i to XCD i % 8
            sub = st64[x:nb:8]
            t0 = sub[:, :, 0].min(); print("   XCD %d: first wave start -> last wave end %d cycles; spread of starts %d" % (x, sub[:, :, last].max() - t0, sub[:, :, 0].max() - t0))
    gf = 2 * cout * (c0 + c1) * 9 * ho * ho * b / 1e9
    nbytes = (b * hs * hs * (c0 + c1) + b * ho * ho * cout * (2 if with_res else 1)) * (2 if dt else 4) + w.numel() * (2 if dt else 4)
    print("c%d+%d o%d h%d%s%s b%d tile %dx%d %s%s: %.1f us per launch (10 back-to-back), %.1f TFLOP/s, %.0f GB/s algorithmic" % (c0, c1, cout, hs, "up" if up else "", " s2" if stride == 2 else "", b, tm, tn,
          "bf16" if dt else "f32", "+res" if with_res else "", ts[len(ts)//2], gf / ts[len(ts)//2], nbytes / ts[len(ts)//2] / 1e3))

main()
