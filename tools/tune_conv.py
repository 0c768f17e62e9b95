#!/usr/bin/env python3
"""Sweep tile shape x split-K of the implicit-GEMM conv kernel over the distinct layer shapes of
the generator through the C ABI's single-conv entry point.  GPU only.
  python tools/tune_conv.py [--batch B] [--out file]
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from livespeechportraits_amd import _native as N  # noqa: E402

# (name, c0, c1, cout, hs, stride, up)
SHAPES = [
    ("64>64@256", 64, 0, 64, 256, 1, 0),
    ("64>128@256s2", 64, 0, 128, 256, 2, 0),
    ("128>128@128", 128, 0, 128, 128, 1, 0),
    ("128>256@128s2", 128, 0, 256, 128, 2, 0),
    ("256>256@64", 256, 0, 256, 64, 1, 0),
    ("256>512@64s2", 256, 0, 512, 64, 2, 0),
    ("512>512@32", 512, 0, 512, 32, 1, 0),
    ("512>512@16", 512, 0, 512, 16, 1, 0),
    ("512>512@8", 512, 0, 512, 8, 1, 0),
    ("512>512@4", 512, 0, 512, 4, 1, 0),
    ("512>512@2", 512, 0, 512, 2, 1, 0),
    ("1024>512@8up9", 512, 512, 512, 8, 1, 1),
    ("1024>512@16up", 512, 512, 512, 16, 1, 2),
    ("1024>256@32up", 512, 512, 256, 32, 1, 2),
    ("512>128@64up", 256, 256, 128, 64, 1, 2),
    ("256>64@128up", 128, 128, 64, 128, 1, 2),
]
DT = 0
TILES = [(128, 128), (128, 64), (64, 128), (64, 64), (32, 128), (32, 64)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--out", default=None)
    ap.add_argument("--only", default=None)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--shape", action="append", default=[], help="extra shape 'cin,cout,h[,stride]' (plain 3x3, one source); repeatable; replaces the built-in list")
    a = ap.parse_args()
    global DT
    DT = 1 if a.dtype == "bf16" else 0
    tdt = torch.bfloat16 if DT else torch.float32
    lib = N.load()
    dev = torch.device("cuda:0")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    lines = []
    shapes = SHAPES
    if a.shape:
        shapes = []
        for spec in a.shape:
            f = [int(v) for v in spec.split(",")]
            shapes.append(("%d>%d@%d%s" % (f[0], f[1], f[2], "s2" if len(f) > 3 and f[3] == 2 else ""), f[0], 0, f[1], f[2], f[3] if len(f) > 3 else 1, 0))
    for name, c0, c1, cout, hs, stride, up in shapes:
        if a.only and a.only not in name:
            continue
        b = a.batch
        x0 = (torch.rand(b, hs, hs, c0, device=dev) - 0.5).to(tdt)
        x1 = (torch.rand(b, hs, hs, c1, device=dev) - 0.5).to(tdt) if c1 else None
        w = ((torch.rand(cout, (16 if up == 2 else 9) * (c0 + c1), device=dev) - 0.5) * 0.05).to(tdt)
        sc = torch.rand(cout, device=dev) + 0.5
        sh = torch.rand(cout, device=dev)
        ho = 2 * hs if up else hs // stride
        out = torch.empty(b, ho, ho, cout, device=dev, dtype=tdt)
        M = b * ho * ho
        kt = (4 if up == 2 else 9) * (c0 + c1) // (64 if DT else 32)
        par = 4 if up == 2 else 1
        flops = 2.0 * M * cout * 9 * (c0 + c1)
        res = []
        for tm, tn in TILES:
            if tn > max(64, cout) or (tm > 64 and M <= 64):
                continue
            tiles = par * -(-(M // par) // tm) * -(-cout // tn)
            for sp in (1, 2, 3, 4, 6, 8, 12, 16, 18, 24, 36, 48, 72):
                if sp > kt // 4 or tiles * sp > 8192 or (tiles * sp < 96 and sp < kt // 2):
                    continue
                sb = lib.lspf2f_conv3x3_scratch_bytes(b, hs, hs, c0, c1, cout, stride, up, tm, tn, sp, 1, DT)
                if sb == 0 and sp > 1:
                    continue
                scratch = torch.empty(max(sb, 4), dtype=torch.uint8, device=dev)
                for g in (1, 2, 4):
                    if g == 2 and (tm, tn) not in ((128, 64), (64, 64)):
                        continue
                    if g == 4 and (tm, tn) not in ((64, 64), (32, 64)):
                        continue

                    def run():
                        N.check(lib.lspf2f_conv3x3(P(x0), P(x1), P(w), P(sc), P(sh), None, P(out), b, hs, hs, c0, c1,
                                                   cout, stride, up, 1, tm, tn, sp, g, DT, P(scratch), scratch.numel(),
                                                   stream))
                    try:
                        for _ in range(3):
                            run()
                    except N.Lspf2fError:
                        continue          # combination not instantiated (e.g. 9-tap upsample form)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    reps = 20
                    e0.record()
                    for _ in range(reps):
                        run()
                    e1.record()
                    torch.cuda.synchronize()
                    us = e0.elapsed_time(e1) * 1e3 / reps
                    res.append((us, tm, tn, sp, tiles * sp, g))
        res.sort()
        best = res[0]
        lines.append("%-16s M=%-7d N=%-4d K=%-5d best %7.1f us %6.1f TF  tile %dx%d split %d g%d (%d WGs)" % (
            name, M, cout, 9 * (c0 + c1), best[0], flops / best[0] / 1e6, best[1], best[2], best[3], best[5], best[4]))
        for us, tm, tn, sp, wgs, g in res[:10]:
            lines.append("      %7.1f us %6.1f TF  %3dx%-3d split %-2d g%d WGs %d" % (us, flops / us / 1e6, tm, tn, sp, g, wgs))
    txt = "\n".join(lines)
    print(txt)
    if a.out:
        open(a.out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
