#!/usr/bin/env python3
"""Where a gated workgroup of wino3x3_chain spends its time (library built with -DLSPF2F_WINO_STAMPS: tools/probes/wino_chain_stamps_job.sh).
  python tools/probes/wino_chain_stamps.py c hs splits nlayers mode
Per layer of ONE chain launch: median / p90 over the waves of the shader cycles between the phases (s_memtime, per-XCD counter: only differences inside a wave are used).
slots: 0 kernel entry, 1 prologue done (descriptors, addresses), 7 gate open (layers > 0), 2 first K-step landed, 3 K loop done, 4 Z patch written, 5 output stored"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livespeechportraits_amd import _native as N   # noqa: E402
import wino_model as WM   # noqa: E402


def main():
    c, hs, sp, nl, mode = [int(x) for x in sys.argv[1:6]]
    lib = N.load()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, hs, hs, c, generator=g).to(dev)
    us = [torch.from_numpy(WM.pack_u((torch.randn(c, c, 3, 3, generator=g) * (0.6 / (3.0 * c ** 0.5))).numpy()).astype(np.float32)).to(dev) for _ in range(nl)]
    outs = [torch.empty(1, hs, hs, c, device=dev) for _ in range(nl)]
    src = [x] + outs[:-1]
    res = [None if k % 2 == 0 else (x if k == 1 else outs[k - 2]) for k in range(nl)]
    used = (lib.lspf2f_wino_chain_scratch_bytes(nl, 1, hs, c, sp) + 255) // 256 * 256
    wgs = (hs // 8) * (hs // 16) * (c // 32) * sp
    blocks = nl * wgs
    scratch = torch.zeros(used + blocks * 4 * 8 * 8, dtype=torch.uint8, device=dev)
    arr = lambda ts: (ctypes.c_void_p * nl)(*[ctypes.c_void_p(t.data_ptr()) if t is not None else None for t in ts])
    none = [None] * nl
    relu = (ctypes.c_int * nl)(*([1] * nl))
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        N.check(lib.lspf2f_wino_chain(nl, arr(src), arr(us), arr(none), arr(none), arr(res), arr(outs), relu, 1, hs, c, sp, mode, ctypes.c_void_p(scratch.data_ptr()), scratch.numel(), st))
    torch.cuda.synchronize()
    raw = scratch[used:].view(torch.int64).view(blocks, 4, 8).cpu().numpy().astype(np.float64)
    if not raw.any():
        print("no stamps: build the library with -DLSPF2F_WINO_STAMPS")
        return
    print("c %d @ %d^2, %d splits, %d layers in one launch (mode %d): cycles, median / p90 over the %d waves of a layer" % (c, hs, sp, nl, mode, wgs * 4))
    print("%-6s %22s %22s %22s %22s %22s %22s" % ("layer", "entry->prologue done", "prologue->gate open", "gate open->1st landed", "K loop", "loop end->stored", "entry->stored"))
    for k in range(nl):
        r = raw[k * wgs:(k + 1) * wgs]
        t0, t1, t2, t3, t5, t7 = r[:, :, 0], r[:, :, 1], r[:, :, 2], r[:, :, 3], r[:, :, 5], r[:, :, 7]
        ok = t5 > 0        # split-K: only the stamps of a workgroup's own pass (every slice stores its slab)
        gate_open = np.where(t7 > 0, t7, t1)
        cols = [t1 - t0, gate_open - t1, t2 - gate_open, t3 - t2, t5 - t3, t5 - t0]
        print("%-6d " % k + " ".join("%10.0f /%10.0f" % (np.median(v[ok]), np.percentile(v[ok], 90)) for v in cols))


if __name__ == "__main__":
    main()
