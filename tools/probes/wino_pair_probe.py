#!/usr/bin/env python3
"""Round 6 probe (VERDICT r5 "next" #1): the convs of a ResidualBlock (models/networks.py:650-675) as ONE launch against one launch per conv.

For each wino3x3<1> shape of the `large` plan at one frame (128 ch @ 128^2, 256 ch @ 64^2 with 2 K splits, 512 ch @ 32^2 with 4) and chain lengths 2 and 4:
  mode 0  one launch of wino3x3<1> per layer (register form, write-through stores, wave priority: exactly what the plans run)
  mode 2  the chain kernel launched once per layer (what its changed prologue costs without any overlap)
  mode 3  ONE launch of nlayers x 512 workgroups, gated on per-tile-block arrival counters (wino.hip, wino3x3_chain), raw patch read with sc1 loads
  mode 1  the same with one agent-scope acquire at the gate and plain loads behind it
Each arm is captured into a hipGraph of REPS chains and replayed; the arms alternate A-B-A-B in one session.  Results of the three modes must be
bit-identical; the scratch (counters) must be left zero and the give-up word must stay 0.

  python tools/probes/wino_pair_probe.py [reps=20] [rounds=5]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from livespeechportraits_amd import _native as N   # noqa: E402
import wino_model as WM   # noqa: E402


def build(lib, dev, c, hs, nlayers, sp, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(1, hs, hs, c, generator=g).to(dev)
    us, scs, shs = [], [], []
    for k in range(nlayers):
        w = (torch.randn(c, c, 3, 3, generator=g) * (0.6 / (3.0 * c ** 0.5))).numpy()
        us.append(torch.from_numpy(WM.pack_u(w).astype(np.float32)).to(dev))
        scs.append((1.0 + 0.05 * torch.randn(c, generator=g)).to(dev))
        shs.append((0.05 * torch.randn(c, generator=g)).to(dev))
    outs = [torch.empty(1, hs, hs, c, device=dev) for _ in range(nlayers)]
    # ResidualBlock: layer 2m = conv a (ReLU), layer 2m + 1 = conv b (+ the block's input, ReLU)
    src = [x] + outs[:-1]
    res = [None if k % 2 == 0 else (x if k == 1 else outs[k - 2]) for k in range(nlayers)]
    sb = lib.lspf2f_wino_chain_scratch_bytes(nlayers, 1, hs, c, sp)
    scratch = torch.zeros(sb, dtype=torch.uint8, device=dev)
    arr = lambda ts: (ctypes.c_void_p * nlayers)(*[ctypes.c_void_p(t.data_ptr()) if t is not None else None for t in ts])
    relu = (ctypes.c_int * nlayers)(*([1] * nlayers))
    args = (arr(src), arr(us), arr(scs), arr(shs), arr(res), arr(outs), relu)
    keep = (x, us, scs, shs, outs, scratch)

    def run(mode, stream):
        N.check(lib.lspf2f_wino_chain(nlayers, *args, 1, hs, c, sp, mode, ctypes.c_void_p(scratch.data_ptr()), scratch.numel(), ctypes.c_void_p(stream)))
    return run, outs, scratch, keep


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    lib = N.load()
    dev = torch.device("cuda:0")
    print("wino3x3 chain probe: %d chains per graph replay, %d A-B rounds; us per CHAIN (graph replay, hip events)" % (reps, rounds))
    for c, hs, sp in ((128, 128, 1), (256, 64, 2), (512, 32, 4)):
        for nl in (2, 4):
            run, outs, scratch, keep = build(lib, dev, c, hs, nl, sp)
            st = torch.cuda.current_stream().cuda_stream
            ref = None
            for mode in (0, 2, 3, 4, 1):
                for o in outs:
                    o.fill_(float("nan"))
                run(mode, st)
                torch.cuda.synchronize()
                got = [o.clone() for o in outs]
                if ref is None:
                    ref = got
                    assert all(torch.isfinite(o).all() for o in got)
                else:
                    for a, b in zip(ref, got):
                        assert torch.equal(a, b), "mode %d differs from one launch per layer (c %d hs %d layers %d)" % (mode, c, hs, nl)
            tail = scratch[(sp > 1) * sp * hs * hs * c * 4:].view(torch.int32)
            assert int(tail.abs().sum()) == 0, "counters / give-up word not zero after the launches: %s" % tail.nonzero().flatten()[:8].tolist()
            graphs = {}
            for mode in (0, 2, 3, 4, 1):
                g = torch.cuda.CUDAGraph()
                s = torch.cuda.Stream()
                with torch.cuda.stream(s):
                    for _ in range(3):
                        run(mode, s.cuda_stream)
                    s.synchronize()
                    with torch.cuda.graph(g, stream=s):
                        for _ in range(reps):
                            run(mode, s.cuda_stream)
                graphs[mode] = g
            times = {0: [], 2: [], 3: [], 4: [], 1: []}
            for _ in range(rounds):
                for mode in (0, 2, 3, 4, 1):
                    g = graphs[mode]
                    g.replay()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5):
                        g.replay()
                    e1.record()
                    torch.cuda.synchronize()
                    times[mode].append(e0.elapsed_time(e1) * 1000.0 / (5 * reps))
            tail = scratch[(sp > 1) * sp * hs * hs * c * 4:].view(torch.int32)
            assert int(tail.abs().sum()) == 0, "counters / give-up word not zero after the timed replays"
            m = {k: float(np.median(v)) for k, v in times.items()}
            print("c %3d @ %3d^2 splits %d, %d layers: per-layer launches %7.2f us [%s] | chain kernel per layer %7.2f | ONE launch, sc1 loads %7.2f | ONE launch, plain loads and no acquire (NOT a valid hand-off) %7.2f | ONE launch, acquire + plain loads %7.2f [%s] -> %+.1f %% per chain, %.2f us per seam" % (
                c, hs, sp, nl, m[0], " ".join("%.1f" % t for t in times[0]), m[2], m[3], m[4], m[1], " ".join("%.1f" % t for t in times[1]),
                (m[1] / m[0] - 1.0) * 100.0, (m[0] - m[1]) / (nl - 1)))
    print("bit-identical: all arms; counters left zero; give-up word 0")


if __name__ == "__main__":
    main()
