#!/usr/bin/env python3
"""Frames/s and the per-kernel-class table of the InstanceNorm variant of the generators (norm_layer=nn.InstanceNorm2d) (GPU):
  python tools/in_bench.py [variant] [batch]
Same timing protocol as bench.py's headline: warm-up, then graph replays of the whole forward between a synchronise pair."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livespeechportraits_amd import synth   # noqa: E402
from livespeechportraits_amd.engine import Engine   # noqa: E402
from livespeechportraits_amd.topology import build_topology   # noqa: E402


def main():
    variant = sys.argv[1] if len(sys.argv) > 1 else "large"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    dev = torch.device("cuda:0")
    topo = build_topology(variant, size=512, norm="instance")
    sd = synth.make_state_dict(topo, 1234)
    eng = Engine(variant, size=512, max_batch=B, norm="instance")
    eng.load_state_dict(sd)
    eng.bind(eng.pack(), dev)
    f, c = synth.make_inputs(B, 512, seed=99, cand_batch=1)
    feat, cand = torch.from_numpy(f).to(dev), torch.from_numpy(c).to(dev)
    out = torch.empty((B, 3, 512, 512), device=dev)
    for _ in range(5):
        eng.forward(feat, cand, out)
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        eng.forward(feat, cand, out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    layers = eng.layers(B)
    print("%s InstanceNorm batch %d: %.1f frames/s, %.3f ms per step, %.1f TFLOP/s algorithmic (%.3f of the fp32 MFMA peak); %d launches-sets" % (
        variant, B, B / dt, dt * 1e3, topo.flops_per_frame() * B / dt / 1e12, topo.flops_per_frame() * B / dt / 1e12 / 157.3, len(layers)))
    classes = {}
    for i, l in enumerate(layers):
        classes.setdefault(l["kernel"].split(" ")[0], []).append(i)
    rows = []
    for name, idxs in classes.items():
        sel = [0] * len(layers)
        for i in idxs:
            sel[i] = 3
        ms = eng.subset_timed(feat, cand, sel, out, reps=10)
        rows.append((ms, name, len(idxs)))
    for ms, name, cnt in sorted(rows, reverse=True):
        print("   %-52s %3d layers %8.4f ms" % (name, cnt, ms))


main()
