#!/usr/bin/env python3
"""Time the `size == 'small'` generator at 512x512 -- host-sequenced launches and the captured-graph replay -- and the oracle on the host."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livespeechportraits_amd import synth
from livespeechportraits_amd.unet_small import SmallUnetEngine
from oracle import unet_small_oracle
dev = torch.device("cuda:0")
sd = synth.make_unet_small_state_dict()
for graph, live in ((False, True), (False, False), (True, True)):
    e = SmallUnetEngine(graph=graph, live_taps=live); e.load_state_dict(sd, "model", dev)
    for B in (1, 8):
        x = torch.from_numpy(synth.symmetric(B * 23 * 512 * 512, 0.6, 3).reshape(B, 23, 512, 512)).to(dev)
        for _ in range(3): e.forward(x)
        torch.cuda.synchronize(); ts = []
        for _ in range(10):
            t0 = time.perf_counter(); e.forward(x); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        ts.sort()
        print("small generator, %s, %s, batch %d: median %.3f ms per forward (min %.3f) = %.0f frames/s" % (
            "graph replay" if graph else "host-sequenced", "16 live K blocks per down-conv" if live else "dense 36-block down-convs", B, 1e3 * ts[5], 1e3 * ts[0], B / ts[5]))
sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
x = torch.from_numpy(synth.symmetric(23 * 512 * 512, 0.6, 3).reshape(1, 23, 512, 512))
unet_small_oracle.generator_forward(sdt, x); t0 = time.perf_counter()
for _ in range(3): unet_small_oracle.generator_forward(sdt, x)
print("oracle on the host (%d threads): %.1f ms per frame" % (torch.get_num_threads(), 1e3 * (time.perf_counter() - t0) / 3))
