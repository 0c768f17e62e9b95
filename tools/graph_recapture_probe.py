#!/usr/bin/env python3
"""What a hipGraph cache miss costs (GPU): engine.forward with 2 input/output buffer pairs (always a cached graph) against 12 (the handle keeps 8: every call re-captures)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livespeechportraits_amd import synth
from livespeechportraits_amd.engine import Engine
from livespeechportraits_amd.topology import build_topology
dev = torch.device("cuda:0")
topo = build_topology("large")
e = Engine("large", max_batch=8)
e.load_state_dict(synth.make_state_dict(topo, 1234))
e.bind(e.pack(), dev)
for B in (1, 8):
    f, c = synth.make_inputs(B, 512, 99, 1)
    c = torch.from_numpy(c).to(dev)
    for nbuf in (2, 12):
        fs = [torch.from_numpy(f).to(dev) for _ in range(nbuf)]
        os_ = [torch.empty((B, 3, 512, 512), device=dev) for _ in range(nbuf)]
        for i in range(2 * nbuf): e.forward(fs[i % nbuf], c, os_[i % nbuf])
        torch.cuda.synchronize(); t0 = time.perf_counter(); n = 48
        for i in range(n): e.forward(fs[i % nbuf], c, os_[i % nbuf])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        print("batch %d, %2d buffer pairs in rotation: %.3f ms per forward" % (B, nbuf, 1e3 * dt), flush=True)
