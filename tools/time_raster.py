#!/usr/bin/env python3
"""Kernel time of lspraster_edge_maps from a graph of 20 back-to-back launches (GPU)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livespeechportraits_amd.feature_map import FeatureMapRasteriser
dev = torch.device("cuda:0"); r = FeatureMapRasteriser(512, 18, dev)
rng = np.random.default_rng(0)
for name, sd in (("random long edges (sigma 61 px)", 0.12), ("face-sized spread (sigma 30 px)", 0.06), ("all edges in one band (sigma 10 px)", 0.02)):
    lm = (256 + rng.normal(0, 512 * sd, (8, 73, 2))).astype(np.float32)
    sh = np.tile(np.stack([np.linspace(0, 512, 18), np.full(18, 460.)], 1)[None], (8, 1, 1)).astype(np.float32)
    pts = torch.from_numpy(np.concatenate([lm, sh], 1)).to(dev).contiguous(); out = torch.empty(8, 1, 512, 512, device=dev)
    for b in (1, 8):
        p = pts[:b].contiguous(); r.rasterise_points(p, out=out[:b]); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20): r.rasterise_points(p, out=out[:b])
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
        for _ in range(5): g.replay()
        e1.record(); torch.cuda.synchronize(); t = e0.elapsed_time(e1) * 10
        print("%-38s batch %d: %6.1f us per launch, output written at %.2f TB/s" % (name, b, t, b * 1048576 / t / 1e6))
