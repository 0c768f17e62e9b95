#!/usr/bin/env python3
"""Frames/s of the batched render loop (render_loop.render_frames: H2D of the feature maps, render + tensor2im, D2H of uint8 frames) with 1 / 2 / 3 batches in flight (GPU).
  python tools/render_loop_time.py [variant] [fp16: 0 | 1] [batch] [frames]"""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import livespeechportraits_amd as L
from livespeechportraits_amd import synth
from livespeechportraits_amd.render_loop import render_frames
from livespeechportraits_amd.topology import build_topology
variant = sys.argv[1] if len(sys.argv) > 1 else "large"
fp16 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 8
nframes = int(sys.argv[4]) if len(sys.argv) > 4 else 256
dev = torch.device("cuda:0")
topo = build_topology(variant)
sd = synth.make_state_dict(topo, 1234)
opt = argparse.Namespace(model="feature2face", gpu_ids=[0], isTrain=False, size=variant, ngf=64, n_downsample_G=8, fp16=fp16, checkpoints_dir="/tmp", name="t", load_epoch="none", verbose=False)
model = L.create_model(opt)
model._g().load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()})
model.eval()
feats, cand = synth.make_inputs(batch, 512, seed=5, cand_batch=1)
c = torch.from_numpy(cand).to(dev)
maps = [torch.from_numpy(feats[i % batch]).pin_memory() for i in range(nframes)]
ref = None
for lanes in (1, 2, 3, 1, 2, 4):
    render_frames(model, iter(maps[:4 * batch]), c, batch=batch, streams=lanes)          # warm-up: handles, graphs, pinned buffers
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = render_frames(model, iter(maps), c, batch=batch, streams=lanes)
    dt = time.perf_counter() - t0
    if ref is None: ref = out
    same = all(np.array_equal(a, b) for a, b in zip(out, ref))
    print("%s fp16=%d, render loop, batch %d, %d batch(es) in flight: %.1f frames/s (%d frames in %.3f s, H2D + render + tensor2im + D2H); same frames: %s" % (variant, fp16, batch, lanes, nframes / dt, nframes, dt, same), flush=True)
