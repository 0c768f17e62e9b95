#!/usr/bin/env python3
"""Where a render-loop iteration's time goes (GPU): each step of render_loop.render_frames alone, synchronised, batch B.   python tools/render_loop_breakdown.py [batch]"""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import livespeechportraits_amd as L
from livespeechportraits_amd import synth
from livespeechportraits_amd.topology import build_topology
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
topo = build_topology("large")
sd = synth.make_state_dict(topo, 1234)
opt = argparse.Namespace(model="feature2face", gpu_ids=[0], isTrain=False, size="large", ngf=64, n_downsample_G=8, fp16=0, checkpoints_dir="/tmp", name="t", load_epoch="none", verbose=False)
model = L.create_model(opt)
model._g().load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()})
model.eval()
feats, cand = synth.make_inputs(B, 512, seed=5, cand_batch=1)
c = torch.from_numpy(cand).to(dev)
maps = [torch.from_numpy(feats[i]).pin_memory() for i in range(B)]
stage = torch.empty((B, 1, 512, 512), pin_memory=True)
devin = torch.empty((B, 1, 512, 512), device=dev)
u8 = torch.empty((B, 512, 512, 3), dtype=torch.uint8, device=dev)
host = torch.empty((B, 512, 512, 3), dtype=torch.uint8, pin_memory=True)
model.inference_image(devin, c, out=u8); torch.cuda.synchronize()
def t(fn, n=50):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
def stage_fn():
    for k in range(B): stage[k].copy_(maps[k])
print("batch %d, ms per iteration:" % B)
print("  gather the maps into pinned memory      %.3f" % t(stage_fn))
print("  H2D of the maps (pinned, non_blocking)  %.3f" % t(lambda: devin.copy_(stage, non_blocking=True)))
print("  torch.stack + pageable .to(device)       %.3f" % t(lambda: torch.stack(maps).to(dev, torch.float32, non_blocking=True)))
print("  inference_image(out=own buffer)          %.3f" % t(lambda: model.inference_image(devin, c, out=u8)))
print("  inference_image (fresh result tensor)    %.3f" % t(lambda: model.inference_image(devin, c)))
e = model._g().netG._engine
print("  engine.forward_image(out_u8=own buffer)  %.3f" % t(lambda: e.forward_image(devin, c, out_u8=u8)))
print("  D2H of the frames (pinned, non_blocking) %.3f" % t(lambda: host.copy_(u8, non_blocking=True)))
print("  numpy copies of the frames               %.3f" % t(lambda: [host[k].numpy().copy() for k in range(B)]))
print("  a new pinned result tensor               %.3f" % t(lambda: torch.empty((B, 512, 512, 3), dtype=torch.uint8, pin_memory=True)))
print("  torch.cuda.Event record + synchronize    %.3f" % t(lambda: (lambda ev: (ev.record(), ev.synchronize()))(torch.cuda.Event())))
