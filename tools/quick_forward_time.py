#!/usr/bin/env python3
"""Whole-forward time of the bench default (large, batch 1, fp32), graph replays between two events -- the quickest way to tell one process environment from another on the same box
(tools/sessions/gpu_r5_numa.sh: kernarg placement, CPU / memory binding).  Prints one line."""
import os, sys, time
t_imp = time.perf_counter()
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livespeechportraits_amd import synth
from livespeechportraits_amd.engine import Engine
from livespeechportraits_amd.topology import build_topology
label = sys.argv[1] if len(sys.argv) > 1 else "default"
dev = torch.device("cuda:0")
topo = build_topology("large")
sd = synth.make_state_dict(topo, 1234)
e = Engine("large", max_batch=1)
e.load_state_dict(sd)
e.bind(e.pack(), dev)
f, c = synth.make_inputs(1, 512, 99, 1)
f, c = torch.from_numpy(f).to(dev), torch.from_numpy(c).to(dev)
out = torch.empty((1, 3, 512, 512), device=dev)
for _ in range(30): e.forward(f, c, out)
torch.cuda.synchronize()
res = []
for _ in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(300): e.forward(f, c, out)
    b.record(); torch.cuda.synchronize()
    res.append(a.elapsed_time(b) / 300)
cpu = os.sched_getaffinity(0)
print("%-44s %.4f ms / forward (%.1f frames/s; runs %s); cpus allowed %d (first %d), running on cpu %s" % (
    label, min(res), 1e3 / min(res), " ".join("%.4f" % r for r in res), len(cpu), min(cpu), open("/proc/self/stat").read().split()[38]), flush=True)
