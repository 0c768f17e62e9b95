#!/usr/bin/env python3
"""Throughput of N CONCURRENT batch-1 forwards (N engines on one packed blob, one stream each) against one stream (GPU).  A batch-1 forward is a chain of 79 dependent launches,
every one a single round of workgroups: its prologues, tails and kernel boundaries overlap nothing.  Independent frames on separate streams can fill those holes.
  python tools/multistream_probe.py [variant] [dtype] [max_streams] [batch] [streams created (and used once) before the lanes' own: 0]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livespeechportraits_amd import synth
from livespeechportraits_amd.engine import Engine
from livespeechportraits_amd.topology import build_topology

variant = sys.argv[1] if len(sys.argv) > 1 else "large"
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 4
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device("cuda:0")
pre = int(sys.argv[5]) if len(sys.argv) > 5 else 0
MB = int(sys.argv[6]) if len(sys.argv) > 6 else 0            # handles planned for up to this many frames (0: the batch itself) -- bench.py's are planned for 8
dummies = [torch.cuda.Stream(dev) for _ in range(pre)]          # does a process's stream history change what the lanes get?  (bench.py measures this late in its life)
for d in dummies:
    with torch.cuda.stream(d):
        torch.zeros(16, device=dev).add_(1)
torch.cuda.synchronize()
topo = build_topology(variant)
sd = synth.make_state_dict(topo, 1234)
first = Engine(variant, dtype=dtype, max_batch=max(B, MB))
first.load_state_dict(sd)
blob = first.pack().to(dev)
engines, streams, ins, outs = [], [], [], []
for i in range(nmax):
    e = first if i == 0 else Engine(variant, dtype=dtype, max_batch=max(B, MB))
    e.bind(blob, dev)                                               # one copy of the weights for all of them
    f, c = synth.make_inputs(B, 512, 99 + i, 1)
    engines.append(e); streams.append(torch.cuda.Stream(dev))
    ins.append((torch.from_numpy(f).to(dev), torch.from_numpy(c).to(dev)))
    outs.append(torch.empty((B, 3, 512, 512), device=dev))
torch.cuda.synchronize()
ref = [engines[i].forward(*ins[i]).clone() for i in range(nmax)]
torch.cuda.synchronize()
for n in range(1, nmax + 1):
    def run(reps):
        for _ in range(reps):
            for i in range(n):
                with torch.cuda.stream(streams[i]):
                    engines[i].forward(ins[i][0], ins[i][1], outs[i])
    run(10); torch.cuda.synchronize()
    reps = 200 if B == 1 else 40
    t0 = time.perf_counter(); run(reps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    same = all(torch.equal(outs[i], ref[i]) for i in range(n))
    print(("[%d streams created before] " % pre if pre else "") + ("[handles planned for %d frames] " % MB if MB else "") + "%s %s batch %d, %d concurrent stream(s): %.1f frames/s (%.4f ms per frame); outputs bit-identical to the single-stream run: %s" % (variant, dtype, B, n, B * n * reps / dt, 1e3 * dt / (B * n * reps), same), flush=True)
