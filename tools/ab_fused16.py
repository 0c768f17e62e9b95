"""A-B of the tune key fused_splitk16 (16-bit plans: 2..8 K-splits combined inside the igemm launch): outputs must be bit-identical (same z order,
same epilogue), time per forward of both arms interleaved.  usage: python tools/ab_fused16.py [variant] [batch] [dtype]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from livespeechportraits_amd import synth
from livespeechportraits_amd.engine import Engine
from livespeechportraits_amd.topology import build_topology

variant = sys.argv[1] if len(sys.argv) > 1 else "normal"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dtype = sys.argv[3] if len(sys.argv) > 3 else "bf16"
dev = torch.device("cuda:0")
topo = build_topology(variant)
sd = synth.scale_last_conv(synth.make_state_dict(topo, 1234), topo, 0.05)
feat, cand = synth.make_inputs(batch, 512, 99, 1)
f, c = torch.from_numpy(feat).to(dev), torch.from_numpy(cand).to(dev)
arms = {}
for key in (0, 1):
    e = Engine(variant, dtype=dtype, max_batch=batch, tune={"fused_splitk16": key})
    e.load_state_dict(sd)            # returns the ignored num_batches_tracked keys
    e.bind(e.pack(), dev)
    arms[key] = (e, e.forward(f, c).clone())
print("bit-identical:", torch.equal(arms[0][1], arms[1][1]), "max-abs diff %.3e" % (arms[0][1].float() - arms[1][1].float()).abs().max().item())
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(2):
    for key in (0, 1):
        e = arms[key][0]
        for _ in range(20): e.forward(f, c)
        torch.cuda.synchronize(); t0.record()
        for _ in range(200): e.forward(f, c)
        t1.record(); torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / 200
        print("fused_splitk16=%d: %.4f ms / forward  (%.1f frames/s)" % (key, ms, batch / ms * 1e3))
