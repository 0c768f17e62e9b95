#!/usr/bin/env python3
"""The first layer alone (GPU): replayed from its own hipGraph (lspf2f_subset_timed), and as N back-to-back eager launches of the whole forward's first layer.
  python tools/first_conv_time.py [variant] [batch] [dtype]          (LSP_HIP_DBG picks the ablation arm of a -DLSPF2F_ABLATE build)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livespeechportraits_amd import synth
from livespeechportraits_amd import distributed as D
from livespeechportraits_amd.engine import Engine
from livespeechportraits_amd.topology import build_topology

def main():
    variant = sys.argv[1] if len(sys.argv) > 1 else "large"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    dtype = sys.argv[3] if len(sys.argv) > 3 else "f32"
    dev = torch.device("cuda:0")
    topo = build_topology(variant, size=512)
    eng = Engine(variant, size=512, max_batch=B, dtype=dtype)
    D.setup_engine(eng, synth.make_state_dict(topo, 1234), dev)
    f, c = synth.make_inputs(B, 512, seed=99, cand_batch=1)
    feat, cand = torch.from_numpy(f).to(dev), torch.from_numpy(c).to(dev)
    out = torch.empty((B, 3, 512, 512), device=dev)
    for _ in range(3): eng.forward(feat, cand, out)
    torch.cuda.synchronize()
    layers = eng.layers(B)
    sel = [0] * len(layers); sel[0] = 3
    one = sorted(eng.subset_timed(feat, cand, sel, out, reps=50) for _ in range(5))[2]
    sel2 = [0] * len(layers); sel2[0] = 3; sel2[1] = 3
    two = sorted(eng.subset_timed(feat, cand, sel2, out, reps=50) for _ in range(5))[2]
    sel3 = [0] * len(layers); sel3[1] = 3
    nxt = sorted(eng.subset_timed(feat, cand, sel3, out, reps=50) for _ in range(5))[2]
    print("dbg=%s %s b%d %s: %s alone %.2f us per replay; with the next layer %.2f us (the next layer alone %.2f)" % (os.environ.get("LSP_HIP_DBG", "0"), variant, B, dtype, layers[0]["kernel"], one * 1e3, two * 1e3, nxt * 1e3))

main()
