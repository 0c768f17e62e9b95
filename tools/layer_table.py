#!/usr/bin/env python3
"""Per-layer table of one plan (GPU): every layer replayed alone from its own hipGraph (lspf2f_subset_timed), with the MFMA issue time its
executed FLOPs need at the measured clock beside it.   python tools/layer_table.py [variant] [batch] [dtype]"""
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
    per_cycle = 65536 if dtype == "f32" else 1048576      # MFMA flops per shader cycle, whole chip (dense)
    tot = 0.0
    print("%-16s %-28s %5s %5s %4s %9s %3s %6s %8s %8s" % ("layer", "kernel", "cin", "cout", "h", "tile", "sk", "blocks", "us", "TF/s ex"))
    for i, l in enumerate(layers):
        sel = [0] * len(layers); sel[i] = 3
        ms = eng.subset_timed(feat, cand, sel, out, reps=20)
        tot += ms
        ex = l["exec_flops_per_frame"] * B
        m = B * (l["h_in"] ** 2 if l["upsample"] else l["h_out"] ** 2)
        blocks = 0
        if l["tile_m"] and l["tile_n"]:
            blocks = -(-m // l["tile_m"]) * -(-l["cout"] // l["tile_n"]) * (4 if l["upsample"] else 1) * max(1, l["split_k"])
        print("%-16s %-28s %5d %5d %4d %4dx%-4d %3d %6d %8.2f %8.1f" % (l["name"], l["kernel"][:28], l["cin"], l["cout"], l["h_out"], l["tile_m"], l["tile_n"],
              l["split_k"], blocks, ms * 1e3, ex / (ms * 1e-3) / 1e12))
    print("sum of single-layer replays: %.3f ms" % tot)

main()
