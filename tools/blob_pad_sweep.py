#!/usr/bin/env python3
"""Does a layer's time depend on WHERE its weights sit relative to the workspace (HBM channel interleave)?  Per-layer eager timings (median of 11) of the
`large` fp32 batch-1 plan for several paddings in front of the packed blob.  python tools/blob_pad_sweep.py [pad_kb ...]"""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livespeechportraits_amd import synth, distributed as D
from livespeechportraits_amd.engine import Engine
from livespeechportraits_amd.topology import build_topology

pads = [int(a) for a in sys.argv[1:]] or [0, 64, 128, 192, 256, 320, 384, 448, 1, 4, 16]
dev = torch.device("cuda:0")
topo = build_topology("large", size=512)
sd = synth.make_state_dict(topo, 1234)
f, c = synth.make_inputs(1, 512, seed=99, cand_batch=1)
feat, cand = torch.from_numpy(f).to(dev), torch.from_numpy(c).to(dev)
rows, names, whole = {}, None, {}
for pad in pads:
    e = Engine("large", size=512, max_batch=1, tune={"blob_pad_kb": pad})
    D.setup_engine(e, sd, dev)
    out = torch.empty((1, 3, 512, 512), device=dev)
    for _ in range(5): e.forward(feat, cand, out)
    per = [e.forward_timed(feat, cand, out)[1] for _ in range(11)]
    rows[pad] = [statistics.median(x[i] for x in per) * 1e3 for i in range(len(per[0]))]
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(200): e.forward(feat, cand, out)
    ev1.record(); torch.cuda.synchronize()
    whole[pad] = ev0.elapsed_time(ev1) / 200 * 1e3
    names = [l["name"] + " " + l["kernel"].split(" ")[0] for l in e.layers(1)]
    print("blob at %#x, workspace at %#x, pad %d KB: whole forward %.1f us" % (e._blob_dev.data_ptr(), e._ws.data_ptr(), pad, whole[pad]), flush=True)
    e.close()
print("%-34s" % "layer" + "".join("%8d" % p for p in pads) + "   spread")
for i, n in enumerate(names):
    v = [rows[p][i] for p in pads]
    if max(v) - min(v) > 1.0:
        print("%-34s" % n + "".join("%8.1f" % x for x in v) + "   %5.1f" % (max(v) - min(v)))
print("%-34s" % "whole forward (graph replay)" + "".join("%8.1f" % whole[p] for p in pads))
