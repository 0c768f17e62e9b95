#!/usr/bin/env python3
"""Which layers change when the stride-2 convs of the small levels run on the K-split full-K kernel (tune key fullk_s2=1)?  Per-layer eager timings
(lspf2f_forward_timed: an event after every layer) of both plans, median of 15, and the whole forward by graph replay.  python tools/s2_layer_delta.py"""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livespeechportraits_amd import synth, distributed as D
from livespeechportraits_amd.engine import Engine
from livespeechportraits_amd.topology import build_topology

dev = torch.device("cuda:0")
topo = build_topology("large", size=512)
sd = synth.make_state_dict(topo, 1234)
f, c = synth.make_inputs(1, 512, seed=99, cand_batch=1)
feat, cand = torch.from_numpy(f).to(dev), torch.from_numpy(c).to(dev)
res = {}
for s2 in (0, 1, 2, 0, 1, 2):
    e = Engine("large", size=512, max_batch=1, tune={"fullk_s2": s2})
    D.setup_engine(e, sd, dev)
    out = torch.empty((1, 3, 512, 512), device=dev)
    for _ in range(5): e.forward(feat, cand, out)
    per = [e.forward_timed(feat, cand, out)[1] for _ in range(15)]
    med = [statistics.median(x[i] for x in per) * 1e3 for i in range(len(per[0]))]
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(200): e.forward(feat, cand, out)
    ev1.record(); torch.cuda.synchronize()
    res.setdefault(s2, []).append((med, ev0.elapsed_time(ev1) / 200 * 1e3, [l["name"] + " " + l["kernel"].split(" ")[0] for l in e.layers(1)]))
    e.close()
for run in range(2):
    for lvl in (1, 2):
        (m0, g0, n0), (m1, g1, n1) = res[0][run], res[lvl][run]
        print("run %d: whole forward (graph replay) fullk_s2=0 %.1f us, fullk_s2=%d %.1f us; sum of eager per-layer times %.1f -> %.1f" % (run, g0, lvl, g1, sum(m0), sum(m1)))
        for i, (a, b) in enumerate(zip(m0, m1)):
            if abs(a - b) > 1.5:
                print("   %-34s -> %-26s %7.2f -> %7.2f us (%+.2f)" % (n0[i], n1[i].split(" ")[1], a, b, b - a))
