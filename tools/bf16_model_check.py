#!/usr/bin/env python3
"""GPU bf16 path vs the exact bf16 storage model (oracle/bf16_model.py) vs the fp32 reference golden: distribution of
absolute differences on the full-size goldens.  python tools/bf16_model_check.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden_problem
from livespeechportraits_amd.engine import Engine
from oracle import bf16_model, torch_oracle

dev = torch.device("cuda:0")
for case in ("normal_512", "large_512"):
    meta, arrays, topo, sd, feat, cand = golden_problem(case)
    e = Engine(topo.variant, size=topo.size, max_batch=1, dtype="bf16")
    e.load_state_dict(sd); e.bind(e.pack(), dev)
    gpu = e.forward(torch.from_numpy(feat).to(dev), torch.from_numpy(cand).to(dev)).cpu().numpy()
    x = torch.cat([torch.from_numpy(feat), torch.from_numpy(cand)], 1)
    nres = 2 if topo.variant == "large" else 1
    model = bf16_model.generator_forward_bf16(torch_oracle.to_torch(sd), x, nres, topo.num_downs).numpy()
    for name, a, b in (("GPU bf16 vs bf16 model", gpu, model), ("GPU bf16 vs fp32 reference", gpu, arrays["out"]), ("bf16 model vs fp32 reference", model, arrays["out"])):
        d = np.abs(a - b).ravel()
        print("%-11s %-30s median %.2e  p99 %.2e  p99.9 %.2e  max %.2e  mean %.2e  frac>1e-4: %.4f" %
              (case, name, np.median(d), np.percentile(d, 99), np.percentile(d, 99.9), d.max(), d.mean(), (d > 1e-4).mean()))
