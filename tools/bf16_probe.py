import sys, json, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from conftest import golden_problem
from livespeechportraits_amd.engine import Engine
dev = torch.device("cuda:0")
for case in ("normal_512", "large_512"):
    meta, arrays, topo, sd, feat, cand = golden_problem(case)
    e = Engine(topo.variant, size=512, max_batch=8, dtype="bf16")
    e.load_state_dict(sd); e.bind(e.pack(), dev)
    f, c = torch.from_numpy(feat).to(dev), torch.from_numpy(cand).to(dev)
    out = e.forward(f, c).cpu().numpy()
    d = np.abs(out - arrays["out"])
    print(case, "bf16 vs fp32 reference: max-abs %.4g mean-abs %.4g p99.9 %.4g  (out std %.3f)" % (d.max(), d.mean(), np.quantile(d, 0.999), arrays["out"].std()))
    from livespeechportraits_amd import synth
    f8 = torch.from_numpy(synth.make_inputs(8, 512, 99, 1)[0]).to(dev)
    o8 = torch.empty(8, 3, 512, 512, device=dev)
    for b, ff, oo in ((1, f, None), (8, f8, o8)):
        for _ in range(5): e.forward(ff, c, oo)
        torch.cuda.synchronize(); t = time.perf_counter(); n = 30
        for _ in range(n): e.forward(ff, c, oo)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        print("   batch %d: %.1f frames/s  (%.3f ms/step, %.0f TFLOP/s algorithmic)" % (b, b * n / dt, 1e3 * dt / n, topo.flops_per_frame() * b * n / dt / 1e12))
