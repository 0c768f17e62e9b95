#!/usr/bin/env python3
"""Time the `size == 'small'` generator at 512x512: the native plan (include/lspunet.h) with each of its A-B arms, the host-sequenced form it replaced (round 3/4),
a per-launch table of the native plan, and the oracle on the host.  Usage: unet_small_time.py [--no-oracle]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livespeechportraits_amd import synth
from livespeechportraits_amd.unet_small import HostSequencedUnetEngine, SmallUnetEngine
dev = torch.device("cuda:0")
sd = synth.make_unet_small_state_dict()


def clock(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def burst(fn, reps=50):
    """device time per forward without the host in between: `reps` forwards enqueued back to back between two events"""
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


xs = {B: torch.from_numpy(synth.symmetric(B * 23 * 512 * 512, 0.6, 3).reshape(B, 23, 512, 512)).to(dev) for B in (1, 8)}
ref = {}
for label, mk in (("host-sequenced (round 4)", lambda: HostSequencedUnetEngine()),
                  ("native plan", lambda: SmallUnetEngine()),
                  ("native, last_direct=0", lambda: SmallUnetEngine(tune="last_direct=0")),
                  ("native, last_direct=0 64x64", lambda: SmallUnetEngine(tune="last_direct=0,last_tile=-1")),
                  ("native, dense0=1", lambda: SmallUnetEngine(tune="dense0=1")),
                  ("native, tiny=0", lambda: SmallUnetEngine(tune="tiny=0")),
                  ("native, fused_prepare=0", lambda: SmallUnetEngine(tune="fused_prepare=0")),
                  ("native, input_pass=0", lambda: SmallUnetEngine(tune="input_pass=0")),
                  ("native, fused_splitk=1", lambda: SmallUnetEngine(tune="fused_splitk=1")),
                  ("native, graph=0", lambda: SmallUnetEngine(tune="graph=0")),
                  ("native plan (again)", lambda: SmallUnetEngine()),
                  ("native plan, fp16 storage", lambda: SmallUnetEngine(dtype="f16"))):
    e = mk(); e.load_state_dict(sd, "model", dev)
    for B in (1, 8):
        x = xs[B]
        out = e.forward(x)
        same = ""
        if B in ref and "fp16" in label: same = " max-abs %.2e vs the fp32 host-sequenced form" % (out - ref[B]).abs().max().item()
        elif B in ref: same = " bit-identical to the host-sequenced form" if torch.equal(out, ref[B]) else " DIFFERS from the host-sequenced form by %.2e" % (out - ref[B]).abs().max().item()
        ref.setdefault(B, out.clone())
        med, mn = clock(lambda: e.forward(x))
        dv = burst(lambda: e.forward(x))
        print("small generator, %-26s batch %d: %.3f ms per forward with a sync each (min %.3f), %.3f ms back to back = %.0f frames/s;%s" % (label + ",", B, 1e3 * med, 1e3 * mn, dv, B / dv * 1e3, same), flush=True)
    if hasattr(e, "close"): e.close()

for dt in ("f32", "f16"):
  e = SmallUnetEngine(dtype=dt); e.load_state_dict(sd, "model", dev)
  for B in (1, 8):
    rows = e.launches(512, B)
    acc = np.zeros(len(rows))
    for _ in range(5):
        ms = []
        e.render(xs[B], None, timed=ms)
        acc += np.array(ms)
    acc /= 5
    print("\nper launch (eager, one event pair each), %s, batch %d: sum %.3f ms" % (dt, B, acc.sum()))
    for r, t in zip(rows, acc):
        print("  %-12s %-72s tile %3dx%-3d splits %-3d %8.1f us" % (r["name"], r["kernel"], r["tile"][0], r["tile"][1], r["split_k"], 1e3 * t))
  e.close()

if "--no-oracle" not in sys.argv:
    from oracle import unet_small_oracle
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    x = xs[1].cpu()
    unet_small_oracle.generator_forward(sdt, x); t0 = time.perf_counter()
    for _ in range(3): unet_small_oracle.generator_forward(sdt, x)
    print("oracle on the host (%d threads): %.1f ms per frame" % (torch.get_num_threads(), 1e3 * (time.perf_counter() - t0) / 3))
