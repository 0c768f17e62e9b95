#!/usr/bin/env python3
"""Time the head-pose generator.  The clip of SURVEY.md 8d config 5 has 687 frames + frame_future 15 = 702 audio rows.
Reports, per kernel, the loop time at several clip lengths and the least-squares fit  loop_ms = fill + per_frame * nframe
(fill = launch + filling the first receptive field; per_frame = one autoregressive step).
  python tools/a2h_time.py [--reps 5] [--kernels pipeline,single]"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livespeechportraits_amd import synth
from livespeechportraits_amd.a2h_engine import HeadposeEngine

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--kernels", default="pipeline,single")
ap.add_argument("--frames", default="87,237,387,687")
a = ap.parse_args()
dev = torch.device("cuda:0"); cfg = dict(synth.A2H_DEFAULTS); ff = 15
lengths = [int(x) for x in a.frames.split(",")]
sd = synth.make_a2h_state_dict(cfg)
for kernel in a.kernels.split(","):
    e = HeadposeEngine(max_audio_frames=max(lengths) + ff, single_workgroup=kernel == "single"); e.load_state_dict(sd); e.bind(dev)
    xs, ys = [], []
    for nframe in lengths:
        audio, pre = synth.make_a2h_inputs(nframe + ff, cfg)
        au, pr = torch.from_numpy(audio).to(dev), torch.from_numpy(pre).to(dev)
        noise = torch.randn(nframe, 12).to(dev)
        e.generate_timed(au, pr, noise, None, 0.3, ff)                       # warm-up
        t = [e.generate_timed(au, pr, noise, None, 0.3, ff)[1:] for _ in range(a.reps)]
        assert e.status() == 0
        loops = sorted(x[1] for x in t)
        print("%-8s nframe %4d: precompute %.3f ms, loop min/median/max %.3f / %.3f / %.3f ms over %d reps"
              % (kernel, nframe, min(x[0] for x in t), loops[0], loops[len(loops) // 2], loops[-1], a.reps))
        xs += [nframe] * a.reps; ys += [x[1] for x in t]
    slope, icpt = np.polyfit(xs, ys, 1)
    print("%-8s fit: loop_ms = %.3f + %.5f * nframe  ->  %.1f us per autoregressive frame, %.2f ms fill; 687-frame clip: %.0f head poses/s"
          % (kernel, icpt, slope, 1e3 * slope, icpt, 687 / ((icpt + slope * 687) * 1e-3)))
