#!/usr/bin/env python3
"""Time the head-pose generator on the demo clip's length (687 frames + frame_future 15 = 702 audio rows,
SURVEY.md 8d config 5).  python tools/a2h_time.py [--n-audio 702] [--reps 5]"""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livespeechportraits_amd import synth
from livespeechportraits_amd.a2h_engine import HeadposeEngine

ap = argparse.ArgumentParser(); ap.add_argument("--n-audio", type=int, default=702); ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0"); cfg = dict(synth.A2H_DEFAULTS); ff = 15
e = HeadposeEngine(max_audio_frames=a.n_audio); e.load_state_dict(synth.make_a2h_state_dict(cfg)); e.bind(dev)
audio, pre = synth.make_a2h_inputs(a.n_audio, cfg)
au, pr = torch.from_numpy(audio).to(dev), torch.from_numpy(pre).to(dev)
nframe = a.n_audio - ff
noise = torch.randn(nframe, 12).to(dev)
steps = e.receptive_field - 1 + nframe
for r in range(a.reps):
    out, pre_ms, loop_ms = e.generate_timed(au, pr, noise, None, 0.3, ff)
    print("rep %d: precompute %.3f ms, loop %.3f ms = %.1f us/step over %d steps (%d frames) -> %.0f head poses/s; weights streamed %.1f GB/s"
          % (r, pre_ms, loop_ms, 1e3 * loop_ms / steps, steps, nframe, nframe / ((pre_ms + loop_ms) * 1e-3),
             steps * 14 * 114688 * 4 / (loop_ms * 1e-3) / 1e9))
t0 = time.time(); o = e.generate(au, pr, noise, None, 0.3, ff); o = o.cpu(); print("end-to-end incl. D2H: %.1f ms" % (1e3 * (time.time() - t0)))
