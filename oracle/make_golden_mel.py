#!/usr/bin/env python3
"""Mel fixtures from the REFERENCE's own code (container only: needs /root/reference): funcs/utils.compute_mel_one_sequence ->
funcs/audio_funcs.Audio2Mel, executed unmodified with two shims for what this image lacks (see oracle/mel_oracle.py):
  * torch.stft is given return_complex=False -- what torch 1.7.1 (cog.yaml:9) did by default; torch 2.10 refuses to guess;
  * librosa (requirements.txt: 0.7.0, absent here) is a stub whose filters.mel is oracle/mel_oracle.slaney_mel_filterbank.
Asserts the restatement bit-identical to the reference functions, then freezes the mel rows.  The filterbank itself stays
parity-unpinned.  `--out` selects the directory."""
import argparse
import json
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)

from livespeechportraits_amd import synth      # noqa: E402
from oracle import mel_oracle                   # noqa: E402

CASES = {"mel_speechlike_1p5s": (24000, 5), "mel_short_tail": (16000 + 777, 6), "mel_silence_and_clicks": (8000, 7)}


def make_audio(n, seed):
    """deterministic, speech-like: a few drifting harmonics + noise bursts, in (-1, 1), float32 (librosa.load returns float32)"""
    t = np.arange(n) / 16000.0
    u = synth.uniform01(n, seed).astype(np.float64)
    f0 = 120 + 40 * np.sin(2 * np.pi * 1.3 * t + seed)
    x = sum(np.sin(2 * np.pi * k * np.cumsum(f0) / 16000.0) / k for k in range(1, 9))
    env = 0.5 * (1 + np.sin(2 * np.pi * 3.1 * t)) ** 2
    x = 0.2 * env * x + 0.05 * (u - 0.5) * (np.sin(2 * np.pi * 0.7 * t) > 0)
    if seed == 7:
        x[:] = 0.0
        x[1000::1777] = 0.9
    return x.astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    a = ap.parse_args()
    if not os.path.isdir(REF):
        raise SystemExit("make_golden_mel.py needs /root/reference (build container only)")
    lib = types.ModuleType("librosa")
    lib.filters = types.ModuleType("librosa.filters")
    lib.filters.mel = lambda sr, n_fft, n_mels, fmin, fmax: mel_oracle.slaney_mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
    sys.modules["librosa"], sys.modules["librosa.filters"] = lib, lib.filters
    for name in ("cv2", "scipy.io", "albumentations", "skimage", "skimage.io"):
        sys.modules.setdefault(name, types.ModuleType(name))
    real_stft = torch.stft
    torch.stft = lambda *args, **kw: real_stft(*args, **dict({"return_complex": False}, **kw))      # only fills in the missing argument
    sys.path.insert(0, REF)
    from funcs import audio_funcs                    # the reference module
    os.makedirs(a.out, exist_ok=True)
    try:
        from funcs import utils
        ref_fn = utils.compute_mel_one_sequence
        how = "funcs.utils.compute_mel_one_sequence"
    except Exception as e:                            # utils.py imports more of the absent stack (sklearn, cv2 ...): fall back to its loop
        print("funcs.utils not importable here (%s): driving the reference Audio2Mel with utils.py:61-83's loop" % type(e).__name__)
        how = "funcs.audio_funcs.Audio2Mel driven by the loop of utils.py:61-83"

        def ref_fn(audio, device="cpu"):
            m = audio_funcs.Audio2Mel(n_fft=512, hop_length=int(16000 / 120), win_length=int(16000 / 60), sampling_rate=16000,
                                      n_mel_channels=80, mel_fmin=90, mel_fmax=7600.0)
            nframe = int(audio.shape[0] / 16000 * 60)
            out = np.zeros([2 * nframe, 80])
            for i in range(2 * nframe):
                st = int(i * (16000 * 0.5 / 60))
                clip = audio[st: st + 266]
                if len(clip) < 266:
                    clip = np.concatenate([clip, np.zeros([266 - len(clip)])])
                out[i] = m(torch.from_numpy(clip).unsqueeze(0).unsqueeze(0).float()).cpu().numpy()[0].T
            return out
    for name, (n, seed) in CASES.items():
        audio = make_audio(n, seed)
        ref = ref_fn(audio)
        ora = mel_oracle.compute_mel_one_sequence(audio)
        assert ref.shape == ora.shape and np.array_equal(ref, ora), np.abs(ref - ora).max()
        print("%-24s %6d samples -> %s, range [%.3f, %.3f]; restatement bit-identical to %s" % (name, n, ref.shape, ref.min(), ref.max(), how))
        np.savez_compressed(os.path.join(a.out, name + ".npz"), mel=ref.astype(np.float32))
        json.dump({"samples": n, "seed": seed, "reference_entry": how, "torch": torch.__version__}, open(os.path.join(a.out, name + ".json"), "w"))
    fb = mel_oracle.slaney_mel_filterbank()
    print("filterbank %s: row sums %.4g..%.4g, nonzero %d (parity-unpinned: librosa absent)" % (fb.shape, fb.sum(1).min(), fb.sum(1).max(), (fb > 0).sum()))


if __name__ == "__main__":
    main()
