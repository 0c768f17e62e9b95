#!/usr/bin/env python3
"""Pins the mel filterbank on REAL librosa the moment it is importable (it is not in this image: the filterbank VALUES of the mel
front-end are UNPINNED until this script has run somewhere and its fixture is committed).

  python oracle/make_golden_melfb.py [--out tests/golden]

Calls what the reference calls -- funcs/audio_funcs.py:33-35: librosa.filters.mel(sr=16000, n_fft=512, n_mels=80, fmin=90, fmax=7600)
(`librosa_mel_fn(sampling_rate, n_fft, n_mel_channels, mel_fmin, mel_fmax)`, positional in librosa 0.7.0, keywords here so that newer
versions accept it) -- checks oracle/mel_oracle.slaney_mel_filterbank against it, and writes tests/golden/melfb_librosa.npz + .json.
tests/test_mel.py::test_filterbank_matches_real_librosa compares the oracle and the library's host copy with the fixture when it exists
and reports "unpinned" (xfail) when it does not.  If librosa is missing the script says so and exits 2."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    a = ap.parse_args()
    try:
        import librosa
    except ImportError:
        print("make_golden_melfb: librosa is not importable here -- nothing written; the mel filterbank stays parity-unpinned "
              "(requirements.txt of the reference pins librosa==0.7.0)", file=sys.stderr)
        return 2
    from oracle import mel_oracle
    fb = np.asarray(librosa.filters.mel(sr=16000, n_fft=512, n_mels=80, fmin=90.0, fmax=7600.0), np.float32)
    ours = mel_oracle.slaney_mel_filterbank()
    err = float(np.abs(fb - ours).max())
    print("oracle/mel_oracle.slaney_mel_filterbank vs librosa %s: max-abs %.3e (filter peak %.3e)" % (librosa.__version__, err, float(fb.max())))
    os.makedirs(a.out, exist_ok=True)
    np.savez_compressed(os.path.join(a.out, "melfb_librosa.npz"), fb=fb)
    json.dump({"librosa_version": librosa.__version__, "oracle_max_abs": err,
               "generator": "oracle/make_golden_melfb.py: librosa.filters.mel(sr=16000, n_fft=512, n_mels=80, fmin=90, fmax=7600), funcs/audio_funcs.py:33-35"},
              open(os.path.join(a.out, "melfb_librosa.json"), "w"), indent=1)
    return 0 if err <= 1e-6 else 1


if __name__ == "__main__":
    sys.exit(main())
