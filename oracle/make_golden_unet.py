#!/usr/bin/env python3
"""ORACLE tooling -- freeze outputs of the reference's OWN Feature2FaceGenerator_Unet (size == 'small') into
tests/golden/unet_*.npz.  Container only.  Asserts oracle/unet_small_oracle.py is bit-identical to the module."""
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

from livespeechportraits_amd import synth               # noqa: E402
from oracle import unet_small_oracle                     # noqa: E402

# name -> (ngf, num_downs, size, batch)
CASES = {"small_512": (64, 8, 512, 1),                   # feature2face_G.py:16-17 with the option defaults
         "small_s64_b2": (32, 5, 64, 2)}


def main():
    if not os.path.isdir(REF):
        raise SystemExit("make_golden_unet.py needs /root/reference (build container only)")
    for name in ("torchvision", "torchvision.models", "cv2"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path.insert(0, REF)
    from models.networks import Feature2FaceGenerator_Unet
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"), help="directory the fixtures are written to")
    out = ap.parse_args().out
    os.makedirs(out, exist_ok=True)
    for name, (ngf, nd, size, batch) in CASES.items():
        net = Feature2FaceGenerator_Unet(input_nc=23, output_nc=3, num_downs=nd, ngf=ngf).eval()
        ref_keys = {k: list(v.shape) for k, v in net.state_dict().items()}
        sd = synth.make_unet_small_state_dict(23, 3, nd, ngf)
        assert {k: v for k, v in ref_keys.items() if not k.endswith("num_batches_tracked")} == {k: list(v.shape) for k, v in sd.items()}, "key map differs"
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        x = torch.from_numpy(synth.symmetric(batch * 23 * size * size, 0.6, synth._stream(5, name)).reshape(batch, 23, size, size))
        with torch.no_grad():
            ref = net(x.clone())                                   # clone: the reference modifies its input in place? (it does not at depth 0)
        ora = unet_small_oracle.generator_forward({k: torch.from_numpy(v) for k, v in sd.items()}, x, nd)
        assert torch.equal(ora, ref), (ora - ref).abs().max()
        r = ref.numpy()
        print("%-13s ngf %d downs %d size %d batch %d: |out| max %.3f std %.3f, saturated %.4f; oracle bit-exact"
              % (name, ngf, nd, size, batch, np.abs(r).max(), r.std(), (np.abs(r) > 0.99).mean()))
        assert (np.abs(r) > 0.99).mean() < 0.01
        np.savez_compressed(os.path.join(out, "unet_%s.npz" % name), out=r)
        json.dump({"ngf": ngf, "num_downs": nd, "size": size, "batch": batch, "weights_seed": 97, "keys": ref_keys},
                  open(os.path.join(out, "unet_%s.json" % name), "w"), indent=0)


if __name__ == "__main__":
    main()
