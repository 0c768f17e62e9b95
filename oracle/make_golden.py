#!/usr/bin/env python3
"""ORACLE tooling -- freeze outputs of the REAL reference into tests/golden/.

Runs only in the build container, where /root/reference exists; the GPU box
never executes this file (it only reads the committed fixtures).

For every case below it
  1. builds the deterministic synthetic state dict (livespeechportraits_amd.synth),
  2. saves it as a '<path>.pkl' checkpoint with the 'module.' key prefix the
     reference's CPU loader strips (models/base_model.py:213-215),
  3. drives the reference's OWN code path end to end: models.create_model(opt)
     -> Feature2FaceModel.setup() (load_networks) -> .eval() -> .inference()
     (models/__init__.py:58-71, models/feature2face_model.py:225-237),
  4. asserts oracle/torch_oracle.py reproduces that output BIT-EXACTLY (same torch
     ops, same order), and that tanh is not saturated (SURVEY.md 8c warning),
  5. writes the reference output (+ strided taps of every level's block output)
     to tests/golden/<case>.npz and the reference's key->shape map to
     tests/golden/keys_<variant>.json.

Only two import-time stubs are needed to import the reference here: ``torchvision``
(models/losses.py:236, training-only VGG loss) and ``cv2`` (util/util.py:10).
"""
import argparse
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)

from livespeechportraits_amd import synth                       # noqa: E402
from livespeechportraits_amd.topology import build_topology     # noqa: E402
from oracle import torch_oracle                                  # noqa: E402

# name -> (variant, ngf, num_downs, size, batch, cand_batch)
CASES = {
    "large_512":  ("large", 64, 8, 512, 1, 1),     # BASELINE.json configs[0]/[1] (May)
    "normal_512": ("normal", 64, 8, 512, 1, 1),    # Obama1 architecture, fp32
    "large_s128_b2": ("large", 32, 6, 128, 2, 1),  # small: batch 2, candidates broadcast
    "normal_s64_b3": ("normal", 32, 5, 64, 3, 3),  # small: per-frame candidates
}


def import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("make_golden.py needs /root/reference (build container only)")
    for name in ("torchvision", "torchvision.models", "cv2"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path.insert(0, REF)
    import models  # noqa: F401  (the reference package)
    return models


def subsample(t: torch.Tensor, budget: int = 16384) -> np.ndarray:
    """[:, ::cs, ::s, ::s] with strides chosen so that the tap stays <= budget floats/frame."""
    _, c, h, _ = t.shape
    s = 1
    while (h // s) * (h // s) > 256 and s < h:
        s *= 2
    cs = 1
    while (c // cs) * (h // s) * (h // s) > budget:
        cs *= 2
    return t[:, ::cs, ::s, ::s].contiguous().numpy(), cs, s


def run_case(models, name, out_dir):
    variant, ngf, num_downs, size, batch, cand_batch = CASES[name]
    topo = build_topology(variant, ngf=ngf, num_downs=num_downs, size=size)
    sd_np = synth.make_state_dict(topo, seed=1234)
    feat_np, cand_np = synth.make_inputs(batch, size, seed=99, cand_batch=cand_batch)
    feat, cand = torch.from_numpy(feat_np), torch.from_numpy(cand_np)

    with tempfile.TemporaryDirectory() as tmp:
        ckpt = os.path.join(tmp, "Feature2Face.pkl")
        torch.save({"module." + k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items()},
                   ckpt)
        opt = argparse.Namespace(model="feature2face", gpu_ids=[], isTrain=False, size=variant,
                                 ngf=ngf, n_downsample_G=num_downs, fp16=0, checkpoints_dir=tmp,
                                 name="golden", load_epoch=ckpt, verbose=False, task="Feature2Face")
        torch.manual_seed(0)
        model = models.create_model(opt)
        model.setup(opt)
        model.eval()

    net = model.Feature2Face_G
    ref_sd = net.state_dict()
    # the checkpoint really was ingested (strict=False would hide a key mismatch)
    assert set(ref_sd.keys()) == set(sd_np.keys()), "key map mismatch vs reference"
    for k, v in sd_np.items():
        assert tuple(ref_sd[k].shape) == tuple(v.shape), k
        if v.dtype == np.float32:
            assert torch.equal(ref_sd[k], torch.from_numpy(v)), k

    # the reference broadcasts nothing: give it a per-frame candidate stack
    cand_full = cand if cand_batch == batch else cand.expand(batch, -1, -1, -1).contiguous()
    ref_out = model.inference(feat, cand_full)
    assert ref_out.shape == (batch, 3, size, size) and ref_out.dtype == torch.float32

    taps = {}
    ora_out = torch_oracle.inference(torch_oracle.to_torch(sd_np), feat, cand_full, topo.nres,
                                     num_downs, taps=taps)
    max_diff = (ora_out - ref_out).abs().max().item()
    assert max_diff == 0.0, "torch_oracle deviates from the reference: %g" % max_diff
    sat = (ref_out.abs() > 0.99).float().mean().item()
    assert sat < 0.01, "tanh saturated on %.1f%% of the output" % (100 * sat)

    arrays = {"out": ref_out.numpy()}
    meta = {"case": name, "variant": variant, "ngf": ngf, "num_downs": num_downs, "size": size,
            "batch": batch, "cand_batch": cand_batch, "weight_seed": 1234, "input_seed": 99,
            "torch": torch.__version__, "out_absmax": ref_out.abs().max().item(),
            "out_std": ref_out.std().item(), "taps": {}}
    for tname, t in taps.items():
        if tname == "pre_tanh":
            continue
        arr, cs, s = subsample(t)
        arrays["tap_" + tname] = arr
        meta["taps"][tname] = {"cstride": cs, "sstride": s, "shape": list(t.shape)}
    np.savez(os.path.join(out_dir, name + ".npz"), **arrays)
    with open(os.path.join(out_dir, name + ".json"), "w") as f:
        json.dump(meta, f, indent=1)

    keys = {k: list(v.shape) for k, v in ref_sd.items()}
    if ngf == 64 and num_downs == 8:
        with open(os.path.join(out_dir, "keys_%s.json" % variant), "w") as f:
            json.dump(keys, f, indent=0)
    print("%-16s ok: out std %.4f absmax %.4f sat %.3f%%  keys %d" %
          (name, meta["out_std"], meta["out_absmax"], 100 * sat, len(keys)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="*", default=list(CASES))
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    models = import_reference()
    for c in a.cases:
        run_case(models, c, a.out)


if __name__ == "__main__":
    main()
