#!/usr/bin/env python3
"""Golden outputs of the reference's generators built with ``norm_layer=nn.InstanceNorm2d`` (container only: needs
/root/reference).  The reference's Feature2Face_G never passes norm_layer (feature2face_G.py:19-21), so this variant is reached
through the constructors themselves (models/networks.py:459 Feature2FaceGenerator_normal, :555 Feature2FaceGenerator_large):
use_bias = True on the level convs (:494 / :590), InstanceNorm2d(affine=False, eps=1e-5) wherever the default has BatchNorm2d.

For every case: build the REFERENCE module, load synthetic weights (incl. the conv biases), run it on CPU, assert that
oracle/torch_oracle.py is bit-identical, freeze the output (asserted free of tanh saturation).  `--out` selects the directory.
"""
import argparse
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)

from livespeechportraits_amd import synth                       # noqa: E402
from livespeechportraits_amd.topology import build_topology     # noqa: E402
from oracle import torch_oracle                                  # noqa: E402

# name -> (variant, ngf, num_downs, size, batch)
CASES = {
    "in_large_512": ("large", 64, 8, 512, 1),
    "in_normal_512": ("normal", 64, 8, 512, 1),
    "in_large_s128_b2": ("large", 32, 6, 128, 2),       # mixes all three statistics routes at a small size
    "in_normal_s192_b3": ("normal", 32, 6, 192, 3),     # extents that are not powers of two (96, 48, 24, 12, 6, 3)
}


LAST_GAIN = 0.35     # keeps the pre-tanh map inside +-2 (no saturation); recorded in the fixture's json


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="*", default=list(CASES))
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    a = ap.parse_args()
    if not os.path.isdir(REF):
        raise SystemExit("make_golden_in.py needs /root/reference (build container only)")
    for name in ("torchvision", "torchvision.models", "cv2"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path.insert(0, REF)
    from models import networks as ref_networks
    os.makedirs(a.out, exist_ok=True)
    for name in a.cases:
        variant, ngf, nd, size, batch = CASES[name]
        topo = build_topology(variant, ngf=ngf, num_downs=nd, size=size, norm="instance")      # keys netG.model... as under Feature2Face_G
        sd = synth.scale_last_conv(synth.make_state_dict(topo, seed=1234), topo, LAST_GAIN)
        ctor = ref_networks.Feature2FaceGenerator_large if variant == "large" else ref_networks.Feature2FaceGenerator_normal
        net = ctor(13, 3, nd, ngf, norm_layer=nn.InstanceNorm2d).eval()
        ref_keys = {k: list(v.shape) for k, v in net.state_dict().items()}
        assert ref_keys == {k[len("netG."):]: list(v.shape) for k, v in sd.items()}, "key map differs from the reference module"
        net.load_state_dict({k[len("netG."):]: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)
        feat, cand = synth.make_inputs(batch, size, seed=99, cand_batch=1)
        x = torch.cat([torch.from_numpy(feat), torch.from_numpy(cand).expand(batch, -1, -1, -1)], 1)
        with torch.no_grad():
            ref = net(x)
        taps = {}
        ora = torch_oracle.generator_forward(torch_oracle.to_torch(sd), x, topo.nres, nd, taps=taps)
        assert torch.equal(ora, ref), "oracle differs from the reference module: %g" % (ora - ref).abs().max().item()
        # The reference's own reproducibility on this case.  InstanceNorm over the 4 / 16 values of the 2x2 / 4x4 levels amplifies fp32
        # rounding differences by up to 1/sqrt(eps) = 316, so two evaluations of the SAME module with different CPU kernels differ
        # far more than in the BatchNorm plans: (a) oneDNN convolutions vs ATen's native ones, (b) fp32 vs a float64 evaluation.
        with torch.backends.mkldnn.flags(enabled=False):
            with torch.no_grad():
                alt = net(x)
        d_alt = (alt - ref).abs()
        sd64 = {k: v.double() for k, v in torch_oracle.to_torch(sd).items()}
        ref64 = torch_oracle.generator_forward(sd64, x.double(), topo.nres, nd).float()
        d_64 = (ref64 - ref).abs()
        # (c) the distance of every fp32 evaluation of the reference module from the float64 one: the default oneDNN kernels (= the fixture), ATen's
        # native convolutions, and oneDNN's channels_last kernels.  Their spread is what "as close to exact arithmetic as the reference" means
        # for this variant; the GPU tests hold the HIP path to the widest of them.
        with torch.no_grad():
            cl = net.to(memory_format=torch.channels_last)(x.contiguous(memory_format=torch.channels_last)).contiguous()
        net.to(memory_format=torch.contiguous_format)
        f64 = {"onednn": [d_64.max().item(), d_64.mean().item()],
               "native": [(alt - ref64).abs().max().item(), (alt - ref64).abs().mean().item()],
               "onednn_channels_last": [(cl - ref64).abs().max().item(), (cl - ref64).abs().mean().item()]}
        print("   reference vs itself with oneDNN off: max %.2e mean %.2e; vs float64: max %.2e mean %.2e; fp32 evaluations vs float64 %s"
              % (d_alt.max(), d_alt.mean(), d_64.max(), d_64.mean(), {k: "%.2e / %.2e" % tuple(v) for k, v in f64.items()}))
        sat = (ref.abs() > 0.99).float().mean().item()
        print("%-18s %s ngf %d downs %d size %d batch %d: |out| max %.3f std %.3f sat %.4f%%, pre-tanh absmax %.2f; oracle bit-exact; %d keys"
              % (name, variant, ngf, nd, size, batch, ref.abs().max(), ref.std(), 100 * sat, taps["pre_tanh"].abs().max(), len(ref_keys)))
        assert sat < 1e-4
        np.savez_compressed(os.path.join(a.out, name + ".npz"), out=ref.numpy())      # no saturation (asserted above): tanh hides nothing
        with open(os.path.join(a.out, name + ".json"), "w") as f:
            json.dump({"variant": variant, "ngf": ngf, "num_downs": nd, "size": size, "batch": batch, "cand_batch": 1, "norm": "instance", "last_gain": LAST_GAIN,
                       "reference_self_distance": {"onednn_off_max": d_alt.max().item(), "onednn_off_mean": d_alt.mean().item(),
                                                   "float64_max": d_64.max().item(), "float64_mean": d_64.mean().item(),
                                                   "fp32_evaluations_vs_float64": f64},
                       "weight_seed": 1234, "input_seed": 99, "torch": torch.__version__, "keys": ref_keys}, f, indent=0)


if __name__ == "__main__":
    main()
