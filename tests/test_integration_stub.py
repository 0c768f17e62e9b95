"""The reference-side binding of INTEGRATION.md section 2 (integration/lsp_hip_generator.py), executed:
  * here in the build container against the REFERENCE's own Feature2FaceModel / Feature2Face_G (CPU: construction, every
    state-dict key through lspf2f_set_tensor, host packing -- the blob must equal the one this package packs);
  * on the GPU box (no reference there) against a stand-in module with the reference's key names: forward vs the golden."""
import argparse
import os
import sys
import types

import numpy as np
import pytest
import torch

from conftest import ROOT, golden_problem

REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "integration"))


def _lib():
    return os.path.join(ROOT, "livespeechportraits_amd", "liblspf2f.so")


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs /root/reference (build container only)")
@pytest.mark.parametrize("variant", ["normal", "large"])
def test_stub_binds_the_reference_model_on_the_host(variant, tmp_path):
    import subprocess
    # the reference's `models` / `util` packages shadow nothing of ours, but keep them out of this interpreter anyway
    code = r"""
import sys, types, argparse, torch, numpy as np
for name in ("torchvision", "torchvision.models", "cv2"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.path.insert(0, %(ref)r); sys.path.insert(0, %(integ)r); sys.path.insert(0, %(root)r)
from models import create_model                                   # the REFERENCE's factory
import lsp_hip_generator as G
opt = argparse.Namespace(model="feature2face", gpu_ids=[], isTrain=False, size=%(variant)r, ngf=64, n_downsample_G=8, fp16=0,
                         checkpoints_dir=%(tmp)r, name="t", load_epoch="none", verbose=False, task="Feature2Face", dataset_mode="face")
torch.manual_seed(3)
model = create_model(opt)                                         # reference Feature2FaceModel, init_weights'd
model.eval()
net = model.Feature2Face_G
gen = G.HipGenerator(net, opt, size=512, max_batch=2, library=%(lib)r)      # host half of install(): every key accepted, packed
from livespeechportraits_amd.engine import Engine
e = Engine(%(variant)r, size=512, max_batch=2)
e.load_state_dict({k: v for k, v in net.state_dict().items()})
assert torch.equal(e.pack(), gen.blob), "stub and package pack different blobs"
nkeys = len([k for k in net.state_dict() if not k.endswith("num_batches_tracked")])
print("OK", type(model).__module__, nkeys, gen.blob.numel())
"""
    p = subprocess.run([sys.executable, "-c", code % {"ref": REF, "integ": os.path.join(ROOT, "integration"), "root": ROOT,
                                                      "variant": variant, "tmp": str(tmp_path), "lib": _lib()}],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    last = p.stdout.strip().splitlines()[-1].split()
    assert last[0] == "OK" and last[1] == "models.feature2face_model"          # it really was the reference's class
    assert int(last[2]) == (368 if variant == "large" else 218)


@pytest.mark.gpu
def test_stub_renders_on_the_gpu(gpu_device):
    import lsp_hip_generator as G
    meta, arrays, topo, sd, feat, cand = golden_problem("normal_512")

    class RefShaped:                              # state_dict() with the reference's DataParallel-style keys
        def state_dict(self):
            return {"module." + k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
    opt = argparse.Namespace(size="normal", ngf=64, n_downsample_G=8, fp16=0)
    model = types.SimpleNamespace(Feature2Face_G=RefShaped(), opt=opt)
    gen = G.install(model, device=str(gpu_device), size=512, max_batch=2, library=_lib())
    assert model.Feature2Face_G is gen
    f, c = torch.from_numpy(feat).to(gpu_device), torch.from_numpy(cand).to(gpu_device)
    out = model.Feature2Face_G(torch.cat([f, c], 1))           # what the reference's inference() does (feature2face_model.py:231-236)
    assert np.abs(out.cpu().numpy() - arrays["out"]).max() <= 5e-5
    assert np.abs(gen.render(f, c).cpu().numpy() - arrays["out"]).max() <= 5e-5
