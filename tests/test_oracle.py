"""CPU: the oracle against the golden vectors frozen from the REAL reference
(oracle/make_golden.py), and the two oracle restatements against each other."""
import numpy as np
import pytest
import torch

from conftest import golden_problem, load_golden


@pytest.mark.parametrize("case", ["normal_s64_b3", "large_s128_b2"])
def test_torch_oracle_reproduces_reference_golden(case):
    from oracle import torch_oracle
    meta, arrays, topo, sd, feat, cand = golden_problem(case)
    cand_full = torch.from_numpy(cand).expand(meta["batch"], -1, -1, -1)
    taps = {}
    out = torch_oracle.inference(torch_oracle.to_torch(sd), torch.from_numpy(feat), cand_full, topo.nres,
                                 topo.num_downs, taps=taps)
    # same torch build as the one that generated the fixture => bit-exact; allow 1e-6 for a
    # different oneDNN dispatch on another host CPU
    assert np.abs(out.numpy() - arrays["out"]).max() <= 1e-6
    for name, info in meta["taps"].items():
        cs, ss = info["cstride"], info["sstride"]
        got = taps[name][:, ::cs, ::ss, ::ss].numpy()
        assert np.abs(got - arrays["tap_" + name]).max() <= 1e-5 * max(1.0, np.abs(got).max())


@pytest.mark.parametrize("case", ["normal_s64_b3", "large_s128_b2"])
def test_c_oracle_matches_reference_golden(case):
    """Plain-C restatement (double accumulation) vs the reference's fp32 output."""
    from oracle import c_oracle
    meta, arrays, topo, sd, feat, cand = golden_problem(case)
    x = np.concatenate([feat, np.broadcast_to(cand, (feat.shape[0],) + cand.shape[1:])], 1)
    out = c_oracle.generator_forward(topo, sd, x)
    assert np.abs(out - arrays["out"]).max() <= 2e-6


def test_c_oracle_conv_matches_torch():
    from oracle import c_oracle
    rng = np.random.default_rng(0)
    for stride in (1, 2):
        x = rng.standard_normal((5, 9, 11)).astype(np.float32)
        w = rng.standard_normal((4, 5, 3, 3)).astype(np.float32)
        ref = torch.nn.functional.conv2d(torch.from_numpy(x)[None].double(), torch.from_numpy(w).double(), None,
                                         stride, 1)[0].float().numpy()
        got = c_oracle.conv3x3(x, w, stride)
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 1e-5


def test_golden_fixtures_are_sane():
    for case in ("large_512", "normal_512", "large_s128_b2", "normal_s64_b3"):
        meta, arrays = load_golden(case)
        out = arrays["out"]
        assert out.shape == (meta["batch"], 3, meta["size"], meta["size"]) and out.dtype == np.float32
        assert np.isfinite(out).all()
        assert np.abs(out).max() < 0.99, "saturated tanh would hide errors (SURVEY.md 8c)"
        assert out.std() > 0.02


def test_oracle_is_not_reachable_from_the_product():
    """The product package must never import the oracle (no CPU fallback)."""
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "livespeechportraits_amd")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "f2f_oracle" not in src, f


def test_tensor2im_oracle_matches_reference_function():
    """oracle/tensor2im_oracle.py against the reference's own util.tensor2im when the reference tree is
    present (build container); always against a hand-computed known-answer vector."""
    import os
    import sys
    import types
    from oracle.tensor2im_oracle import tensor2im
    x = np.array([-1.0, -0.999, -0.5, 0.0, 0.0039, 0.5, 0.999, 1.0, 1.5, -2.0, 0.2, -0.2], np.float32)
    chw = np.stack([x.reshape(3, 4)] * 3)
    got = tensor2im(chw)
    assert got.shape == (3, 4, 3) and got.dtype == np.uint8
    exp = np.clip((x.astype(np.float32) + 1) / 2.0 * 255.0, 0, 255).astype(np.uint8)
    assert np.array_equal(got[:, :, 0].reshape(-1), exp)
    assert list(exp[[0, 3, 7, 8, 9]]) == [0, 127, 255, 255, 0]
    if os.path.isdir("/root/reference/util"):
        sys.modules.setdefault("cv2", types.ModuleType("cv2"))
        sys.path.insert(0, "/root/reference")
        try:
            from util import util as ref_util
        finally:
            sys.path.pop(0)
        rnd = np.random.default_rng(3).uniform(-1.2, 1.2, (3, 17, 19)).astype(np.float32)
        assert np.array_equal(ref_util.tensor2im(torch.from_numpy(rnd)), tensor2im(rnd))
