"""The norm_layer=nn.InstanceNorm2d variant of the generators (reference constructors
models/networks.py:459, :555, use_bias :494 / :590, ResidualBlock :650-668).  Goldens = outputs of the REFERENCE modules built
with that argument (oracle/make_golden_in.py asserts the oracle bit-identical to them).  Tolerance: the north-star 1e-3 max-abs
fp32 -- WHERE THE REFERENCE ITSELF IS REPRODUCIBLE TO THAT.  Errors are ~100x those of the BatchNorm plans by nature: at the 2x2 / 4x4
levels a channel's statistics come from 4 / 16 values, and where those are nearly equal 1/sqrt(var + 1e-5) amplifies fp32 summation-order
noise up to 316x per layer.  The fixture records the reference module's distance from ITSELF evaluated with other CPU kernels (oneDNN
disabled) and from a float64 evaluation: on `in_large_512` the reference differs from itself by 1.8e-3 max / 1.9e-4 mean, so no
implementation can be held to 1e-3 there.  The bar: within 1.5x of the reference's own self-distance (max and mean), and 1e-3 wherever
that self-distance leaves room for it."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import GOLDEN

CASES = ["in_large_s128_b2", "in_normal_s192_b3", "in_normal_512", "in_large_512"]
TOL = 1e-3


def problem(case):
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.topology import build_topology
    meta = json.load(open(os.path.join(GOLDEN, case + ".json")))
    ref = np.load(os.path.join(GOLDEN, case + ".npz"))["out"]
    topo = build_topology(meta["variant"], ngf=meta["ngf"], num_downs=meta["num_downs"], size=meta["size"], norm="instance")
    sd = synth.scale_last_conv(synth.make_state_dict(topo, meta["weight_seed"]), topo, meta["last_gain"])
    feat, cand = synth.make_inputs(meta["batch"], meta["size"], meta["input_seed"], 1)
    return meta, ref, topo, sd, feat, cand


# ---- CPU -----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", CASES[:2])
def test_oracle_reproduces_the_reference_golden(case):
    from oracle import torch_oracle
    meta, ref, topo, sd, feat, cand = problem(case)
    x = torch.cat([torch.from_numpy(feat), torch.from_numpy(cand).expand(meta["batch"], -1, -1, -1)], 1)
    out = torch_oracle.generator_forward(torch_oracle.to_torch(sd), x, topo.nres, topo.num_downs).numpy()
    assert np.abs(out - ref).max() <= 1e-6            # bit-identical under the torch build that made the fixture
    assert np.abs(ref).max() < 0.99


def test_plan_and_containers_carry_the_reference_keys():
    from livespeechportraits_amd import networks
    from livespeechportraits_amd.engine import Engine
    for case in ("in_large_512", "in_normal_512"):
        meta = json.load(open(os.path.join(GOLDEN, case + ".json")))
        e = Engine(meta["variant"], norm="instance")
        want = {"netG." + k: tuple(v) for k, v in meta["keys"].items()}           # dumped from the reference module
        assert e.expected_tensors() == want
        ctor = networks.Feature2FaceGenerator_large if meta["variant"] == "large" else networks.Feature2FaceGenerator_normal
        g = ctor(13, 3, 8, 64, norm_layer=nn.InstanceNorm2d)
        assert {k: tuple(v.shape) for k, v in g.state_dict().items()} == {k: tuple(v) for k, v in meta["keys"].items()}
        routes = {l["kernel"] for l in e.layers(1)}
        assert any("stats" in r for r in routes) and any("in_small" in r for r in routes) and any("in_reduce_stats" in r for r in routes)
    with pytest.raises(Exception):
        Engine("normal", norm="instance", dtype="bf16")            # fp32 only
    with pytest.raises(NotImplementedError):
        networks.Feature2FaceGenerator_normal(norm_layer=nn.GroupNorm)


# ---- GPU -----------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_matches_reference_golden(case, gpu_device):
    from livespeechportraits_amd.engine import Engine
    meta, ref, topo, sd, feat, cand = problem(case)
    e = Engine(meta["variant"], 13, 1, 3, meta["ngf"], meta["num_downs"], meta["size"], max_batch=meta["batch"], norm="instance")
    assert not e.load_state_dict(sd)
    e.bind(e.pack(), gpu_device)
    f, c = torch.from_numpy(feat).to(gpu_device), torch.from_numpy(cand).to(gpu_device)
    out = e.forward(f, c)
    err = np.abs(out.cpu().numpy() - ref)
    print("\n%s: max-abs vs the reference module %.2e (mean %.2e) -- the contract is %.0e: %s; routes %s" % (
        case, err.max(), err.mean(), TOL, "inside" if err.max() <= TOL else "OUTSIDE (see the reference's own spread below: a wider, labelled bound applies)",
        sorted({l["kernel"] for l in e.layers(meta["batch"]) if "in_" in l["kernel"]})))
    # the reference's own distance from exact arithmetic on this case (float64 oracle), for scale
    from oracle import torch_oracle
    sd64 = {k: torch.from_numpy(np.ascontiguousarray(v)).double() for k, v in sd.items() if v.dtype == np.float32}
    x64 = torch.cat([f.cpu(), c.cpu().expand(meta["batch"], -1, -1, -1)], 1).double()
    # GPU vs float64 and reference vs float64, at every size (a few seconds of float64 on the host at 512x512).  "The reference" is not one
    # number here: the fixture records the distance from float64 of three fp32 evaluations of the SAME reference module under the same torch
    # (default oneDNN kernels = the golden, ATen's native convolutions, oneDNN channels_last) -- on in_large_512 they range from 8.7e-4 to
    # 2.1e-3 max-abs, because InstanceNorm amplifies whichever fp32 rounding the high-resolution levels made (running the <= 64x64 levels in
    # float64 changes none of them: oracle/make_golden_in.py).  The HIP path must be no further from exact arithmetic than the
    # widest of the reference's own evaluations, max and mean.
    ref64 = torch_oracle.generator_forward(sd64, x64, topo.nres, topo.num_downs).float().numpy()
    r64, g64 = np.abs(ref - ref64), np.abs(out.cpu().numpy() - ref64)
    evals = meta["reference_self_distance"]["fp32_evaluations_vs_float64"]
    print("   vs the float64 evaluation -- reference (golden) max %.2e mean %.2e; its other fp32 evaluations %s; ours: max %.2e mean %.2e" % (
        r64.max(), r64.mean(), {k: "%.2e / %.2e" % tuple(v) for k, v in evals.items() if k != "onednn"}, g64.max(), g64.mean()))
    assert abs(r64.max() - evals["onednn"][0]) <= 1e-6                       # the fixture's own number, recomputed
    assert g64.max() <= 1.2 * max(v[0] for v in evals.values()) and g64.mean() <= 1.2 * max(v[1] for v in evals.values())
    self_d = meta["reference_self_distance"]
    print("   the reference vs itself (oneDNN off): max %.2e mean %.2e" % (self_d["onednn_off_max"], self_d["onednn_off_mean"]))
    assert err.max() <= 1.5 * self_d["onednn_off_max"] and err.mean() <= 1.5 * self_d["onednn_off_mean"]
    if self_d["onednn_off_max"] <= TOL / 1.5:
        assert err.max() <= TOL
    assert torch.equal(e.forward(f, c), out)                    # fixed summation order: bit-reproducible
    # frames are independent: statistics are per (frame, channel)
    if meta["batch"] > 1:
        one = e.forward(f[1:2].contiguous(), c)
        assert (one[0] - out[1]).abs().max().item() <= 1.5 * self_d["onednn_off_max"]     # other tiles / split-K at batch 1, amplified as above
    u8 = e.forward_image(f, c)
    want = ((out.permute(0, 2, 3, 1) + 1.0) / 2.0 * 255.0).clamp(0, 255).to(torch.uint8)
    assert (u8.int() - want.int()).abs().max().item() <= 1


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["in_large_s128_b2", "in_normal_512"])
def test_instance_norm_plans_through_the_f4x4_kernel(case, gpu_device):
    """The opt-in Winograd F(4x4,3x3) route serves the InstanceNorm plans like F(2x2) does (raw conv output + bias, statistics and normalisation
    behind it), inside the 1e-3 contract against the reference-generated golden."""
    from livespeechportraits_amd.engine import Engine
    meta, ref, topo, sd, feat, cand = problem(case)
    e = Engine(meta["variant"], 13, 1, 3, meta["ngf"], meta["num_downs"], meta["size"], max_batch=meta["batch"], norm="instance", wino4=True)
    assert not e.load_state_dict(sd)
    e.bind(e.pack(), gpu_device)
    kernels = [l["kernel"] for l in e.layers(meta["batch"])]
    assert any(k.startswith("wino4_3x3+in_") for k in kernels)
    out = e.forward(torch.from_numpy(feat).to(gpu_device), torch.from_numpy(cand).to(gpu_device)).cpu().numpy()
    err = np.abs(out - ref)
    self_d = meta["reference_self_distance"]
    print("\n%s through wino4_3x3: max-abs vs the reference module %.2e (mean %.2e); the reference vs itself %.2e" % (case, err.max(), err.mean(), self_d["onednn_off_max"]))
    # Measured: 4.3e-4 / 2.9e-4 here against 1.25e-4 / 7.7e-5 through the default route -- F(4x4)'s transforms round coarser than F(2x2)'s
    # (profiles/r04_wino4x4_error.txt) and InstanceNorm amplifies that like any other rounding.  The opt-in route is held to the north-star
    # contract itself, not to the reference's own spread as the default route is.
    assert err.max() <= TOL and err.mean() <= TOL / 10


@pytest.mark.gpu
def test_batch8_and_the_parameter_container(gpu_device):
    """the reference-named constructor with norm_layer=nn.InstanceNorm2d, batch 8 at full size vs the live oracle (every frame)"""
    from livespeechportraits_amd import networks, synth
    from oracle import torch_oracle
    meta, ref, topo, sd, _, cand = problem("in_normal_512")
    g = networks.Feature2FaceGenerator_normal(13, 3, 8, 64, norm_layer=nn.InstanceNorm2d)
    g.load_state_dict({k[len("netG."):]: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()})
    g = g.to(gpu_device)
    feat, _ = synth.make_inputs(8, 512, seed=7, cand_batch=1)
    out = g.render(torch.from_numpy(feat).to(gpu_device), torch.from_numpy(cand).to(gpu_device)).cpu().numpy()
    sdt = torch_oracle.to_torch(sd)
    for i in range(8):
        want = torch_oracle.inference(sdt, torch.from_numpy(feat[i:i + 1]), torch.from_numpy(cand), 1, 8).numpy()
        err = np.abs(out[i:i + 1] - want).max()
        print("frame %d: %.2e" % (i, err))
        assert err <= TOL, i            # other inputs than the fixture's: the contract itself (the `normal` nets leave room for it)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["in_large_s128_b2", "in_large_512"])
def test_round5_arms_of_the_one_launch_route(case, gpu_device):
    """in_small with its rows resident in registers (one read of the slab instead of three) performs the same operations in the same order as the three-pass form: the same
    bits.  conv3x3_smallm normalising in its own epilogue (a workgroup holds every pixel of its channels) sums a frame's <= 16 values in another order than in_small's
    shuffle tree: held to the golden like the default, and to a hair of the unfused arm."""
    from livespeechportraits_amd.engine import Engine
    meta, ref, topo, sd, feat, cand = problem(case)
    f, c = torch.from_numpy(feat).to(gpu_device), torch.from_numpy(cand).to(gpu_device)
    outs = {}
    for name, tune in (("default", None), ("three_pass", {"in_small_regs": 0}), ("unfused", {"in_smallm_fused": 0}), ("unfused_three_pass", {"in_smallm_fused": 0, "in_small_regs": 0}),
                       ("reduce_above_256", {"in_small_max_hw": 256})):
        e = Engine(meta["variant"], 13, 1, 3, meta["ngf"], meta["num_downs"], meta["size"], max_batch=meta["batch"], norm="instance", tune=tune)
        e.load_state_dict(sd)
        e.bind(e.pack(), gpu_device)
        outs[name] = e.forward(f, c).clone()
        kern = sorted({l["kernel"] for l in e.layers(meta["batch"]) if l["kernel"].startswith("conv3x3_smallm")})
        assert kern and all(("(in)" in k) == (name in ("default", "three_pass", "reduce_above_256")) for k in kern if k != "conv3x3_smallm"), (name, kern)
        e.close()
    assert torch.equal(outs["default"], outs["three_pass"]) and torch.equal(outs["unfused"], outs["unfused_three_pass"])
    self_d = meta["reference_self_distance"]
    for name, o in outs.items():
        err = np.abs(o.cpu().numpy() - ref)
        print("%s %-20s max-abs vs the reference module %.2e (mean %.2e); vs the default arm %.2e" % (case, name, err.max(), err.mean(), (o - outs["default"]).abs().max().item()))
        assert err.max() <= 1.5 * self_d["onednn_off_max"] and err.mean() <= 1.5 * self_d["onednn_off_mean"]
