"""CPU: host logic -- key map, C++ plan, BN-fold packer, C-ABI surface, model wrapper API,
error behaviour.  No GPU compute is called here."""
import argparse
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT


def opt_ns(**kw):
    d = dict(model="feature2face", gpu_ids=[], isTrain=False, size="large", ngf=64, n_downsample_G=8, fp16=0,
             checkpoints_dir="/tmp", name="t", load_epoch="none", verbose=False)
    d.update(kw)
    return argparse.Namespace(**d)


# ---- C ABI surface -------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    from livespeechportraits_amd import _native as N
    hdr = open(os.path.join(ROOT, "include", "lspf2f.h")).read()
    declared = set(re.findall(r"\b(lspf2f_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "header parse failed"
    lib = ctypes.CDLL(N.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), "liblspf2f.so does not export %s" % name
    assert declared == set(N.SIGNATURES), "ctypes binding and header disagree: %s" % (declared ^ set(N.SIGNATURES))
    assert N.load().lspf2f_abi_version() == N.ABI_VERSION


def test_create_rejects_unsupported_configs():
    from livespeechportraits_amd import _native as N
    from livespeechportraits_amd.engine import Engine
    with pytest.raises(ValueError):
        Engine("small")
    for kw in (dict(ngf=48), dict(num_downs=4), dict(size=500), dict(output_nc=5)):
        with pytest.raises(N.Lspf2fError) as ei:
            Engine("large", **kw)
        assert ei.value.code == -2
    e = Engine("normal", ngf=32, num_downs=5, size=64)
    with pytest.raises(N.Lspf2fError):   # forward before weights are bound -> STATE error, not a crash
        N.check(e.lib.lspf2f_forward(e._h, ctypes.c_void_p(8), ctypes.c_void_p(8), 1, ctypes.c_void_p(8), 1, None))
    # the hazard-test entry point: a null handle and a handle without a workspace are errors with a message, never a crash; the Python wrapper says so before the C call
    n = ctypes.c_uint32(7)
    assert e.lib.lspf2f_debug_poison(None, 0xFF, None, ctypes.byref(n)) == -1 and b"null handle" in e.lib.lspf2f_last_error()
    assert e.lib.lspf2f_debug_poison(e._h, 0xFF, None, ctypes.byref(n)) == -5 and b"workspace not bound" in e.lib.lspf2f_last_error()
    with pytest.raises(RuntimeError, match="no workspace bound"):
        e.debug_poison()
    # unknown tune keys are refused (a typo must not silently run the default)
    with pytest.raises(N.Lspf2fError):
        Engine("normal", ngf=32, num_downs=5, size=64, tune={"wino_pri": 1})
    for k in ("wino_prio", "out_wt", "fullk16", "fullk16_min_frames", "rowlast_fused"):
        Engine("normal", ngf=64, num_downs=5, size=64, dtype="bf16" if "16" in k or "rowlast" in k else "f32", tune={k: 0}).close()


# ---- key map / plan ---------------------------------------------------------------------------
@pytest.mark.parametrize("variant", ["large", "normal"])
def test_key_map_matches_reference_dump(variant):
    """Python topology == C++ plan == nn.Module skeleton == keys dumped from the reference."""
    from livespeechportraits_amd.engine import Engine
    from livespeechportraits_amd.feature2face_G import Feature2Face_G
    from livespeechportraits_amd.topology import build_topology
    ref = json.load(open(os.path.join(GOLDEN, "keys_%s.json" % variant)))
    topo = build_topology(variant)
    assert {k: list(v) for k, v in topo.tensors.items()} == ref
    assert list(topo.tensors) == list(ref)                       # same order as the reference state dict
    eng = Engine(variant)
    exp = eng.expected_tensors()
    assert {k: list(v) for k, v in exp.items()} == {k: v for k, v in ref.items() if not k.endswith("num_batches_tracked")}
    mod = Feature2Face_G(opt_ns(size=variant))
    sd = mod.state_dict()
    assert list(sd) == list(ref) and all(list(sd[k].shape) == ref[k] for k in ref)


def test_plan_matches_survey_workload_constants():
    """76/46 convs, 249.77/166.08 GFLOP, 820.5/499.0 MB activations, 486.9/304.7 MB weights."""
    from livespeechportraits_amd.engine import Engine
    from livespeechportraits_amd.topology import build_topology
    for variant, nconv, gflop, act_mb, w_mb in (("large", 76, 249.76, 820.5, 486.9), ("normal", 46, 166.08, 499.0, 304.7)):
        layers = Engine(variant).layers(1)
        topo = build_topology(variant)
        assert len(layers) == nconv == len(topo.convs)
        assert abs(sum(l["flops_per_frame"] for l in layers) / 1e9 - gflop) < 0.01
        assert abs(sum(l["act_bytes_per_frame"] for l in layers) / 1e6 - act_mb) < 0.1
        assert abs(topo.weight_elems() * 4 / 1e6 - w_mb) < 0.1
        assert topo.flops_per_frame() == sum(l["flops_per_frame"] for l in layers)
        for l, c in zip(layers, topo.convs):
            assert (l["cin"], l["cout"], l["h_in"], l["h_out"], l["stride"]) == (c.cin, c.cout, c.h_in, c.h_out, c.stride)
            assert bool(l["upsample"]) == c.upsample and bool(l["residual"]) == c.residual and bool(l["concat"]) == c.concat
        # sub-pixel up-convs issue 4/9 of the algorithmic FLOPs
        assert sum(l["exec_flops_per_frame"] for l in layers) < sum(l["flops_per_frame"] for l in layers)


def test_workspace_reuse_and_growth():
    from livespeechportraits_amd.engine import Engine
    e = Engine("large", max_batch=8)
    keep = Engine("large", max_batch=8, keep_intermediates=True)
    w1, w8 = e.workspace_bytes(1), e.workspace_bytes(8)
    assert w1 < keep.workspace_bytes(1) / 2, "liveness reuse should at least halve the arena"
    fixed = 2 * 256 * 256 * 64 * 4 + 16384 * 4   # batch-independent head: two candidate-share slots + split-K arrival counters (plan.h)
    assert 6 * (w1 - fixed) < w8 - fixed < 9 * (w1 - fixed)
    # every layer output lies inside the arena, 256-byte aligned
    for l in keep.layers(2):
        if l["out_offset"] >= 0:
            assert l["out_offset"] % 256 == 0
            assert l["out_offset"] + 2 * l["cout"] * l["h_out"] ** 2 * 4 <= keep.workspace_bytes(2)


# ---- packer -----------------------------------------------------------------------------------
def test_pack_weights_matches_numpy_restatement():
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.engine import Engine
    topo, sd = synth.synthetic("large", ngf=32, num_downs=6, size=128)
    # all_forms: the blob of a plain handle carries only the weight forms its batch range reads (a Winograd layer then has no row copy); this
    # test checks the ROW form of every layer
    e = Engine("large", ngf=32, num_downs=6, size=128, tune={"all_forms": 1})
    assert e.packed_bytes() > Engine("large", ngf=32, num_downs=6, size=128).packed_bytes()
    e.load_state_dict(sd)
    blob = e.pack().numpy()
    f32 = lambda off, n: blob[off:off + 4 * n].view(np.float32)
    layers = e.layers(1)
    seen_sub = seen_nine = False
    for l, c in zip(layers, topo.convs):
        w = sd[c.weight_key].astype(np.float64)                      # OIHW
        n = c.cin * c.cout
        if l["kernel"] == "first_conv":
            exp = w.transpose(1, 2, 3, 0).reshape(-1)                # [ci][ky][kx][co]
        elif (l["weight_bytes"] == 16 * n * 4 and not l["kernel"].startswith("wino3x3")) or l["kernel"].startswith("winoup3x3"):   # sub-pixel up-conv (incl. the last layer, and the up-convs that run on winoup3x3: their sub-pixel copy stays at w_offset); the Winograd
            # layers also read 16 values per (co, ci) but keep the 9-tap copy at w_offset (their G g G^T copy: tests/test_wino_cpu.py)
            seen_sub = True
            grp = {0: [[0], [1, 2]], 1: [[0, 1], [2]]}
            exp = np.zeros((4, c.cout, 2, 2, c.cin))
            for py in (0, 1):
                for px in (0, 1):
                    for a in (0, 1):
                        for b in (0, 1):
                            exp[py * 2 + px, :, a, b, :] = sum(w[:, :, ky, kx] for ky in grp[py][a] for kx in grp[px][b])
            exp = exp.reshape(-1)
        else:
            seen_nine = True
            exp = w.transpose(0, 2, 3, 1).reshape(-1)                # [co][ky][kx][ci]
        got = f32(l["w_offset"], exp.size)
        assert np.array_equal(got, exp.astype(np.float32)), l["name"]
        if c.bn_key:
            g, b = sd[c.bn_key + ".weight"].astype(np.float64), sd[c.bn_key + ".bias"].astype(np.float64)
            m, v = sd[c.bn_key + ".running_mean"].astype(np.float64), sd[c.bn_key + ".running_var"].astype(np.float64)
            s = g / np.sqrt(v + 1e-5)
            assert np.array_equal(f32(l["scale_offset"], c.cout), s.astype(np.float32))
            assert np.array_equal(f32(l["shift_offset"], c.cout), (b - m * s).astype(np.float32))
        else:
            assert l["scale_offset"] == -1
    assert seen_sub and seen_nine


def test_pack_refuses_missing_and_unknown_tensors():
    from livespeechportraits_amd import _native as N, synth
    from livespeechportraits_amd.engine import Engine
    topo, sd = synth.synthetic("normal", ngf=32, num_downs=5, size=64)
    e = Engine("normal", ngf=32, num_downs=5, size=64)
    partial = dict(sd)
    del partial["netG.model.model.0.weight"]
    with pytest.raises(KeyError):
        e.load_state_dict(partial)                                   # the reference would stay silent (strict=False)
    e.load_state_dict(partial, strict=False)
    with pytest.raises(N.Lspf2fError) as ei:
        e.pack()
    assert ei.value.code == -3 and "netG.model.model.0.weight" in str(ei.value)
    with pytest.raises(N.Lspf2fError):
        N.check(e.lib.lspf2f_set_tensor(e._h, b"netG.bogus", np.zeros(4, np.float32).ctypes.data, 4))
    bad = dict(sd)
    bad["netG.model.model.0.weight"] = np.zeros((3, 3), np.float32)
    with pytest.raises(ValueError):
        e.load_state_dict(bad)
    extra = e.load_state_dict({("module." + k): v for k, v in sd.items()})   # DataParallel-prefixed keys are accepted
    assert all(k.endswith("num_batches_tracked") for k in extra)
    assert e.pack().numel() == e.packed_bytes()


# ---- model wrapper (the reference's API surface) -------------------------------------------------
def test_create_model_and_checkpoint_roundtrip(tmp_path):
    import livespeechportraits_amd as L
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.topology import build_topology
    topo = build_topology("normal", ngf=32, num_downs=5, size=64)
    sd = synth.make_state_dict(topo)
    ckpt = str(tmp_path / "Feature2Face.pkl")
    torch.save({"module." + k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, ckpt)
    opt = opt_ns(size="normal", ngf=32, n_downsample_G=5, load_epoch=ckpt, checkpoints_dir=str(tmp_path))
    m = L.create_model(opt)
    assert type(m).__name__ == "Feature2FaceModel" and m.model_names == ["Feature2Face_G"]
    m.setup(opt)
    m.eval()
    got = m.Feature2Face_G.state_dict()
    for k, v in sd.items():
        assert np.array_equal(got[k].numpy(), v), k
    # save_networks writes '<epoch>_<name>.pkl' that load_networks reads back
    m.save_networks("7")
    assert os.path.exists(os.path.join(str(tmp_path), "t", "7_Feature2Face_G.pkl"))
    # missing checkpoint at inference time -> ValueError (base_model.py:221-223)
    with pytest.raises(ValueError):
        L.create_model(opt_ns(size="normal", ngf=32, n_downsample_G=5, load_epoch="nope", checkpoints_dir=str(tmp_path))).setup(
            opt_ns(load_epoch="nope", checkpoints_dir=str(tmp_path)))
    # a checkpoint that lacks generator tensors is an error, not silent garbage
    torch.save({"module.netG.model.model.0.weight": torch.zeros(32, 13, 3, 3)}, ckpt)
    with pytest.raises(KeyError):
        m.load_networks(ckpt)


def test_inference_has_no_cpu_fallback():
    import livespeechportraits_amd as L
    m = L.create_model(opt_ns(size="normal", ngf=32, n_downsample_G=5))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.inference(torch.zeros(1, 1, 64, 64), torch.zeros(1, 12, 64, 64))
    small = L.create_model(opt_ns(size="small"))                    # the pix2pix U-Net variant is supported, but never on the CPU
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        small.inference(torch.zeros(1, 1, 64, 64), torch.zeros(1, 22, 64, 64))
    with pytest.raises(ValueError):
        L.create_model(opt_ns(size="medium"))
    with pytest.raises(NotImplementedError):
        L.create_model(opt_ns(isTrain=True))
    with pytest.raises(NotImplementedError):
        L.create_model(opt_ns(model="pix2pix"))                     # not one of the three models this package replaces
    with pytest.raises(RuntimeError, match="no CPU path"):
        L.create_model(opt_ns(model="audio2feature", feature_decoder="LSTM"))      # supported, but never on the CPU


def test_synth_is_deterministic_and_matches_recipe():
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.topology import build_topology
    topo = build_topology("normal", ngf=32, num_downs=5, size=64)
    a, b = synth.make_state_dict(topo, 1234), synth.make_state_dict(topo, 1234)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    w = a["netG.model.model.2.block.0.weight"]
    assert abs(w.std() - 0.02) < 2e-3 and abs(w.mean()) < 2e-3
    v = a["netG.model.model.2.block.1.running_var"]
    assert v.min() >= 0.9 and v.max() <= 1.6
    feat, cand = synth.make_inputs(2, 64)
    assert set(np.unique(feat)) <= {0.0, 1.0} and 0.01 < feat.mean() < 0.06
    assert cand.min() >= -1 and cand.max() < 1
    assert float(synth.uniform01(4, 7)[0]) == float(synth.uniform01(1, 7)[0])


def test_batched_render_loop_bookkeeping():
    """render_frames chunks the frame stream, keeps order, handles a ragged last batch (CPU, stand-in model)."""
    from livespeechportraits_amd.render_loop import batched, render_frames
    assert [len(c) for c in batched(range(11), 4)] == [4, 4, 3]

    class Fake:
        calls = []
        def inference_image(self, maps, cand):
            Fake.calls.append(maps.shape[0])
            return (maps[:, 0, :, :, None] * 10).to(torch.uint8).expand(-1, -1, -1, 3).contiguous()

    maps = [torch.full((1, 4, 4), float(i)) for i in range(7)]
    got = render_frames(Fake(), iter(maps), torch.zeros(1, 12, 4, 4), batch=3)
    assert Fake.calls == [3, 3, 1] and len(got) == 7
    assert [int(f[0, 0, 0]) for f in got] == [10 * i for i in range(7)]
    assert got[0].shape == (4, 4, 3) and got[0].dtype == np.uint8
    seen = []
    render_frames(Fake(), iter(maps), torch.zeros(1, 12, 4, 4), batch=4, on_frame=lambda i, a: seen.append((i, int(a[0, 0, 0]))))
    assert seen == [(i, 10 * i) for i in range(7)]


def test_bf16_plans_route_64_channel_layers_to_the_row_kernel(monkeypatch):
    """in bf16 plans the 64 -> 64 and 128 -> 128 stride-1 convs run on rowconv64 / rowconv128 (strip height by batch), nothing else does, fp32 plans never do,
    and LSP_HIP_ROWCONV=0 (read at create) puts them back on the implicit GEMM.  The packer's fragment-ordered copy of their weights (right
    behind the row layout in the blob, plan.cpp) is checked against the numpy restatement used by the GPU test."""
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.engine import Engine
    e = Engine("normal", dtype="bf16", max_batch=8)
    for batch, rows in ((1, 4), (2, 8), (8, 32)):
        names = [(l["name"], l["tile_m"] // 64) for l in e.layers(batch) if l["kernel"] == "rowconv64"]
        assert [n for n, _ in names] == ["L0.d.res0.a", "L0.d.res0.b", "L1.u.res0.a", "L1.u.res0.b"]
        assert all(r == rows for _, r in names)
    wide = [(l["name"], l["tile_m"] // 32) for l in e.layers(8) if l["kernel"] == "rowconv128"]
    assert wide == [(n, 16) for n in ("L1.d.res0.a", "L1.d.res0.b", "L2.u.res0.a", "L2.u.res0.b")]
    for l in e.layers(8):
        if l["kernel"] == "rowconv64":
            assert (l["cin"], l["cout"], l["stride"], l["upsample"], l["h_out"]) == (64, 64, 1, 0, 256)
        if l["kernel"] == "rowconv128":
            assert (l["cin"], l["cout"], l["stride"], l["upsample"], l["h_out"]) == (128, 128, 1, 0, 128)
    assert not any(l["kernel"].startswith("rowconv") for l in Engine("normal", max_batch=8).layers(8))
    monkeypatch.setenv("LSP_HIP_ROWCONV", "0")
    assert not any(l["kernel"].startswith("rowconv") for l in Engine("normal", dtype="bf16", max_batch=8).layers(8))
    monkeypatch.delenv("LSP_HIP_ROWCONV")

    topo, sd = synth.synthetic("normal", ngf=64, num_downs=5, size=128)      # 64-channel level at 64x64
    s = Engine("normal", ngf=64, num_downs=5, size=128, dtype="bf16", tune={"all_forms": 1})      # rows AND fragments in the blob
    s.load_state_dict(sd)
    blob = s.pack().numpy()
    checked = 0
    kinds = set()
    for i, (l, c) in enumerate(zip(s.layers(1), topo.convs)):
        if not l["kernel"].startswith("rowconv"):
            continue
        ch = l["cin"]
        nbytes = ch * 9 * ch * 2
        fo = s.form_offset(i, "row")
        assert fo > l["w_offset"] >= 0
        rows = blob[l["w_offset"]: l["w_offset"] + nbytes].view(np.uint16).reshape(ch // 32, 32, 9, ch // 16, 2, 8)   # [nb][ch][tap][kc][hi][e]
        frag = blob[fo: fo + nbytes].view(np.uint16).reshape(ch // 32, 9, ch // 16, 2, 32, 8)
        assert np.array_equal(frag, rows.transpose(0, 2, 3, 4, 1, 5)), l["name"]
        checked += 1
        kinds.add(l["kernel"])
    assert checked >= 4 and kinds == {"rowconv64", "rowconv128"}


def test_bf16_plans_route_small_512_channel_layers_to_the_band_kernel(monkeypatch):
    """from 128 workgroups up, bf16 plans run the 512 -> 512 stride-1 convs of the 16x16 / 8x8 levels on bandconv512; the
    packer's fragment-ordered copy (right behind the row layout in the blob) equals the numpy restatement the GPU test uses."""
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.engine import Engine
    e = Engine("normal", dtype="bf16", max_batch=8)
    band = [l["name"] for l in e.layers(8) if l["kernel"] == "bandconv512"]
    assert band == ["L4.d.res0.a", "L4.d.res0.b", "L5.d.res0.a", "L5.d.res0.b", "L6.u.res0.a", "L6.u.res0.b", "L5.u.res0.a", "L5.u.res0.b"]
    for l in e.layers(8):
        if l["kernel"] == "bandconv512":
            assert (l["cin"], l["cout"], l["stride"], l["upsample"]) == (512, 512, 1, 0) and l["h_out"] in (8, 16)
    assert not any(l["kernel"] == "bandconv512" for l in e.layers(1))            # 64 / 32 workgroups: stays on the implicit GEMM
    assert [l["name"] for l in e.layers(2) if l["kernel"] == "bandconv512"] == ["L4.d.res0.a", "L4.d.res0.b", "L5.u.res0.a", "L5.u.res0.b"]
    assert not any(l["kernel"] == "bandconv512" for l in Engine("normal", max_batch=8).layers(8))
    monkeypatch.setenv("LSP_HIP_BANDCONV", "0")
    assert not any(l["kernel"] == "bandconv512" for l in Engine("normal", dtype="bf16", max_batch=8).layers(8))
    monkeypatch.delenv("LSP_HIP_BANDCONV")

    topo, sd = synth.synthetic("normal", ngf=64, num_downs=6, size=256)      # 512 channels at 16x16 and 8x8
    s = Engine("normal", ngf=64, num_downs=6, size=256, dtype="bf16", max_batch=8, tune={"all_forms": 1})      # rows AND fragments in the blob
    s.load_state_dict(sd)
    blob = s.pack().numpy()
    checked = 0
    for i, l in enumerate(s.layers(8)):
        if l["kernel"] != "bandconv512":
            continue
        cout, nbytes = l["cout"], l["cout"] * 9 * 512 * 2
        fo = s.form_offset(i, "band")
        assert fo > l["w_offset"] >= 0
        rows = blob[l["w_offset"]: l["w_offset"] + nbytes].view(np.uint16).reshape(cout // 32, 32, 9, 4, 8, 2, 8)   # [cs][ch][tap][q][kc][hi][e]
        frag = blob[fo: fo + nbytes].view(np.uint16).reshape(cout // 32, 4, 9, 8, 2, 32, 8)
        assert np.array_equal(frag, rows.transpose(0, 3, 2, 4, 5, 1, 6)), l["name"]
        checked += 1
    assert checked >= 2


def test_16bit_plans_route_the_64_and_32_levels_to_the_patch_staged_kernel():
    """DESIGN.md 4.12 (round 6): from the batch that fills the chip with 256-pixel tiles the stride-1 single-source convs of the 64x64 / 32x32 levels of a 16-bit plan run on
    conv3x3_patch16 (the ResidualBlock convs, models/networks.py:650-675) -- 128 channels per workgroup when that gives >= patch16_min_blocks workgroups, else 64; it reads the
    implicit GEMM's own weight rows, so the blob carries no extra form for it; `patch16=0` puts them back."""
    from livespeechportraits_amd.engine import Engine
    e = Engine("normal", dtype="bf16", max_batch=8)
    at8 = {l["name"]: l for l in e.layers(8)}
    names = [n for n, l in at8.items() if l["kernel"] == "conv3x3_patch16"]
    assert names == ["L2.d.res0.a", "L2.d.res0.b", "L3.d.res0.a", "L3.d.res0.b", "L4.u.res0.a", "L4.u.res0.b", "L3.u.res0.a", "L3.u.res0.b"]
    assert (at8["L2.d.res0.a"]["tile_m"], at8["L2.d.res0.a"]["tile_n"]) == (256, 128)      # 8 x 16 pixel tiles x 2 channel tiles = 256 workgroups
    assert (at8["L3.d.res0.a"]["tile_m"], at8["L3.d.res0.a"]["tile_n"]) == (256, 64)       # 8 x 4 x 8 = 256
    assert all(at8[n]["split_k"] == 1 for n in names)
    assert not any(l["kernel"] == "conv3x3_patch16" for l in e.layers(1))                  # one frame: 16 / 4 pixel tiles, the implicit GEMM keeps them
    assert not any(l["kernel"] == "conv3x3_patch16" for l in Engine("normal", max_batch=8).layers(8))      # fp32 plans: Winograd
    assert not any(l["kernel"] == "conv3x3_patch16" for l in Engine("normal", dtype="bf16", max_batch=8, tune={"patch16": 0}).layers(8))
    assert len([l for l in Engine("large", dtype="f16", max_batch=8).layers(8) if l["kernel"] == "conv3x3_patch16"]) == 16
    # its sub-pixel up-conv form takes the up-convs over 16x16 / 32x32 / 64x64 sources (L4.up: one whole low-res frame per tile, 64 channels per workgroup; L3.up, L2.up); L5.up
    # (8x8 source) and L1.up (rowup256) keep their kernels
    ups = [n for n, l in at8.items() if l["kernel"] == "conv3x3_patchup16"]
    assert ups == ["L4.up", "L3.up", "L2.up"] and (at8["L3.up"]["tile_m"], at8["L3.up"]["tile_n"]) == (256, 128) and at8["L4.up"]["tile_n"] == 64
    assert at8["L5.up"]["kernel"].startswith("igemm3x3") and at8["L1.up"]["kernel"] == "rowup256"
    off = {l["name"]: l["kernel"] for l in Engine("normal", dtype="bf16", max_batch=8, tune={"patchup16": 0}).layers(8)}
    assert off["L3.up"].startswith("igemm3x3") and off["L2.d.res0.a"] == "conv3x3_patch16"


def test_16bit_plans_route_the_smallest_levels_to_the_full_k_kernel():
    """DESIGN.md 4.5 (round 5): from 2 frames up the 16-bit plans run the 4x4 / 2x2 levels and the convs that write 8x8 through a stride or an upsample on
    conv3x3_fullk16 -- whole K per workgroup, no split-K, no splitk_reduce launch (19 -> 8 of them in configs[2]); the packer's tile-blocked copy equals the
    numpy restatement the GPU test uses (tests/test_gpu_conv.py pack_fullk16)."""
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.engine import Engine
    e = Engine("normal", dtype="bf16", max_batch=8)
    at8 = e.layers(8)
    fk = [l["name"] for l in at8 if l["kernel"] == "conv3x3_fullk16"]
    assert fk == ["L5.down", "L6.down", "L6.d.res0.a", "L6.d.res0.b", "L7.down", "L7.d.res0.a", "L7.d.res0.b", "L7.up", "L7.u.res0.a", "L7.u.res0.b", "L6.up"]
    assert sum("splitk_reduce" in l["kernel"] for l in at8) == 2                      # <= 8 (19 in round 4; 7 in round 5; the four 32x32 ResidualBlock convs left their 2 K-splits for conv3x3_patch16 in round 6, L4.up for conv3x3_patchup16)
    for l in at8:
        if l["kernel"] == "conv3x3_fullk16":
            assert l["split_k"] == 1 and l["h_out"] in (2, 4, 8) and l["cout"] == 512
    assert not any(l["kernel"] == "conv3x3_fullk16" for l in e.layers(1))            # one frame: the tiny-M kernel / the implicit GEMM keep these levels
    assert len([l for l in e.layers(2) if l["kernel"] == "conv3x3_fullk16"]) >= 6
    assert not any(l["kernel"] == "conv3x3_fullk16" for l in Engine("normal", max_batch=8).layers(8))          # fp32 plans have conv3x3_fullk
    assert not any(l["kernel"] == "conv3x3_fullk16" for l in Engine("normal", dtype="bf16", max_batch=8, tune={"fullk16": 0}).layers(8))
    all8 = [l["name"] for l in Engine("normal", dtype="f16", max_batch=8, tune={"fullk16": 7}).layers(8) if l["kernel"] == "conv3x3_fullk16"]
    assert len(all8) == 15 and "L5.d.res0.a" in all8                                  # bit 2: the stride-1 8x8 layers too (bandconv512 by default)

    topo, sd = synth.synthetic("normal", ngf=64, num_downs=6, size=128)      # 512 channels at 8x8, 4x4, 2x2
    s = Engine("normal", ngf=64, num_downs=6, size=128, dtype="bf16", max_batch=4, tune={"all_forms": 1})      # rows AND the tile-blocked copy in the blob
    s.load_state_dict(sd)
    blob = s.pack().numpy()
    checked = 0
    for i, l in enumerate(s.layers(4)):
        if l["kernel"] != "conv3x3_fullk16":
            continue
        cout, cin = l["cout"], l["cin"]
        nch, c0 = (2, cin // 2) if l["concat"] else (1, cin)
        G, nbytes = c0 // 128, cout * 9 * cin * 2
        fo = s.form_offset(i, "fullk")
        assert fo >= 0 and l["w_offset"] >= 0 and fo != l["w_offset"]
        rows = blob[l["w_offset"]: l["w_offset"] + nbytes].view(np.uint16).reshape(cout // 16, 16, 9, nch, 4, 4, G, 8)     # [nt][li][tap][src][kq][wave][g][e]
        tiled = blob[fo: fo + nbytes].view(np.uint16).reshape(cout // 16, nch, 9, 4, G, 4, 16, 8)                         # [nt][src][tap][wave][g][kq][li][e]
        assert np.array_equal(tiled, rows.transpose(0, 3, 2, 5, 6, 4, 1, 7)), l["name"]
        checked += 1
    assert checked >= 6


def test_bf16_plans_route_the_edge_layers_of_the_256_level_to_row_kernels(monkeypatch):
    """in bf16 plans the last conv (GEMM form over two 64-channel sources) runs on rowlast128 and L1.up (two 128-channel
    sources -> 64) on rowup256 from 8-row strips up; the fragment-ordered copy of L1.up's weights equals the numpy restatement of the GPU test."""
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.engine import Engine
    e = Engine("normal", dtype="bf16", max_batch=8)
    l8 = e.layers(8)
    assert "rowlast128" in l8[-1]["kernel"]
    assert [l["name"] for l in l8 if l["kernel"] == "rowup256"] == ["L1.up"]
    assert [l["tile_m"] // 32 for l in l8 if l["kernel"] == "rowup256"] == [32]
    assert not any(l["kernel"] == "rowup256" for l in e.layers(1))
    assert "rowlast128" not in Engine("normal", max_batch=8).layers(8)[-1]["kernel"]        # fp32 plans: the direct kernels
    monkeypatch.setenv("LSP_HIP_ROWUP", "0"); monkeypatch.setenv("LSP_HIP_ROWLAST", "0")
    off = Engine("normal", dtype="bf16", max_batch=8).layers(8)
    monkeypatch.delenv("LSP_HIP_ROWUP"); monkeypatch.delenv("LSP_HIP_ROWLAST")
    assert not any(l["kernel"] == "rowup256" for l in off) and "rowlast128" not in off[-1]["kernel"]

    # the blob of `e` (max_batch 8) carries L1.up's rows (one-frame plans: implicit GEMM) and its rowup256 fragments (8-frame plans)
    e.load_state_dict(synth.make_state_dict(__import__("livespeechportraits_amd.topology", fromlist=["build_topology"]).build_topology("normal"), 7))
    blob = e.pack().numpy()
    # rowlast128's operand: the GEMM form's 12 rows ([4 parities x 3][9][128]) in the PAIRED order n' = ((py * cout + co) * 2 + px), so that a lane's accumulator registers are
    # (px 0, px 1) pairs and the fused epilogue writes 8-byte runs of an NCHW row (round 5); rows 12..15 of the 16-row MFMA operand are zero
    il = len(l8) - 1
    g_off, rl_off = e.form_offset(il, "gemm_last"), e.form_offset(il, "rowlast")
    assert g_off >= 0 and rl_off >= 0
    g = blob[g_off: g_off + 12 * 9 * 128 * 2].view(np.uint16).reshape(12, 9, 4, 4, 8)                     # [n][tap][kc][k-group][e]
    frag = blob[rl_off: rl_off + 9 * 4 * 64 * 8 * 2].view(np.uint16).reshape(9, 4, 4, 16, 8)              # [tap][kc][lane >> 4][lane & 15][e]
    for npr in range(16):
        q, px = npr >> 1, npr & 1
        want = g[(q // 3 * 2 + px) * 3 + q % 3].transpose(0, 1, 2, 3) if npr < 12 else np.zeros((9, 4, 4, 8), np.uint16)
        assert np.array_equal(frag[:, :, :, npr, :], want), npr
    assert g.any()
    i, l = [(i, x) for i, x in enumerate(l8) if x["kernel"] == "rowup256"][0]
    nbytes = 16 * 64 * 256 * 2
    fo = e.form_offset(i, "rowup")
    assert fo > l["w_offset"] >= 0
    rows = blob[l["w_offset"]: l["w_offset"] + nbytes].view(np.uint16).reshape(4, 2, 32, 4, 16, 2, 8)        # [par][nb][ch][tap][kc][hi][e]
    frag = blob[fo: fo + nbytes].view(np.uint16).reshape(2, 4, 4, 16, 2, 32, 8)
    assert np.array_equal(frag, rows.transpose(1, 0, 3, 4, 5, 2, 6))


def test_fp16_plan_takes_the_16bit_kernels_and_packs_rne_half():
    """dtype 'f16' (the reference's opt.fp16): 16-bit K-tiles and -- since the row / band kernels are templated on the storage type -- the same
    kernel per layer as the bf16 plan; no Winograd (fp32 only); conv weights narrowed to IEEE half with round-to-nearest-even by the host packer."""
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.engine import Engine
    for b in (1, 8):
        k16 = [l["kernel"] for l in Engine("normal", max_batch=8, dtype="f16").layers(b)]
        kbf = [l["kernel"] for l in Engine("normal", max_batch=8, dtype="bf16").layers(b)]
        assert k16 == kbf and not any(k.startswith("wino3x3") for k in k16)
    assert any(k.startswith("rowconv") for k in k16) and any(k == "bandconv512" for k in k16) and any(k == "rowup256" for k in k16)
    topo, sd = synth.synthetic("normal", ngf=64, num_downs=5, size=64)
    e = Engine("normal", ngf=64, num_downs=5, size=64, max_batch=8, dtype="f16", tune={"all_forms": 1})
    e.load_state_dict(sd)
    blob = e.pack().numpy()
    checked = 0
    for l, c in zip(e.layers(1), topo.convs):
        if not l["kernel"].startswith(("igemm3x3", "conv3x3_smallm")) or (l["upsample"] and l["h_out"] >= 32):
            continue
        w = sd[c.weight_key]
        exp = w.transpose(0, 2, 3, 1).astype(np.float16).reshape(-1)               # numpy casts with round-to-nearest-even
        got = blob[l["w_offset"]: l["w_offset"] + 2 * exp.size].view(np.float16)
        if l["weight_bytes"] == 2 * exp.size:
            assert np.array_equal(got, exp), l["name"]
            checked += 1
    assert checked >= 5
    assert Engine("normal", dtype="f16").packed_bytes() == Engine("normal", dtype="bf16").packed_bytes()
    with pytest.raises(Exception):
        Engine("normal", ngf=32, num_downs=5, size=64, dtype="f16")                  # a 16-bit K-tile is 64 channels


def test_full_k_kernel_splits_k_where_its_tiles_fill_half_the_chip(monkeypatch):
    """8x8 outputs at batch 1: 4 x 512 / 16 = 128 tiles for 256 CUs -> the full-K kernel runs with K in two halves (split_k == 2; single sources read
    as two half-sources, L6.up by source).  Not at 16x16 (256 tiles), not from batch 2 on, not with LSP_HIP_FULLK_SPLIT=0 (read BEFORE the plan is made)."""
    from livespeechportraits_amd.engine import Engine
    e = Engine("large", max_batch=2)
    one = {l["name"]: l for l in e.layers(1) if l["kernel"] == "conv3x3_fullk" and l["stride"] == 1}      # (the stride-2 ones: next test)
    assert len(one) == 18
    assert sorted(n for n, l in one.items() if l["split_k"] == 2) == sorted(n for n, l in one.items() if l["h_out"] == 8) and "L6.up" in one
    assert all(l["split_k"] == 1 for l in one.values() if l["h_out"] == 16)
    assert all(l["split_k"] == 1 for l in e.layers(2) if l["kernel"] == "conv3x3_fullk" and l["stride"] == 1)
    e.close()
    monkeypatch.setenv("LSP_HIP_FULLK_SPLIT", "0")
    off = Engine("large")
    assert all(l["split_k"] == 1 for l in off.layers(1) if l["kernel"] == "conv3x3_fullk") and not any(l["kernel"] == "conv3x3_fullk" and l["stride"] == 2 for l in off.layers(1))
    off.close()


def test_stride2_convs_of_the_small_levels_can_run_on_the_k_split_full_k_kernel(monkeypatch):
    """L4 / L5 / L6.down at batch 1 (BatchNorm plans): half the channels of their source band fit LDS, so conv3x3_fullk with K in two halves can run them.
    The whole forward measured slower with it (profiles/r03_fullk_stride2_ab.txt), so the planner only does so under LSP_HIP_FULLK_S2=1 (read before the
    plan is made); never from batch 2 on or under InstanceNorm plans."""
    from livespeechportraits_amd.engine import Engine
    downs = lambda ls: {l["name"]: (l["kernel"], l["split_k"]) for l in ls if l["name"] in ("L3.down", "L4.down", "L5.down", "L6.down")}
    assert all(k.startswith("igemm3x3") for k, _ in downs(Engine("large").layers(1)).values())
    monkeypatch.setenv("LSP_HIP_FULLK_S2", "1")
    e = Engine("large", max_batch=2)
    one = downs(e.layers(1))
    assert one["L4.down"] == one["L5.down"] == one["L6.down"] == ("conv3x3_fullk", 2) and one["L3.down"][0].startswith("igemm3x3")
    assert all(k.startswith("igemm3x3") for k, _ in downs(e.layers(2)).values())
    e.close()
    assert all(k.startswith("igemm3x3") for k, _ in downs(Engine("large", norm="instance").layers(1)).values())


def test_the_blob_carries_only_the_weight_forms_the_handle_can_run():
    """the packed blob (what rank 0 broadcasts and every rank keeps in HBM) holds, per layer, only the weight forms the plans of
    batch 1 .. max_batch read -- the planner is run for each of them at create.  `large` fp32 512x512 went from 878 MB (every form) to 565 MB
    for a one-frame handle; a layer every plan runs on the Winograd kernel has no 9-tap rows, the 16x16 layers of a one-frame handle no G g G^T."""
    from livespeechportraits_amd import _native as N
    from livespeechportraits_amd.engine import Engine
    e1, e8, all8 = Engine("large", max_batch=1), Engine("large", max_batch=8), Engine("large", max_batch=8, tune={"all_forms": 1})
    assert e1.packed_bytes() <= 600e6 < 720e6 and e1.packed_bytes() < e8.packed_bytes() < all8.packed_bytes()
    used_form = lambda l: ("wino4" if l["kernel"].startswith("wino4") else "wino" if l["kernel"].startswith("wino3x3") else "winoup" if l["kernel"].startswith("winoup")
                           else None)
    for e, batches in ((e1, (1,)), (e8, range(1, 9))):
        need = [set() for _ in e.layers(1)]
        for b in batches:
            for i, l in enumerate(e.layers(b)):
                f = used_form(l)
                if f is None and l["kernel"].startswith("conv3x3_fullk"):
                    f = "fullk2" if l["split_k"] == 2 and not l["concat"] else "fullk"
                need[i].add(f or "rows")
        for i, forms in enumerate(need):
            have = {f for f in ("rows", "fullk", "fullk2", "wino", "wino4", "winoup") if e.form_offset(i, f) >= 0}
            assert have == forms or e.layers(1)[i]["kernel"].startswith(("first_conv", "last_conv")), (i, e.layers(1)[i]["name"], have, forms)
    # what a plan reads is there: every layer's own form offset is valid for every batch of the range
    for b in range(1, 9):
        for i, l in enumerate(e8.layers(b)):
            f = used_form(l)
            if f:
                assert e8.form_offset(i, f) >= 0, (b, l["name"], f)
    for e in (e1, e8, all8):
        e.close()


def test_workspace_bytes_covers_every_smaller_batch():
    """The plans of different batch sizes pick different kernels, so the scratch a batch needs is not monotonic in it (with the planner of the day it was
    found, a 5-frame handle of a small net needed more for 3 frames -- an 18-way split-K slab -- than for 5, where the up-conv ran on winoup3x3): lspf2f_workspace_bytes(b)
    covers every batch of 1 .. b frames, which is what lets a host bind ONE workspace for max_batch (it failed on the GPU before: -5 STATE)."""
    from livespeechportraits_amd import _native as N
    from livespeechportraits_amd.engine import Engine
    for cfg in (("normal", 13, 1, 3, 32, 5, 64), ("large", 13, 1, 3, 32, 6, 128), ("normal", 13, 1, 3, 64, 8, 512)):
        e = Engine(*cfg, max_batch=8)
        need = [int(N.load().lspf2f_workspace_bytes(e._h, b)) for b in range(1, 9)]
        assert need == sorted(need) and need[0] > 0, (cfg, need)      # never less than any smaller batch needs
        e.close()


def test_a_second_pack_needs_the_tensors_again():
    """lspf2f_pack_weights frees the handle's fp32 copies of the state dict (the packed blob is what lives on); packing the same handle again without setting the
    tensors again must answer MISSING_TENSOR -- it dereferenced the freed vectors before round 4's last day -- and works, bit for bit, once they are set again."""
    from livespeechportraits_amd import _native as N, synth
    from livespeechportraits_amd.engine import Engine
    topo, sd = synth.synthetic("normal", ngf=32, num_downs=5, size=64)
    e = Engine("normal", ngf=32, num_downs=5, size=64)
    e.load_state_dict(sd)
    first = e.pack().clone()
    with pytest.raises(N.Lspf2fError, match="MISSING_TENSOR"):
        e.pack()
    e.load_state_dict(sd)
    assert torch.equal(e.pack(), first)
    e.close()


def test_planner_choices_of_the_small_levels_per_batch():
    """Which kernel takes the 16x16 ... 2x2 levels of `large` depends on the frame count (round 4 measured every cell: profiles/r04_fullk_small_levels_batch.txt):
    the tiny-M kernel while the level has <= 16 output pixels, the full-K kernel above that (one launch, no reduction), wino3x3 for the 16x16 level from 4 frames,
    winoup3x3 for the up-conv that writes 16x16 from 2 frames; the stride-2 convs stay on the implicit GEMM."""
    from livespeechportraits_amd.engine import Engine
    e = Engine("large", max_batch=8)
    want = {
        #     L4.d.res (16x16)   L5.d.res (8x8)    L6.d.res (4x4)     L7.d.res (2x2)     L7.up (-> 4x4)     L5.up (-> 16x16)
        1: ("conv3x3_fullk", "conv3x3_fullk", "conv3x3_smallm", "conv3x3_smallm", "conv3x3_smallm", "conv3x3_fullk"),
        2: ("conv3x3_fullk", "conv3x3_fullk", "conv3x3_fullk", "conv3x3_smallm", "conv3x3_fullk", "winoup3x3<1>"),
        4: ("wino3x3<1>", "conv3x3_fullk", "conv3x3_fullk", "conv3x3_smallm", "conv3x3_fullk", "winoup3x3<1>"),
        5: ("wino3x3<1>", "conv3x3_fullk", "conv3x3_fullk", "conv3x3_fullk", "conv3x3_fullk", "winoup3x3<1>"),
        8: ("wino3x3<1>", "conv3x3_fullk", "conv3x3_fullk", "conv3x3_fullk", "conv3x3_fullk", "winoup3x3<1>"),
    }
    names = ("L4.d.res0.a", "L5.d.res0.a", "L6.d.res0.a", "L7.d.res0.a", "L7.up", "L5.up")
    for b, exp in want.items():
        by = {l["name"]: l["kernel"].split(" ")[0] for l in e.layers(b)}
        assert tuple(by[n] for n in names) == exp, (b, [by[n] for n in names])
        assert all(by[n].startswith("igemm3x3") for n in ("L5.down", "L6.down")), b
    # one-frame handles carry no full-K tiles for the 4x4 / 2x2 levels (the blob holds only what the batch range reads)
    assert Engine("large", max_batch=1).packed_bytes() < Engine("large", max_batch=2).packed_bytes() < e.packed_bytes()
    e.close()
