"""Head-pose generator: oracle vs the frozen reference outputs, host logic, C ABI surface.
No GPU needed.  Goldens come from the real reference (oracle/make_golden_a2h.py)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def load_case(name):
    from livespeechportraits_amd import synth
    meta = json.load(open(os.path.join(GOLD, "a2h_%s.json" % name)))
    z = np.load(os.path.join(GOLD, "a2h_%s.npz" % name))
    cfg = meta["cfg"]
    sd = synth.make_a2h_state_dict(cfg, seed=meta["weights_seed"])
    audio, pre = synth.make_a2h_inputs(meta["nframe"] + meta["frame_future"], cfg, seed=meta["inputs_seed"])
    return meta, cfg, sd, audio, pre, z["out"], z["noise"], z["expq"]


@pytest.mark.parametrize("name", ["nc2_l4b1", "l2_l5b2", "default_n48"])
def test_oracle_reproduces_reference(name):
    from oracle import a2h_oracle
    meta, cfg, sd, audio, pre, ref, noise, expq = load_case(name)
    out = a2h_oracle.generate_sequences(sd, cfg, audio, pre, noise, expq, meta["sigma_scale"], meta["frame_future"])
    # same torch build as the one that froze the goldens => bit-identical; tolerate other CPU kernels
    assert np.abs(out - ref).max() <= 2e-5
    assert np.abs(ref).max() > 0.5, "degenerate golden"


@pytest.mark.parametrize("name", ["nc2_l4b1", "l2_l5b2", "default_n48"])
def test_streaming_evaluation_equals_sliding_window(name):
    """The algorithm of csrc/a2h.hip (dilation queues, one position per step) against the reference outputs."""
    from oracle import a2h_oracle
    meta, cfg, sd, audio, pre, ref, noise, expq = load_case(name)
    out = a2h_oracle.stream(sd, cfg, audio, pre, noise, expq, meta["sigma_scale"], meta["frame_future"])
    assert np.abs(out - ref).max() <= 2e-5


def test_rng_stream_matches_reference_draws():
    """draw_gmm_noise under the golden's torch seed == the draws the reference consumed."""
    from livespeechportraits_amd.audio2headpose_model import draw_gmm_noise
    for name in ("default_n48", "nc2_l4b1"):
        meta, cfg, *_rest, noise, expq = load_case(name)
        torch.manual_seed(meta["torch_seed"])
        n, q = draw_gmm_noise(meta["nframe"], cfg["ncenter"], cfg["ndim"])
        assert np.array_equal(n.numpy(), noise) and np.array_equal(q.numpy(), expq)


def test_multinomial_single_sample_is_argmax_of_prob_over_exponential():
    for nc in (1, 2, 5):
        for t in range(50):
            p = torch.softmax(torch.randn(1, nc, generator=torch.Generator().manual_seed(t)), 1)
            torch.manual_seed(t); a = torch.multinomial(p, 1, replacement=True); r1 = torch.randn(1, 12)
            torch.manual_seed(t); q = torch.empty(1, nc).exponential_(1); b = torch.argmax(p / q, 1, keepdim=True); r2 = torch.randn(1, 12)
            assert torch.equal(a, b) and torch.equal(r1, r2)


def test_state_dict_keys_match_reference():
    from livespeechportraits_amd import synth
    ref = json.load(open(os.path.join(GOLD, "keys_a2h.json")))
    mine = {k: list(v) for k, v in synth.a2h_shapes(synth.A2H_DEFAULTS).items()}
    assert {k: v for k, v in ref.items() if not k.endswith("num_batches_tracked")} == mine
    import argparse
    from livespeechportraits_amd.audio2headpose import Audio2Headpose
    opt = argparse.Namespace(loss="GMM", A2H_GMM_ndim=12, A2H_GMM_ncenter=1, APC_hidden_size=512, A2H_wavenet_residual_layers=7,
                             A2H_wavenet_residual_blocks=2, A2H_wavenet_residual_channels=128, A2H_wavenet_dilation_channels=128,
                             A2H_wavenet_skip_channels=256, A2H_wavenet_kernel_size=2, time_frame_length=1, A2H_wavenet_use_bias=True,
                             A2H_wavenet_input_channels=12, A2H_wavenet_cond_channels=512)
    net = Audio2Headpose(opt)
    assert {k: list(v.shape) for k, v in net.state_dict().items()} == ref          # incl. num_batches_tracked
    assert net.WaveNet.receptive_field == 255


# ---- C ABI surface ---------------------------------------------------------------------------
def test_library_exports_every_lspa2h_symbol():
    from livespeechportraits_amd import _native as N
    hdr = open(os.path.join(ROOT, "include", "lspa2h.h")).read()
    declared = set(re.findall(r"\b(lspa2h_[a-z0-9_]+)\s*\(", hdr))
    assert declared
    lib = ctypes.CDLL(N.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), "liblspf2f.so does not export %s" % name
    assert declared == set(N.A2H_SIGNATURES), declared ^ set(N.A2H_SIGNATURES)
    assert N.load().lspa2h_abi_version() == N.A2H_ABI_VERSION


def test_create_validates_config():
    from livespeechportraits_amd import _native as N
    from livespeechportraits_amd.a2h_engine import HeadposeEngine
    e = HeadposeEngine()
    assert e.receptive_field == 255 and len(e.tensor_keys()) == 184
    assert HeadposeEngine(residual_layers=4, residual_blocks=1).receptive_field == 16
    for bad in (dict(residual_channels=64), dict(skip_channels=128), dict(kernel_size=3), dict(input_channels=6),
                dict(ndim=20, input_channels=20), dict(ncenter=9), dict(residual_layers=9, residual_blocks=2),   # queues > LDS
                dict(cond_channels=256)):
        with pytest.raises(N.Lspa2hError):
            HeadposeEngine(**bad)
    with pytest.raises(ValueError):
        HeadposeEngine(loss="L1")


def test_weight_ingress_errors_and_packed_layout():
    from livespeechportraits_amd import _native as N, synth
    from livespeechportraits_amd.a2h_engine import HeadposeEngine
    cfg = dict(synth.A2H_DEFAULTS, residual_layers=3, residual_blocks=1)
    sd = synth.make_a2h_state_dict(cfg)
    e = HeadposeEngine(residual_layers=3, residual_blocks=1)
    lib = e.lib
    nbytes = lib.lspa2h_packed_bytes(e.h)
    buf = np.zeros(nbytes // 4, np.float32)
    with pytest.raises(N.Lspa2hError, match="tensor not set"):           # the reference's strict=False is silent
        N.check_a2h(lib.lspa2h_pack_weights(e.h, buf.ctypes.data_as(ctypes.c_void_p), nbytes))
    with pytest.raises(KeyError):
        e.load_state_dict({k: v for k, v in sd.items() if "skip_conv" not in k})
    w = np.zeros(3, np.float32)
    with pytest.raises(N.Lspa2hError, match="unknown tensor key"):
        N.check_a2h(lib.lspa2h_set_tensor(e.h, b"WaveNet.nope", w.ctypes.data_as(ctypes.c_void_p), 3))
    with pytest.raises(N.Lspa2hError, match="element count"):
        N.check_a2h(lib.lspa2h_set_tensor(e.h, b"WaveNet.start_conv1.bias", w.ctypes.data_as(ctypes.c_void_p), 3))
    e.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in sd.items()})        # DataParallel-style keys
    N.check_a2h(lib.lspa2h_pack_weights(e.h, buf.ctypes.data_as(ctypes.c_void_p), nbytes))
    # every filter/gate/residual/skip weight of layer 1 appears exactly where the kernel's thread map expects it:
    # find the fg block of layer 1 by content, then check the documented index formula on random entries
    fw, gw = sd["WaveNet.residual_blocks.1.filter_conv.weight"], sd["WaveNet.residual_blocks.1.gate_conv.weight"]
    t, j, q, el = 8 * 5 + 6, 3, 2, 1             # thread (u=5, part=6), item 3 = gate[u+64], float4 q=2, element 1
    c = 6 * 32 + q * 4 + el                      # column 201 -> tap 1, input channel 73
    want = gw[5 + 64, c - 128, 1]
    hits = np.flatnonzero(buf == want)
    assert len(hits) >= 1
    base = [h - (((j * 8 + q) * 512 + t) * 4 + el) for h in hits]
    fg1 = [b for b in base if b >= 0 and b % 64 == 0]
    assert fg1, "gate weight not at the packed position"
    b0 = fg1[0]
    for (tt, jj, qq, ee) in [(0, 0, 0, 0), (511, 3, 7, 3), (77, 1, 4, 2), (300, 2, 0, 1)]:
        cc = (tt & 7) * 32 + qq * 4 + ee
        src = (gw if jj & 1 else fw)[(tt >> 3) + (jj >> 1) * 64, cc & 127, 1 if cc >= 128 else 0]
        assert buf[b0 + ((jj * 8 + qq) * 512 + tt) * 4 + ee] == src
    # BatchNorm1d fold: scale = gamma / sqrt(var + 1e-5)
    s = sd["audio_downsample.1.weight"] / np.sqrt(sd["audio_downsample.1.running_var"].astype(np.float64) + 1e-5)
    assert np.isin(np.float32(s[:8]), buf).all()


def test_model_requires_a_device_and_wavenet_decoder():
    import argparse
    from livespeechportraits_amd.models import create_model
    base = dict(model="audio2headpose", gpu_ids=[], isTrain=False, checkpoints_dir="/tmp", name="t", load_epoch="none",
                feature_decoder="WaveNet", loss="GMM")
    with pytest.raises(RuntimeError, match="no CPU path"):
        create_model(argparse.Namespace(**base))
    with pytest.raises(NotImplementedError):
        create_model(argparse.Namespace(**dict(base, feature_decoder="GRU", gpu_ids=[0])))      # the reference knows WaveNet and LSTM
