"""Recurrent audio stages: APC GRU stack and Audio2Feature (MLP + LSTM + MLP).
Goldens are outputs of the reference's own classes (oracle/make_golden_rnn.py)."""
import argparse
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
TOL = 5e-5      # fp32 recurrences over up to 300 steps x 3 layers; measured values are printed


def gold(name):
    return json.load(open(os.path.join(GOLD, "rnn_%s.json" % name))), np.load(os.path.join(GOLD, "rnn_%s.npz" % name))["out"]


def a2f_opt(ff, gpu_ids, ckpt="none"):
    return argparse.Namespace(model="audio2feature", gpu_ids=gpu_ids, isTrain=False, checkpoints_dir="/tmp", name="a2f", load_epoch=ckpt,
                              verbose=False, feature_decoder="LSTM", loss="L2", A2L_GMM_ndim=75, A2L_GMM_ncenter=1, predict_length=1,
                              APC_hidden_size=512, frame_future=ff)


# ---- CPU ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["apc_t300", "apc_t1"])
def test_apc_oracle_reproduces_reference(name):
    from livespeechportraits_amd import synth
    from oracle import rnn_oracle
    meta, ref = gold(name)
    out = rnn_oracle.apc_forward(synth.make_apc_state_dict(), synth.make_mel(meta["T"]))
    assert np.abs(out - ref).max() <= 2e-6
    # and the published equations, in float64, for the first layer
    sd = synth.make_apc_state_dict()
    l0 = rnn_oracle.gru_cell_reference({k[7:]: v for k, v in sd.items() if k.startswith("rnns.0.")}, synth.make_mel(meta["T"]), 512)
    g = torch.nn.GRU(80, 512, batch_first=True)
    g.load_state_dict({k[7:]: torch.from_numpy(v) for k, v in sd.items() if k.startswith("rnns.0.")})
    assert np.abs(g(torch.from_numpy(synth.make_mel(meta["T"])).unsqueeze(0))[0][0].detach().numpy() - l0).max() <= 2e-6


@pytest.mark.parametrize("name", ["a2f_n120", "a2f_n40_ff0"])
def test_a2f_oracle_reproduces_reference(name):
    from livespeechportraits_amd import synth
    from oracle import rnn_oracle
    meta, ref = gold(name)
    feats = synth.symmetric(meta["rows"] * 512, 0.5, meta["feats_stream"]).reshape(meta["rows"], 512)
    out = rnn_oracle.a2f_generate(synth.make_a2f_state_dict(), feats, meta["frame_future"])
    assert out.shape == ref.shape and np.abs(out - ref).max() <= 2e-6


def test_state_dict_keys_match_reference():
    from livespeechportraits_amd.apc import APC_encoder
    from livespeechportraits_amd.audio2feature import Audio2Feature
    meta, _ = gold("apc_t300")
    assert {k: list(v.shape) for k, v in APC_encoder(80, 512, 3, False).state_dict().items()} == meta["keys"]
    meta, _ = gold("a2f_n120")
    assert {k: list(v.shape) for k, v in Audio2Feature(a2f_opt(18, [])).state_dict().items()} == meta["keys"]
    with pytest.raises(NotImplementedError):
        APC_encoder(80, 512, 3, True)


def test_library_exports_every_lsprnn_symbol_and_validates_config():
    from livespeechportraits_amd import _native as N
    from livespeechportraits_amd.rnn_engine import RecurrentEngine
    hdr = open(os.path.join(ROOT, "include", "lsprnn.h")).read()
    declared = set(re.findall(r"\b(lsprnn_[a-z0-9_]+)\s*\(", hdr))
    lib = ctypes.CDLL(N.LIB_PATH)
    assert declared and all(hasattr(lib, n) for n in declared)
    assert declared == set(N.RNN_SIGNATURES), declared ^ set(N.RNN_SIGNATURES)
    assert set(RecurrentEngine("GRU", 2, 80, 512).tensor_keys()) == {"%s_l%d" % (n, l) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh") for l in (0, 1)}
    for bad in (dict(hidden_size=128), dict(input_size=81), dict(num_layers=0), dict(num_layers=9)):
        kw = dict(cell="LSTM", num_layers=3, input_size=512, hidden_size=256)
        kw.update(bad)
        with pytest.raises(N.LsprnnError):
            RecurrentEngine(**kw)
    with pytest.raises(ValueError):
        RecurrentEngine("RNN", 1, 80, 512)
    e = RecurrentEngine("GRU", 1, 80, 512)
    with pytest.raises(KeyError):
        e.load_state_dict({"weight_ih_l0": np.zeros((1536, 80), np.float32)})


def test_packed_recurrent_weights_layout():
    from livespeechportraits_amd import _native as N, synth
    from livespeechportraits_amd.rnn_engine import RecurrentEngine
    sd = synth.make_rnn_state_dict("LSTM", 1, 512, 256)
    e = RecurrentEngine("LSTM", 1, 512, 256)
    e.load_state_dict(sd)
    n = e.lib.lsprnn_packed_bytes(e.h)
    buf = np.zeros(n // 4, np.float32)
    N.check_rnn(e.lib.lsprnn_pack_weights(e.h, buf.ctypes.data_as(ctypes.c_void_p), n))
    whh, bih, bhh = sd["weight_hh_l0"], sd["bias_ih_l0"], sd["bias_hh_l0"]
    # thread t of workgroup w holds unit w*64 + t//8, columns (t%8)*32 + q*4 + e of gate g at float4 ((w*4+g)*8+q)*512 + t
    w, g, q, t, el = 2, 3, 5, 8 * 9 + 6, 2
    want = whh[g * 256 + w * 64 + t // 8, (t % 8) * 32 + q * 4 + el]
    hits = np.flatnonzero(buf == want)
    base = [h - ((((w * 4 + g) * 8 + q) * 512 + t) * 4 + el) for h in hits]
    base = [b for b in base if b >= 0 and b % 64 == 0]
    assert base
    for (ww, gg, qq, tt, ee) in [(0, 0, 0, 0, 0), (3, 3, 7, 511, 3), (1, 2, 4, 100, 1)]:
        assert buf[base[0] + ((((ww * 4 + gg) * 8 + qq) * 512 + tt) * 4 + ee)] == whh[gg * 256 + ww * 64 + tt // 8, (tt % 8) * 32 + qq * 4 + ee]
    assert np.isin(np.float32(bih[:16] + bhh[:16]), buf).all()


# ---- GPU ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["apc_t300", "apc_t1"])
def test_apc_encoder_matches_reference(name):
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.apc import APC_encoder
    meta, ref = gold(name)
    dev = torch.device("cuda:0")
    net = APC_encoder(80, 512, 3, False)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_apc_state_dict().items()})
    net = net.to(dev).eval()
    mel = torch.from_numpy(synth.make_mel(meta["T"])).to(dev).unsqueeze(0)
    with torch.no_grad():
        out = net.forward(mel, torch.Tensor([meta["T"]]))[0]      # demo.py:188-190
    assert net._engine.status() == 0
    err = np.abs(out.cpu().numpy() - ref).max()
    print("\n[rnn %s] APC GRU x3 max-abs vs reference %.3e (|ref| max %.2f)" % (name, err, np.abs(ref).max()))
    assert out.shape == ref.shape and err <= TOL
    out2 = net.forward(mel, torch.Tensor([meta["T"]]))[0]
    assert torch.equal(out, out2), "not deterministic"


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["a2f_n120", "a2f_n40_ff0"])
def test_audio2feature_model_matches_reference(name, tmp_path):
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.models import create_model
    meta, ref = gold(name)
    ckpt = os.path.join(tmp_path, "Audio2Feature.pkl")
    torch.save({"module." + k: torch.from_numpy(v) for k, v in synth.make_a2f_state_dict().items()}, ckpt)
    opt = a2f_opt(meta["frame_future"], [0], ckpt)
    m = create_model(opt)
    m.setup(opt)
    m.eval()
    feats = synth.symmetric(meta["rows"] * 512, 0.5, meta["feats_stream"]).reshape(meta["rows"], 512)
    out = m.generate_sequences(feats, 16000, 60, fill_zero=True, opt=opt)
    err = np.abs(out - ref).max()
    print("\n[rnn %s] Audio2Feature max-abs vs reference %.3e (|ref| max %.2f)" % (name, err, np.abs(ref).max()))
    assert out.shape == ref.shape and out.dtype == np.float32 and err <= TOL


@pytest.mark.gpu
@pytest.mark.parametrize("route", ["wave", "layers"])
def test_both_recurrent_routes_match_reference(route, monkeypatch):
    """The default for stacks is the wavefront kernel (all layers in one launch); LSP_RNN_KERNEL=layers runs one launch
    per layer.  Different summation orders, same goldens."""
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.apc import APC_encoder
    from livespeechportraits_amd.rnn_engine import RecurrentEngine
    monkeypatch.setenv("LSP_RNN_KERNEL", route)
    dev = torch.device("cuda:0")
    meta, ref = gold("apc_t300")
    net = APC_encoder(80, 512, 3, False)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_apc_state_dict().items()})
    net = net.to(dev).eval()
    out = net.forward(torch.from_numpy(synth.make_mel(300)).to(dev).unsqueeze(0), torch.Tensor([300]))[0]
    assert net._engine.status() == 0
    e1 = np.abs(out.cpu().numpy() - ref).max()
    sd = synth.make_rnn_state_dict("LSTM", 3, 512, 256, seed=23, prefix="")
    lstm = torch.nn.LSTM(512, 256, num_layers=3, batch_first=True)
    lstm.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    x = synth.symmetric(200 * 512, 0.5, 77).reshape(200, 512)
    with torch.no_grad():
        want = lstm(torch.from_numpy(x).unsqueeze(0))[0][0].numpy()
    e = RecurrentEngine("LSTM", 3, 512, 256, max_steps=256)
    e.load_state_dict(sd); e.bind(dev)
    got = e.forward(torch.from_numpy(x).to(dev))
    assert e.status() == 0
    e2 = np.abs(got.cpu().numpy() - want).max()
    print("\n[rnn route %s] GRU x3 vs reference %.2e, LSTM x3 vs torch %.2e" % (route, e1, e2))
    assert e1 <= TOL and e2 <= TOL


@pytest.mark.gpu
def test_lstm_and_gru_engines_against_torch_other_shapes():
    """hidden 256 GRU and hidden 512 LSTM (the two template instances no reference module uses), ragged lengths."""
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.rnn_engine import RecurrentEngine
    dev = torch.device("cuda:0")
    for cell, layers, n_in, H, T in (("GRU", 2, 36, 256, 77), ("LSTM", 2, 64, 512, 130), ("LSTM", 1, 512, 256, 1)):
        sd = synth.make_rnn_state_dict(cell, layers, n_in, H, seed=5)
        mod = (torch.nn.GRU if cell == "GRU" else torch.nn.LSTM)(n_in, H, num_layers=layers, batch_first=True)
        mod.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        x = synth.symmetric(T * n_in, 1.0, 9).reshape(T, n_in)
        with torch.no_grad():
            want = mod(torch.from_numpy(x).unsqueeze(0))[0][0].numpy()
        e = RecurrentEngine(cell, layers, n_in, H, max_steps=256)
        e.load_state_dict(sd)
        e.bind(dev)
        got = e.forward(torch.from_numpy(x).to(dev))
        assert e.status() == 0
        assert np.abs(got.cpu().numpy() - want).max() <= TOL, (cell, H)


@pytest.mark.gpu
def test_linear_matches_torch():
    from livespeechportraits_amd.rnn_engine import Linear
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    for M, N_, K, bn, leaky in ((7, 75, 512, False, False), (130, 512, 1024, True, True), (1, 33, 36, True, False)):
        lin = torch.nn.Linear(K, N_)
        b = torch.nn.BatchNorm1d(N_).eval()
        with torch.no_grad():
            b.weight.normal_(1, 0.1, generator=g); b.bias.normal_(0, 0.1, generator=g)
            b.running_mean.normal_(0, 0.1, generator=g); b.running_var.uniform_(0.5, 1.5, generator=g)
        x = torch.randn(M, K, generator=g)
        with torch.no_grad():
            want = lin(x)
            if bn:
                want = b(want)
            if leaky:
                want = torch.nn.functional.leaky_relu(want, 0.2)
        mine = Linear(lin.weight.detach().numpy(), lin.bias.detach().numpy(),
                      (b.weight.detach().numpy(), b.bias.detach().numpy(), b.running_mean.numpy(), b.running_var.numpy()) if bn else None, leaky, dev)
        got = mine(x.to(dev)).cpu()
        assert (got - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())
