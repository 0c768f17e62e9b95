"""GPU parity of the head-pose generator (csrc/a2h.hip through include/lspa2h.h) against outputs of the real
reference frozen in tests/golden/a2h_*.npz, and against the oracle on live inputs."""
import argparse
import os

import numpy as np
import pytest
import torch

from test_a2h_cpu import load_case

pytestmark = pytest.mark.gpu
TOL = 2e-4          # fp32, 14 gated layers, feedback over up to 300 frames; measured values are printed


def make_engine(cfg, sd, dev, max_audio_frames=2048, single_workgroup=False):
    from livespeechportraits_amd.a2h_engine import HeadposeEngine
    e = HeadposeEngine(**{k: cfg[k] for k in ("residual_layers", "residual_blocks", "residual_channels", "dilation_channels",
                                              "skip_channels", "kernel_size", "input_channels", "cond_channels", "hidden_size",
                                              "ncenter", "ndim", "loss")}, max_audio_frames=max_audio_frames,
                       single_workgroup=single_workgroup)
    e.load_state_dict(sd)
    e.bind(dev)
    return e


def run(e, cfg, audio, pre, noise, expq, sigma, ff, dev):
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    out = e.generate(d(audio), d(pre), d(noise) if cfg["loss"] == "GMM" else None,
                     d(expq) if cfg["ncenter"] > 1 and cfg["loss"] == "GMM" else None, sigma, ff)
    assert e.status() == 0, "an inter-workgroup hand-off timed out"
    return out.cpu().numpy()


@pytest.mark.parametrize("kernel", ["pipeline", "single_workgroup"])
@pytest.mark.parametrize("name", ["default_n48", "default_n300", "nc2_l4b1", "l2_l5b2"])
def test_generate_matches_reference_golden(name, kernel):
    dev = torch.device("cuda:0")
    meta, cfg, sd, audio, pre, ref, noise, expq = load_case(name)
    e = make_engine(cfg, sd, dev, single_workgroup=kernel == "single_workgroup")
    out = run(e, cfg, audio, pre, noise, expq, meta["sigma_scale"], meta["frame_future"], dev)
    err = np.abs(out - ref).max()
    print("\n[a2h %s %s] max-abs vs reference %.3e (|ref| max %.2f)" % (name, kernel, err, np.abs(ref).max()))
    assert out.shape == ref.shape and err <= TOL
    # deterministic: same call twice -> identical bits (no atomics, fixed reduction order)
    out2 = run(e, cfg, audio, pre, noise, expq, meta["sigma_scale"], meta["frame_future"], dev)
    assert np.array_equal(out, out2)


def test_both_kernels_agree_bit_for_bit():
    """Same thread maps and summation order in both kernels: any stale, torn or misrouted hand-off granule of the
    pipelined kernel would show up as a difference from the single-workgroup one."""
    dev = torch.device("cuda:0")
    for name in ("default_n300", "nc2_l4b1"):
        meta, cfg, sd, audio, pre, ref, noise, expq = load_case(name)
        outs = [run(make_engine(cfg, sd, dev, single_workgroup=s), cfg, audio, pre, noise, expq, meta["sigma_scale"],
                    meta["frame_future"], dev) for s in (False, True)]
        assert np.array_equal(outs[0], outs[1])


def test_cond_features_match_oracle():
    from oracle import a2h_oracle
    dev = torch.device("cuda:0")
    meta, cfg, sd, audio, pre, ref, noise, expq = load_case("default_n48")
    e = make_engine(cfg, sd, dev)
    run(e, cfg, audio, pre, noise, expq, 0.3, meta["frame_future"], dev)
    W = a2h_oracle._t(sd)
    want = a2h_oracle._downsample(W, torch.from_numpy(audio)).numpy()
    got = e.debug_cond().cpu().numpy()
    assert got.shape == want.shape and np.abs(got - want).max() <= 1e-4 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("single", [False, True])
def test_live_oracle_other_lengths_and_sigma_zero(single):
    """nframe 1 (shortest), ragged n_audio (not a multiple of the GEMM tile), sigma 0 with no noise tensor."""
    from livespeechportraits_amd import synth
    from oracle import a2h_oracle
    dev = torch.device("cuda:0")
    cfg = dict(synth.A2H_DEFAULTS, residual_layers=5, residual_blocks=2)
    sd = synth.make_a2h_state_dict(cfg, seed=9)
    e = make_engine(cfg, sd, dev, max_audio_frames=100, single_workgroup=single)
    for n_audio, ff in ((100, 15), (1, 0), (70, 5), (100, 15)):      # longer call first: its mailbox slots must not leak into later calls
        audio, pre = synth.make_a2h_inputs(n_audio, cfg, seed=n_audio)
        nframe = n_audio - ff
        g = torch.Generator().manual_seed(n_audio)
        noise = torch.randn(nframe, 12, generator=g).numpy()
        want = a2h_oracle.stream(sd, cfg, audio, pre, noise, np.ones((nframe, 1)), 0.4, ff)
        got = run(e, cfg, audio, pre, noise, None, 0.4, ff, dev)
        assert np.abs(got - want).max() <= TOL
        want0 = a2h_oracle.stream(sd, cfg, audio, pre, np.zeros((nframe, 12)), np.ones((nframe, 1)), 0.0, ff)
        got0 = e.generate(torch.from_numpy(audio).to(dev), torch.from_numpy(pre).to(dev), None, None, 0.0, ff).cpu().numpy()
        assert e.status() == 0
        assert np.abs(got0 - want0).max() <= TOL


def test_argument_errors():
    from livespeechportraits_amd import _native as N, synth
    dev = torch.device("cuda:0")
    cfg = dict(synth.A2H_DEFAULTS, residual_layers=3, residual_blocks=1, ncenter=2)
    e = make_engine(cfg, synth.make_a2h_state_dict(cfg), dev, max_audio_frames=64)
    audio, pre = synth.make_a2h_inputs(40, cfg)
    a, p = torch.from_numpy(audio).to(dev), torch.from_numpy(pre).to(dev)
    noise = torch.zeros(38, 12, device=dev)
    with pytest.raises(N.Lspa2hError, match="expq"):
        e.generate(a, p, noise, None, 0.3, 2)                       # ncenter 2 needs the Exp(1) draws
    with pytest.raises(ValueError):
        e.generate(a, p, torch.zeros(40, 12, device=dev), None, 0.3, 2)     # noise rows != nframe
    with pytest.raises(ValueError):
        e.generate(a.cpu(), p, None, None, 0.0, 2)                  # no CPU path
    big, _ = synth.make_a2h_inputs(65, cfg)
    with pytest.raises(N.Lspa2hError, match="max_audio_frames"):
        e.generate(torch.from_numpy(big).to(dev), p, None, torch.ones(65, 2, device=dev), 0.0, 0)
    with pytest.raises(N.Lspa2hError, match="nframe|null argument"):       # empty output tensor -> null pointer
        e.generate(a, p, None, torch.ones(0, 2, device=dev), 0.0, 40)      # nframe 0


def test_drop_in_model_reproduces_seeded_reference_run(tmp_path):
    """create_model(opt) -> setup (checkpoint with 'module.' keys) -> generate_sequences under the golden's torch seed
    == the reference's own seeded run (same CPU RNG stream, same numbers)."""
    from livespeechportraits_amd.models import create_model
    meta, cfg, sd, audio, pre, ref, noise, expq = load_case("default_n48")
    ckpt = os.path.join(tmp_path, "Audio2Headpose.pkl")
    torch.save({"module." + k: torch.from_numpy(v) for k, v in sd.items()}, ckpt)
    opt = argparse.Namespace(
        model="audio2headpose", gpu_ids=[0], isTrain=False, checkpoints_dir=str(tmp_path), name="x", load_epoch=ckpt, verbose=False,
        feature_decoder="WaveNet", loss="GMM", A2H_GMM_ndim=12, A2H_GMM_ncenter=1, APC_hidden_size=512,
        A2H_wavenet_residual_layers=7, A2H_wavenet_residual_blocks=2, A2H_wavenet_residual_channels=128,
        A2H_wavenet_dilation_channels=128, A2H_wavenet_skip_channels=256, A2H_wavenet_kernel_size=2, time_frame_length=1,
        A2H_wavenet_use_bias=True, A2H_wavenet_input_channels=12, A2H_wavenet_cond_channels=512, frame_future=meta["frame_future"])
    m = create_model(opt)
    m.setup(opt)
    m.eval()
    net = m.Audio2Headpose.module
    opt.A2H_receptive_field = net.WaveNet.receptive_field              # demo.py:164
    torch.manual_seed(meta["torch_seed"])
    out = m.generate_sequences(audio.reshape(-1, 2, 512), pre, fill_zero=True, sigma_scale=meta["sigma_scale"], opt=opt)
    assert out.shape == ref.shape and out.dtype == np.float64
    assert np.abs(out - ref).max() <= TOL
    assert m.generate_sequences(audio, pre, fill_zero=False, opt=opt) is None      # reference :166-167


@pytest.mark.parametrize("name", ["lstm_nc1", "lstm_nc3", "lstm_l2"])
def test_lstm_decoder_reproduces_seeded_reference_run(name, tmp_path):
    """feature_decoder == 'LSTM' (reference audio2headpose.py:56-100, audio2headpose_model.py:189-202): one forward over
    all audio rows, one vectorised Sample_GMM; under the golden's torch seed the numbers are the reference's."""
    import json
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.models import create_model
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    meta = json.load(open(os.path.join(gold, "a2h_%s.json" % name)))
    ref = np.load(os.path.join(gold, "a2h_%s.npz" % name))["out"]
    cfg = meta["cfg"]
    sd = synth.make_a2h_lstm_state_dict(512, cfg["ncenter"], 12, cfg["loss"], seed=meta["weights_seed"])
    ckpt = os.path.join(tmp_path, "Audio2Headpose.pkl")
    torch.save({"module." + k: torch.from_numpy(v) for k, v in sd.items()}, ckpt)
    opt = argparse.Namespace(model="audio2headpose", gpu_ids=[0], isTrain=False, checkpoints_dir=str(tmp_path), name="x", load_epoch=ckpt,
                             verbose=False, feature_decoder="LSTM", loss=cfg["loss"], A2H_GMM_ndim=12, A2H_GMM_ncenter=cfg["ncenter"],
                             APC_hidden_size=512, frame_future=15)
    m = create_model(opt)
    m.setup(opt)
    m.eval()
    audio, pre = synth.make_a2h_inputs(meta["rows"], cfg, seed=meta["inputs_seed"])
    torch.manual_seed(meta["torch_seed"])
    out = m.generate_sequences(audio, pre, fill_zero=True, sigma_scale=meta["sigma_scale"], opt=opt)
    err = np.abs(out - ref).max()
    print("\n[a2h %s] LSTM decoder max-abs vs reference %.3e" % (name, err))
    assert out.shape == ref.shape == (meta["rows"], 12) and out.dtype == np.float32 and err <= 5e-5


def test_sample_gmm_entry_point():
    import ctypes
    from livespeechportraits_amd import _native as N
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    rows, nc, nd = 37, 4, 12
    params = torch.randn(rows, (2 * nd + 1) * nc, generator=g)
    noise, expq = torch.randn(rows, nd, generator=g), torch.empty(rows, nc).exponential_(1, generator=g)
    prob = torch.softmax(params[:, :nc], 1)
    idx = torch.argmax(prob / expq, 1)
    r = torch.arange(rows)
    mu = params[:, nc:nc + nc * nd].reshape(rows, nc, nd)[r, idx]
    sg = (torch.exp(-params[:, nc + nc * nd:]) * 0.7).reshape(rows, nc, nd)[r, idx]
    want = noise * sg + mu
    out = torch.empty(rows, nd, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    pd, nz, eq = params.to(dev), noise.to(dev), expq.to(dev)
    N.check_a2h(N.load().lspa2h_sample_gmm(p(pd), rows, nc, nd, p(nz), p(eq), ctypes.c_float(0.7), p(out), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    assert (out.cpu() - want).abs().max() <= 1e-5
    with pytest.raises(N.Lspa2hError, match="expq"):
        N.check_a2h(N.load().lspa2h_sample_gmm(p(pd), rows, nc, nd, p(nz), None, ctypes.c_float(0.7), p(out), None))
