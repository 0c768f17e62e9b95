"""Hand-off robustness of the multi-workgroup kernels (csrc/a2h.hip a2h_pipe, csrc/rnn.hip rnn_layer): the protocols must
not depend on timing, placement or an idle GPU.  Every repetition must be bit-identical to a quiet run and report no
timeout while the renderer saturates the device on another stream (cdna_hip_programming.md Guideline 16, pitfall 3:
a test on an idle, L1-cold GPU cannot see a missing acquire)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _renderer(dev):
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.engine import Engine
    from livespeechportraits_amd.topology import build_topology
    topo = build_topology("normal", size=512)
    e = Engine("normal", size=512, max_batch=8)
    e.load_state_dict(synth.make_state_dict(topo, 1234))
    e.bind(e.pack(), dev)
    feat, cand = synth.make_inputs(8, 512, seed=99, cand_batch=1)
    return e, torch.from_numpy(feat).to(dev), torch.from_numpy(cand).to(dev)


def test_headpose_pipeline_under_load_is_bit_stable():
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.a2h_engine import HeadposeEngine
    dev = torch.device("cuda:0")
    cfg, ff, nframe = dict(synth.A2H_DEFAULTS), 15, 400
    e = HeadposeEngine(max_audio_frames=nframe + ff)
    e.load_state_dict(synth.make_a2h_state_dict(cfg))
    e.bind(dev)
    audio, pre = synth.make_a2h_inputs(nframe + ff, cfg)
    au, pr = torch.from_numpy(audio).to(dev), torch.from_numpy(pre).to(dev)
    noise = torch.from_numpy(synth.symmetric(nframe * 12, 1.0, 5).reshape(nframe, 12)).to(dev)
    quiet = e.generate(au, pr, noise, None, 0.3, ff)
    assert e.status() == 0
    quiet = quiet.cpu().numpy()
    ref_single = HeadposeEngine(max_audio_frames=nframe + ff, single_workgroup=True)
    ref_single.load_state_dict(synth.make_a2h_state_dict(cfg)); ref_single.bind(dev)
    assert np.array_equal(ref_single.generate(au, pr, noise, None, 0.3, ff).cpu().numpy(), quiet)

    rend, f8, c = _renderer(dev)
    side = torch.cuda.Stream(dev)
    o8 = torch.empty((8, 3, 512, 512), device=dev)
    for rep in range(12):
        with torch.cuda.stream(side):                      # ~15 ms of renderer work queued per repetition: the GPU stays busy
            for _ in range(4):
                rend.forward(f8, c, o8)
        out = e.generate(au, pr, noise, None, 0.3, ff)
        assert e.status() == 0, "hand-off timed out under load (rep %d)" % rep
        assert np.array_equal(out.cpu().numpy(), quiet), "repetition %d differs from the quiet run" % rep
    torch.cuda.synchronize()


def test_recurrent_stacks_under_load_are_bit_stable():
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.rnn_engine import RecurrentEngine
    dev = torch.device("cuda:0")
    rend, f8, c = _renderer(dev)
    side = torch.cuda.Stream(dev)
    o8 = torch.empty((8, 3, 512, 512), device=dev)
    for cell, n_in, H, T in (("GRU", 80, 512, 700), ("LSTM", 512, 256, 500)):
        e = RecurrentEngine(cell, 3, n_in, H, max_steps=T)
        e.load_state_dict(synth.make_rnn_state_dict(cell, 3, n_in, H))
        e.bind(dev)
        x = torch.from_numpy(synth.symmetric(T * n_in, 0.7, 3).reshape(T, n_in)).to(dev)
        quiet = e.forward(x)
        assert e.status() == 0
        quiet = quiet.cpu().numpy()
        for rep in range(12):
            with torch.cuda.stream(side):
                for _ in range(3):
                    rend.forward(f8, c, o8)
            out = e.forward(x)
            assert e.status() == 0, "%s hand-off timed out under load (rep %d)" % (cell, rep)
            assert np.array_equal(out.cpu().numpy(), quiet), "%s repetition %d differs from the quiet run" % (cell, rep)
    torch.cuda.synchronize()


def test_checked_calls_retry_a_lost_handoff_and_return_the_same_numbers(monkeypatch):
    """generate_checked / forward_checked (what the model classes call): a time-out reported by the status word -- simulated here on the
    first attempt, it needs a saturated device for tens of milliseconds to happen for real -- makes them drain the device and run the call
    again; the retry is bit-identical because the kernels are deterministic and the random draws are inputs.  A persistent time-out raises."""
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.a2h_engine import HeadposeEngine
    from livespeechportraits_amd.rnn_engine import RecurrentEngine
    dev = torch.device("cuda:0")
    cfg, ff, nframe = dict(synth.A2H_DEFAULTS), 15, 60
    e = HeadposeEngine(max_audio_frames=nframe + ff)
    e.load_state_dict(synth.make_a2h_state_dict(cfg))
    e.bind(dev)
    audio, pre = synth.make_a2h_inputs(nframe + ff, cfg)
    au, pr = torch.from_numpy(audio).to(dev), torch.from_numpy(pre).to(dev)
    noise = torch.from_numpy(synth.symmetric(nframe * 12, 1.0, 5).reshape(nframe, 12)).to(dev)
    quiet = e.generate_checked(au, pr, noise, None, 0.3, ff).cpu().numpy()
    real_status = e.status
    calls = {"n": 0}

    def flaky(device=None):
        calls["n"] += 1
        code = real_status(device)
        return 0x77 if calls["n"] == 1 else code
    monkeypatch.setattr(e, "status", flaky)
    again = e.generate_checked(au, pr, noise, None, 0.3, ff).cpu().numpy()
    assert calls["n"] == 2 and np.array_equal(again, quiet)
    monkeypatch.setattr(e, "status", lambda device=None: 0x77)
    with pytest.raises(RuntimeError, match="timed out 3 times"):
        e.generate_checked(au, pr, noise, None, 0.3, ff)

    g = RecurrentEngine("GRU", 3, 80, 512, max_steps=256)
    g.load_state_dict(_gru_sd(synth))
    g.bind(dev)
    x = torch.from_numpy(synth.make_mel(200)).to(dev)
    q = g.forward_checked(x).cpu().numpy()
    rs = g.status
    n = {"n": 0}

    def flaky2():
        n["n"] += 1
        c = rs()
        return 5 if n["n"] == 1 else c
    monkeypatch.setattr(g, "status", flaky2)
    assert np.array_equal(g.forward_checked(x).cpu().numpy(), q) and n["n"] == 2


def _gru_sd(synth):
    """the APC encoder's three single-layer GRUs as one 3-layer stack's tensors (the mapping livespeechportraits_amd/apc.py uses)"""
    sd = synth.make_apc_state_dict()
    out = {}
    for k, v in sd.items():             # rnns.<i>.weight_ih_l0 -> weight_ih_l<i>
        i = int(k.split(".")[1])
        out[k.split(".")[2].replace("_l0", "_l%d" % i)] = v
    return out
