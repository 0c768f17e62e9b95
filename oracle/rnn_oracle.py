"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the reference's recurrent audio stages.  The arithmetic of nn.GRU / nn.LSTM lives in PyTorch, so
the restatement instantiates torch's own modules with the given weights and follows the reference's call sequence:
  apc_forward()          models/networks.py:36-66   APC_encoder.forward (pack_padded_sequence of ONE full-length
                                                     sequence is the identity, so each layer is rnn(x)[0])
  a2f_forward()          models/audio2feature.py:57-72  Audio2Feature.forward, LSTM decoder
  a2f_generate()         models/audio2feature_model.py:98-137  generate_sequences
  gru_cell_reference()   the published GRU equations, step by step in float64 numpy -- an independent check
oracle/make_golden_rnn.py asserts apc_forward / a2f_generate are bit-identical to the real reference classes."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def _t(v):
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))


def apc_forward(sd, mel, hidden=512, layers=3):
    """mel [T, mel_dim] -> [T, hidden]"""
    x = _t(mel).float().unsqueeze(0)
    with torch.no_grad():
        for i in range(layers):
            g = nn.GRU(input_size=x.shape[-1], hidden_size=hidden, batch_first=True)
            g.load_state_dict({k: _t(sd["rnns.%d.%s" % (i, k)]) for k in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")})
            x, _ = g(x)
    return x[0].numpy()


def a2f_forward(sd, feats, hidden=512):
    """feats [T2, hidden] -> [T2 // 2, out]"""
    W = {k: _t(v).float() for k, v in sd.items()}
    x = _t(feats).float().reshape(-1, 2 * hidden)
    bn = lambda h, p: F.batch_norm(h, W[p + ".running_mean"], W[p + ".running_var"], W[p + ".weight"], W[p + ".bias"], False, 0.1, 1e-5)
    with torch.no_grad():
        h = F.leaky_relu(bn(F.linear(x, W["downsample.0.weight"], W["downsample.0.bias"]), "downsample.1"), 0.2)
        h = F.linear(h, W["downsample.3.weight"], W["downsample.3.bias"])
        lstm = nn.LSTM(input_size=hidden, hidden_size=256, num_layers=3, batch_first=True)
        lstm.load_state_dict({k[5:]: v for k, v in W.items() if k.startswith("LSTM.")})
        h, _ = lstm(h.unsqueeze(0))
        h = h.reshape(-1, 256)
        h = F.leaky_relu(bn(F.linear(h, W["fc.0.weight"], W["fc.0.bias"]), "fc.1"), 0.2)
        h = F.leaky_relu(bn(F.linear(h, W["fc.3.weight"], W["fc.3.bias"]), "fc.4"), 0.2)
        return F.linear(h, W["fc.6.weight"], W["fc.6.bias"]).numpy()


def a2f_generate(sd, audio_feats, frame_future, hidden=512):
    feats = np.asarray(audio_feats, np.float32)
    nframe = int(feats.shape[0] / 2)
    if frame_future:
        feats = np.concatenate([feats, np.repeat(feats[-1], 2 * frame_future).reshape(-1, 2 * frame_future).T])
    preds = a2f_forward(sd, feats, hidden)
    preds = preds[frame_future:] if frame_future else preds
    assert preds.shape[0] == nframe
    return preds


def gru_cell_reference(sd, x, hidden, prefix="", layer=0):
    """One GRU layer over x [T, in] in float64 from the equations in include/lsprnn.h."""
    g = lambda n: np.asarray(sd["%s%s_l%d" % (prefix, n, layer)], np.float64)
    wi, wh, bi, bh = g("weight_ih"), g("weight_hh"), g("bias_ih"), g("bias_hh")
    H = hidden
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    h = np.zeros(H)
    out = np.zeros((x.shape[0], H))
    for t in range(x.shape[0]):
        gi = wi @ np.asarray(x[t], np.float64) + bi
        gh = wh @ h + bh
        r = sig(gi[:H] + gh[:H]); z = sig(gi[H:2 * H] + gh[H:2 * H])
        n = np.tanh(gi[2 * H:] + r * gh[2 * H:])
        h = (1 - z) * n + z * h
        out[t] = h
    return out
