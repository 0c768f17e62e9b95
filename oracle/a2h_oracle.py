"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the reference's head-pose generation loop:

  generate_sequences()   models/audio2headpose_model.py:133-187  sliding window, one frame per iteration
  _forward()             models/audio2headpose.py:40-52           audio_downsample MLP + WaveNet
  _wavenet()             models/networks.py:186-214               start convs, residual blocks, end convs
  _block()               models/networks.py:299-326               pad, dilated filter/gate convs, cond, gated unit
  sample_gmm()           models/losses.py:68-112                   Sample_GMM with the random draws passed in

Same torch ops in the same order as the reference (F.pad + F.conv1d with dilation, eval BatchNorm1d via
F.batch_norm), so on CPU it is bit-identical to it when given the reference's random draws;
oracle/make_golden_a2h.py asserts that against the real reference and freezes its outputs in tests/golden/.

``stream()`` is the incremental evaluation the HIP kernel uses (per-layer dilation queues), in numpy float64:
a second, independent route to the same numbers."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def _t(sd):
    return {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))).float() for k, v in sd.items()}


def _block(W, p, x, cond, dilation):                       # networks.py:299-326
    x_pad = F.pad(x, (dilation, 0))                        # (kernel_size - 1) * dilation zeros on the left, :259
    filt = F.conv1d(x_pad, W[p + "filter_conv.weight"], W[p + "filter_conv.bias"], dilation=dilation)
    gate = F.conv1d(x_pad, W[p + "gate_conv.weight"], W[p + "gate_conv.bias"], dilation=dilation)
    filt = filt + F.conv1d(cond, W[p + "cond_filter_conv.weight"], W[p + "cond_filter_conv.bias"])
    gate = gate + F.conv1d(cond, W[p + "cond_gate_conv.weight"], W[p + "cond_gate_conv.bias"])
    z = torch.tanh(filt) * torch.sigmoid(gate)
    residual = F.conv1d(z, W[p + "residual_conv.weight"], W[p + "residual_conv.bias"]) + x
    skip = F.conv1d(z, W[p + "skip_conv.weight"], W[p + "skip_conv.bias"])
    return residual, skip


def _wavenet(W, cfg, x, cond, output_length=1):            # networks.py:186-214 (dropout2D is identity in eval)
    act = lambda t: F.leaky_relu(t, 0.2)
    x = act(F.conv1d(x, W["WaveNet.start_conv1.weight"], W["WaveNet.start_conv1.bias"]))
    x = act(F.conv1d(x, W["WaveNet.start_conv2.weight"], W["WaveNet.start_conv2.bias"]))
    skip = 0
    i = 0
    for _b in range(cfg["residual_blocks"]):
        for l in range(cfg["residual_layers"]):
            x, s = _block(W, "WaveNet.residual_blocks.%d." % i, x, cond, 2 ** l)
            skip = skip + s
            i += 1
    res = F.conv1d(act(skip), W["WaveNet.end_conv_1.weight"], W["WaveNet.end_conv_1.bias"])
    res = F.conv1d(act(res), W["WaveNet.end_conv_2.weight"], W["WaveNet.end_conv_2.bias"])
    return res[:, :, -output_length:].transpose(1, 2)      # [b, T, nout]


def _downsample(W, feats):                                  # audio2headpose.py:16-21, 47
    h = F.linear(feats, W["audio_downsample.0.weight"], W["audio_downsample.0.bias"])
    h = F.batch_norm(h, W["audio_downsample.1.running_mean"], W["audio_downsample.1.running_var"],
                     W["audio_downsample.1.weight"], W["audio_downsample.1.bias"], False, 0.1, 1e-5)
    return F.linear(F.leaky_relu(h, 0.2), W["audio_downsample.3.weight"], W["audio_downsample.3.bias"])


def _forward(W, cfg, history, audio_feats):                 # audio2headpose.py:40-52
    bs, item_len, ndim = audio_feats.shape
    down = _downsample(W, audio_feats.reshape(-1, ndim)).reshape(bs, item_len, -1)
    return _wavenet(W, cfg, history.permute(0, 2, 1), down.transpose(1, 2))


def receptive_field(cfg) -> int:                            # networks.py:150-166
    return 1 + cfg["residual_blocks"] * (cfg["kernel_size"] - 1) * (2 ** cfg["residual_layers"] - 1)


def sample_gmm(params, ncenter, ndim, sigma_scale, randn_row, expq_row):   # losses.py:68-112, b*T == 1
    g = params.reshape(-1, (2 * ndim + 1) * ncenter)
    prob = F.softmax(g[:, :ncenter], dim=1)
    idx = int(torch.argmax(prob / expq_row.reshape(1, -1), dim=1))          # == torch.multinomial(prob, 1, True)
    mu = g[:, ncenter: ncenter + ncenter * ndim]
    sigma = torch.exp(-g[:, ncenter + ncenter * ndim:]) * sigma_scale
    sel_sigma = sigma[0, idx * ndim:(idx + 1) * ndim]
    sel_mu = mu[0, idx * ndim:(idx + 1) * ndim]
    return (randn_row * sel_sigma + sel_mu).reshape(1, 1, -1)


def generate_sequences(sd, cfg, audio_feats, pre_headpose, noise, expq, sigma_scale, frame_future):
    """audio2headpose_model.py:133-187 with fill_zero=True.  noise [nframe, ndim], expq [nframe, ncenter]."""
    W = _t(sd)
    R = receptive_field(cfg)
    nd, nc = cfg["ndim"], cfg["ncenter"]
    audio = np.asarray(audio_feats, np.float32).reshape(-1, 2 * cfg["hidden_size"])
    nframe = audio.shape[0] - frame_future
    out = np.zeros([nframe, nd])
    insert = np.repeat(audio[0], R - 1).reshape(-1, R - 1).T                   # :155-157
    audio = np.concatenate([insert, audio])
    hist = np.repeat(np.asarray(pre_headpose, np.float32), R).reshape(-1, R).T   # :159-161
    hist = torch.from_numpy(hist).unsqueeze(0).float()
    with torch.no_grad():
        for i in range(nframe):
            feats = torch.from_numpy(audio[i + frame_future: i + frame_future + R]).unsqueeze(0).float()   # :172
            preds = _forward(W, cfg, hist, feats)
            if cfg["loss"] == "GMM":
                pred = sample_gmm(preds, nc, nd, sigma_scale, torch.as_tensor(noise[i]).float(), torch.as_tensor(expq[i]).float())
            else:
                pred = preds
            out[i] = pred[0, 0].numpy()
            hist = torch.cat((hist[:, 1:, :], pred), dim=1)                     # :186
    return out


def stream(sd, cfg, audio_feats, pre_headpose, noise, expq, sigma_scale, frame_future, dtype=np.float64):
    """Incremental evaluation (one position per step, dilation queues) -- the algorithm of csrc/a2h.hip."""
    W = {k: np.asarray(v, dtype) for k, v in sd.items()}
    R = receptive_field(cfg)
    nd, nc, L = cfg["ndim"], cfg["ncenter"], cfg["residual_layers"] * cfg["residual_blocks"]
    audio = np.asarray(audio_feats, dtype).reshape(-1, 2 * cfg["hidden_size"])
    nframe = audio.shape[0] - frame_future
    lrelu = lambda v: np.where(v > 0, v, 0.2 * v)
    h = audio @ W["audio_downsample.0.weight"].T + W["audio_downsample.0.bias"]
    h = (h - W["audio_downsample.1.running_mean"]) / np.sqrt(W["audio_downsample.1.running_var"] + 1e-5) \
        * W["audio_downsample.1.weight"] + W["audio_downsample.1.bias"]
    cond = lrelu(h) @ W["audio_downsample.3.weight"].T + W["audio_downsample.3.bias"]
    dil = [2 ** (i % cfg["residual_layers"]) for i in range(L)]
    queues = [np.zeros((d, cfg["residual_channels"]), dtype) for d in dil]
    out = np.zeros((nframe, nd))
    x_in = np.asarray(pre_headpose, dtype)
    for s in range(R - 1 + nframe):
        c = cond[max(s + frame_future - (R - 1), 0)]
        x = lrelu(W["WaveNet.start_conv1.weight"][:, :, 0] @ x_in + W["WaveNet.start_conv1.bias"])
        x = lrelu(W["WaveNet.start_conv2.weight"][:, :, 0] @ x + W["WaveNet.start_conv2.bias"])
        skip = 0.0
        for l in range(L):
            p = "WaveNet.residual_blocks.%d." % l
            slot = s % dil[l]
            old = queues[l][slot].copy()
            queues[l][slot] = x
            f = W[p + "filter_conv.weight"][:, :, 0] @ old + W[p + "filter_conv.weight"][:, :, 1] @ x + W[p + "filter_conv.bias"] \
                + W[p + "cond_filter_conv.weight"][:, :, 0] @ c + W[p + "cond_filter_conv.bias"]
            g = W[p + "gate_conv.weight"][:, :, 0] @ old + W[p + "gate_conv.weight"][:, :, 1] @ x + W[p + "gate_conv.bias"] \
                + W[p + "cond_gate_conv.weight"][:, :, 0] @ c + W[p + "cond_gate_conv.bias"]
            z = np.tanh(f) / (1.0 + np.exp(-g))
            skip = skip + W[p + "skip_conv.weight"][:, :, 0] @ z + W[p + "skip_conv.bias"]
            x = W[p + "residual_conv.weight"][:, :, 0] @ z + W[p + "residual_conv.bias"] + x
        i = s - (R - 1)
        if i < 0:
            continue
        r = W["WaveNet.end_conv_1.weight"][:, :, 0] @ lrelu(skip) + W["WaveNet.end_conv_1.bias"]
        r = W["WaveNet.end_conv_2.weight"][:, :, 0] @ lrelu(r) + W["WaveNet.end_conv_2.bias"]
        if cfg["loss"] == "GMM":
            logits = r[:nc]
            prob = np.exp(logits - logits.max()); prob /= prob.sum()
            idx = int(np.argmax(prob / np.asarray(expq[i], dtype)))
            mu = r[nc + idx * nd: nc + (idx + 1) * nd]
            sigma = np.exp(-r[nc + nc * nd + idx * nd: nc + nc * nd + (idx + 1) * nd]) * sigma_scale
            smp = np.asarray(noise[i], dtype) * sigma + mu
        else:
            smp = r[:nd]
        out[i] = smp
        x_in = smp
    return out


def lstm_generate(sd, cfg, audio_feats, noise, expq, sigma_scale):
    """The LSTM decoder branch: Audio2Headpose_LSTM.forward (models/audio2headpose.py:88-99) over ALL audio rows, then one
    Sample_GMM over all rows (models/audio2headpose_model.py:189-202).  noise [rows, ndim], expq [rows, ncenter]."""
    import torch.nn as nn
    W = _t(sd)
    H, nd, nc = cfg["hidden_size"], cfg["ndim"], cfg["ncenter"]
    x = torch.from_numpy(np.asarray(audio_feats, np.float32).reshape(-1, 2 * H))
    bn = lambda h, p: F.batch_norm(h, W[p + ".running_mean"], W[p + ".running_var"], W[p + ".weight"], W[p + ".bias"], False, 0.1, 1e-5)
    with torch.no_grad():
        h = F.leaky_relu(bn(F.linear(x, W["audio_downsample.0.weight"], W["audio_downsample.0.bias"]), "audio_downsample.1"), 0.2)
        h = F.linear(h, W["audio_downsample.3.weight"], W["audio_downsample.3.bias"])
        lstm = nn.LSTM(input_size=H, hidden_size=256, num_layers=3, batch_first=True)
        lstm.load_state_dict({k[5:]: v for k, v in W.items() if k.startswith("LSTM.")})
        h, _ = lstm(h.unsqueeze(0))
        h = h.reshape(-1, 256)
        h = F.leaky_relu(bn(F.linear(h, W["fc.0.weight"], W["fc.0.bias"]), "fc.1"), 0.2)
        h = F.leaky_relu(bn(F.linear(h, W["fc.3.weight"], W["fc.3.bias"]), "fc.4"), 0.2)
        g = F.linear(h, W["fc.6.weight"], W["fc.6.bias"])                 # [rows, nout]
        if cfg["loss"] != "GMM":
            return g.numpy()
        prob = F.softmax(g[:, :nc], dim=1)
        idx = torch.argmax(prob / torch.as_tensor(expq).float(), dim=1)      # == torch.multinomial(prob, 1, True)
        mu = g[:, nc: nc + nc * nd].reshape(-1, nc, nd)
        sigma = (torch.exp(-g[:, nc + nc * nd:]) * sigma_scale).reshape(-1, nc, nd)
        r = torch.arange(g.shape[0])
        return (torch.as_tensor(noise).float() * sigma[r, idx] + mu[r, idx]).numpy()
