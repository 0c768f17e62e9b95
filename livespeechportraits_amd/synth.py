"""Deterministic synthetic weights and inputs for the feature2face generator.

No checkpoint of the reference is obtainable offline (README.md:52 points at
Google Drive), so every parity test, the smoke test and bench.py run on
synthetic state dicts of the reference's exact architecture.  The generator is
a counter-based hash (splitmix64 finaliser over the element index), so the same
(seed, key) always yields the same tensor on any machine and any numpy/torch
version -- the golden fixtures under tests/golden/ depend on that.

Statistics follow the recipe SURVEY.md 8c validated not to saturate tanh:
  conv weight   zero mean, std 0.02      (reference init: networks.py:360, N(0, 0.02))
  BN weight     mean 1,   std 0.02       (networks.py:374)
  BN bias       std 0.05                 (perturbed: the reference inits it to 0)
  running_mean  std 0.05
  running_var   U(0.9, 1.6)
Uniform distributions of matching variance stand in for the normals; the
distribution shape is irrelevant for parity.
"""
from __future__ import annotations

import zlib
from typing import Dict, Tuple

import numpy as np

from .topology import Topology, build_topology

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    x ^= x >> np.uint64(30)
    x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27)
    x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    return x


def uniform01(n: int, stream: int) -> np.ndarray:
    """n float32 values in [0,1), a pure function of (stream, index)."""
    if n == 0:
        return np.zeros(0, np.float32)
    half = (n + 1) // 2
    with np.errstate(over="ignore"):
        ctr = np.arange(half, dtype=np.uint64) + (np.uint64(stream & 0xFFFFFFFF) << np.uint64(32))
        h = _splitmix(ctr)
    out = np.empty(half * 2, np.float32)
    out[0::2] = (h >> np.uint64(40)).astype(np.float32)            # top 24 bits
    out[1::2] = ((h >> np.uint64(16)) & np.uint64(0xFFFFFF)).astype(np.float32)
    out *= np.float32(1.0 / (1 << 24))
    return out[:n]


def _stream(seed: int, key: str) -> int:
    return (zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF


def symmetric(n: int, std: float, stream: int) -> np.ndarray:
    """zero-mean uniform with the given standard deviation."""
    a = np.float32(std * np.sqrt(3.0))
    return (uniform01(n, stream) * np.float32(2.0) - np.float32(1.0)) * a


def make_state_dict(topo: Topology, seed: int = 1234) -> Dict[str, np.ndarray]:
    """numpy state dict with the reference's key names (no 'module.' prefix)."""
    sd: Dict[str, np.ndarray] = {}
    for key, shape in topo.tensors.items():
        n = int(np.prod(shape)) if shape else 1
        s = _stream(seed, key)
        if key.endswith("num_batches_tracked"):
            sd[key] = np.zeros((), np.int64)
        elif key.endswith("running_var"):
            sd[key] = (uniform01(n, s) * np.float32(0.7) + np.float32(0.9)).reshape(shape)
        elif key.endswith("running_mean") or key.endswith(".bias"):
            sd[key] = symmetric(n, 0.05, s).reshape(shape)
        elif len(shape) == 1:  # BN weight
            sd[key] = (np.float32(1.0) + symmetric(n, 0.02, s)).reshape(shape)
        else:                  # conv weight, OIHW
            sd[key] = symmetric(n, 0.02, s).reshape(shape)
    return sd


def scale_last_conv(sd: Dict[str, np.ndarray], topo: Topology, gain: float) -> Dict[str, np.ndarray]:
    """Scale the weight of the outermost up-conv (the layer in front of tanh).  With run-time normalisation (InstanceNorm
    variant) the activations in front of it have unit variance, and the N(0, 0.02)-like weights of the recipe would push a
    percent of the output into tanh saturation, where errors hide."""
    key = topo.convs[-1].weight_key
    out = dict(sd)
    out[key] = (sd[key] * np.float32(gain)).astype(np.float32)
    return out


def make_inputs(batch: int, size: int = 512, seed: int = 99, cand_batch: int = 1,
                cand_channels: int = 12) -> Tuple[np.ndarray, np.ndarray]:
    """(feature_map [B,1,S,S] with exact {0,1} values -- face_dataset.py:280 divides a
    uint8 0/255 raster by 255 --, cand_image [Bc,12,S,S] in [-1,1) -- demo.py:89-95)."""
    feat = np.empty((batch, 1, size, size), np.float32)
    for b in range(batch):
        u = uniform01(size * size, _stream(seed + b, "feature_map"))
        feat[b, 0] = (u > np.float32(0.97)).astype(np.float32).reshape(size, size)
    cand = np.empty((cand_batch, cand_channels, size, size), np.float32)
    for b in range(cand_batch):
        u = uniform01(cand_channels * size * size, _stream(seed + 1000 + b, "cand_image"))
        cand[b] = (u * np.float32(2.0) - np.float32(1.0)).reshape(cand_channels, size, size)
    return feat, cand


def synthetic(variant: str = "large", ngf: int = 64, num_downs: int = 8, size: int = 512,
              seed: int = 1234):
    topo = build_topology(variant, ngf=ngf, num_downs=num_downs, size=size)
    return topo, make_state_dict(topo, seed)


# ---- Audio2Headpose ------------------------------------------------------
A2H_DEFAULTS = dict(residual_layers=7, residual_blocks=2, residual_channels=128, dilation_channels=128,
                    skip_channels=256, kernel_size=2, input_channels=12, cond_channels=512, hidden_size=512,
                    ncenter=1, ndim=12, loss="GMM")


def a2h_shapes(cfg: Dict) -> Dict[str, Tuple[int, ...]]:
    """state-dict key -> shape of the reference's Audio2Headpose (models/audio2headpose.py:8-37;
    key list dumped from the instantiated reference module, see oracle/make_golden_a2h.py)."""
    H, nd, nc = cfg["hidden_size"], cfg["ndim"], cfg["ncenter"]
    res, dil, skip, k, cond = cfg["residual_channels"], cfg["dilation_channels"], cfg["skip_channels"], cfg["kernel_size"], cfg["cond_channels"]
    out = (2 * nd + 1) * nc if cfg["loss"] == "GMM" else nd
    s = {"audio_downsample.0.weight": (H, 2 * H), "audio_downsample.0.bias": (H,),
         "audio_downsample.1.weight": (H,), "audio_downsample.1.bias": (H,),
         "audio_downsample.1.running_mean": (H,), "audio_downsample.1.running_var": (H,),
         "audio_downsample.3.weight": (H, H), "audio_downsample.3.bias": (H,),
         "WaveNet.start_conv1.weight": (res, cfg["input_channels"], 1), "WaveNet.start_conv1.bias": (res,),
         "WaveNet.start_conv2.weight": (res, res, 1), "WaveNet.start_conv2.bias": (res,)}
    for i in range(cfg["residual_layers"] * cfg["residual_blocks"]):
        p = "WaveNet.residual_blocks.%d." % i
        s.update({p + "filter_conv.weight": (dil, res, k), p + "filter_conv.bias": (dil,),
                  p + "gate_conv.weight": (dil, res, k), p + "gate_conv.bias": (dil,),
                  p + "residual_conv.weight": (res, dil, 1), p + "residual_conv.bias": (res,),
                  p + "skip_conv.weight": (skip, dil, 1), p + "skip_conv.bias": (skip,),
                  p + "cond_filter_conv.weight": (dil, cond, 1), p + "cond_filter_conv.bias": (dil,),
                  p + "cond_gate_conv.weight": (dil, cond, 1), p + "cond_gate_conv.bias": (dil,)})
    s.update({"WaveNet.end_conv_1.weight": (out, skip, 1), "WaveNet.end_conv_1.bias": (out,),
              "WaveNet.end_conv_2.weight": (out, out, 1), "WaveNet.end_conv_2.bias": (out,)})
    return s


def make_a2h_state_dict(cfg: Dict, seed: int = 4321) -> Dict[str, np.ndarray]:
    """Weights of std 1/sqrt(fan_in) (the reference's N(0, 0.02) init gives a near-constant network whose
    feedback path would not be exercised), biases std 0.05, BN statistics perturbed as for the renderer."""
    sd = {}
    for key, shape in a2h_shapes(cfg).items():
        n = int(np.prod(shape))
        st = _stream(seed, key)
        if key.endswith("running_var"):
            v = uniform01(n, st) * np.float32(0.7) + np.float32(0.9)
        elif key.endswith("running_mean"):
            v = symmetric(n, 0.05, st)
        elif len(shape) >= 2:
            v = symmetric(n, 1.0 / float(np.sqrt(np.prod(shape[1:]))), st)
        elif key == "audio_downsample.1.weight":
            v = np.float32(1.0) + symmetric(n, 0.02, st)
        else:
            v = symmetric(n, 0.05, st)
        sd[key] = v.reshape(shape).astype(np.float32)
    return sd


def make_a2h_inputs(n_audio: int, cfg: Dict, seed: int = 17) -> Tuple[np.ndarray, np.ndarray]:
    """(audio_feats [n_audio, 2*hidden], pre_headpose [ndim]) -- APC-feature-like values of std 0.5."""
    audio = symmetric(n_audio * 2 * cfg["hidden_size"], 0.5, _stream(seed, "a2h.audio")).reshape(n_audio, -1)
    pre = symmetric(cfg["ndim"], 0.1, _stream(seed, "a2h.pre"))
    return audio.astype(np.float32), pre.astype(np.float32)


# ---- APC feature database for the manifold projection ---------------------
def make_feature_database(m: int, n: int, d: int = 512, intrinsic: int = 0, seed: int = 3, noise: float = 0.05
                          ) -> Tuple[np.ndarray, np.ndarray]:
    """(database [m, d], query features [n, d]).  intrinsic == 0: isotropic points (well-conditioned neighbourhoods);
    intrinsic > 0: points near an ``intrinsic``-dimensional linear manifold plus ``noise`` isotropic noise, the shape real
    APC features have (neighbour differences become nearly dependent, the normal equations ill-conditioned).
    Queries are noisy copies of database rows, so every query has genuine near neighbours."""
    def gauss(count, key):      # sum of 4 uniforms: bell-shaped, variance 1
        u = symmetric(count * 4, 1.0, _stream(seed, key)).reshape(count, 4)
        return (u.sum(1) * np.float32(0.5)).astype(np.float32)
    if intrinsic:
        basis = gauss(intrinsic * d, "lle.basis").reshape(intrinsic, d) / np.float32(np.sqrt(intrinsic))
        db = gauss(m * intrinsic, "lle.coef").reshape(m, intrinsic) @ basis + np.float32(noise) * gauss(m * d, "lle.noise").reshape(m, d)
    else:
        db = gauss(m * d, "lle.db").reshape(m, d)
    pick = (uniform01(n, _stream(seed, "lle.pick")) * m).astype(np.int64) % m
    scale = np.float32(noise if intrinsic else 0.4)
    q = db[pick] * np.float32(0.8) + scale * gauss(n * d, "lle.q").reshape(n, d)
    return np.ascontiguousarray(db, np.float32), np.ascontiguousarray(q, np.float32)


# ---- recurrent stacks of the audio front-end ----------------------------------
def make_rnn_state_dict(cell: str, num_layers: int, input_size: int, hidden_size: int, seed: int = 11, prefix: str = "") -> Dict[str, np.ndarray]:
    """torch.nn.GRU / nn.LSTM parameter names; PyTorch's own init range U(-1/sqrt(H), 1/sqrt(H)) (scaled 1.6x so the
    gates leave their linear region and the state carries memory over many steps)."""
    gates = 3 if cell == "GRU" else 4
    k = 1.6 / float(np.sqrt(hidden_size))
    sd = {}
    for l in range(num_layers):
        n_in = input_size if l == 0 else hidden_size
        for name, shape in (("weight_ih", (gates * hidden_size, n_in)), ("weight_hh", (gates * hidden_size, hidden_size)),
                            ("bias_ih", (gates * hidden_size,)), ("bias_hh", (gates * hidden_size,))):
            key = "%s%s_l%d" % (prefix, name, l)
            n = int(np.prod(shape))
            sd[key] = ((uniform01(n, _stream(seed, key)) * np.float32(2.0) - np.float32(1.0)) * np.float32(k)).reshape(shape).astype(np.float32)
    return sd


def make_apc_state_dict(mel_dim: int = 80, hidden: int = 512, layers: int = 3, seed: int = 11) -> Dict[str, np.ndarray]:
    """Keys of the reference's APC_encoder: one single-layer nn.GRU per entry of ``rnns`` (models/networks.py:33-34)."""
    sd = {}
    for i in range(layers):
        one = make_rnn_state_dict("GRU", 1, mel_dim if i == 0 else hidden, hidden, seed=seed + i)
        sd.update({"rnns.%d.%s" % (i, k): v for k, v in one.items()})
    return sd


def a2f_shapes(hidden: int = 512, out: int = 75) -> Dict[str, Tuple[int, ...]]:
    s = {"downsample.0.weight": (hidden, 2 * hidden), "downsample.0.bias": (hidden,),
         "downsample.3.weight": (hidden, hidden), "downsample.3.bias": (hidden,),
         "fc.0.weight": (512, 256), "fc.0.bias": (512,), "fc.3.weight": (512, 512), "fc.3.bias": (512,),
         "fc.6.weight": (out, 512), "fc.6.bias": (out,)}
    for bn, n in (("downsample.1", hidden), ("fc.1", 512), ("fc.4", 512)):
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            s["%s.%s" % (bn, leaf)] = (n,)
    return s


def make_a2f_state_dict(hidden: int = 512, out: int = 75, seed: int = 23) -> Dict[str, np.ndarray]:
    """Audio2Feature (LSTM decoder) keys: downsample.*, LSTM.weight_ih_l<k> ..., fc.*  (models/audio2feature.py:35-54)."""
    sd = {}
    for key, shape in a2f_shapes(hidden, out).items():
        n = int(np.prod(shape))
        st = _stream(seed, key)
        if key.endswith("running_var"):
            v = uniform01(n, st) * np.float32(0.7) + np.float32(0.9)
        elif key.endswith("running_mean"):
            v = symmetric(n, 0.05, st)
        elif len(shape) == 2:
            v = symmetric(n, 1.0 / float(np.sqrt(shape[1])), st)
        elif key.endswith(".1.weight") or key.endswith(".4.weight"):
            v = np.float32(1.0) + symmetric(n, 0.02, st)
        else:
            v = symmetric(n, 0.05, st)
        sd[key] = v.reshape(shape).astype(np.float32)
    sd.update(make_rnn_state_dict("LSTM", 3, hidden, 256, seed=seed, prefix="LSTM."))
    return sd


def make_mel(T: int, mel_dim: int = 80, seed: int = 31) -> np.ndarray:
    """log-mel-like input [T, mel_dim]: smooth in time (a random walk), unit scale."""
    steps = symmetric(T * mel_dim, 0.35, _stream(seed, "mel")).reshape(T, mel_dim)
    x = np.cumsum(steps, 0)
    return (x - x.mean(0)).astype(np.float32) * np.float32(0.5)


# ---- the 'small' U-Net generator ----------------------------------------------
def unet_small_shapes(input_nc: int = 23, output_nc: int = 3, num_downs: int = 8, ngf: int = 64, prefix: str = "model"):
    """state-dict key -> shape of Feature2FaceGenerator_Unet (models/networks.py:680-692); same walk as the oracle's block_keys."""
    chans = [ngf * min(2 ** i, 8) for i in range(num_downs)]              # inner channels: ngf, 2ngf, 4ngf, 8ngf, 8ngf, ...
    shapes = {}
    pfx = prefix + ".model"
    for depth in range(num_downs):
        outer, inner = depth == 0, depth == num_downs - 1
        cin = input_nc if outer else chans[depth - 1]
        cout_up = output_nc if outer else chans[depth - 1]
        c = chans[depth]
        bn = lambda key, n: shapes.update({key + ".weight": (n,), key + ".bias": (n,), key + ".running_mean": (n,), key + ".running_var": (n,)})
        if outer:
            shapes[pfx + ".0.weight"] = (c, cin, 4, 4)
            shapes[pfx + ".3.weight"] = (2 * c, cout_up, 4, 4)            # ConvTranspose2d weight: [in][out][kh][kw]
            shapes[pfx + ".3.bias"] = (cout_up,)
            pfx += ".1.model"
        elif inner:
            shapes[pfx + ".1.weight"] = (c, cin, 4, 4)
            shapes[pfx + ".3.weight"] = (c, cout_up, 4, 4)
            bn(pfx + ".4", cout_up)
        else:
            shapes[pfx + ".1.weight"] = (c, cin, 4, 4)
            bn(pfx + ".2", c)
            shapes[pfx + ".5.weight"] = (2 * c, cout_up, 4, 4)
            bn(pfx + ".6", cout_up)
            pfx += ".3.model"
    return shapes


def make_unet_small_state_dict(input_nc: int = 23, output_nc: int = 3, num_downs: int = 8, ngf: int = 64, seed: int = 97,
                               prefix: str = "model") -> Dict[str, np.ndarray]:
    """Conv weights with std sqrt(2 / fan_in) (the activations keep unit scale through 16 layers, tanh not saturated),
    BN statistics perturbed as for the residual generators."""
    sd = {}
    for key, shape in unet_small_shapes(input_nc, output_nc, num_downs, ngf, prefix).items():
        n = int(np.prod(shape))
        st = _stream(seed, key)
        if key.endswith("running_var"):
            v = uniform01(n, st) * np.float32(0.7) + np.float32(0.9)
        elif key.endswith("running_mean") or key.endswith(".bias"):
            v = symmetric(n, 0.05, st)
        elif len(shape) == 4:
            is_t = key.endswith(".3.weight") or key.endswith(".5.weight")          # transposed conv: fan_in = in * 4 taps per output
            fan_in = shape[0] * 4 if is_t else shape[1] * 16
            gain = 0.5 if key == prefix + ".model.3.weight" else 1.0                # keep the frame away from tanh saturation
            v = symmetric(n, gain * float(np.sqrt(2.0 / fan_in)), st)
        else:
            v = np.float32(1.0) + symmetric(n, 0.02, st)
        sd[key] = v.reshape(shape).astype(np.float32)
    return sd


def make_a2h_lstm_state_dict(hidden: int = 512, ncenter: int = 1, ndim: int = 12, loss: str = "GMM", seed: int = 41) -> Dict[str, np.ndarray]:
    """Audio2Headpose_LSTM keys (models/audio2headpose.py:56-86): the Audio2Feature layout with ``audio_downsample`` in place
    of ``downsample`` and a GMM-parameter output."""
    out = (2 * ndim + 1) * ncenter if loss == "GMM" else ndim
    sd = {}
    for k, v in make_a2f_state_dict(hidden, out, seed=seed).items():
        sd[("audio_" + k) if k.startswith("downsample.") else k] = v
    return sd
