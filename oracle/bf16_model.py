"""ORACLE (test infrastructure only) -- an exact numerical MODEL of the bf16 storage path (LSPF2F_DTYPE_BF16).

The reference has no bf16 path (only fp16 autocast, feature2face_G.py:28-30), so nothing of the reference can pin it.
This file states precisely what the kernels are supposed to compute, so that the GPU path can be held to a definite
specification instead of a loose distance from the fp32 reference:

  * every activation tensor the kernels keep in the workspace is bf16: each layer's epilogue result
    (acc * scale + shift, + residual, ReLU) is rounded to nearest-even bf16 once, on store;
  * conv weights of the implicit-GEMM layers are bf16 (rounded once by the host packer); up-convs writing >= 32x32 and
    the last conv use the sub-pixel form -- the 3x3 taps that alias onto the same low-res pixel are summed in double,
    rounded to fp32, THEN to bf16 (csrc/plan.cpp pack());
  * the first conv reads the fp32 API tensors with fp32 weights; the last conv returns fp32 (pre-tanh never rounded);
  * accumulation, BatchNorm scale/shift (folded in double, stored fp32), residual add, ReLU and tanh are fp32.
Accumulation ORDER is not part of the model: GPU and model differ by fp32 rounding before the bf16 rounding, which now
and then flips a bf16 result by one unit in the last place.  Tests therefore look at the distribution of differences
(almost all near zero, a thin tail), not only at the maximum."""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

UP4_MIN_EXTENT = 32          # csrc/plan.h kUp4MinExtent


def rb(t: torch.Tensor) -> torch.Tensor:
    """round to nearest-even bf16, back to fp32"""
    return t.to(torch.bfloat16).to(torch.float32)


def _affine(sd, key):
    g, b = sd[key + ".weight"].double(), sd[key + ".bias"].double()
    m, v = sd[key + ".running_mean"].double(), sd[key + ".running_var"].double()
    s = g / torch.sqrt(v + 1e-5)
    return s.float().view(1, -1, 1, 1), (b - m * s).float().view(1, -1, 1, 1)


def _fold(w: torch.Tensor) -> torch.Tensor:
    """[co][ci][3][3] -> [4 parities][co][ci][2][2]: pre-summed taps of Upsample(x2, nearest) + Conv3x3 (plan.cpp pack())."""
    lo = [[0, 1], [0, 2]]
    hi = [[0, 2], [1, 2]]
    wd = w.double()
    out = torch.zeros((4,) + tuple(w.shape[:2]) + (2, 2), dtype=torch.float64)
    for py in range(2):
        for px in range(2):
            for a in range(2):
                for b in range(2):
                    out[py * 2 + px, :, :, a, b] = wd[:, :, lo[py][a]:hi[py][a] + 1, lo[px][b]:hi[px][b] + 1].sum((2, 3))
    return out.float()


def _up_subpixel(x: torch.Tensor, wf: torch.Tensor) -> torch.Tensor:
    """out[2y+py, 2x+px] = sum_{a,b} wf[par][.., a, b] * x[y+py-1+a, x+px-1+b]  (zero outside)"""
    B, _, H, W = x.shape
    xp = F.pad(x, (1, 1, 1, 1))
    out = torch.empty((B, wf.shape[1], 2 * H, 2 * W), dtype=torch.float32)
    for py in range(2):
        for px in range(2):
            out[:, :, py::2, px::2] = F.conv2d(xp[:, :, py:py + H + 1, px:px + W + 1], wf[py * 2 + px])
    return out


def generator_forward_bf16(sd: Dict[str, torch.Tensor], x: torch.Tensor, nres: int, num_downs: int = 8,
                           prefix: str = "netG.model", pre_tanh: bool = False) -> torch.Tensor:
    def conv(h, key, stride=1):                       # implicit-GEMM layer: bf16 weights, bf16 inputs, fp32 accumulate
        return F.conv2d(h, rb(sd[key]), None, stride, 1)

    def res(h, key):
        s1, t1 = _affine(sd, key + ".block.1")
        s2, t2 = _affine(sd, key + ".block.4")
        a = rb(F.relu(conv(h, key + ".block.0.weight") * s1 + t1))
        return rb(F.relu(conv(a, key + ".block.3.weight") * s2 + t2 + h))

    def level(xin, pfx, depth):
        outer, inner = depth == 0, depth == num_downs - 1
        i = 0
        wkey = "%s.model.%d.weight" % (pfx, i)
        if outer:
            h = F.conv2d(xin, sd[wkey], None, 2, 1)               # first conv: fp32 inputs and weights
        else:
            h = conv(xin, wkey, 2)
        i += 1
        if not (outer or inner):
            s, t = _affine(sd, "%s.model.%d" % (pfx, i))
            h = h * s + t
            i += 1
        h = rb(F.relu(h))
        i += 1
        for _ in range(nres):
            h = res(h, "%s.model.%d" % (pfx, i))
            i += 1
        if not inner:
            h = level(h, "%s.model.%d" % (pfx, i), depth + 1)
            i += 1
        i += 1                                                     # Upsample
        w = sd["%s.model.%d.weight" % (pfx, i)]
        ho = 2 * h.shape[-1]
        if outer or ho >= UP4_MIN_EXTENT:
            y = _up_subpixel(h, rb(_fold(w)))                      # sub-pixel form, folded taps rounded to bf16
        else:
            y = F.conv2d(F.interpolate(h, scale_factor=2, mode="nearest"), rb(w), None, 1, 1)
        i += 1
        if outer:
            return y                                               # fp32, never rounded
        s, t = _affine(sd, "%s.model.%d" % (pfx, i))
        y = rb(F.relu(y * s + t))
        i += 2
        for _ in range(nres):
            y = res(y, "%s.model.%d" % (pfx, i))
            i += 1
        return torch.cat([xin, y], 1)

    with torch.no_grad():
        y = level(x.float(), prefix, 0)
        return y if pre_tanh else torch.tanh(y)
