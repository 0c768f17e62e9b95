"""ORACLE (test infrastructure only) -- CPU restatement of the reference's `size == 'small'` generator,
Feature2FaceGenerator_Unet / UnetSkipConnectionBlock (models/networks.py:680-769), driven by its state-dict keys.

The reference builds every block as nn.Sequential with IN-PLACE activations (LeakyReLU(0.2, True) before each down-conv,
ReLU(True) before each up-conv, :737-741) and returns cat([x, model(x)], 1) (:767).  Because the LeakyReLU runs in place
on x before the cat reads it, and the parent's ReLU then runs in place on the concatenated tensor, the effective dataflow is

    d1 = conv(input)                              outermost: no activation in front, no norm
    dk = [BN](conv(lrelu(d(k-1))))                innermost: no norm
    u_n = BN(convT(relu(d_n)))                    innermost
    uk  = BN(convT(cat[relu(dk), relu(u(k+1))]))  relu(lrelu(d)) == relu(d)
    out = tanh(convT(cat[relu(d1), relu(u2)]) + bias)
which is what this file computes with the same torch ops (F.conv2d k4 s2 p1, F.conv_transpose2d k4 s2 p1, eval
batch_norm).  oracle/make_golden_unet.py checks it against the real reference module bit for bit."""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F


def _bn(x, sd, key):
    return F.batch_norm(x, sd[key + ".running_mean"], sd[key + ".running_var"], sd[key + ".weight"], sd[key + ".bias"],
                        training=False, eps=1e-5)


def block_keys(num_downs: int, prefix: str = "model"):
    """[(down conv key, down BN key or None, up conv key, up BN key or None, prefix of the submodule)] outermost first."""
    out = []
    pfx = prefix + ".model"
    for depth in range(num_downs):
        outer, inner = depth == 0, depth == num_downs - 1
        if outer:        # [downconv 0, submodule 1, uprelu 2, upconv 3, tanh 4]
            out.append((pfx + ".0", None, pfx + ".3", None))
            pfx = pfx + ".1.model"
        elif inner:      # [downrelu 0, downconv 1, uprelu 2, upconv 3, upnorm 4]
            out.append((pfx + ".1", None, pfx + ".3", pfx + ".4"))
        else:            # [downrelu 0, downconv 1, downnorm 2, submodule 3, uprelu 4, upconv 5, upnorm 6]
            out.append((pfx + ".1", pfx + ".2", pfx + ".5", pfx + ".6"))
            pfx = pfx + ".3.model"
    return out


def generator_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, num_downs: int = 8, prefix: str = "model") -> torch.Tensor:
    keys = block_keys(num_downs, prefix)
    with torch.no_grad():
        d = []
        h = x
        for k, (dc, dbn, _uc, _ubn) in enumerate(keys):
            if k > 0:
                h = F.leaky_relu(h, 0.2)
            h = F.conv2d(h, sd[dc + ".weight"], None, 2, 1)
            if dbn:
                h = _bn(h, sd, dbn)
            d.append(h)
        u = None
        for k in range(num_downs - 1, -1, -1):
            _dc, _dbn, uc, ubn = keys[k]
            src = F.relu(d[k]) if u is None else torch.cat([F.relu(d[k]), F.relu(u)], 1)
            u = F.conv_transpose2d(src, sd[uc + ".weight"], sd.get(uc + ".bias"), 2, 1)
            if ubn:
                u = _bn(u, sd, ubn)
        return torch.tanh(u)
