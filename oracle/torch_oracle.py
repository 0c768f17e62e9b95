"""ORACLE (test infrastructure, not product) -- CPU restatement of the reference's
feature2face generator forward, driven purely by the checkpoint's state-dict keys.

Only tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg may
import this file.  The product path (livespeechportraits_amd/) never does and has
no CPU fallback.

What it restates (reference file:line, /root/reference):
  models/feature2face_model.py:225-237  inference(): cat([feature_map, cand_image], 1) -> G
  models/feature2face_G.py:27-34        Feature2Face_G.forward -> self.netG(input)
  models/networks.py:575-579 / 479-483  generator forward: model(x) then torch.tanh
  models/networks.py:642-646 / 546-550  skip block: outermost -> model(x), else cat([x, model(x)], 1)
  models/networks.py:592-640            per-level Sequential order (down conv s2, [BN], ReLU,
                                        res blocks, submodule, Upsample x2 nearest, up conv,
                                        [BN, ReLU, res blocks])
  models/networks.py:670-675            ResidualBlock: conv-BN-ReLU-conv-BN, += x, ReLU

Where the arithmetic lives: the reference has no kernels of its own; Conv2d /
BatchNorm2d(eval) / Upsample(nearest) / ReLU / cat / tanh are PyTorch (pinned
torch==1.7.1 in the reference's cog.yaml:9; this image has torch 2.10.0+rocm7.0,
CPU path = ATen/oneDNN).  This file calls the same torch.nn.functional ops in the
same order, so on CPU it is bit-identical to the reference modules run under the
same torch build -- oracle/make_golden.py asserts exactly that in the container
where /root/reference exists, and freezes reference outputs into tests/golden/.
The independent plain-C restatement of the arithmetic is oracle/f2f_oracle.c.

Pinning status: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4), so parity is pinned on outputs of the reference itself,
generated in-container by oracle/make_golden.py (committed script + fixtures).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def _bn(x: torch.Tensor, sd: Dict[str, torch.Tensor], key: str) -> torch.Tensor:
    if (key + ".running_mean") not in sd:
        # norm_layer=nn.InstanceNorm2d (networks.py:459, :555): affine=False, track_running_stats=False, eps=1e-5 -> no tensors
        return F.instance_norm(x, None, None, None, None, True, 0.1, BN_EPS)
    return F.batch_norm(x, sd[key + ".running_mean"], sd[key + ".running_var"],
                        sd[key + ".weight"], sd[key + ".bias"], training=False, eps=BN_EPS)


def _res(x: torch.Tensor, sd, key: str) -> torch.Tensor:
    h = F.conv2d(x, sd[key + ".block.0.weight"], None, 1, 1)
    h = F.relu(_bn(h, sd, key + ".block.1"))
    h = F.conv2d(h, sd[key + ".block.3.weight"], None, 1, 1)
    h = _bn(h, sd, key + ".block.4")
    return F.relu(h + x)


def generator_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, nres: int, num_downs: int = 8,
                      prefix: str = "netG.model", taps: Optional[Dict[str, torch.Tensor]] = None,
                      pre_tanh: bool = False) -> torch.Tensor:
    """Forward of Feature2FaceGenerator_{normal (nres=1), large (nres=2)}.

    ``taps`` (optional dict) receives the per-level block outputs ("L<d>.out", the
    cat([x, model(x)]) tensors) and the pre-tanh frame ("pre_tanh").
    """

    def level(x: torch.Tensor, pfx: str, depth: int) -> torch.Tensor:
        outer = depth == 0
        inner = depth == num_downs - 1
        i = 0
        h = F.conv2d(x, sd["%s.model.%d.weight" % (pfx, i)], sd.get("%s.model.%d.bias" % (pfx, i)), 2, 1)   # bias: InstanceNorm variant
        i += 1
        if not (outer or inner):
            h = _bn(h, sd, "%s.model.%d" % (pfx, i))
            i += 1
        h = F.relu(h)
        i += 1
        for _ in range(nres):
            h = _res(h, sd, "%s.model.%d" % (pfx, i))
            i += 1
        if not inner:
            h = level(h, "%s.model.%d" % (pfx, i), depth + 1)
            i += 1
        h = F.interpolate(h, scale_factor=2, mode="nearest")
        i += 1
        h = F.conv2d(h, sd["%s.model.%d.weight" % (pfx, i)], sd.get("%s.model.%d.bias" % (pfx, i)), 1, 1)
        i += 1
        if outer:
            return h
        h = F.relu(_bn(h, sd, "%s.model.%d" % (pfx, i)))
        i += 2
        for _ in range(nres):
            h = _res(h, sd, "%s.model.%d" % (pfx, i))
            i += 1
        out = torch.cat([x, h], 1)
        if taps is not None:
            taps["L%d.out" % depth] = out
        return out

    with torch.no_grad():
        y = level(x, prefix, 0)
        if taps is not None:
            taps["pre_tanh"] = y
        return y if pre_tanh else torch.tanh(y)


def inference(sd: Dict[str, torch.Tensor], feature_map: torch.Tensor,
              cand_image: Optional[torch.Tensor], nres: int, num_downs: int = 8,
              **kw) -> torch.Tensor:
    """feature2face_model.py:225-237 (fp32 branch): cat on dim 1 unless cand_image is None."""
    x = feature_map if cand_image is None else torch.cat([feature_map, cand_image], dim=1)
    return generator_forward(sd, x, nres, num_downs, **kw)


def to_torch(sd_np) -> Dict[str, torch.Tensor]:
    import numpy as np
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items()}


def time_cpu(sd, x, nres, num_downs=8, repeats: int = 4, threads: Optional[int] = None,
             budget_s: float = 25.0):
    """cpu_baseline helper: (min s/frame, median s/frame, threads used) of this restatement on
    the host cores.  The GPU box advertises more logical CPUs than the
    pod may use (256 threads ran 20x slower than 32), so unless ``threads`` is given the thread
    count is calibrated: 1 warm-up + 1 timed frame at 8, 16, 32, 64 ... threads (bounded by the
    affinity mask), keeping the fastest, then ``repeats`` timed frames within ``budget_s``."""
    import os
    import time
    import statistics

    def one():
        t0 = time.perf_counter()
        generator_forward(sd, x, nres, num_downs)
        return time.perf_counter() - t0

    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        avail = os.cpu_count() or 1
    t_start = time.perf_counter()
    if threads is None:
        best_t, best_n = None, None
        n = min(8, avail)
        while True:
            torch.set_num_threads(n)
            one()
            t = one()
            if best_t is None or t < best_t:
                best_t, best_n = t, n
            elif t > 1.3 * best_t:
                break
            if n >= min(avail, 128) or time.perf_counter() - t_start > budget_s / 2:
                break
            n = min(2 * n, avail)
        threads = best_n
    torch.set_num_threads(threads)
    one()
    ts: List[float] = []
    for _ in range(repeats):
        ts.append(one())
        if time.perf_counter() - t_start > budget_s and len(ts) >= 2:
            break
    return min(ts), statistics.median(ts), threads
