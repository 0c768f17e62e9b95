"""Static description of the feature2face generator (the residual U-Net).

This is the host-side statement of WHAT the reference network is -- the
state-dict key map, the conv list with shapes, and the algorithmic work per
frame -- written as data instead of as an nn.Module tree.  The C++ plan
builder (csrc/plan.cpp) derives the same list independently; a CPU test
cross-checks the two and the key list dumped from the reference itself
(tests/golden/keys_*.json).

Reference (read-only, /root/reference):
  models/networks.py:458-483   Feature2FaceGenerator_normal  (1 ResidualBlock per side)
  models/networks.py:554-579   Feature2FaceGenerator_large   (2 ResidualBlocks per side)
  models/networks.py:489-550 / 585-646   ResUnetSkipConnectionBlock{_small,}
  models/networks.py:650-675   ResidualBlock
  models/feature2face_G.py:16-21   variant selection ('normal' / 'large'), attribute name ``netG``

Level ``d`` (0 = outermost) of the nest owns an ``nn.Sequential`` called
``model`` whose integer indices are what the checkpoint keys carry:

  idx  outermost            middle                       innermost
  0    down conv s2         down conv s2                 down conv s2
  +1   (relu)               BN                           (relu)
  +1   res x n              (relu)                       res x n
  ..   SUB                  res x n, SUB                 (upsample)
  ..   (upsample)           (upsample)                   up conv
  ..   up conv              up conv, BN, (relu), res x n BN, (relu), res x n

Only parametrised entries appear in the state dict; the parameter-free ones
(ReLU, Upsample) still consume an index.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

VARIANTS = {"normal": 1, "large": 2}   # -> residual blocks per side
BN_EPS = 1e-5                           # nn.BatchNorm2d default, networks.py uses the default


@dataclass
class ConvSpec:
    """One 3x3 convolution of the generator, in execution order."""
    name: str                 # e.g. "L1.down", "L0.res0.a", "L3.up"
    weight_key: str           # state-dict key of the OIHW weight
    bn_key: Optional[str]     # state-dict prefix of the BatchNorm2d that follows (or None)
    cin: int
    cout: int
    h_in: int                 # spatial size of the tensor the conv actually reads (pre-upsample)
    h_out: int
    stride: int
    upsample: bool            # nearest x2 in front of the conv (reference: nn.Upsample)
    concat: bool              # input is cat([skip, below]) of two cin/2 tensors
    residual: bool            # second conv of a ResidualBlock: += block input before ReLU
    relu: bool
    tanh: bool = False

    @property
    def flops(self) -> int:   # per frame
        return 2 * self.cout * self.cin * 9 * self.h_out * self.h_out


@dataclass
class Topology:
    variant: str
    input_nc: int
    output_nc: int
    ngf: int
    num_downs: int
    size: int
    convs: List[ConvSpec] = field(default_factory=list)
    tensors: Dict[str, Tuple[int, ...]] = field(default_factory=dict)   # state-dict key -> shape
    norm: str = "batch"       # "batch": nn.BatchNorm2d (the default of the constructors); "instance": norm_layer=nn.InstanceNorm2d

    @property
    def nres(self) -> int:
        return VARIANTS[self.variant]

    def flops_per_frame(self) -> int:
        return sum(c.flops for c in self.convs)

    def weight_elems(self) -> int:
        return sum(c.cin * c.cout * 9 for c in self.convs)

    def activation_bytes_per_frame(self, elt: int = 4) -> int:
        """Algorithmic HBM bytes of activations per frame: every conv
        reads its unique input once (the pre-upsample tensor; both halves of a concat),
        writes its output once; the residual conv also reads the block input."""
        total = 0
        for c in self.convs:
            total += c.cin * c.h_in * c.h_in
            total += c.cout * c.h_out * c.h_out
            if c.residual:
                total += c.cout * c.h_out * c.h_out
        return total * elt


def level_channels(depth: int, ngf: int, input_nc: int, output_nc: int) -> Tuple[int, int, int]:
    """(input channels of the down conv, inner channels, output channels of the up conv)
    for nesting depth ``depth`` -- networks.py:557-570."""
    mult_out = min(2 ** max(depth - 1, 0), 8)
    mult_in = min(2 ** depth, 8)
    if depth == 0:
        return input_nc, ngf, output_nc
    return ngf * mult_out, ngf * mult_in, ngf * mult_out


def build_topology(variant: str = "large", input_nc: int = 13, output_nc: int = 3,
                   ngf: int = 64, num_downs: int = 8, size: int = 512,
                   prefix: str = "netG.model", norm: str = "batch") -> Topology:
    """``norm="instance"`` describes the same nets built with ``norm_layer=nn.InstanceNorm2d`` (networks.py:459, :555): the
    norms own no tensors (affine=False, no running stats) and the level convs -- not the ResidualBlock convs -- have a bias
    (``use_bias``, networks.py:494 / :590)."""
    if norm not in ("batch", "instance"):
        raise ValueError("norm must be 'batch' or 'instance'")
    inst = norm == "instance"
    if variant not in VARIANTS:
        raise ValueError("variant must be 'normal' or 'large' (the 'small' pix2pix U-Net of "
                         "networks.py:680-769 is not on the shipped path), got %r" % (variant,))
    if num_downs < 5:
        raise ValueError("num_downs must be >= 5 (networks.py:563 builds num_downs-5 middle blocks)")
    if size % (1 << num_downs) != 0:
        raise ValueError("size must be a multiple of 2**num_downs")
    topo = Topology(variant, input_nc, output_nc, ngf, num_downs, size, norm=norm)
    nres = topo.nres

    def add_bn(key: str, c: int):
        if inst:
            return
        topo.tensors[key + ".weight"] = (c,)
        topo.tensors[key + ".bias"] = (c,)
        topo.tensors[key + ".running_mean"] = (c,)
        topo.tensors[key + ".running_var"] = (c,)
        topo.tensors[key + ".num_batches_tracked"] = ()

    def add_res(lname: str, key: str, c: int, h: int, r: int):
        # ResidualBlock.block: 0 conv, 1 BN, 2 relu, 3 conv, 4 BN   (networks.py:662-668)
        for part, ci, bi, residual in (("a", 0, 1, False), ("b", 3, 4, True)):
            wk = "%s.block.%d.weight" % (key, ci)
            bk = "%s.block.%d" % (key, bi)
            topo.tensors[wk] = (c, c, 3, 3)
            add_bn(bk, c)
            topo.convs.append(ConvSpec("%s.res%d.%s" % (lname, r, part), wk, bk, c, c, h, h, 1,
                                       False, False, residual, True))

    def walk(depth: int, pfx: str, h_in: int):
        outermost = depth == 0
        innermost = depth == num_downs - 1
        cin, inner, cout = level_channels(depth, ngf, input_nc, output_nc)
        h = h_in // 2
        i = 0
        wk = "%s.model.%d.weight" % (pfx, i)
        topo.tensors[wk] = (inner, cin, 3, 3)
        if inst:
            topo.tensors["%s.model.%d.bias" % (pfx, i)] = (inner,)
        i += 1
        bn_key = None
        if not (outermost or innermost):
            bn_key = "%s.model.%d" % (pfx, i)
            add_bn(bn_key, inner)
            i += 1
        topo.convs.append(ConvSpec("L%d.down" % depth, wk, bn_key, cin, inner, h_in, h, 2,
                                   False, False, False, True))
        i += 1  # relu
        for r in range(nres):
            add_res("L%d.d" % depth, "%s.model.%d" % (pfx, i), inner, h, r)
            i += 1
        if not innermost:
            walk(depth + 1, "%s.model.%d" % (pfx, i), h)
            i += 1
        i += 1  # upsample
        up_cin = inner if innermost else inner * 2
        wk = "%s.model.%d.weight" % (pfx, i)
        topo.tensors[wk] = (cout, up_cin, 3, 3)
        if inst:
            topo.tensors["%s.model.%d.bias" % (pfx, i)] = (cout,)
        i += 1
        bn_key = None
        if not outermost:
            bn_key = "%s.model.%d" % (pfx, i)
            add_bn(bn_key, cout)
            i += 2  # BN, relu
        topo.convs.append(ConvSpec("L%d.up" % depth, wk, bn_key, up_cin, cout, h, h_in, 1,
                                   True, not innermost, False, not outermost, tanh=outermost))
        if not outermost:
            for r in range(nres):
                add_res("L%d.u" % depth, "%s.model.%d" % (pfx, i), cout, h_in, r)
                i += 1

    walk(0, prefix, size)
    return topo


def expected_keys(variant: str, **kw) -> List[str]:
    return list(build_topology(variant, **kw).tensors.keys())
