"""Per-kernel parity: the fused implicit-GEMM 3x3 conv (lspf2f_conv3x3, the unit the generator
is made of) against torch.nn.functional on the host CPU, in every mode the network uses it:
stride 1/2, nearest-x2 upsample prologue, two-source (concat) K loop, folded-BN epilogue,
residual add, ReLU, split-K, every instantiated tile shape, ragged M / N edges.
A/B are random and asymmetric, so a transposed C/D mapping cannot pass.
"""
import ctypes
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def run_conv(dev, x0, x1, w, scale, shift, res, stride, up, relu, tile=(0, 0), split_k=0, k_group=0, dtype=0):
    """x0/x1: NCHW cpu tensors; w OIHW. Returns NCHW cpu tensor computed by the HIP kernel."""
    from livespeechportraits_amd import _native as N
    lib = N.load()
    b, c0, hs, ws = x0.shape
    c1 = x1.shape[1] if x1 is not None else 0
    cout = w.shape[0]
    tdt = {1: torch.bfloat16, 2: torch.float16}.get(dtype, torch.float32)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(dev).to(tdt)
    d0 = nhwc(x0)
    d1 = nhwc(x1) if x1 is not None else None
    if k_group == -1 and tile[0] > 3000:
        wp = pack_rowup(w).to(dev).to(tdt)                    # the up-conv row kernel's fragment order (bf16 | fp16)
    elif k_group == -1 and tile[0] == 2000:
        wp = pack_bandconv(w).to(dev).to(tdt)                 # the band kernel's fragment order
    elif k_group == -1 and tile[0] > 1000:
        wp = pack_rowconv(w).to(dev).to(tdt)                  # the row kernel's MFMA-fragment order
    elif k_group == -1 and dtype != 0:
        wp = pack_fullk16(w, c0, 2 if c1 else 1).to(dev).to(tdt)   # the 16-bit full-K kernel's tile-blocked layout (fullk16.hip)
    elif k_group == -1:
        wp = pack_fullk(w, c0, 2 if c1 else 1).to(dev)       # the full-K kernel's tile-blocked layout
    elif up == 2:
        wp = pack_subpixel(w).to(dev).to(tdt)                # [parity][co][a][b][ci]
    else:
        wp = w.permute(0, 2, 3, 1).contiguous().to(dev).to(tdt)   # [co][ky][kx][ci]
    dsc = scale.to(dev) if scale is not None else None
    dsh = shift.to(dev) if shift is not None else None
    ho = 2 * hs if up else (hs + stride - 1) // stride
    dres = nhwc(res) if res is not None else None
    out = torch.full((b, ho, ho, cout), float("nan"), device=dev, dtype=tdt)
    sb = lib.lspf2f_conv3x3_scratch_bytes(b, hs, ws, c0, c1, cout, stride, int(up), tile[0], tile[1], split_k, k_group, dtype)
    scratch = torch.empty(max(sb, 4), dtype=torch.uint8, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    rc = lib.lspf2f_conv3x3(p(d0), p(d1), p(wp), p(dsc), p(dsh), p(dres), p(out), b, hs, ws, c0, c1, cout,
                            stride, int(up), int(relu), tile[0], tile[1], split_k, k_group, dtype, p(scratch), scratch.numel(),
                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    N.check(rc)
    torch.cuda.synchronize()
    return out.float().permute(0, 3, 1, 2).contiguous().cpu()


def pack_fullk(w, c0, nch):
    """Independent (torch indexing) statement of the tile-blocked weight layout of conv3x3_fullk:
    [cout/16][source * 9 + tap][wave 4][g][lane 64][4], lane = li + 16 kq holds channels 4 (kq 4G + wave G + g) .. + 3 of source
    `source`, tap `tap`, output row 16 nt + li.  w: OIHW."""
    cout = w.shape[0]
    G = c0 // 64
    rows = w.permute(0, 2, 3, 1).reshape(cout, 9, nch, c0)                                  # [n][tap][source][c]
    nt, T, wv, g, lane = torch.meshgrid(torch.arange(cout // 16), torch.arange(nch * 9), torch.arange(4), torch.arange(G),
                                        torch.arange(64), indexing="ij")
    li, kq = lane % 16, lane // 16
    c = 4 * (kq * 4 * G + wv * G + g)
    n, src, tap = nt * 16 + li, T // 9, T % 9
    return torch.stack([rows[n, tap, src, c + e] for e in range(4)], -1).contiguous().float()


def pack_fullk16(w, c0, nch):
    """The 16-bit twin (conv3x3_fullk16): [cout/16][source * 9 + tap][wave 4][g][lane 64][8] with G = c0 / 128 groups; lane = li + 16 kq holds channels
    8 (kq 4G + wave G + g) .. + 7 of source `source`, tap `tap`, output row 16 nt + li (pack_fullk16_weights on the host).  fp32 here; run_conv narrows."""
    cout = w.shape[0]
    G = c0 // 128
    rows = w.permute(0, 2, 3, 1).reshape(cout, 9, nch, c0)                                  # [n][tap][source][c]
    nt, T, wv, g, lane = torch.meshgrid(torch.arange(cout // 16), torch.arange(nch * 9), torch.arange(4), torch.arange(G),
                                        torch.arange(64), indexing="ij")
    li, kq = lane % 16, lane // 16
    c = 8 * (kq * 4 * G + wv * G + g)
    n, src, tap = nt * 16 + li, T // 9, T % 9
    return torch.stack([rows[n, tap, src, c + e] for e in range(8)], -1).contiguous().float()


def pack_rowconv(w):
    """OIHW [C][C][3][3] (C = 64 | 128) -> bf16 [nb C/32][tap 9][kc C/16][lane 64][8]: A-fragment (nb, tap, kc), lane = channel
    nb*32 + (lane & 31), k = kc*16 + 8*(lane >> 5) .. +7 (Plan::pack does the same on the host, pack_rowconv_weights)"""
    c = w.shape[0]
    rows = w.permute(0, 2, 3, 1).reshape(c // 32, 32, 9, c // 16, 2, 8)      # [nb][ch][tap][kc][hi][e]
    return rows.permute(0, 2, 3, 4, 1, 5).contiguous()                       # [nb][tap][kc][hi][ch][e]: lane = hi*32 + ch (fp32; run_conv narrows)


def pack_rowup(w):
    """OIHW [64][256][3][3] -> the sub-pixel form [par][co][a][b][ci] -> bf16 [nb 2][par 4][tap 4][kc 16][lane 64][8]
    (pack_rowup_weights on the host)"""
    sub = pack_subpixel(w).reshape(4, 2, 32, 4, 16, 2, 8)                       # [par][nb][ch][tap][kc][hi][e]
    return sub.permute(1, 0, 3, 4, 5, 2, 6).contiguous()                         # [nb][par][tap][kc][hi][ch][e]


def pack_bandconv(w):
    """OIHW [Cout][512][3][3] -> bf16 [slice Cout/32][K quarter 4][tap 9][kc 8][lane 64][8]: lane = channel slice*32 + (lane & 31),
    k = input channel q*128 + kc*16 + 8*(lane >> 5) .. +7 of the tap (pack_bandconv_weights on the host)"""
    cout = w.shape[0]
    rows = w.permute(0, 2, 3, 1).reshape(cout // 32, 32, 9, 4, 8, 2, 8)        # [cs][ch][tap][q][kc][hi][e]
    return rows.permute(0, 3, 2, 4, 5, 1, 6).contiguous()                        # [cs][q][tap][kc][hi][ch][e]


def pack_subpixel(w):
    """Independent (numpy-free, torch float64) statement of the sub-pixel weight fold: for output
    parity (py, px) of Upsample(x2, nearest) + Conv3x3, the 3x3 taps that read the same source pixel
    are summed.  [co][ci][3][3] -> [4][co][2][2][ci]"""
    groups = {0: [[0], [1, 2]], 1: [[0, 1], [2]]}   # parity -> tap groups for a = 0, 1
    wd = w.double()
    out = torch.zeros(4, w.shape[0], 2, 2, w.shape[1], dtype=torch.float64)
    for py in (0, 1):
        for px in (0, 1):
            for a in (0, 1):
                for b in (0, 1):
                    acc = 0
                    for ky in groups[py][a]:
                        for kx in groups[px][b]:
                            acc = acc + wd[:, :, ky, kx]
                    out[py * 2 + px, :, a, b, :] = acc
    return out.float().contiguous()


def ref_conv(x0, x1, w, scale, shift, res, stride, up, relu):
    x = x0 if x1 is None else torch.cat([x0, x1], 1)
    x = x.double()
    if up:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    y = F.conv2d(x, w.double(), None, stride, 1)
    if scale is not None:
        y = y * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if res is not None:
        y = y + res.double()
    if relu:
        y = F.relu(y)
    return y.float()


def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(*shape, generator=g) * 2 - 1


CASES = [
    # b, c0, c1, cout, hs, stride, up, bn, res, relu, tile, split
    (1, 64, 0, 64, 32, 1, False, True, False, True, (0, 0), 0),        # res-block conv a
    (1, 64, 0, 64, 32, 1, False, True, True, True, (0, 0), 0),         # res-block conv b (+x)
    (2, 64, 0, 128, 32, 2, False, True, False, True, (0, 0), 0),       # down conv
    (1, 128, 0, 128, 16, 2, False, False, False, True, (0, 0), 0),     # innermost down conv (no BN)
    (1, 64, 64, 32, 16, 1, True, True, False, True, (0, 0), 0),        # up conv: upsample + concat
    (1, 128, 0, 128, 2, 1, True, True, False, True, (0, 0), 0),        # innermost up conv, tiny spatial
    (1, 512, 0, 512, 2, 1, False, True, True, True, (0, 0), 0),        # 2x2 spatial weight-streaming
    (1, 512, 0, 512, 4, 2, False, True, False, True, (0, 0), 0),       # 4 -> 2
    (3, 32, 0, 32, 8, 1, False, True, False, False, (0, 0), 0),        # ngf=32-style, Cout < tile
    (1, 64, 0, 96, 24, 1, False, True, False, True, (0, 0), 0),        # ragged N (96) and M (576)
]
for tm, tn in [(128, 128), (128, 64), (64, 128), (64, 64), (32, 128), (32, 64)]:
    CASES.append((1, 64, 32, 160, 20, 1, False, True, True, True, (tm, tn), 1))   # ragged everything
    up_ok = (tm, tn) == (64, 64)   # the 9-tap gather form is instantiated for 64x64 / 32x64(g4) only
    CASES.append((1, 96, 0, 128, 12, 1, up_ok, True, False, True, (tm, tn), 3))   # forced split-K
    CASES.append((2, 64, 0, 64, 18, 2, False, False, False, False, (tm, tn), 2))


# K-tiles-per-step variants (k_group 2 / 4), including ranges that are not a multiple of the
# group (tail tiles are zero-filled) and the single-step (one LDS buffer) path
GROUP_CASES = []
for (tm, tn), g in [((128, 64), 2), ((64, 64), 2), ((64, 64), 4), ((32, 64), 4)]:
    GROUP_CASES.append((1, 64, 32, 96, 10, 1, False, True, True, True, (tm, tn), 1, g))   # 27 tiles, 1 split
    GROUP_CASES.append((1, 64, 0, 64, 6, 1, g == 4, True, False, True, (tm, tn), 5, g))   # 18 tiles / 5 splits = 4,4,4,4,2
    GROUP_CASES.append((1, 128, 0, 64, 4, 2, False, False, False, False, (tm, tn), 9, g))  # 36 tiles / 9 = 4 each: single step


@pytest.mark.parametrize("cfg", GROUP_CASES, ids=lambda c: "c%d+%d_o%d_h%d_t%dx%d_k%d_g%d" % (
    c[1], c[2], c[3], c[4], c[10][0], c[10][1], c[11], c[12]))
def test_conv3x3_k_groups(cfg, gpu_device):
    b, c0, c1, cout, hs, stride, up, bn, res, relu, tile, split, g = cfg
    x0 = rnd(b, c0, hs, hs, seed=11)
    x1 = rnd(b, c1, hs, hs, seed=12) if c1 else None
    w = rnd(cout, c0 + c1, 3, 3, seed=13) * 0.05
    scale = rnd(cout, seed=14) * 0.5 + 1.0 if bn else None
    shift = rnd(cout, seed=15) * 0.1 if bn else None
    ho = 2 * hs if up else (hs + stride - 1) // stride
    r = rnd(b, cout, ho, ho, seed=16) if res else None
    got = run_conv(gpu_device, x0, x1, w, scale, shift, r, stride, up, relu, tile, split, g)
    ref = ref_conv(x0, x1, w, scale, shift, r, stride, up, relu)
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() <= 2e-5


FULLK_CASES = [
    # b, c0, c1, cout, hs, up, bn, res, relu, tile       (tile 16x16 / 32x16 = the full-K single-launch kernel)
    (1, 512, 0, 512, 16, False, True, True, True, (32, 16)),      # L4 / L5 res conv b at batch 1: 256 tiles of 32 px x 16 ch
    (1, 512, 0, 512, 16, False, True, False, True, (16, 16)),     # the same layer in 16-pixel tiles
    (1, 512, 0, 512, 8, False, True, True, True, (16, 16)),       # 8x8 res conv: a 16-pixel block is two rows
    (1, 512, 512, 512, 8, True, True, False, True, (32, 16)),     # L5.up: nearest x2 (8 -> 16) + concat, two staged sources
    (1, 512, 512, 512, 4, True, True, False, True, (16, 16)),     # L6.up: 4 -> 8
    (2, 256, 0, 128, 4, False, True, True, False, (16, 16)),      # 4x4: one block per frame, G = 4, no ReLU
    (3, 128, 0, 256, 2, False, False, False, True, (16, 16)),     # 2x2: 4 of a block's 16 pixels are real; G = 2; no BN
    (2, 256, 256, 128, 2, True, True, False, True, (16, 16)),     # 2 -> 4 with concat
    (5, 128, 0, 128, 16, False, True, True, True, (32, 16)),      # batch 5: tiles index (frame, row pair)
    (1, 512, 0, 512, 16, False, True, True, True, (0, 0)),        # planner's own choice for the batch-1 16x16 layer
]


@pytest.mark.parametrize("cfg", FULLK_CASES, ids=lambda c: "b%d_c%d+%d_o%d_h%d%s_t%dx%d" % (
    c[0], c[1], c[2], c[3], c[4], "up" if c[5] else "", c[9][0], c[9][1]))
def test_conv3x3_full_k_kernel(cfg, gpu_device):
    b, c0, c1, cout, hs, up, bn, res, relu, tile = cfg
    x0 = rnd(b, c0, hs, hs, seed=31)
    x1 = rnd(b, c1, hs, hs, seed=32) if c1 else None
    w = rnd(cout, c0 + c1, 3, 3, seed=33) * 0.05
    scale = rnd(cout, seed=34) * 0.5 + 1.0 if bn else None
    shift = rnd(cout, seed=35) * 0.1 if bn else None
    ho = 2 * hs if up else hs
    r = rnd(b, cout, ho, ho, seed=36) if res else None
    got = run_conv(gpu_device, x0, x1, w, scale, shift, r, 1, up, relu, tile)
    ref = ref_conv(x0, x1, w, scale, shift, r, 1, up, relu)
    assert torch.isfinite(got).all()
    err = (got - ref).abs().max().item()
    assert err <= 3e-5, err                       # K up to 9216 fp32 products of O(0.05) weights
    again = run_conv(gpu_device, x0, x1, w, scale, shift, r, 1, up, relu, tile)
    assert torch.equal(again, got)                # fixed summation order
    if tile != (0, 0):
        # the shipped form: weights in the tile-blocked layout (k_group = -1); same K order, so bit-identical to the row-layout form
        tiled = run_conv(gpu_device, x0, x1, w, scale, shift, r, 1, up, relu, tile, k_group=-1)
        assert torch.equal(tiled, got)


FULLK_SPLIT_CASES = [
    # b, c0, c1, cout, hs, up, bn, res, relu      (K in two halves over twice the workgroups, combined in the launch)
    (1, 512, 0, 512, 8, False, True, True, True, 16),      # the 8x8 res convs at batch 1: one source read as two 256-channel half-sources
    (1, 512, 512, 512, 4, True, True, False, True, 16),    # L6.up: 4 -> 8 over the concat, half = source
    (1, 256, 0, 128, 8, False, False, True, False, 16),    # 128-channel halves (G = 2), no BN, no ReLU
    (2, 256, 256, 128, 4, False, True, False, True, 16),   # 4x4 frames, batch 2, two sources, no upsample
    (1, 512, 0, 128, 16, False, True, True, True, 16),     # 16x16 in 16-pixel tiles: 16 tiles per frame
    (1, 512, 0, 512, 16, False, True, True, True, 32),     # 16x16 in 32-pixel tiles (two pixel blocks per workgroup): the 16x16 res convs at batch 1
    (1, 512, 512, 256, 8, True, True, False, True, 32),    # L5.up-shaped: 8 -> 16 over the concat, 32-pixel tiles
]


@pytest.mark.parametrize("cfg", FULLK_SPLIT_CASES, ids=lambda c: "b%d_c%d+%d_o%d_h%d%s_t%d" % (c[0], c[1], c[2], c[3], c[4], "up" if c[5] else "", c[9]))
def test_conv3x3_full_k_kernel_k_split(cfg, gpu_device):
    """conv3x3_fullk with its K split in two: workgroup (tile, z) reads source z of a concat input, or channel half z of a single source (weights packed
    as two half-sources), the second arriver adds the two partial tiles in z order.  Against the fp64 convolution, against the unsplit kernel (another
    summation order: a few ulp), twice (fixed order: bit-identical; the arrival counters must be back at zero)."""
    from livespeechportraits_amd import _native as N
    b, c0, c1, cout, hs, up, bn, res, relu, tm = cfg
    pbk = tm // 16
    x0 = rnd(b, c0, hs, hs, seed=51)
    x1 = rnd(b, c1, hs, hs, seed=52) if c1 else None
    w = rnd(cout, c0 + c1, 3, 3, seed=53) * 0.05
    scale = rnd(cout, seed=54) * 0.5 + 1.0 if bn else None
    shift = rnd(cout, seed=55) * 0.1 if bn else None
    ho = 2 * hs if up else hs
    r = rnd(b, cout, ho, ho, seed=56) if res else None
    ref = ref_conv(x0, x1, w, scale, shift, r, 1, up, relu)
    whole = run_conv(gpu_device, x0, x1, w, scale, shift, r, 1, up, relu, (tm, 16), k_group=-1)
    lib, dev = N.load(), gpu_device
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(dev)
    d0, d1 = nhwc(x0), nhwc(x1) if x1 is not None else None
    wp = (pack_fullk(w, c0, 2) if c1 else pack_fullk(w, c0 // 2, 2)).to(dev)
    dsc, dsh = (scale.to(dev), shift.to(dev)) if bn else (None, None)
    dres = nhwc(r) if r is not None else None
    sb = lib.lspf2f_conv3x3_scratch_bytes(b, hs, hs, c0, c1, cout, 1, int(up), tm, 16, 2, -1, 0)
    ntile = b * (ho * ho // (16 * pbk)) * (cout // 16)
    assert sb == ntile * (2 * pbk * 256 * 4 + 4)
    scratch = torch.zeros(sb, dtype=torch.uint8, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    outs = []
    for _ in range(2):
        out = torch.full((b, ho, ho, cout), float("nan"), device=dev)
        N.check(lib.lspf2f_conv3x3(p(d0), p(d1), p(wp), p(dsc), p(dsh), p(dres), p(out), b, hs, hs, c0, c1, cout, 1, int(up), int(relu), tm, 16, 2, -1, 0,
                                   p(scratch), scratch.numel(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        outs.append(out.permute(0, 3, 1, 2).contiguous().cpu())
        assert int(scratch[ntile * 2 * pbk * 1024:].view(torch.int32).abs().sum().item()) == 0      # counters left at zero
    got = outs[0]
    assert torch.isfinite(got).all() and torch.equal(outs[0], outs[1])
    assert (got - ref).abs().max().item() <= 3e-5
    assert (got - whole).abs().max().item() <= 1.5e-5           # another summation order over K = 4608 / 9216 products, outputs of a few units


FULLK_S2_CASES = [
    # b, c0, cout, hs          (stride-2 convs of the small levels on the K-split full-K kernel; hs -> hs / 2)
    (1, 512, 512, 32),      # L4.down: 16x16 out, 16-pixel tiles = one output row, 3 source rows
    (1, 512, 512, 16),      # L5.down: 8x8 out, 2 output rows per tile, 5 source rows
    (1, 512, 512, 8),       # L6.down: 4x4 out, the whole frame in one tile, every source row
    (1, 256, 128, 16),      # 128-channel halves, one N-slice group
    (2, 256, 256, 8),       # two frames: tiles index (frame, row block)
]


@pytest.mark.parametrize("cfg", FULLK_S2_CASES, ids=lambda c: "b%d_c%d_o%d_h%d" % c)
def test_conv3x3_full_k_kernel_stride2(cfg, gpu_device):
    """conv3x3_fullk with stride 2 (K-split form only: half the channels of the 2 nr + 1 source rows fit LDS): against the fp64 convolution, against the
    implicit GEMM on the same problem, bit-identical repeats, counters back at zero."""
    from livespeechportraits_amd import _native as N
    b, c0, cout, hs = cfg
    ho = hs // 2
    x0 = rnd(b, c0, hs, hs, seed=61)
    w = rnd(cout, c0, 3, 3, seed=63) * 0.05
    scale, shift = rnd(cout, seed=64) * 0.5 + 1.0, rnd(cout, seed=65) * 0.1
    ref = ref_conv(x0, None, w, scale, shift, None, 2, False, True)
    gemm = run_conv(gpu_device, x0, None, w, scale, shift, None, 2, False, True)             # the implicit GEMM in the tiling its planner picks
    lib, dev = N.load(), gpu_device
    d0 = x0.permute(0, 2, 3, 1).contiguous().to(dev)
    wp = pack_fullk(w, c0 // 2, 2).to(dev)
    dsc, dsh = scale.to(dev), shift.to(dev)
    sb = lib.lspf2f_conv3x3_scratch_bytes(b, hs, hs, c0, 0, cout, 2, 0, 16, 16, 2, -1, 0)
    ntile = b * (ho * ho // 16) * (cout // 16)
    assert sb == ntile * (2 * 256 * 4 + 4)
    scratch = torch.zeros(sb, dtype=torch.uint8, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    outs = []
    for _ in range(2):
        out = torch.full((b, ho, ho, cout), float("nan"), device=dev)
        N.check(lib.lspf2f_conv3x3(p(d0), None, p(wp), p(dsc), p(dsh), None, p(out), b, hs, hs, c0, 0, cout, 2, 0, 1, 16, 16, 2, -1, 0,
                                   p(scratch), scratch.numel(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        outs.append(out.permute(0, 3, 1, 2).contiguous().cpu())
        assert int(scratch[ntile * 2 * 1024:].view(torch.int32).abs().sum().item()) == 0
    got = outs[0]
    assert torch.isfinite(got).all() and torch.equal(outs[0], outs[1])
    assert (got - ref).abs().max().item() <= 3e-5
    assert (got - gemm).abs().max().item() <= 1.5e-5


TINY_CASES = [
    # b, cin, cout, hs, stride, up, bn, res, relu     (tile 1x1 = the single-launch tiny-M kernel)
    (1, 512, 512, 4, 1, False, True, True, True),     # L6 res conv b
    (1, 512, 512, 2, 1, False, True, False, True),    # L7 res conv a (2x2: every tap row/col hits padding)
    (1, 512, 512, 4, 2, False, False, False, True),   # L7.down (4 -> 2, no BN)
    (1, 512, 512, 2, 1, True, True, False, True),     # L7.up (9-tap nearest-x2 gather, 2 -> 4)
    (2, 256, 64, 2, 1, False, True, True, False),     # batch folded into M (8 pixels), Cin 256, no ReLU
    (4, 256, 96, 2, 1, False, False, False, False),   # M = 16 exactly, Cout not a multiple of 64
    (1, 256, 36, 1, 1, True, True, False, True),      # 1x1 source upsampled to 2x2
]


@pytest.mark.parametrize("cfg", TINY_CASES, ids=lambda c: "b%d_c%d_o%d_h%d_s%d_up%d_bn%d_res%d_relu%d" % c)
def test_conv3x3_tiny_m_kernel(cfg, gpu_device):
    b, cin, cout, hs, stride, up, bn, res, relu = cfg
    x0 = rnd(b, cin, hs, hs, seed=31)
    w = rnd(cout, cin, 3, 3, seed=33) * 0.05
    scale = rnd(cout, seed=34) * 0.5 + 1.0 if bn else None
    shift = rnd(cout, seed=35) * 0.1 if bn else None
    ho = 2 * hs if up else (hs + stride - 1) // stride
    r = rnd(b, cout, ho, ho, seed=36) if res else None
    got = run_conv(gpu_device, x0, None, w, scale, shift, r, stride, int(up), relu, (1, 1))
    ref = ref_conv(x0, None, w, scale, shift, r, stride, up, relu)
    assert got.shape == ref.shape and torch.isfinite(got).all()
    assert (got - ref).abs().max().item() <= 2e-5
    # the MFMA path computes the same thing
    mf = run_conv(gpu_device, x0, None, w, scale, shift, r, stride, int(up), relu, (32, 64), 0, 4)
    assert (got - mf).abs().max().item() <= 2e-5


def test_conv3x3_tiny_m_rejects_unsupported(gpu_device):
    from livespeechportraits_amd import _native as N
    with pytest.raises(N.Lspf2fError):      # 8x8 output = 64 pixels > 16
        run_conv(gpu_device, rnd(1, 256, 8, 8), None, rnd(64, 256, 3, 3), None, None, None, 1, 0, False, (1, 1))
    with pytest.raises(N.Lspf2fError):      # Cin not a multiple of 256
        run_conv(gpu_device, rnd(1, 128, 2, 2), None, rnd(64, 128, 3, 3), None, None, None, 1, 0, False, (1, 1))


SUBPIXEL_CASES = [
    # b, c0, c1, cout, hs, tile, split, group
    (1, 64, 64, 64, 16, (0, 0), 0, 0),        # concat up-conv, planner's choice
    (2, 32, 0, 96, 9, (64, 64), 1, 1),        # odd extent, ragged tiles, batch 2
    (1, 128, 128, 32, 8, (128, 64), 2, 2),    # split-K + K groups: partials land on scattered output rows
    (1, 64, 0, 128, 4, (64, 128), 4, 1),
    (1, 32, 32, 64, 2, (32, 64), 2, 4),
]


@pytest.mark.parametrize("cfg", SUBPIXEL_CASES, ids=lambda c: "b%d_c%d+%d_o%d_h%d_t%dx%d_k%d_g%d" % (
    c[0], c[1], c[2], c[3], c[4], c[5][0], c[5][1], c[6], c[7]))
def test_conv3x3_subpixel_upsample(cfg, gpu_device):
    """upsample=2 (4 parities x 2x2 taps, pre-summed weights) == Upsample(2,'nearest') + Conv3x3."""
    b, c0, c1, cout, hs, tile, split, g = cfg
    x0 = rnd(b, c0, hs, hs, seed=21)
    x1 = rnd(b, c1, hs, hs, seed=22) if c1 else None
    w = rnd(cout, c0 + c1, 3, 3, seed=23) * 0.05
    scale, shift = rnd(cout, seed=24) * 0.5 + 1.0, rnd(cout, seed=25) * 0.1
    got = run_conv(gpu_device, x0, x1, w, scale, shift, None, 1, 2, True, tile, split, g)
    ref = ref_conv(x0, x1, w, scale, shift, None, 1, True, True)
    got9 = run_conv(gpu_device, x0, x1, w, scale, shift, None, 1, 1, True)
    assert got.shape == ref.shape and torch.isfinite(got).all()
    assert (got - ref).abs().max().item() <= 2e-5
    assert (got - got9).abs().max().item() <= 2e-5   # both forms of the same op agree


@pytest.mark.parametrize("cfg", CASES, ids=lambda c: "b%d_c%d+%d_o%d_h%d_s%d_up%d_bn%d_res%d_relu%d_t%dx%d_k%d" % (
    c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], c[9], c[10][0], c[10][1], c[11]))
def test_conv3x3_modes(cfg, gpu_device):
    b, c0, c1, cout, hs, stride, up, bn, res, relu, tile, split = cfg
    x0 = rnd(b, c0, hs, hs, seed=1)
    x1 = rnd(b, c1, hs, hs, seed=2) if c1 else None
    w = rnd(cout, c0 + c1, 3, 3, seed=3) * 0.05
    scale = rnd(cout, seed=4) * 0.5 + 1.0 if bn else None
    shift = rnd(cout, seed=5) * 0.1 if bn else None
    ho = 2 * hs if up else (hs + stride - 1) // stride
    r = rnd(b, cout, ho, ho, seed=6) if res else None
    got = run_conv(gpu_device, x0, x1, w, scale, shift, r, stride, up, relu, tile, split)
    ref = ref_conv(x0, x1, w, scale, shift, r, stride, up, relu)
    assert got.shape == ref.shape
    assert torch.isfinite(got).all(), "kernel left unwritten (NaN) outputs"
    err = (got - ref).abs().max().item()
    assert err <= 2e-5, err


def test_conv3x3_impulse_layout(gpu_device):
    """Known-answer test: a one-hot input and a one-hot weight tap must land on exactly one
    output pixel/channel -- catches any transposed fragment or K-permutation mismatch."""
    c, h = 64, 8
    x = torch.zeros(1, c, h, h)
    x[0, 37, 5, 2] = 2.0
    w = torch.zeros(64, c, 3, 3)
    w[11, 37, 0, 2] = 3.0          # out[11][oy][ox] += 3 * in[37][oy-1][ox+1]
    got = run_conv(gpu_device, x, None, w, None, None, None, 1, False, False)
    exp = torch.zeros(1, 64, h, h)
    exp[0, 11, 6, 1] = 6.0
    assert torch.equal(got, exp)


def test_conv3x3_rejects_bad_arguments(gpu_device):
    from livespeechportraits_amd import _native as N
    x = rnd(1, 48, 8, 8)          # 48 % 32 != 0
    w = rnd(64, 48, 3, 3)
    with pytest.raises(N.Lspf2fError):
        run_conv(gpu_device, x, None, w, None, None, None, 1, False, False)


def bf16r(t):
    """round to bf16 and back (what the bf16 path stores in HBM)"""
    return t.to(torch.bfloat16).float() if t is not None else None


BF16_CASES = [
    # b, c0, c1, cout, hs, stride, up(0/1/2), res, tile, split, group
    (1, 64, 0, 64, 32, 1, 0, True, (0, 0), 0, 0),
    (2, 64, 64, 128, 16, 1, 2, False, (0, 0), 0, 0),       # sub-pixel concat up-conv
    (1, 128, 0, 256, 16, 2, 0, False, (64, 128), 1, 1),
    (1, 512, 0, 512, 4, 1, 0, True, (1, 1), 0, 0),         # tiny-M kernel
    (1, 512, 0, 512, 2, 1, 1, False, (32, 64), 0, 4),      # 9-tap upsample gather, split-K + reduce
    (1, 256, 0, 192, 12, 1, 0, True, (64, 64), 3, 2),      # ragged, split-K, K groups
]


@pytest.mark.parametrize("cfg", BF16_CASES, ids=lambda c: "b%d_c%d+%d_o%d_h%d_s%d_up%d_res%d_t%dx%d_k%d_g%d" % (
    c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8][0], c[8][1], c[9], c[10]))
def test_conv3x3_bf16_storage(cfg, gpu_device):
    """bf16 activations/weights in HBM, fp32 accumulate + epilogue.  Reference = the same conv on the
    bf16-ROUNDED operands in fp64, output rounded to bf16: the kernel may differ by one bf16 ulp of the
    result (fp32 accumulation order), i.e. 2^-8 relative."""
    b, c0, c1, cout, hs, stride, up, res, tile, split, g = cfg
    x0 = bf16r(rnd(b, c0, hs, hs, seed=41))
    x1 = bf16r(rnd(b, c1, hs, hs, seed=42)) if c1 else None
    w = rnd(cout, c0 + c1, 3, 3, seed=43) * 0.05
    scale, shift = rnd(cout, seed=44) * 0.5 + 1.0, rnd(cout, seed=45) * 0.1
    ho = 2 * hs if up else (hs + stride - 1) // stride
    r = bf16r(rnd(b, cout, ho, ho, seed=46)) if res else None
    got = run_conv(gpu_device, x0, x1, w, scale, shift, r, stride, up, True, tile, split, g, dtype=1)
    wq = bf16r(pack_subpixel(w)) if up == 2 else bf16r(w)
    if up == 2:   # reference for the folded weights: run the 4 parity convs explicitly
        x = x0 if x1 is None else torch.cat([x0, x1], 1)
        xp = F.pad(x.double(), (1, 1, 1, 1))
        ref = torch.zeros(b, cout, 2 * hs, 2 * hs, dtype=torch.float64)
        for py in (0, 1):
            for px in (0, 1):
                acc = 0
                for a in (0, 1):
                    for bb in (0, 1):
                        patch = xp[:, :, a + py: a + py + hs, bb + px: bb + px + hs]
                        acc = acc + torch.einsum("bchw,oc->bohw", patch, wq[py * 2 + px, :, a, bb, :].double())
                ref[:, :, py::2, px::2] = acc
        ref = F.relu(ref * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)).float()
    else:
        ref = ref_conv(x0, x1, wq, scale, shift, r, stride, bool(up), True)
    assert torch.isfinite(got).all()
    tol = (ref.abs() * 2.0 ** -8 + 1e-3)
    assert ((got - ref).abs() <= tol).all(), (got - ref).abs().max().item()


@pytest.mark.parametrize("cfg", BF16_CASES, ids=lambda c: "b%d_c%d+%d_o%d_h%d_s%d_up%d_res%d_t%dx%d_k%d_g%d" % (
    c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8][0], c[8][1], c[9], c[10]))
def test_conv3x3_fp16_storage(cfg, gpu_device):
    """fp16 (IEEE binary16) activations / weights in HBM, fp32 accumulate + epilogue: the storage path behind the reference's opt.fp16
    (models/feature2face_G.py:28-30).  Reference = the same conv on the fp16-ROUNDED operands in fp64; the kernel may differ by one
    fp16 ulp of the result (2^-11 relative) plus the fp32 accumulation-order noise."""
    b, c0, c1, cout, hs, stride, up, res, tile, split, g = cfg
    h16 = lambda t: t.half().float() if t is not None else None
    x0 = h16(rnd(b, c0, hs, hs, seed=41))
    x1 = h16(rnd(b, c1, hs, hs, seed=42)) if c1 else None
    w = rnd(cout, c0 + c1, 3, 3, seed=43) * 0.05
    scale, shift = rnd(cout, seed=44) * 0.5 + 1.0, rnd(cout, seed=45) * 0.1
    ho = 2 * hs if up else (hs + stride - 1) // stride
    r = h16(rnd(b, cout, ho, ho, seed=46)) if res else None
    got = run_conv(gpu_device, x0, x1, w, scale, shift, r, stride, up, True, tile, split, g, dtype=2)
    wq = h16(pack_subpixel(w)) if up == 2 else h16(w)
    if up == 2:
        x = x0 if x1 is None else torch.cat([x0, x1], 1)
        xp = F.pad(x.double(), (1, 1, 1, 1))
        ref = torch.zeros(b, cout, 2 * hs, 2 * hs, dtype=torch.float64)
        for py in (0, 1):
            for px in (0, 1):
                acc = 0
                for a in (0, 1):
                    for bb in (0, 1):
                        patch = xp[:, :, a + py: a + py + hs, bb + px: bb + px + hs]
                        acc = acc + torch.einsum("bchw,oc->bohw", patch, wq[py * 2 + px, :, a, bb, :].double())
                ref[:, :, py::2, px::2] = acc
        ref = F.relu(ref * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)).float()
    else:
        ref = ref_conv(x0, x1, wq, scale, shift, r, stride, bool(up), True)
    assert torch.isfinite(got).all()
    tol = (ref.abs() * 2.0 ** -11 + 2e-4)
    assert ((got - ref).abs() <= tol).all(), (got - ref).abs().max().item()


ROWCONV_CASES = [
    # channels, b, h, rows per strip, residual, relu
    (64, 1, 64, 16, True, True),      # 4 strips per column
    (64, 2, 64, 7, False, True),      # ragged: the last strip has one real row
    (64, 1, 128, 4, True, False),
    (64, 1, 64, 1, True, True),       # one row per strip: every input row is a halo row of its neighbours
    (64, 1, 256, 32, True, True),     # the shipped size: 8 strips of 32 rows x 4 columns of 64-pixel strips (34 row steps, rounded to 36)
    (128, 1, 32, 16, True, True),     # 128 -> 128: one column of 32-pixel strips
    (128, 2, 64, 7, False, True),     # ragged
    (128, 1, 64, 1, True, False),
    (128, 1, 128, 16, True, True),    # the shipped size at batch 8: 8 strips of 16 rows x 4 columns
]


@pytest.mark.parametrize("cfg", ROWCONV_CASES, ids=lambda c: "c%d_b%d_h%d_r%d_res%d_relu%d" % c)
def test_conv3x3_rows_kernel_bf16(cfg, gpu_device):
    """The weights-stationary kernels of the bf16 plans (rowconv.hip, 64 -> 64 and 128 -> 128) against the fp64 conv of the bf16-rounded
    operands (one bf16 ulp of the result), and against the implicit-GEMM kernel on the same inputs: same MFMA, same accumulation order ->
    the same bits."""
    c, b, h, rows, res, relu = cfg
    x0 = bf16r(rnd(b, c, h, h, seed=71))
    w = rnd(c, c, 3, 3, seed=72) * 0.05
    scale, shift = rnd(c, seed=73) * 0.5 + 1.0, rnd(c, seed=74) * 0.1
    r = bf16r(rnd(b, c, h, h, seed=75)) if res else None
    got = run_conv(gpu_device, x0, None, w, scale, shift, r, 1, 0, relu, (1000 + rows, c), 0, 0, dtype=1)
    ref = ref_conv(x0, None, bf16r(w), scale, shift, r, 1, False, relu)
    assert torch.isfinite(got).all()
    tol = (ref.abs() * 2.0 ** -8 + 1e-3)
    assert ((got - ref).abs() <= tol).all(), (got - ref).abs().max().item()
    other = run_conv(gpu_device, x0, None, w, scale, shift, r, 1, 0, relu, (64, 64), 1, 1, dtype=1)
    assert torch.equal(got, other), (got - other).abs().max().item()
    frag = run_conv(gpu_device, x0, None, w, scale, shift, r, 1, 0, relu, (1000 + rows, c), 0, -1, dtype=1)   # the shipped weight layout
    assert torch.equal(got, frag)


def test_conv3x3_rows_kernel_rejects_other_shapes(gpu_device):
    from livespeechportraits_amd import _native as N
    x = bf16r(rnd(1, 128, 64, 64))
    with pytest.raises(N.Lspf2fError):       # 128 in, 64 out
        run_conv(gpu_device, x, None, rnd(64, 128, 3, 3), None, None, None, 1, 0, False, (1016, 64), 0, 0, dtype=1)
    with pytest.raises(N.Lspf2fError):       # 128 -> 128 data on the 64-channel kernel
        run_conv(gpu_device, x, None, rnd(128, 128, 3, 3), None, None, None, 1, 0, False, (1016, 64), 0, 0, dtype=1)
    with pytest.raises(N.Lspf2fError):       # fp32 storage
        run_conv(gpu_device, rnd(1, 64, 64, 64), None, rnd(64, 64, 3, 3), None, None, None, 1, 0, False, (1016, 64), 0, 0, dtype=0)


PATCH16_CASES = [
    # b, cin, cout, h, tile width, channels per workgroup, residual, relu
    (1, 256, 256, 64, 64, 128, True, True),       # the shipped shape of the 64x64 level (16 pixel tiles x 2 channel tiles per frame)
    (2, 128, 256, 64, 64, 64, False, True),       # two channel blocks, cin != cout, 64 channels per workgroup
    (1, 192, 128, 128, 64, 128, True, False),     # two pixel tiles per row: interior tile edges read real neighbours, frame edges zeros; odd number of channel blocks
    (1, 512, 512, 32, 32, 64, True, True),        # the shipped shape of the 32x32 level (8 rows x 32 pixels per tile)
    (3, 128, 128, 32, 32, 128, False, False),     # 12 workgroups: the XCD chunking with a remainder
    (1, 64 * 5, 64, 8, 32, 64, False, True),      # must be refused (8 rows < a tile's 8 rows is fine, 8 % 8 == 0 -> accepted?) -- see the reject test for real refusals
]
PATCH16_CASES = PATCH16_CASES[:5]


@pytest.mark.parametrize("dtype", [1, 2], ids=["bf16", "f16"])
@pytest.mark.parametrize("cfg", PATCH16_CASES, ids=lambda c: "b%d_c%d_o%d_h%d_tw%d_bn%d_res%d_relu%d" % c)
def test_conv3x3_patch_kernel_16bit(cfg, dtype, gpu_device):
    """The patch-staged kernel of the 16-bit plans' 64x64 / 32x32 levels (patch16.hip) against the fp64 conv of the rounded operands (one 16-bit ulp of the
    result) and against the implicit-GEMM kernel on the same inputs (same products, channel-block-major instead of tap-major summation: one ulp, not bit-equal);
    repeated launches bit-identical."""
    b, cin, cout, h, tw, bn, res, relu = cfg
    rt = (lambda t: t.half().float()) if dtype == 2 else bf16r
    x0 = rt(rnd(b, cin, h, h, seed=171))
    w = rnd(cout, cin, 3, 3, seed=172) * 0.05
    scale, shift = rnd(cout, seed=173) * 0.5 + 1.0, rnd(cout, seed=174) * 0.1
    r = rt(rnd(b, cout, h, h, seed=175)) if res else None
    got = run_conv(gpu_device, x0, None, w, scale, shift, r, 1, 0, relu, (7000 + tw, bn), 0, 0, dtype=dtype)
    ref = ref_conv(x0, None, rt(w), scale, shift, r, 1, False, relu)
    assert torch.isfinite(got).all()
    tol = (ref.abs() * 2.0 ** -8 + 1e-3) if dtype == 1 else (ref.abs() * 2.0 ** -11 + 2e-4)
    assert ((got - ref).abs() <= tol).all(), (got - ref).abs().max().item()
    other = run_conv(gpu_device, x0, None, w, scale, shift, r, 1, 0, relu, (128, 128), 1, 1, dtype=dtype)
    assert ((got - other).abs() <= 2 * tol).all(), (got - other).abs().max().item()
    again = run_conv(gpu_device, x0, None, w, scale, shift, r, 1, 0, relu, (7000 + tw, bn), 0, 0, dtype=dtype)
    assert torch.equal(got, again)
    if bn == 64:      # tile_n 64 = the deep-ring form (conv3x3_patch16d), 65 = 64 channels per workgroup in the first form: same products in the same order -> the same bits
        first = run_conv(gpu_device, x0, None, w, scale, shift, r, 1, 0, relu, (7000 + tw, 65), 0, 0, dtype=dtype)
        assert torch.equal(got, first), (got - first).abs().max().item()


PATCHUP16_CASES = [
    # b, c0, c1, cout, low-res h, tile width, channels per workgroup, relu, residual (at the output's resolution)
    (1, 128, 128, 128, 32, 32, 128, True, False),       # two sources, 8 x 32-pixel tiles
    (2, 256, 0, 128, 32, 32, 64, False, True),          # one source, 64 channels per workgroup, residual
    (1, 64, 64, 64, 64, 64, 64, True, False),           # 4 x 64-pixel tiles, two channel blocks (one per source)
    (1, 192, 192, 256, 64, 64, 128, True, True),        # six channel blocks: the ring slot of tap 0 walks 0, 1, 2, 0, 1, 2
    (3, 128, 128, 128, 32, 32, 128, True, False),       # 48 workgroups: XCD chunks with a remainder
    (2, 128, 128, 128, 16, 16, 64, True, False),        # 16 x 16-pixel tiles (a whole low-res frame): a wave's 32-pixel block is two half-rows
    (1, 256, 256, 64, 16, 16, 64, False, True),
]


@pytest.mark.parametrize("dtype", [1, 2], ids=["bf16", "f16"])
@pytest.mark.parametrize("cfg", PATCHUP16_CASES, ids=lambda c: "b%d_c%d+%d_o%d_h%d_tw%d_bn%d_relu%d_res%d" % c)
def test_conv3x3_patch_kernel_upconv_16bit(cfg, dtype, gpu_device):
    """The sub-pixel up-conv form of the patch-staged kernel (conv3x3_patchup16): Upsample x2 + conv3x3 over the concat, against the fp64 evaluation of the four parity
    convolutions on the rounded operands (one 16-bit ulp) and against the implicit GEMM's up4 path (other summation order: two ulps); repeats bit-identical."""
    b, c0, c1, cout, hs, tw, bn, relu, res = cfg
    rt = (lambda t: t.half().float()) if dtype == 2 else bf16r
    x0 = rt(rnd(b, c0, hs, hs, seed=181))
    x1 = rt(rnd(b, c1, hs, hs, seed=182)) if c1 else None
    w = rnd(cout, c0 + c1, 3, 3, seed=183) * 0.05
    scale, shift = rnd(cout, seed=184) * 0.5 + 1.0, rnd(cout, seed=185) * 0.1
    r = rt(rnd(b, cout, 2 * hs, 2 * hs, seed=186)) if res else None
    got = run_conv(gpu_device, x0, x1, w, scale, shift, r, 1, 2, relu, (7100 + tw, bn), 0, 0, dtype=dtype)
    wq = rt(pack_subpixel(w))
    x = x0 if x1 is None else torch.cat([x0, x1], 1)
    xp = F.pad(x.double(), (1, 1, 1, 1))
    ref = torch.zeros(b, cout, 2 * hs, 2 * hs, dtype=torch.float64)
    for py in (0, 1):
        for px in (0, 1):
            acc = 0
            for a in (0, 1):
                for bb in (0, 1):
                    patch = xp[:, :, a + py: a + py + hs, bb + px: bb + px + hs]
                    acc = acc + torch.einsum("bchw,oc->bohw", patch, wq[py * 2 + px, :, a, bb, :].double())
            ref[:, :, py::2, px::2] = acc
    ref = ref * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if r is not None:
        ref = ref + r.double()
    ref = (F.relu(ref) if relu else ref).float()
    assert torch.isfinite(got).all()
    tol = (ref.abs() * 2.0 ** -8 + 1e-3) if dtype == 1 else (ref.abs() * 2.0 ** -11 + 2e-4)
    assert ((got - ref).abs() <= tol).all(), (got - ref).abs().max().item()
    other = run_conv(gpu_device, x0, x1, w, scale, shift, r, 1, 2, relu, (128, 128), 1, 1, dtype=dtype)
    assert ((got - other).abs() <= 2 * tol).all(), (got - other).abs().max().item()
    again = run_conv(gpu_device, x0, x1, w, scale, shift, r, 1, 2, relu, (7100 + tw, bn), 0, 0, dtype=dtype)
    assert torch.equal(got, again)


def test_conv3x3_patch_kernel_rejects_other_shapes(gpu_device):
    from livespeechportraits_amd import _native as N
    x = rnd(1, 128, 64, 64, seed=1); w = rnd(128, 128, 3, 3, seed=2)
    with pytest.raises(N.Lspf2fError):       # fp32 storage
        run_conv(gpu_device, x, None, w, None, None, None, 1, 0, False, (7064, 128), 0, 0, dtype=0)
    with pytest.raises(N.Lspf2fError):       # stride 2
        run_conv(gpu_device, x, None, w, None, None, None, 2, 0, False, (7064, 128), 0, 0, dtype=1)
    with pytest.raises(N.Lspf2fError):       # two sources
        run_conv(gpu_device, x, x, rnd(128, 256, 3, 3, seed=3), None, None, None, 1, 0, False, (7064, 128), 0, 0, dtype=1)
    with pytest.raises(N.Lspf2fError):       # a frame narrower than the tile
        run_conv(gpu_device, rnd(1, 128, 32, 32, seed=4), None, w, None, None, None, 1, 0, False, (7064, 128), 0, 0, dtype=1)
    with pytest.raises(N.Lspf2fError):       # channels not a multiple of 64
        run_conv(gpu_device, rnd(1, 160, 64, 64, seed=5), None, rnd(128, 160, 3, 3, seed=6), None, None, None, 1, 0, False, (7064, 128), 0, 0, dtype=1)


BANDCONV_CASES = [
    # b, h (= w), cout, residual, relu
    (1, 16, 64, True, True),
    (2, 16, 512, False, True),     # 2 frames x 4 tiles x 16 slices
    (3, 8, 96, True, False),       # 8x8 level: 2 tiles per frame
    (8, 8, 512, True, True),       # the shipped shapes at 8 frames
    (8, 16, 512, True, True),
    (8, 4, 512, True, True),       # 4x4 level: a tile = 2 whole frames
    (3, 4, 64, False, True),       # ... the second tile half empty
    (8, 2, 512, True, False),      # 2x2 level: a tile = 8 whole frames
    (5, 2, 96, True, True),        # ... 5 of them real
    (9, 2, 32, False, False),      # ... two tiles
]


@pytest.mark.parametrize("cfg", BANDCONV_CASES, ids=lambda c: "b%d_h%d_o%d_res%d_relu%d" % c)
def test_conv3x3_band_kernel_bf16(cfg, gpu_device):
    """The activation-stationary kernel of the bf16 plans for the 512-channel layers at 16x16 / 8x8 (bandconv.hip) against the fp64 conv of
    the bf16-rounded operands: one bf16 ulp of the result (its K split by channel quarter is a different fp32 summation order from the
    implicit GEMM's, so the two agree to that ulp, not bit for bit)."""
    b, h, cout, res, relu = cfg
    x0 = bf16r(rnd(b, 512, h, h, seed=81))
    w = rnd(cout, 512, 3, 3, seed=82) * 0.02
    scale, shift = rnd(cout, seed=83) * 0.5 + 1.0, rnd(cout, seed=84) * 0.1
    r = bf16r(rnd(b, cout, h, h, seed=85)) if res else None
    got = run_conv(gpu_device, x0, None, w, scale, shift, r, 1, 0, relu, (2000, 32), 0, -1, dtype=1)
    ref = ref_conv(x0, None, bf16r(w), scale, shift, r, 1, False, relu)
    assert torch.isfinite(got).all()
    tol = (ref.abs() * 2.0 ** -8 + 1e-3)
    assert ((got - ref).abs() <= tol).all(), (got - ref).abs().max().item()
    again = run_conv(gpu_device, x0, None, w, scale, shift, r, 1, 0, relu, (2000, 32), 0, -1, dtype=1)
    assert torch.equal(got, again)                                   # fixed summation order


def test_conv3x3_band_kernel_rejects_other_shapes(gpu_device):
    from livespeechportraits_amd import _native as N
    with pytest.raises(N.Lspf2fError):       # 32x32 frames
        run_conv(gpu_device, bf16r(rnd(1, 512, 32, 32)), None, rnd(64, 512, 3, 3), None, None, None, 1, 0, False, (2000, 32), 0, -1, dtype=1)
    with pytest.raises(N.Lspf2fError):       # stride 2
        run_conv(gpu_device, bf16r(rnd(1, 512, 16, 16)), None, rnd(64, 512, 3, 3), None, None, None, 2, 0, False, (2000, 32), 0, -1, dtype=1)


ROWUP_CASES = [
    # b, low-res h (= w), low-res rows per strip, relu
    (1, 32, 8, True),
    (2, 64, 6, True),        # ragged: 64 = 10 strips of 6 + 4
    (1, 64, 2, False),
    (2, 128, 32, True),      # the shipped shape (8 frames: 256 workgroups)
]


@pytest.mark.parametrize("cfg", ROWUP_CASES, ids=lambda c: "b%d_h%d_r%d_relu%d" % c)
def test_conv3x3_upconv_row_kernel_bf16(cfg, gpu_device):
    """rowup256 (rowconv.hip): Upsample x2 + conv3x3 over the concat of two 128-channel sources -> 64 channels in sub-pixel form, against the
    implicit-GEMM kernel's sub-pixel path on the same inputs (same taps, same order, same MFMA -> the same bits) and against the fp64
    reference of the folded bf16 weights (one bf16 ulp)."""
    b, h, rows, relu = cfg
    x0, x1 = bf16r(rnd(b, 128, h, h, seed=91)), bf16r(rnd(b, 128, h, h, seed=92))
    w = rnd(64, 256, 3, 3, seed=93) * 0.05
    scale, shift = rnd(64, seed=94) * 0.5 + 1.0, rnd(64, seed=95) * 0.1
    got = run_conv(gpu_device, x0, x1, w, scale, shift, None, 1, 2, relu, (3000 + rows, 64), 0, -1, dtype=1)
    other = run_conv(gpu_device, x0, x1, w, scale, shift, None, 1, 2, relu, (128, 64), 1, 1, dtype=1)
    assert torch.isfinite(got).all()
    assert torch.equal(got, other), (got - other).abs().max().item()
    wq = bf16r(pack_subpixel(w))
    xp = F.pad(torch.cat([x0, x1], 1).double(), (1, 1, 1, 1))
    ref = torch.zeros(b, 64, 2 * h, 2 * h, dtype=torch.float64)
    for py in (0, 1):
        for px in (0, 1):
            acc = 0
            for a in (0, 1):
                for bb in (0, 1):
                    acc = acc + torch.einsum("bchw,oc->bohw", xp[:, :, a + py: a + py + h, bb + px: bb + px + h], wq[py * 2 + px, :, a, bb, :].double())
            ref[:, :, py::2, px::2] = acc
    ref = ref * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    ref = (F.relu(ref) if relu else ref).float()
    tol = (ref.abs() * 2.0 ** -8 + 1e-3)
    assert ((got - ref).abs() <= tol).all(), (got - ref).abs().max().item()


# ---- Winograd F(2x2, 3x3) kernel (csrc/wino.hip): the stride-1 residual convs of the >= 32x32 levels in fp32 plans ----------------
def pack_wino(w):
    """OIHW -> G g G^T in the kernel's fragment order; the numpy statement of the layout (tools/wino_model.py), not the library's packer"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import wino_model
    return torch.from_numpy(wino_model.pack_u(w.numpy()))


def run_wino(dev, x, w, scale, shift, res, relu, nb, splits):
    from livespeechportraits_amd import _native as N
    lib = N.load()
    b, c, h, _ = x.shape
    cout = w.shape[0]
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(dev)
    d0, wp = nhwc(x), pack_wino(w).to(dev)
    dsc = scale.to(dev) if scale is not None else None
    dsh = shift.to(dev) if shift is not None else None
    dres = nhwc(res) if res is not None else None
    out = torch.full((b, h, h, cout), float("nan"), device=dev)
    sb = lib.lspf2f_conv3x3_scratch_bytes(b, h, h, c, 0, cout, 1, 0, 4000 + nb, 0, splits, -1, 0)       # nb = 3: tile 4003, nb = 1 with U in registers
    scratch = torch.zeros(max(sb, 4), dtype=torch.uint8, device=dev)      # slabs + arrival counters (zero on entry, left zero by the kernel)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    N.check(lib.lspf2f_conv3x3(p(d0), None, p(wp), p(dsc), p(dsh), p(dres), p(out), b, h, h, c, 0, cout, 1, 0, int(relu), 4000 + nb, 0, splits, -1, 0,
                               p(scratch), scratch.numel(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    return out.permute(0, 3, 1, 2).contiguous().cpu()


WINO_CASES = [
    # b, c, cout, h, nb, splits, epilogue (scale/shift + residual + relu)
    (1, 32, 32, 32, 1, 1, False),
    (1, 32, 64, 32, 2, 1, True),
    (2, 64, 64, 32, 2, 1, True),
    (3, 40, 96, 48, 1, 1, True),          # 48 = 3 tile-block columns, channel counts that are no power of two
    (1, 64, 64, 64, 1, 2, True),
    (1, 128, 64, 32, 2, 4, True),
    (2, 256, 32, 16, 1, 8, True),         # 16x16 frames: one tile-block column, halo on every side
    (1, 104, 32, 32, 1, 3, False),        # 13 K-steps in 3 slices of 5, 5, 3
    (1, 64, 64, 256, 2, 1, True),         # the four shapes of the `large` plan at batch 1
    (1, 128, 128, 128, 1, 1, True),
    (1, 256, 256, 64, 1, 2, True),
    (1, 512, 512, 32, 1, 4, True),
    # tile 4003 ("nb" 3): one channel block per wave with the U fragments in registers -- the form the plans take
    (1, 32, 32, 32, 3, 1, False),         # 4 K-steps
    (1, 8, 32, 16, 3, 1, False),          # one K-step: nothing is ever prefetched
    (1, 16, 32, 16, 3, 1, True),          # two
    (3, 40, 96, 48, 3, 1, True),          # five
    (1, 64, 64, 64, 3, 2, True),
    (2, 256, 32, 16, 3, 8, True),
    (1, 104, 32, 32, 3, 3, False),        # slices of 5, 5, 3 steps
    (1, 128, 128, 128, 3, 1, True),
    (1, 256, 256, 64, 3, 2, True),
    (1, 512, 512, 32, 3, 4, True),
]
# tile 4004 ("nb" 4): the register form with FOUR register sets (operands three K-steps ahead; tune key wino_ureg=2) -- an A-B arm prepared at the end of round 4
# without a GPU left to run it; LSP_TEST_UR4=1 adds its cases (tools/sessions/gpu_r5_ur4.sh), they become permanent once they have passed on the hardware
if os.environ.get("LSP_TEST_UR4"):
    WINO_CASES += [c[:4] + (4,) + c[5:] for c in WINO_CASES if c[4] == 3] + [(1, 24, 32, 16, 4, 1, True), (1, 32, 32, 16, 4, 1, False), (1, 48, 32, 16, 4, 2, True)]   # + three, four K-steps; slices of 3


@pytest.mark.parametrize("cfg", WINO_CASES, ids=lambda c: "b%d_c%d_o%d_h%d_nb%d_s%d%s" % (c[:6] + ("_ep" if c[6] else "",)))
def test_conv3x3_winograd(cfg, gpu_device):
    """wino3x3 against the fp64 convolution.  Winograd trades multiplies for additions on the inputs, so its fp32 rounding error is a few
    times the direct form's; the bound is relative to the layer's output range and the direct kernel's error on the same problem is printed
    beside it.  Whole-network parity (the goldens, <= 5e-5) runs through this kernel too."""
    b, c, cout, h, nb, splits, ep = cfg
    g = torch.Generator().manual_seed(1000 + c + cout + h)
    x = torch.randn(b, c, h, h, generator=g)
    w = torch.randn(cout, c, 3, 3, generator=g) / (3.0 * c ** 0.5)
    scale = torch.rand(cout, generator=g) + 0.5 if ep else None
    shift = torch.randn(cout, generator=g) * 0.1 if ep else None
    res = torch.randn(b, cout, h, h, generator=g) if ep else None
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    if ep:
        ref = torch.relu(ref * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1) + res.double())
    got = run_wino(gpu_device, x, w, scale, shift, res, ep, nb, splits)
    assert torch.isfinite(got).all(), "kernel left unwritten (NaN) outputs"
    err = (got.double() - ref).abs().max().item()
    direct = None
    if c % 32 == 0 and h <= 64:
        direct = (run_conv(gpu_device, x, None, w, scale, shift, res, 1, False, ep).double() - ref).abs().max().item()
    print("\nwino %s: max-abs %.2e (output range %.2f); implicit GEMM on the same problem: %s" % (
        cfg, err, ref.abs().max().item(), "%.2e" % direct if direct is not None else "-"))
    assert err <= 1e-5 * max(1.0, ref.abs().max().item()), err
    if splits > 1:       # the in-launch combine sums the slices in a fixed order, whichever workgroup arrives last
        again = run_wino(gpu_device, x, w, scale, shift, res, ep, nb, splits)
        assert torch.equal(got, again)


def test_conv3x3_winograd_impulse_layout(gpu_device):
    """One-hot input and one-hot tap land on exactly one output value, for taps and pixels on tile and tile-block seams."""
    c, h = 16, 32
    for (ci, y, x_, co, ky, kx) in [(5, 7, 15, 11, 0, 2), (13, 8, 16, 40, 2, 0), (0, 0, 0, 63, 1, 1), (9, 31, 31, 32, 0, 0)]:
        x = torch.zeros(1, c, h, h)
        x[0, ci, y, x_] = 2.0
        w = torch.zeros(64, c, 3, 3)
        w[co, ci, ky, kx] = 3.0           # out[co][oy][ox] += 3 * in[ci][oy + ky - 1][ox + kx - 1]
        exp = F.conv2d(x, w, None, 1, 1)
        got = run_wino(gpu_device, x, w, None, None, None, False, 2, 1)
        assert torch.equal(got, exp), (ci, y, x_, co, ky, kx)


def test_conv3x3_winograd_rejects_unsupported_shapes(gpu_device):
    from livespeechportraits_amd import _native as N
    w = rnd(32, 32, 3, 3)
    with pytest.raises(N.Lspf2fError):
        run_wino(gpu_device, rnd(1, 32, 24, 24), w, None, None, None, False, 1, 1)       # 24 % 16 != 0
    with pytest.raises(N.Lspf2fError):
        run_wino(gpu_device, rnd(1, 32, 32, 32), w, None, None, None, False, 2, 1)       # cout 32 needs nb = 1


# ---- wino3x3_chain (round 6 probe, csrc/wino.hip): the convs of one / two ResidualBlocks in ONE launch behind per-tile-block arrival counters ---------------
def run_wino_chain(dev, lib, nl, c, h, splits, mode, x, ws, scs, shs, keep=None):
    """ResidualBlock semantics (networks.py:650-675): layer 2m = conv a -> BN -> ReLU, layer 2m + 1 = conv b -> BN -> += the block's input -> ReLU.  Returns the
    NHWC device outputs of every layer; `keep` carries the device buffers across calls (so a second call with other inputs reuses every address)."""
    from livespeechportraits_amd import _native as N
    if keep is None:
        keep = {"x": torch.empty(1, h, h, c, device=dev), "outs": [torch.empty(1, h, h, c, device=dev) for _ in range(nl)],
                "scratch": torch.zeros(lib.lspf2f_wino_chain_scratch_bytes(nl, 1, h, c, splits), dtype=torch.uint8, device=dev)}
    keep["x"].copy_(x.permute(0, 2, 3, 1))
    for o in keep["outs"]:
        o.fill_(float("nan"))
    dx, outs, scratch = keep["x"], keep["outs"], keep["scratch"]
    us = [pack_wino(w).float().to(dev) for w in ws]
    dsc, dsh = [t.to(dev) for t in scs], [t.to(dev) for t in shs]
    src = [dx] + outs[:-1]
    res = [None if k % 2 == 0 else (dx if k == 1 else outs[k - 2]) for k in range(nl)]
    arr = lambda ts: (ctypes.c_void_p * nl)(*[ctypes.c_void_p(t.data_ptr()) if t is not None else None for t in ts])
    relu = (ctypes.c_int * nl)(*([1] * nl))
    N.check(lib.lspf2f_wino_chain(nl, arr(src), arr(us), arr(dsc), arr(dsh), arr(res), arr(outs), relu, 1, h, c, splits, mode,
                                  ctypes.c_void_p(scratch.data_ptr()), scratch.numel(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    tail = scratch[(splits > 1) * splits * h * h * c * 4:].view(torch.int32)
    assert not tail.any(), "split-K tickets / gate counters not left at zero, or a gate gave up (last word): %s" % tail.nonzero().flatten()[:8].tolist()
    return [o.clone() for o in outs], keep


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(2, 64, 32, 1), (4, 64, 64, 1), (3, 96, 32, 2), (4, 128, 128, 1), (2, 256, 64, 2), (4, 512, 32, 4)],
                         ids=lambda c: "l%d_c%d_h%d_s%d" % c)
def test_wino_chain_matches_one_launch_per_layer_bit_for_bit(cfg, gpu_device):
    """One launch of nlayers x (workgroups of a layer) workgroups, each layer's workgroups gated on the arrival counters of the tile-blocks they read, must give the
    bits of one wino3x3<1> launch per layer -- twice, with DIFFERENT inputs on the SAME buffers (a consumer that read a line before its producer wrote it would return
    the first call's values), for both valid visibility forms (1: acquire + plain loads, 3: sc1 loads); and the result is the ResidualBlock of the reference (fp64 torch)."""
    from livespeechportraits_amd import _native as N
    lib = N.load()
    nl, c, h, splits = cfg
    g = torch.Generator().manual_seed(77 + nl + c + h)
    ws = [torch.randn(c, c, 3, 3, generator=g) * (0.7 / (3.0 * c ** 0.5)) for _ in range(nl)]
    scs = [torch.rand(c, generator=g) + 0.5 for _ in range(nl)]
    shs = [torch.randn(c, generator=g) * 0.1 for _ in range(nl)]
    keeps = {}
    for rnd_i in range(2):
        x = torch.randn(1, c, h, h, generator=g)
        ref, cur, blk_in = [], x.double(), x.double()
        for k in range(nl):
            y = F.conv2d(cur, ws[k].double(), None, 1, 1) * scs[k].double().view(1, -1, 1, 1) + shs[k].double().view(1, -1, 1, 1)
            if k % 2 == 1:
                y = y + blk_in
            cur = torch.relu(y)
            if k % 2 == 1:
                blk_in = cur
            ref.append(cur)
        base, keeps[0] = run_wino_chain(gpu_device, lib, nl, c, h, splits, 0, x, ws, scs, shs, keeps.get(0))
        for k in range(nl):
            got = base[k].permute(0, 3, 1, 2).double().cpu()
            assert torch.isfinite(got).all()
            assert (got - ref[k]).abs().max().item() <= 2e-5 * max(1.0, ref[k].abs().max().item()), (k, (got - ref[k]).abs().max().item())
        for mode in (1, 3, 2):
            outs, keeps[mode] = run_wino_chain(gpu_device, lib, nl, c, h, splits, mode, x, ws, scs, shs, keeps.get(mode))
            for k in range(nl):
                assert torch.equal(outs[k], base[k]), "mode %d, layer %d, round %d differs from one launch per layer" % (mode, k, rnd_i)


@pytest.mark.gpu
def test_wino_chain_rejects_aliased_buffers_and_bad_arguments(gpu_device):
    from livespeechportraits_amd import _native as N
    lib = N.load()
    c, h, nl = 64, 32, 2
    dev = gpu_device
    x, o = torch.zeros(1, h, h, c, device=dev), torch.zeros(1, h, h, c, device=dev)
    u = torch.zeros(16 * c * c, device=dev)
    scratch = torch.zeros(lib.lspf2f_wino_chain_scratch_bytes(nl, 1, h, c, 1), dtype=torch.uint8, device=dev)
    arr = lambda ts: (ctypes.c_void_p * nl)(*[ctypes.c_void_p(t.data_ptr()) if t is not None else None for t in ts])
    relu = (ctypes.c_int * nl)(1, 1)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    none = [None, None]
    call = lambda src, outs, mode, sb=None: lib.lspf2f_wino_chain(nl, arr(src), arr([u, u]), arr(none), arr(none), arr(none), arr(outs), relu, 1, h, c, 1, mode,
                                                                  ctypes.c_void_p(scratch.data_ptr()), scratch.numel() if sb is None else sb, st)
    assert call([x, o], [o, x], 1) != 0          # layer 1 would overwrite layer 0's input
    assert call([x, o], [o, o], 1) != 0          # two layers, one output
    assert call([x, x], [o, x], 1) != 0          # layer 1 does not read layer 0's output
    assert call([x, o], [o, torch.zeros_like(o)], 7) != 0
    assert call([x, o], [o, torch.zeros_like(o)], 1, 8) != 0      # scratch too small
    torch.cuda.synchronize()


# ---- Winograd F(4x4, 3x3) kernel (csrc/wino4.hip): the same layers with 36 multiplies per 4x4 outputs ---------------------------------------
def run_wino4(dev, x, w, scale, shift, res, relu, splits):
    import os
    import sys
    from livespeechportraits_amd import _native as N
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import wino_model
    lib = N.load()
    b, c, h, _ = x.shape
    cout = w.shape[0]
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(dev)
    d0, wp = nhwc(x), torch.from_numpy(wino_model.pack_u4(w.numpy())).to(dev)      # the numpy statement of the layout, not the library's packer
    dsc = scale.to(dev) if scale is not None else None
    dsh = shift.to(dev) if shift is not None else None
    dres = nhwc(res) if res is not None else None
    out = torch.full((b, h, h, cout), float("nan"), device=dev)
    sb = lib.lspf2f_conv3x3_scratch_bytes(b, h, h, c, 0, cout, 1, 0, 6001, 0, splits, -1, 0)
    scratch = torch.zeros(max(sb, 4), dtype=torch.uint8, device=dev)      # slabs + arrival counters (zero on entry, left zero by the kernel)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    N.check(lib.lspf2f_conv3x3(p(d0), None, p(wp), p(dsc), p(dsh), p(dres), p(out), b, h, h, c, 0, cout, 1, 0, int(relu), 6001, 0, splits, -1, 0,
                               p(scratch), scratch.numel(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    if splits > 1:
        assert not scratch[sb - 4 * (b * (h // 16) * (h // 32) * (cout // 32)):].any(), "arrival counters not left at zero"
    return out.permute(0, 3, 1, 2).contiguous().cpu()


WINO4_CASES = [
    # b, c, cout, h, splits, epilogue (scale/shift + residual + relu)
    (1, 8, 32, 32, 1, False),             # ONE K-step: the prologue's operands are all there is
    (1, 24, 32, 32, 1, True),             # three steps: the two-step unrolled loop ends on its first half
    (1, 32, 32, 32, 1, False),
    (1, 32, 64, 32, 1, True),
    (2, 64, 64, 32, 1, True),
    (3, 40, 96, 64, 1, True),             # 64 = 4 x 2 tile-blocks per frame, channel counts that are no power of two
    (1, 64, 64, 64, 2, True),
    (1, 128, 64, 32, 4, True),
    (2, 256, 32, 32, 8, True),
    (1, 104, 32, 32, 3, False),           # 13 K-steps in 3 slices of 5, 5, 3
    (1, 64, 64, 256, 1, True),            # the four shapes of the `large` plan at batch 1, with the planner's splits
    (1, 128, 128, 128, 2, True),
    (1, 256, 256, 64, 4, True),
    (1, 512, 512, 32, 8, True),
]


@pytest.mark.parametrize("cfg", WINO4_CASES, ids=lambda c: "b%d_c%d_o%d_h%d_s%d%s" % (c[:5] + ("_ep" if c[5] else "",)))
def test_conv3x3_winograd4(cfg, gpu_device):
    """wino4_3x3 against the fp64 convolution.  F(4x4, 3x3) multiplies its inputs by 4, 5 and 8 before it cancels them, so on DENSE unit-variance data
    its fp32 error is ~20x the direct form's (fp32 emulation on these very problems: 1.6e-5 .. 7.7e-5 on outputs of range 4-5; F(2x2): 0.9-3.6e-6);
    the bound is 4e-5 x range.  What the network sees is measured end to end (tools/wino4_error.py, profiles/r04_wino4x4_error.txt: 1.9e-6 on the `large`
    golden) and asserted by the golden / batch-8 tests, which run through this kernel."""
    b, c, cout, h, splits, ep = cfg
    g = torch.Generator().manual_seed(2000 + c + cout + h)
    x = torch.randn(b, c, h, h, generator=g)
    w = torch.randn(cout, c, 3, 3, generator=g) / (3.0 * c ** 0.5)
    scale = torch.rand(cout, generator=g) + 0.5 if ep else None
    shift = torch.randn(cout, generator=g) * 0.1 if ep else None
    res = torch.randn(b, cout, h, h, generator=g) if ep else None
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    if ep:
        ref = torch.relu(ref * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1) + res.double())
    got = run_wino4(gpu_device, x, w, scale, shift, res, ep, splits)
    assert torch.isfinite(got).all(), "kernel left unwritten (NaN) outputs"
    err = (got.double() - ref).abs().max().item()
    f2 = (run_wino(gpu_device, x, w, scale, shift, res, ep, 1, 1).double() - ref).abs().max().item()
    print("\nwino4 %s: max-abs %.2e (output range %.2f); F(2x2,3x3) on the same problem: %.2e" % (cfg, err, ref.abs().max().item(), f2))
    assert err <= 4e-5 * max(1.0, ref.abs().max().item()), err
    if splits > 1:       # the in-launch combine sums the slices in a fixed order, whichever workgroup arrives last
        again = run_wino4(gpu_device, x, w, scale, shift, res, ep, splits)
        assert torch.equal(got, again)


def test_conv3x3_winograd4_impulse_layout(gpu_device):
    """One-hot input and one-hot tap land on exactly one output value, for taps and pixels on tile and tile-block seams (the transforms of a
    one-hot patch are exact in fp32: small integers times one weight)."""
    c, h = 16, 64
    for (ci, y, x_, co, ky, kx) in [(5, 15, 31, 11, 0, 2), (13, 16, 32, 40, 2, 0), (0, 0, 0, 63, 1, 1), (9, 63, 63, 32, 0, 0), (3, 17, 30, 1, 2, 2)]:
        x = torch.zeros(1, c, h, h)
        x[0, ci, y, x_] = 2.0
        w = torch.zeros(64, c, 3, 3)
        w[co, ci, ky, kx] = 3.0           # out[co][oy][ox] += 3 * in[ci][oy + ky - 1][ox + kx - 1]
        exp = F.conv2d(x, w, None, 1, 1)
        got = run_wino4(gpu_device, x, w, None, None, None, False, 1)
        assert (got - exp).abs().max().item() <= 1e-5, (ci, y, x_, co, ky, kx)
        assert (got[exp == 0].abs() <= 1e-5).all()


def test_conv3x3_winograd4_rejects_unsupported_shapes(gpu_device):
    from livespeechportraits_amd import _native as N
    w = rnd(32, 32, 3, 3)
    with pytest.raises(N.Lspf2fError):
        run_wino4(gpu_device, rnd(1, 32, 16, 16), w, None, None, None, False, 1)       # 16 % 32 != 0: a tile-block is 16 x 32 pixels
    with pytest.raises(N.Lspf2fError):
        run_wino4(gpu_device, rnd(1, 32, 32, 32), rnd(48, 32, 3, 3), None, None, None, False, 1)       # cout % 32 != 0


# ---- the 16-bit row / band kernels in fp16 storage (round 3: the same kernels, templated on the storage type, serve opt.fp16 plans) ----
F16_SPECIAL = [
    # kind, args
    ("rows", (64, 1, 64, 16, True, True)),
    ("rows", (128, 2, 64, 7, False, True)),
    ("band", (2, 16, 512, True, True)),
    ("band", (3, 8, 96, True, False)),
    ("rowup", (2, 32, 8)),
]


@pytest.mark.parametrize("kind,cfg", F16_SPECIAL, ids=lambda v: v if isinstance(v, str) else "_".join(str(int(x)) for x in v))
def test_16bit_kernels_in_fp16_storage(kind, cfg, gpu_device):
    """rowconv64 / rowconv128 / bandconv512 / rowup256 with IEEE-half operands (v_mfma_f32_32x32x16_f16): against the fp64 conv of the
    fp16-rounded operands within one fp16 ulp, and (row kernels: same MFMA, same accumulation order) bit-equal to the implicit GEMM."""
    h16 = lambda t: t.half().float() if t is not None else None
    if kind == "rows":
        c, b, h, rows, res, relu = cfg
        x0 = h16(rnd(b, c, h, h, seed=71))
        w = rnd(c, c, 3, 3, seed=72) * 0.05
        scale, shift = rnd(c, seed=73) * 0.5 + 1.0, rnd(c, seed=74) * 0.1
        r = h16(rnd(b, c, h, h, seed=75)) if res else None
        got = run_conv(gpu_device, x0, None, w, scale, shift, r, 1, 0, relu, (1000 + rows, c), 0, -1, dtype=2)
        ref = ref_conv(x0, None, h16(w), scale, shift, r, 1, False, relu)
        other = run_conv(gpu_device, x0, None, w, scale, shift, r, 1, 0, relu, (64, 64), 1, 1, dtype=2)
        assert torch.equal(got, other)
    elif kind == "band":
        b, h, cout, res, relu = cfg
        x0 = h16(rnd(b, 512, h, h, seed=81))
        w = rnd(cout, 512, 3, 3, seed=82) * 0.02
        scale, shift = rnd(cout, seed=83) * 0.5 + 1.0, rnd(cout, seed=84) * 0.1
        r = h16(rnd(b, cout, h, h, seed=85)) if res else None
        got = run_conv(gpu_device, x0, None, w, scale, shift, r, 1, 0, relu, (2000, 0), 0, -1, dtype=2)
        ref = ref_conv(x0, None, h16(w), scale, shift, r, 1, False, relu)
    else:
        b, hs, rows = cfg
        x0, x1 = h16(rnd(b, 128, hs, hs, seed=91)), h16(rnd(b, 128, hs, hs, seed=92))
        w = rnd(64, 256, 3, 3, seed=93) * 0.03
        scale, shift = rnd(64, seed=94) * 0.5 + 1.0, rnd(64, seed=95) * 0.1
        got = run_conv(gpu_device, x0, x1, w, scale, shift, None, 1, 2, True, (3000 + rows, 64), 0, -1, dtype=2)
        other = run_conv(gpu_device, x0, x1, w, scale, shift, None, 1, 2, True, (128, 64), 1, 1, dtype=2)     # the igemm's sub-pixel path
        assert torch.equal(got, other)
        ref = other
    assert torch.isfinite(got).all()
    tol = (ref.abs() * 2.0 ** -11 + 3e-4)
    assert ((got - ref).abs() <= tol).all(), (got - ref).abs().max().item()


# ---- up-conv Winograd form (csrc/winoup.hip): Upsample x2 + conv3x3 over two equally wide sources, 9 multiplies per 2x2 outputs ------------
def run_winoup(dev, x0, x1, w, scale, shift, relu, nb, splits):
    import os
    import sys
    from livespeechportraits_amd import _native as N
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import wino_model
    lib = N.load()
    b, c0, hs, _ = x0.shape
    c1 = x1.shape[1] if x1 is not None else 0
    cout = w.shape[0]
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(dev)
    d0, d1 = nhwc(x0), nhwc(x1) if x1 is not None else None
    wp = torch.from_numpy(wino_model.pack_u_up(w.numpy())).to(dev)            # the numpy statement of the layout, not the library's packer
    dsc = scale.to(dev) if scale is not None else None
    dsh = shift.to(dev) if shift is not None else None
    out = torch.full((b, 2 * hs, 2 * hs, cout), float("nan"), device=dev)
    sb = lib.lspf2f_conv3x3_scratch_bytes(b, hs, hs, c0, c1, cout, 1, 1, 5000 + nb, 0, splits, -1, 0)
    scratch = torch.zeros(max(sb, 4), dtype=torch.uint8, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    N.check(lib.lspf2f_conv3x3(p(d0), p(d1), p(wp), p(dsc), p(dsh), None, p(out), b, hs, hs, c0, c1, cout, 1, 1, int(relu), 5000 + nb, 0, splits, -1, 0,
                               p(scratch), scratch.numel(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    return out.permute(0, 3, 1, 2).contiguous().cpu()


WINOUP_CASES = [
    # b, c0, c1, cout, hs, nb, splits, epilogue
    (1, 16, 16, 32, 8, 1, 1, False),
    (2, 32, 32, 64, 16, 2, 1, True),
    (1, 24, 0, 32, 8, 1, 1, True),           # one source
    (3, 40, 40, 96, 24, 1, 2, True),         # extents and channel counts that are no power of two; 10 K-steps in 2 slices
    (1, 64, 64, 64, 16, 2, 4, True),
    (1, 128, 128, 64, 128, 2, 1, True),      # the four shapes of the `large` plan at batch 1
    (1, 256, 256, 128, 64, 2, 2, True),
    (1, 512, 512, 256, 32, 2, 4, True),
    (1, 512, 512, 512, 16, 2, 8, True),
]


@pytest.mark.parametrize("cfg", WINOUP_CASES, ids=lambda c: "b%d_c%d+%d_o%d_h%d_nb%d_s%d%s" % (c[:7] + ("_ep" if c[7] else "",)))
def test_conv3x3_winograd_upconv(cfg, gpu_device):
    """winoup3x3 against the fp64 reference (nearest x2 upsample, 3x3 conv over the channel concat, folded BN, ReLU), the sub-pixel implicit GEMM on
    the same problem printed beside it."""
    b, c0, c1, cout, hs, nb, splits, ep = cfg
    g = torch.Generator().manual_seed(2000 + c0 + cout + hs)
    x0 = torch.randn(b, c0, hs, hs, generator=g)
    x1 = torch.randn(b, c1, hs, hs, generator=g) if c1 else None
    w = torch.randn(cout, c0 + c1, 3, 3, generator=g) / (3.0 * (c0 + c1) ** 0.5)
    scale = torch.rand(cout, generator=g) + 0.5 if ep else None
    shift = torch.randn(cout, generator=g) * 0.1 if ep else None
    ref = ref_conv(x0, x1, w, scale, shift, None, 1, True, ep).double()
    ref = F.conv2d(F.interpolate((x0 if x1 is None else torch.cat([x0, x1], 1)).double(), scale_factor=2, mode="nearest"), w.double(), None, 1, 1)
    if ep:
        ref = torch.relu(ref * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
    got = run_winoup(gpu_device, x0, x1, w, scale, shift, ep, nb, splits)
    assert torch.isfinite(got).all(), "kernel left unwritten (NaN) outputs"
    err = (got.double() - ref).abs().max().item()
    direct = None
    if (c0 % 32 == 0 and c1 % 32 == 0) and hs <= 32:
        direct = (run_conv(gpu_device, x0, x1, w, scale, shift, None, 1, 2, ep).double() - ref).abs().max().item()
    print("\nwinoup %s: max-abs %.2e (output range %.2f); sub-pixel implicit GEMM on the same problem: %s" % (
        cfg, err, ref.abs().max().item(), "%.2e" % direct if direct is not None else "-"))
    assert err <= 1e-5 * max(1.0, ref.abs().max().item()), err
    if splits > 1:
        assert torch.equal(got, run_winoup(gpu_device, x0, x1, w, scale, shift, ep, nb, splits))


def test_conv3x3_winograd_upconv_impulse_layout(gpu_device):
    """One-hot source pixel and one-hot tap: exactly the outputs of the upsampled convolution, on tile-block seams and borders."""
    c, hs = 8, 16
    for (ci, y, x_, co, ky, kx) in [(2, 3, 7, 5, 0, 2), (9, 4, 8, 31, 2, 0), (0, 0, 0, 0, 1, 1), (15, 15, 15, 17, 0, 0)]:
        x0, x1 = torch.zeros(1, c, hs, hs), torch.zeros(1, c, hs, hs)
        (x0 if ci < c else x1)[0, ci % c, y, x_] = 2.0
        w = torch.zeros(32, 2 * c, 3, 3)
        w[co, ci, ky, kx] = 3.0
        exp = F.conv2d(F.interpolate(torch.cat([x0, x1], 1), scale_factor=2, mode="nearest"), w, None, 1, 1)
        got = run_winoup(gpu_device, x0, x1, w, None, None, False, 1, 1)
        assert torch.equal(got, exp), (ci, y, x_, co, ky, kx)


# ---- the 16-bit full-K kernel (csrc/fullk16.hip): the 8x8 / 4x4 / 2x2 levels of the bf16 / fp16 plans from 2 frames up ---------------------------------
FULLK16_CASES = [
    # b, c0, c1, cout, hs, stride, up, res, relu, tile_m      (tile_n = 16)
    (8, 512, 0, 512, 4, 1, False, True, True, 16),        # L6 / L7.u res convs at 8 frames: a tile = one 4x4 frame
    (8, 512, 0, 512, 2, 1, False, True, True, 16),        # L7 res convs: a tile = one 2x2 frame, 4 of its 16 rows real
    (8, 512, 0, 512, 8, 2, False, False, True, 16),       # L6.down: 8 -> 4, stride 2, the whole 8x8 source is the band
    (8, 512, 0, 512, 4, 2, False, False, True, 16),       # L7.down: 4 -> 2 (no BN in the net; scale/shift still exercised here)
    (8, 512, 0, 512, 2, 1, True, False, True, 16),        # L7.up: 2 -> 4, nearest x2 in front, one source
    (8, 512, 512, 512, 4, 1, True, False, True, 16),      # L6.up: 4 -> 8 over the concat, two staged sources, 72 weight loads per lane
    (8, 512, 0, 512, 16, 2, False, False, True, 16),      # L5.down: 16 -> 8, stride 2, 5-row bands
    (8, 512, 0, 512, 8, 1, False, True, True, 16),        # the 8x8 res convs (bandconv512 in the default plan)
    (3, 256, 0, 128, 4, 1, False, True, False, 16),       # 256 channels (G = 2: the band goes through registers), 3 frames, no ReLU
    (2, 256, 256, 256, 2, 1, True, False, True, 16),      # ... two sources, upsampled
    (5, 256, 0, 128, 16, 1, False, True, True, 32),       # 16x16 in 32-pixel tiles (two pixel blocks per workgroup)
    (2, 512, 0, 128, 16, 2, False, False, True, 16),      # 32 -> 16, stride 2: 3-row bands of 32 pixels
]


@pytest.mark.parametrize("dtype", [1, 2], ids=["bf16", "f16"])
@pytest.mark.parametrize("cfg", FULLK16_CASES, ids=lambda c: "b%d_c%d+%d_o%d_h%d_s%d%s_t%d" % (c[0], c[1], c[2], c[3], c[4], c[5], "up" if c[6] else "", c[9]))
def test_conv3x3_full_k_kernel_16bit(cfg, dtype, gpu_device):
    """conv3x3_fullk16 against the fp64 convolution of the 16-bit-rounded operands: within one ulp of the 16-bit result (its K order differs from the implicit
    GEMM's); the tile-blocked weight layout the plans bind gives the same bits as the row layout; repeats are bit-identical (fixed summation order); and the
    implicit GEMM on the same problem agrees to that ulp."""
    b, c0, c1, cout, hs, stride, up, res, relu, tm = cfg
    rt = bf16r if dtype == 1 else (lambda t: t.to(torch.float16).float() if t is not None else None)
    ulp = 2.0 ** -8 if dtype == 1 else 2.0 ** -11
    x0 = rt(rnd(b, c0, hs, hs, seed=131))
    x1 = rt(rnd(b, c1, hs, hs, seed=132)) if c1 else None
    w = rnd(cout, c0 + c1, 3, 3, seed=133) * 0.02
    scale, shift = rnd(cout, seed=134) * 0.5 + 1.0, rnd(cout, seed=135) * 0.1
    ho = 2 * hs if up else hs // stride
    r = rt(rnd(b, cout, ho, ho, seed=136)) if res else None
    got = run_conv(gpu_device, x0, x1, w, scale, shift, r, stride, int(up), relu, (tm, 16), 0, 0, dtype=dtype)
    ref = ref_conv(x0, x1, rt(w), scale, shift, r, stride, up, relu)
    assert torch.isfinite(got).all()
    tol = ref.abs() * ulp + 1e-3
    assert ((got - ref).abs() <= tol).all(), (got - ref).abs().max().item()
    tiled = run_conv(gpu_device, x0, x1, w, scale, shift, r, stride, int(up), relu, (tm, 16), 0, -1, dtype=dtype)
    assert torch.equal(tiled, got)                                   # same K order in both weight layouts
    assert torch.equal(run_conv(gpu_device, x0, x1, w, scale, shift, r, stride, int(up), relu, (tm, 16), 0, -1, dtype=dtype), tiled)
    gemm = run_conv(gpu_device, x0, x1, w, scale, shift, r, stride, int(up), relu, (64, 64), 0, 0, dtype=dtype)
    assert ((got - gemm).abs() <= 2 * tol).all(), (got - gemm).abs().max().item()


def test_conv3x3_full_k_kernel_16bit_rejects_other_shapes(gpu_device):
    from livespeechportraits_amd import _native as N
    x, w = bf16r(rnd(2, 512, 8, 8, seed=1)), rnd(128, 512, 3, 3, seed=2) * 0.02
    for kw in (dict(x0=bf16r(rnd(2, 128, 8, 8, seed=1)), w=rnd(128, 128, 3, 3, seed=2)),      # 128 channels: not a multiple of the 256 a lane group spans
               dict(x0=x, w=rnd(64, 512, 3, 3, seed=2)),                                        # Cout not a multiple of 128
               dict(x0=bf16r(rnd(2, 512, 32, 32, seed=1)), w=w)):                               # 32x32: not a small level
        with pytest.raises(N.Lspf2fError, match="16-bit full-K"):
            run_conv(gpu_device, kw["x0"], None, kw["w"], None, None, None, 1, 0, True, (16, 16), 0, 0, dtype=1)
    with pytest.raises(N.Lspf2fError, match="16-bit full-K"):               # stride 2 over a concat input
        run_conv(gpu_device, x, x, rnd(128, 1024, 3, 3, seed=2), None, None, None, 2, 0, True, (16, 16), 0, 0, dtype=1)
