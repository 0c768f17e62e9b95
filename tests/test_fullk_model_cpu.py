"""CPU: the addressing of conv3x3_fullk (csrc/fullk.hip) restated in numpy -- which source rows a tile stages (its band), where a lane finds the source
pixel behind (output pixel, tap), and that padding taps land on the zero pixel -- for stride 1, the 9-tap nearest x2 upsample and (round 3) stride 2.
The restatement follows the kernel line by line (band rows, `rowoff` / `coloff`, `aoff`); it is checked against torch's convolution."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F


def fullk_addressing_conv(x, w, stride, up, pb):
    """x [Hs][Ws] (one channel), w [3][3] -> output [Ho][Wo] through the kernel's band + offset arithmetic"""
    Hs, Ws = x.shape
    Ho = 2 * Hs if up else Hs // stride
    Wo = Ho
    wo_log2 = int(np.log2(Wo))
    rpb = 16 >> wo_log2
    nr = pb * rpb
    tiles = (Ho + nr - 1) // nr
    out = np.zeros((Ho, Wo), np.float64)
    for mt in range(tiles):
        r0 = mt * nr
        if up:
            u0 = max(r0 - 1, 0); u1 = min(r0 + nr, 2 * Hs - 1)
            sy0, sy1 = u0 >> 1, (u1 >> 1) + 1
        else:
            S = 2 if stride == 2 else 1
            lo, hi = S * r0 - 1, S * (r0 + nr - 1) + 2
            sy0, sy1 = max(lo, 0), min(hi, Hs)
        want = nr // 2 + 2 if up else (2 if stride == 2 else 1) * (nr - 1) + 3
        assert sy1 - sy0 <= min(want, Hs)                      # what the launcher sizes the LDS band for
        npix = (sy1 - sy0) * Ws
        band = np.concatenate([x[sy0:sy1].reshape(-1), [0.0]])   # the zero pixel sits at index npix
        hl, wl = (2 * Hs, 2 * Ws) if up else (Hs, Ws)
        S = 2 if (not up and stride == 2) else 1
        for pl in range(16 * pb):
            oy, ox = r0 + (pl >> wo_log2), pl & (Wo - 1)
            if oy >= Ho:
                continue
            acc = 0.0
            for t in range(9):
                d0, d1 = t // 3, t % 3
                uy, ux = S * oy + d0 - 1, S * ox + d1 - 1
                rowoff = (((uy >> 1) if up else uy) - sy0) * Ws if (0 <= uy < hl) else -1
                coloff = ((ux >> 1) if up else ux) if (0 <= ux < wl) else -1
                a = npix if (rowoff < 0 or coloff < 0) else rowoff + coloff
                assert 0 <= a <= npix
                acc += band[a] * w[d0, d1]
            out[oy, ox] = acc
    return out


@pytest.mark.parametrize("hs,stride,up,pb", [(16, 1, False, 2), (16, 1, False, 1), (8, 1, False, 1), (4, 1, True, 1), (8, 1, True, 2), (2, 1, False, 1),
                                             (32, 2, False, 1), (16, 2, False, 1), (8, 2, False, 1), (4, 2, False, 1)])
def test_band_and_tap_offsets_equal_the_convolution(hs, stride, up, pb):
    rng = np.random.default_rng(hs * 10 + stride + 3 * up)
    x, w = rng.standard_normal((hs, hs)), rng.standard_normal((3, 3))
    xt = torch.from_numpy(x)[None, None]
    if up:
        xt = F.interpolate(xt, scale_factor=2, mode="nearest")
    ref = F.conv2d(xt, torch.from_numpy(w)[None, None], None, stride, 1)[0, 0].numpy()
    got = fullk_addressing_conv(x, w, stride, up, pb)
    assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-12
