"""Lane-level numpy model of last_conv_mfma (csrc/edge_layers.hip): the staged tile with its XOR-swizzled 16-byte slots, the per-lane operand addresses of
v_mfma_f32_4x4x1_16b_f32 (16 blocks = 4 pixel groups x 4 output parities; rows = 4 pixels, columns = 3 channels + 1 pad) and the accumulator -> output map,
checked against the sub-pixel form computed directly.  Also checks that every ds_read_b128 lane group of the A and B reads touches 16 distinct 16-byte bank
slots (or identical addresses).  Run: python tools/lastconv_model.py"""
import numpy as np

TR, TC, WP = 8, 32, 40          # tile rows / cols (source pixels), staged row pitch in pixels
GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS += [[l + 32 for l in g] for g in GROUPS]


def stage(src, b, y0, x0, half):
    """LDS image of one stage: src [B][Hs][Ws][64] -> bytes as float array [10*WP][8 slots][4]"""
    Hs, Ws = src.shape[1:3]
    lds = np.zeros((10 * WP, 8, 4), np.float32)
    for rr in range(10):
        for cc in range(34):
            y, x = y0 - 1 + rr, x0 - 1 + cc
            if 0 <= y < Hs and 0 <= x < Ws:
                p = rr * WP + cc
                for slot in range(8):
                    q = slot ^ ((p >> 1) & 7)
                    lds[p, slot] = src[b, y, x, half * 32 + 4 * q: half * 32 + 4 * q + 4]
    return lds


def a_addr(lane, wave, tau, t, q):
    g, par, m = lane >> 4, (lane >> 2) & 3, lane & 3
    py, px = par >> 1, par & 1
    R, Cb = 2 * wave + (tau >> 1), 16 * (tau & 1)
    p = (R + (t >> 1) + py) * WP + Cb + 4 * g + m + (t & 1) + px
    return p, q ^ ((p >> 1) & 7)


def check_banks():
    for wave in range(4):
        for tau in range(4):
            for t in range(4):
                for q in range(8):
                    for grp in GROUPS:
                        seen = {}
                        for lane in grp:
                            p, s = a_addr(lane, wave, tau, t, q)
                            byte = p * 128 + s * 16
                            bank = (byte // 16) % 16
                            assert seen.setdefault(bank, byte) == byte, ("A bank conflict", wave, tau, t, q, lane)
    for t in range(4):
        for q in range(32):
            for grp in GROUPS:
                seen = {}
                for lane in grp:
                    par, n = (lane >> 2) & 3, lane & 3
                    byte = ((par * 4 + n) * 516 + t * 128 + 4 * q) * 4
                    bank = (byte // 16) % 16
                    assert seen.setdefault(bank, byte) == byte, ("B bank conflict", t, q, lane)
    print("A and B reads: every ds_read_b128 lane group on 16 distinct bank slots")


def run(B=1, Hs=16, Ws=64, cout=3, seed=0):
    rng = np.random.default_rng(seed)
    s0 = rng.standard_normal((B, Hs, Ws, 64)).astype(np.float32)
    s1 = rng.standard_normal((B, Hs, Ws, 64)).astype(np.float32)
    w = rng.standard_normal((4, cout, 4, 128)).astype(np.float32) * 0.1        # [par][co][t][cin]
    wl = np.zeros((16, 516), np.float32)                                        # LDS copy: row par*4+n, [t*128 + c]; rows n >= cout stay zero
    for par in range(4):
        for n in range(cout):
            wl[par * 4 + n, :512] = w[par, n].reshape(512)
    out = np.zeros((B, cout, 2 * Hs, 2 * Ws), np.float64)
    for b in range(B):
        for y0 in range(0, Hs, TR):
            for x0 in range(0, Ws, TC):
                acc = np.zeros((4, 4, 64, 4), np.float64)                       # [wave][tau][lane][reg]
                for st in range(4):
                    src, half = (s0, s1)[st >> 1], st & 1
                    lds = stage(src, b, y0, x0, half)
                    cst = (st >> 1) * 64 + half * 32
                    for wave in range(4):
                        for tau in range(4):
                            for t in range(4):
                                for q in range(8):
                                    A = np.zeros((64, 4)); Bv = np.zeros((64, 4))
                                    for lane in range(64):
                                        p, s = a_addr(lane, wave, tau, t, q)
                                        A[lane] = lds[p, s]
                                        par, n = (lane >> 2) & 3, lane & 3
                                        Bv[lane] = wl[par * 4 + n, t * 128 + cst + 4 * q: t * 128 + cst + 4 * q + 4]
                                    for ch in range(4):                          # one v_mfma_f32_4x4x1_16b per channel: D[blk][i][j] += A[4 blk + i] * B[4 blk + j]
                                        for lane in range(64):
                                            blk = lane >> 2
                                            for r in range(4):
                                                acc[wave, tau, lane, r] += A[4 * blk + r, ch] * Bv[lane, ch]
                for wave in range(4):
                    for tau in range(4):
                        for lane in range(64):
                            g, par, n = lane >> 4, (lane >> 2) & 3, lane & 3
                            if n >= cout:
                                continue
                            R, Cb = 2 * wave + (tau >> 1), 16 * (tau & 1)
                            for r in range(4):
                                Y, X = 2 * (y0 + R) + (par >> 1), 2 * (x0 + Cb + 4 * g + r) + (par & 1)
                                out[b, n, Y, X] = acc[wave, tau, lane, r]
    # direct sub-pixel form
    ref = np.zeros_like(out)
    cat = np.concatenate([s0, s1], -1).astype(np.float64)
    pad = np.pad(cat, ((0, 0), (1, 1), (1, 1), (0, 0)))
    for par in range(4):
        py, px = par >> 1, par & 1
        for t in range(4):
            ty, tx = t >> 1, t & 1
            sl = pad[:, ty + py: ty + py + Hs, tx + px: tx + px + Ws, :]        # source pixel (y + ty - 1 + py, x + tx - 1 + px)
            ref[:, :, py::2, px::2] += np.einsum("byxc,nc->bnyx", sl, w[par, :, t, :].astype(np.float64))
    err = np.abs(out - ref).max()
    print("model vs direct sub-pixel form: max abs", err)
    assert err < 1e-9


if __name__ == "__main__":
    check_banks()
    run()

