"""Lane-level numpy model of csrc/wino.hip (Winograd F(2x2,3x3) stride-1 conv, fp32) -- test infrastructure.

It replays, index for index, what the kernel does with its data: the LDS-DMA chunk map of the raw input patch, the per-wave
fragment reads, the input transform of ξ-row `wave`, the k pairing of v_mfma_f32_32x32x2_f32, the accumulator layout, the
epilogue patch and the output scatter.  tests/test_wino_cpu.py runs it on the weights packed by the library's own host packer
(lspf2f_pack_wino_weights) and compares with a float64 direct convolution: a wrong index anywhere shows up on the CPU, before
any GPU time is spent.  Arithmetic is float64 here (this file checks the data flow, not rounding).

Geometry (must match wino.hip):
  tile-block   = 4 x 8 Winograd tiles = 8 x 16 output pixels; raw patch 10 x 18 pixels
  MFMA row r   = tile (ty = r >> 3, tx = r & 7); lane l: row l & 31, k-quad q = l >> 5
  K-step       = 8 channels: lane quad q holds channels 8s + 4q .. +3, MFMA t (0..3) contracts channels {8s + t, 8s + 4 + t}
  raw chunk ci = ((pary*2 + parx)*2 + q)*45 + hy*9 + hx  <->  patch pixel (2hy + pary, 2hx + parx), channel quad q
  U fragments  = [n-block][xi-row i][k-step s][j][lane][4]: lane l -> n = 32*nblock + (l & 31), channels 8s + 4(l >> 5) + 0..3
"""
import numpy as np

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)

ROWS = {0: (0, 2, 1.0, -1.0), 1: (1, 2, 1.0, 1.0), 2: (1, 2, -1.0, 1.0), 3: (1, 3, 1.0, -1.0)}   # xi-row i: t = sa*d[ra] + sb*d[rb]


def pack_u(w_oihw):
    """numpy restatement of pack_wino_weights(): OIHW [N][C][3][3] -> fragment order (float64 in, float32 out)."""
    n_out, c_in = w_oihw.shape[:2]
    u = np.einsum("ia,ncab,jb->ijcn", G, w_oihw.astype(np.float64), G)          # [4][4][C][N]
    out = np.zeros((n_out // 32, 4, c_in // 8, 4, 64, 4), np.float32)
    lane = np.arange(64)
    for t in range(4):
        ch = 4 * (lane >> 5) + t                                                # channel within the k-step
        for s in range(c_in // 8):
            for nb in range(n_out // 32):
                out[nb, :, s, :, :, t] = u[:, :, 8 * s + ch, 32 * nb + (lane & 31)]
    return out.reshape(-1)


def chunk_decode(ci):
    par, rem = divmod(ci, 90)
    q, r2 = divmod(rem, 45)
    hy, hx = divmod(r2, 9)
    return par >> 1, par & 1, q, hy, hx


def chunk_index(py, px, q):
    return (((py & 1) * 2 + (px & 1)) * 2 + q) * 45 + (py >> 1) * 9 + (px >> 1)


def conv_model(x_nhwc, u_packed, n_out, scale=None, shift=None, residual=None, relu=False):
    B, H, W, C = x_nhwc.shape
    assert H % 8 == 0 and W % 16 == 0 and C % 8 == 0 and n_out % 32 == 0
    U = u_packed.reshape(n_out // 32, 4, C // 8, 4, 64, 4).astype(np.float64)
    out = np.zeros((B, H, W, n_out))
    lane = np.arange(64)
    r, q = lane & 31, lane >> 5
    ty, tx = r >> 3, r & 7
    for b in range(B):
        for by in range(H // 8):
            for bx in range(W // 16):
                Y0, X0 = 8 * by, 16 * bx
                for nblk in range(n_out // 32):
                    acc = np.zeros((4, 4, 32, 32))                                  # [i][j][row (tile)][col (n)]
                    for s in range(C // 8):
                        # ---- LDS-DMA of the raw patch: 6 pieces of 64 chunks x 4 floats
                        lds = np.zeros((384, 4))
                        for ci in range(360):
                            pary, parx, qq, hy, hx = chunk_decode(ci)
                            y, x = Y0 - 1 + 2 * hy + pary, X0 - 1 + 2 * hx + parx
                            if 0 <= y < H and 0 <= x < W:
                                lds[ci] = x_nhwc[b, y, x, 8 * s + 4 * qq: 8 * s + 4 * qq + 4]
                        for i in range(4):                                          # wave = xi-row
                            ra, rb, sa, sb = ROWS[i]
                            d = np.zeros((2, 4, 64, 4))
                            for k, dy in enumerate((ra, rb)):
                                for dx in range(4):
                                    d[k, dx] = lds[chunk_index(2 * ty + dy, 2 * tx + dx, q)]
                            t = sa * d[0] + sb * d[1]                               # [4 cols][lane][4 ch]
                            v = [t[0] - t[2], t[1] + t[2], t[2] - t[1], t[1] - t[3]]
                            for j in range(4):
                                ufrag = U[nblk, i, s, j]                            # [lane][4]
                                for tt in range(4):
                                    a_op, b_op = v[j][:, tt], ufrag[:, tt]          # one value per lane
                                    # v_mfma_f32_32x32x2_f32: D[row][col] += sum_k A[row][k] * B[k][col], lane -> (row | col = l & 31, k = l >> 5)
                                    A = np.zeros((32, 2)); Bm = np.zeros((2, 32))
                                    A[r, q] = a_op; Bm[q, r] = b_op
                                    acc[i, j] += A @ Bm
                    # ---- epilogue: in-wave column transform, cross-wave row transform through the LDS patch
                    z = np.zeros((4, 2, 32, 32))
                    for i in range(4):
                        z[i, 0] = acc[i, 0] + acc[i, 1] + acc[i, 2]
                        z[i, 1] = acc[i, 1] - acc[i, 2] - acc[i, 3]
                    for row in range(32):                                           # thread (row = tile, channel quad)
                        tty, ttx = row >> 3, row & 7
                        for bcol in range(2):
                            y0v = z[0, bcol, row] + z[1, bcol, row] + z[2, bcol, row]
                            y1v = z[1, bcol, row] - z[2, bcol, row] - z[3, bcol, row]
                            out[b, Y0 + 2 * tty, X0 + 2 * ttx + bcol, 32 * nblk: 32 * nblk + 32] = y0v
                            out[b, Y0 + 2 * tty + 1, X0 + 2 * ttx + bcol, 32 * nblk: 32 * nblk + 32] = y1v
    if scale is not None:
        out = out * scale + shift
    if residual is not None:
        out = out + residual
    if relu:
        out = np.maximum(out, 0)
    return out


# ---- Upsample(x2, nearest) + Conv3x3 as a 9-multiply Winograd form (csrc/winoup.hip) -------------------------------------------------
# The 4x4 patch of the UPSAMPLED image under an (even-aligned) 2x2 output tile has rows (a, b, b, c) = source rows (y-1, y, y+1): row 2 of
# B^T d B vanishes and rows 0, 1, 3 are a - b, 2b, b - c -- so a tile is one SOURCE pixel with its 3x3 neighbourhood, and only the 9 positions
# (i, j) in {0, 1, 3}^2 of the transformed tile are multiplied.  The factors 2 are folded into the weights.
UP_IDX = (0, 1, 3)
UP_SCALE = {0: 1.0, 1: 2.0, 3: 1.0}


def pack_u_up(w_oihw):
    """numpy restatement of pack_winoup_weights(): OIHW [N][C][3][3] -> [N/32][xi-row 3][C/8][j 3][64 lanes][4], U' = c_i c_j (G g G^T)[i][j]"""
    n_out, c_in = w_oihw.shape[:2]
    u = np.einsum("ia,ncab,jb->ijcn", G, w_oihw.astype(np.float64), G)
    out = np.zeros((n_out // 32, 3, c_in // 8, 3, 64, 4), np.float32)
    lane = np.arange(64)
    for wi, i in enumerate(UP_IDX):
        for wj, j in enumerate(UP_IDX):
            for t in range(4):
                ch = 4 * (lane >> 5) + t
                for s in range(c_in // 8):
                    for nb in range(n_out // 32):
                        out[nb, wi, s, wj, :, t] = UP_SCALE[i] * UP_SCALE[j] * u[i, j, 8 * s + ch, 32 * nb + (lane & 31)]
    return out.reshape(-1)


def upconv_model(x0, x1, u_packed, n_out):
    """lane-level model of winoup3x3: x0 / x1 NHWC sources (x1 may be None), output [B][2H][2W][n_out]"""
    B, H, W, C0 = x0.shape
    C1 = 0 if x1 is None else x1.shape[3]
    C = C0 + C1
    assert H % 4 == 0 and W % 8 == 0 and C0 % 8 == 0 and C1 % 8 == 0 and n_out % 32 == 0
    U = u_packed.reshape(n_out // 32, 3, C // 8, 3, 64, 4).astype(np.float64)
    out = np.zeros((B, 2 * H, 2 * W, n_out))
    lane = np.arange(64)
    r, q = lane & 31, lane >> 5
    ty, tx = r >> 3, r & 7
    for b in range(B):
        for by in range(H // 4):
            for bx in range(W // 8):
                Y0, X0 = 4 * by, 8 * bx
                for nblk in range(n_out // 32):
                    acc = np.zeros((3, 3, 32, 32))
                    for s in range(C // 8):
                        src, c0 = (x0, 8 * s) if 8 * s < C0 else (x1, 8 * s - C0)
                        lds = np.zeros((128, 4))                                    # chunk = (q*6 + py)*10 + px
                        for ci in range(120):
                            qq, rem = divmod(ci, 60)
                            py, px = divmod(rem, 10)
                            y, x = Y0 - 1 + py, X0 - 1 + px
                            if 0 <= y < H and 0 <= x < W:
                                lds[ci] = src[b, y, x, c0 + 4 * qq: c0 + 4 * qq + 4]
                        rd = lambda dy, dx: lds[(q * 6 + ty + dy) * 10 + tx + dx]
                        for w in range(3):                                          # wave = xi-row index UP_IDX[w]
                            if w == 0:
                                T = [rd(0, dx) - rd(1, dx) for dx in range(3)]
                            elif w == 1:
                                T = [rd(1, dx) for dx in range(3)]
                            else:
                                T = [rd(1, dx) - rd(2, dx) for dx in range(3)]
                            v = [T[0] - T[1], T[1], T[1] - T[2]]
                            for j in range(3):
                                ufrag = U[nblk, w, s, j]
                                for tt in range(4):
                                    A = np.zeros((32, 2)); Bm = np.zeros((2, 32))
                                    A[r, q] = v[j][:, tt]; Bm[q, r] = ufrag[:, tt]
                                    acc[w, j] += A @ Bm
                    z = np.zeros((3, 2, 32, 32))
                    for w in range(3):
                        z[w, 0] = acc[w, 0] + acc[w, 1]
                        z[w, 1] = acc[w, 1] - acc[w, 2]
                    for row in range(32):
                        tty, ttx = row >> 3, row & 7
                        for bcol in range(2):
                            oy, ox = 2 * (Y0 + tty), 2 * (X0 + ttx) + bcol
                            out[b, oy, ox, 32 * nblk: 32 * nblk + 32] = z[0, bcol, row] + z[1, bcol, row]
                            out[b, oy + 1, ox, 32 * nblk: 32 * nblk + 32] = z[1, bcol, row] - z[2, bcol, row]
    return out


def upconv_direct(x0, x1, w_oihw):
    x = x0 if x1 is None else np.concatenate([x0, x1], 3)
    up = x.repeat(2, axis=1).repeat(2, axis=2)
    return conv_direct(up, w_oihw)


# ---- Winograd F(4x4, 3x3) (csrc/wino4.hip) ----------------------------------------------------------------------------------------------
# Geometry (must match wino4.hip):
#   tile-block = 4 x 8 tiles of 4 x 4 outputs = 16 x 32 pixels; raw patch 18 x 34 pixels; MFMA row r = tile (ty = r >> 3, tx = r & 7)
#   wave (a, b) = (w >> 1, w & 1) owns positions rows 3a..3a+2, columns 3b..3b+2 of the 6x6 transformed tile, f = 3 i + j locally
#   raw chunk = plane_base(py & 3, px & 3) + ((py >> 2) * hxn + (px >> 2)) * 2 + q: 16 planes of hyn x hxn pixels x 2 quads, a pixel's quads adjacent
#   U = [n-block][wave][k-step][f][lane][4]
BT4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=np.float64)
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=np.float64)
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)


def w4_hyn(pary):
    return 5 if pary < 2 else 4


def w4_hxn(parx):
    return 9 if parx < 2 else 8


def w4_plane_base(pary, parx):
    return (0, 340, 680, 952)[pary] + (0, 18, 36, 52)[parx] * w4_hyn(pary)


def w4_chunk(py, px, q):
    pary, parx = py & 3, px & 3
    return w4_plane_base(pary, parx) + ((py >> 2) * w4_hxn(parx) + (px >> 2)) * 2 + q


def w4_chunk_decode(ci):
    """the kernel's prologue arithmetic: chunk id -> (py, px, q)"""
    pary = 3 if ci >= 952 else 2 if ci >= 680 else 1 if ci >= 340 else 0
    rem1 = ci - (0, 340, 680, 952)[pary]
    hyn = w4_hyn(pary)
    parx = 3 if rem1 >= 52 * hyn else 2 if rem1 >= 36 * hyn else 1 if rem1 >= 18 * hyn else 0
    rem2 = rem1 - (0, 18, 36, 52)[parx] * hyn
    hxn = w4_hxn(parx)
    r3, q = divmod(rem2, 2)
    hy, hx = divmod(r3, hxn)
    return 4 * hy + pary, 4 * hx + parx, q


def pack_u4(w_oihw):
    """numpy restatement of pack_wino4_weights(): OIHW [N][C][3][3] -> [N/32][wave 4][C/8][f 9][64 lanes][4] (float64 in, float32 out)"""
    n_out, c_in = w_oihw.shape[:2]
    u = np.einsum("ia,ncab,jb->ijcn", G4, w_oihw.astype(np.float64), G4)        # [6][6][C][N]
    out = np.zeros((n_out // 32, 4, c_in // 8, 9, 64, 4), np.float32)
    lane = np.arange(64)
    for i in range(6):
        for j in range(6):
            wave, f = (i // 3) * 2 + j // 3, (i % 3) * 3 + j % 3
            for t in range(4):
                ch = 4 * (lane >> 5) + t
                for s in range(c_in // 8):
                    for nb in range(n_out // 32):
                        out[nb, wave, s, f, :, t] = u[i, j, 8 * s + ch, 32 * nb + (lane & 31)]
    return out.reshape(-1)


def _bt3(h, w):
    """three rows of B^T on the 5-sample window w (list of arrays) = d[h .. h + 4]; the kernel's operation order"""
    if h == 0:
        p, q = w[4] - 4 * w[2], w[3] - 4 * w[1]
        return [4 * w[0] + (-5 * w[2] + w[4]), p + q, p - q]
    c, g = w[3] - w[1], w[2] - w[0]
    return [c + 2 * g, c - 2 * g, 4 * w[0] + (-5 * w[2] + w[4])]


def _at4(m):
    s1, d1, s2, d2 = m[1] + m[2], m[1] - m[2], m[3] + m[4], m[3] - m[4]
    return [m[0] + s1 + s2, d1 + 2 * d2, s1 + 4 * s2, d1 + 8 * d2 + m[5]]


def conv4_model(x_nhwc, u_packed, n_out, scale=None, shift=None, residual=None, relu=False):
    """lane-level model of wino4_3x3 (float64 arithmetic: this checks the data flow, not rounding)"""
    B, H, W, C = x_nhwc.shape
    assert H % 16 == 0 and W % 32 == 0 and C % 8 == 0 and n_out % 32 == 0
    U = u_packed.reshape(n_out // 32, 4, C // 8, 9, 64, 4).astype(np.float64)
    out = np.zeros((B, H, W, n_out))
    lane = np.arange(64)
    r, q = lane & 31, lane >> 5
    ty, tx = r >> 3, r & 7
    for b in range(B):
        for by in range(H // 16):
            for bx in range(W // 32):
                Y0, X0 = 16 * by, 32 * bx
                for nblk in range(n_out // 32):
                    acc = np.zeros((4, 9, 32, 32))                                  # [wave][f][row (tile)][col (n)]
                    for s in range(C // 8):
                        lds = np.zeros((1280, 4))                                   # 20 pieces of 64 chunks x 4 floats
                        for ci in range(1224):
                            py, px, qq = w4_chunk_decode(ci)
                            assert w4_chunk(py, px, qq) == ci
                            y, x = Y0 - 1 + py, X0 - 1 + px
                            if 0 <= y < H and 0 <= x < W:
                                lds[ci] = x_nhwc[b, y, x, 8 * s + 4 * qq: 8 * s + 4 * qq + 4]
                        for wave in range(4):
                            a, bb = wave >> 1, wave & 1
                            t = [[None] * 5 for _ in range(3)]
                            for xc in range(5):
                                d = [lds[[w4_chunk(4 * ty[l] + a + yy, 4 * tx[l] + bb + xc, q[l]) for l in range(64)]] for yy in range(5)]
                                t[0][xc], t[1][xc], t[2][xc] = _bt3(a, d)
                            v = []
                            for i in range(3):
                                v += _bt3(bb, t[i])                                  # f = 3 i + j
                            for f in range(9):
                                ufrag = U[nblk, wave, s, f]
                                for tt in range(4):
                                    Am = np.zeros((32, 2)); Bm = np.zeros((2, 32))
                                    Am[r, q] = v[f][:, tt]; Bm[q, r] = ufrag[:, tt]
                                    acc[wave, f] += Am @ Bm
                    # ---- epilogue: patch [position][tile][n], then A^T M A per (tile, channel)
                    M = np.zeros((6, 6, 32, 32))
                    for wave in range(4):
                        a, bb = wave >> 1, wave & 1
                        for f in range(9):
                            M[3 * a + f // 3, 3 * bb + f % 3] = acc[wave, f]
                    z = [_at4([M[p, c] for p in range(6)]) for c in range(6)]       # z[c][i]
                    for i in range(4):
                        yrow = _at4([z[c][i] for c in range(6)])                    # [j] -> [tile][n]
                        for j in range(4):
                            for tile in range(32):
                                out[b, Y0 + 4 * (tile >> 3) + i, X0 + 4 * (tile & 7) + j, 32 * nblk: 32 * nblk + 32] = yrow[j][tile]
    if scale is not None:
        out = out * scale + shift
    if residual is not None:
        out = out + residual
    if relu:
        out = np.maximum(out, 0)
    return out


def conv_direct(x_nhwc, w_oihw):
    B, H, W, C = x_nhwc.shape
    xp = np.zeros((B, H + 2, W + 2, C)); xp[:, 1:-1, 1:-1] = x_nhwc
    out = np.zeros((B, H, W, w_oihw.shape[0]))
    for ky in range(3):
        for kx in range(3):
            out += np.einsum("bhwc,nc->bhwn", xp[:, ky:ky + H, kx:kx + W], w_oihw[:, :, ky, kx].astype(np.float64))
    return out


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 8, 32, 16))
    w = rng.standard_normal((64, 16, 3, 3)).astype(np.float32)
    got = conv_model(x, pack_u(w), 64)
    ref = conv_direct(x, w)
    print("max abs diff", np.abs(got - ref).max(), "of", np.abs(ref).max())
    x0, x1 = rng.standard_normal((1, 4, 16, 8)), rng.standard_normal((1, 4, 16, 8))
    w2 = rng.standard_normal((32, 16, 3, 3)).astype(np.float32)
    x4 = rng.standard_normal((1, 16, 32, 8))
    w4 = rng.standard_normal((32, 8, 3, 3)).astype(np.float32)
    print("F(4x4) max abs diff", np.abs(conv4_model(x4, pack_u4(w4), 32) - conv_direct(x4, w4)).max())
    print("up-conv max abs diff", np.abs(upconv_model(x0, x1, pack_u_up(w2), 32) - upconv_direct(x0, x1, w2)).max())
