"""The reference's `size == 'small'` generator.  Goldens: the reference's own
Feature2FaceGenerator_Unet (oracle/make_golden_unet.py)."""
import argparse
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
TOL = 1e-4           # fp32, 16 layers; measured values are printed


def load_case(name):
    from livespeechportraits_amd import synth
    meta = json.load(open(os.path.join(GOLD, "unet_%s.json" % name)))
    ref = np.load(os.path.join(GOLD, "unet_%s.npz" % name))["out"]
    sd = synth.make_unet_small_state_dict(23, 3, meta["num_downs"], meta["ngf"], seed=meta["weights_seed"])
    n = meta["batch"] * 23 * meta["size"] ** 2
    x = synth.symmetric(n, 0.6, synth._stream(5, name)).reshape(meta["batch"], 23, meta["size"], meta["size"])
    return meta, sd, x, ref


# ---- CPU ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["small_s64_b2", "small_512"])
def test_oracle_reproduces_reference(name):
    from oracle import unet_small_oracle
    meta, sd, x, ref = load_case(name)
    out = unet_small_oracle.generator_forward({k: torch.from_numpy(v) for k, v in sd.items()}, torch.from_numpy(x), meta["num_downs"]).numpy()
    assert out.shape == ref.shape and np.abs(out - ref).max() <= 2e-6
    assert (np.abs(ref) > 0.99).mean() < 0.01 and ref.std() > 0.1            # not saturated, not degenerate


def test_container_keys_and_weight_mappings():
    from livespeechportraits_amd.unet_small import Feature2FaceGenerator_Unet, pack_down, pack_last, pack_up
    meta = json.load(open(os.path.join(GOLD, "unet_small_512.json")))
    assert {k: list(v.shape) for k, v in Feature2FaceGenerator_Unet(23, 3, 8, 64).state_dict().items()} == meta["keys"]
    g = torch.Generator().manual_seed(0)
    x, w = torch.randn(2, 5, 8, 8, generator=g), torch.randn(7, 5, 4, 4, generator=g)
    y = torch.zeros(2, 24, 4, 4)                                              # space-to-depth, 4 padding channels
    for dy in range(2):
        for dx in range(2):
            y[:, (dy * 2 + dx) * 5:(dy * 2 + dx + 1) * 5] = x[:, :, dy::2, dx::2]
    w3 = torch.from_numpy(pack_down(w.numpy(), 24)).permute(0, 3, 1, 2)
    assert (F.conv2d(y, w3, None, 1, 1) - F.conv2d(x, w, None, 2, 1)).abs().max() < 1e-5      # Conv2d(k4,s2,p1)
    xt, wt = torch.randn(2, 6, 4, 4, generator=g), torch.randn(6, 3, 4, 4, generator=g)
    ref = F.conv_transpose2d(xt, wt, None, 2, 1)
    sub = torch.from_numpy(pack_up(wt.numpy()))
    xp, out = F.pad(xt, (1, 1, 1, 1)), torch.zeros_like(ref)
    for py in range(2):
        for px in range(2):
            out[:, :, py::2, px::2] = F.conv2d(xp[:, :, py:py + 5, px:px + 5], sub[py * 2 + px].permute(0, 3, 1, 2))
    assert (out - ref).abs().max() < 1e-5                                      # ConvTranspose2d(k4,s2,p1) == sub-pixel form
    gq = F.conv2d(xt, torch.from_numpy(pack_last(wt.numpy())).permute(0, 3, 1, 2), None, 1, 1)
    out2 = torch.zeros_like(ref)
    for par in range(4):
        out2[:, :, par // 2::2, par % 2::2] = gq[:, par * 3:(par + 1) * 3]
    assert (out2 - ref).abs().max() < 1e-5                                     # GEMM form + pixel shuffle


def test_live_tap_packing_is_the_dense_form_without_its_zero_blocks():
    """lspf2f_conv3x3(k_group = -4): of the 36 (tap, channel quarter) blocks of the 3x3 / space-to-depth form of Conv2d(k4, s2, p1) only 16 carry weights.  The K
    cursor of the kernel walks taps 0..8 and, inside a tap, the quarters whose bit is set in the mask api.cpp builds (tap row 0 -> sub-row 1, row 1 -> both, row 2 ->
    sub-row 0; columns alike); pack_down_live must hold exactly the non-zero blocks of pack_down in that order, and the rest of pack_down must be zero."""
    from livespeechportraits_amd.unet_small import pack_down, pack_down_live
    rng = np.random.default_rng(3)
    co, ci = 6, 32
    w = rng.standard_normal((co, ci, 4, 4)).astype(np.float32)
    dense, live = pack_down(w, 4 * ci), pack_down_live(w)
    sub = {0: (1,), 1: (0, 1), 2: (0,)}                      # the rule of api.cpp (static const unsigned sub[3] = {2, 3, 1} as bit sets over dy)
    k = 0
    for tap in range(9):
        for q in range(4):
            blk = dense[:, tap // 3, tap % 3, q * ci:(q + 1) * ci]
            if (q >> 1) in sub[tap // 3] and (q & 1) in sub[tap % 3]:
                assert np.array_equal(live[:, k, :], blk) and np.abs(blk).min() > 0
                k += 1
            else:
                assert not blk.any()
    assert k == 16 and live.shape == (co, 16, ci)


@pytest.mark.parametrize("ci,bke", [(32, 32), (64, 32), (128, 32), (512, 32)])
def test_masked_k_cursor_walks_exactly_the_live_tiles(ci, bke):
    """A restatement of the K cursor igemm3x3<..., KM = true> runs (csrc/igemm.hip: skip_dead, the walk to a split-K slice's first tile, the advance per K-tile) with
    the mask api.cpp builds for k_group = -4: the tiles it visits, in order, are the K-tiles pack_down_live stores, they cover every non-zero weight of the dense form
    once, and any split of the range into slices visits the same sequence."""
    from livespeechportraits_amd.unet_small import pack_down, pack_down_live
    cin, kblk = 4 * ci, ci
    sub = (2, 3, 1)
    kmask = 0
    for ty in range(3):
        for tx in range(3):
            for dy in range(2):
                for dx in range(2):
                    if (sub[ty] >> dy) & 1 and (sub[tx] >> dx) & 1:
                        kmask |= 1 << ((ty * 3 + tx) * 4 + dy * 2 + dx)
    total = 16 * kblk // bke

    def walk(kt_begin, kt_end):
        st = {"tap": 0, "c": 0}

        def skip_dead():
            while st["tap"] < 9 and not (kmask >> (st["tap"] * 4 + st["c"] // kblk)) & 1:
                st["c"] += kblk
                if st["c"] >= cin:
                    st["c"] = 0; st["tap"] += 1

        def advance():
            st["c"] += bke
            if st["c"] == cin:
                st["c"] = 0; st["tap"] += 1
            if st["c"] % kblk == 0:
                skip_dead()
        skip_dead()
        for _ in range(kt_begin):
            advance()
        out = []
        for _ in range(kt_begin, kt_end):
            out.append((st["tap"], st["c"]))
            advance()
        return out
    full = walk(0, total)
    assert len(full) == total and all(t < 9 for t, _ in full)
    rng = np.random.default_rng(ci)
    w = rng.standard_normal((3, ci, 4, 4)).astype(np.float32)
    dense, live = pack_down(w, cin).reshape(3, 9, cin), pack_down_live(w).reshape(3, -1)
    covered = np.zeros((9, cin), bool)
    for kt, (tap, c) in enumerate(full):
        assert np.array_equal(live[:, kt * bke:(kt + 1) * bke], dense[:, tap, c:c + bke]), (kt, tap, c)
        assert not covered[tap, c:c + bke].any()
        covered[tap, c:c + bke] = True
    assert np.array_equal(covered, np.abs(dense).sum(0) > 0)
    for splits in (2, 3, 5, 7, 16):
        per = (total + splits - 1) // splits
        seq = []
        for z in range(splits):
            seq += walk(min(z * per, total), min((z + 1) * per, total))
        assert seq == full, splits


# ---- the native plan (include/lspunet.h, csrc/unet.hip): host side, no device ---------------------------------
def _unet_handle(input_nc=23, feat_nc=None, output_nc=3, ngf=64, num_downs=8, size=512, max_batch=2, tune=None, flags=0, dtype=0):
    import ctypes
    from livespeechportraits_amd import _native as N
    lib = N.load()
    cfg = N.UnetConfig(N.UNET_ABI_VERSION, input_nc, input_nc if feat_nc is None else feat_nc, output_nc, ngf, num_downs, size, max_batch, dtype, flags)
    h = ctypes.c_void_p()
    rc = lib.lspunet_create(ctypes.byref(cfg), tune.encode() if tune else None, ctypes.byref(h))
    return lib, h, rc


def test_lspunet_header_binding_and_library_agree():
    import re
    from livespeechportraits_amd import _native as N
    lib = N.load()
    text = open(os.path.join(ROOT, "include", "lspunet.h")).read()
    declared = set(re.findall(r"\b(lspunet_[a-z0-9_]+)\s*\(", text))
    assert declared == set(N.UNET_SIGNATURES), declared ^ set(N.UNET_SIGNATURES)
    assert all(hasattr(lib, n) for n in declared)
    assert lib.lspunet_abi_version() == N.UNET_ABI_VERSION


def test_native_plan_expects_the_reference_keys_and_packs_like_the_numpy_packers():
    """lspunet_pack_weights against the numpy restatements this file checks against torch's own convolutions (pack_down / pack_down_live / pack_up / pack_last)
    and a float64 BatchNorm fold; the expected tensors are exactly the reference module's state-dict entries (golden `keys`)."""
    import ctypes
    from livespeechportraits_amd import synth, _native as N
    from livespeechportraits_amd.unet_small import _fold_bn, block_keys, pack_down, pack_down_live, pack_last, pack_up
    nd, ngf, inc = 6, 32, 23
    lib, h, rc = _unet_handle(inc, None, 3, ngf, nd, 128, 2)
    assert rc == 0, lib.lspunet_last_error()
    sd = synth.make_unet_small_state_dict(inc, 3, nd, ngf, seed=11)
    name, dims, ndim = ctypes.c_char_p(), (ctypes.c_int64 * 4)(), ctypes.c_int()
    keys = {}
    for i in range(lib.lspunet_num_tensors(h)):
        assert lib.lspunet_tensor_info(h, i, ctypes.byref(name), ctypes.byref(dims), ctypes.byref(ndim)) == 0
        keys[name.value.decode()] = [dims[j] for j in range(ndim.value)]
    assert keys == {k: list(v.shape) for k, v in sd.items()}
    meta = json.load(open(os.path.join(GOLD, "unet_small_512.json")))
    lib8, h8, _ = _unet_handle()
    k8 = {}
    for i in range(lib8.lspunet_num_tensors(h8)):
        lib8.lspunet_tensor_info(h8, i, ctypes.byref(name), ctypes.byref(dims), ctypes.byref(ndim))
        k8[name.value.decode()] = [dims[j] for j in range(ndim.value)]
    assert k8 == {k: v for k, v in meta["keys"].items() if not k.endswith("num_batches_tracked")}       # the reference module's own keys and shapes
    lib8.lspunet_destroy(h8)
    nb = lib.lspunet_packed_bytes(h)
    blob = np.zeros(nb, np.uint8)
    assert lib.lspunet_pack_weights(h, blob.ctypes.data_as(ctypes.c_void_p), nb) == -3                 # nothing supplied yet: MISSING_TENSOR
    assert b"missing" in lib.lspunet_last_error()
    assert lib.lspunet_set_tensor(h, b"model.model.9.weight", sd["model.model.0.weight"].ctypes.data_as(ctypes.c_void_p), 1) == -1      # unknown key
    assert lib.lspunet_set_tensor(h, b"model.model.0.weight", sd["model.model.0.weight"].ctypes.data_as(ctypes.c_void_p), 7) == -4      # wrong size
    for k, v in sd.items():
        a = np.ascontiguousarray(v, np.float32)
        assert lib.lspunet_set_tensor(h, k.encode(), a.ctypes.data_as(ctypes.c_void_p), a.size) == 0, k
    assert lib.lspunet_pack_weights(h, blob.ctypes.data_as(ctypes.c_void_p), nb - 1) == -1             # arena too small
    assert lib.lspunet_pack_weights(h, blob.ctypes.data_as(ctypes.c_void_p), nb) == 0
    f = blob.view(np.float32)
    off = 0

    def take(n):
        nonlocal off
        a = f[off // 4: off // 4 + n].copy()
        off += (4 * n + 255) // 256 * 256
        return a
    chans = [ngf * min(2 ** i, 8) for i in range(nd)]
    for k, (dc, dbn, uc, ubn) in enumerate(block_keys(nd, "model")):
        cin, cout = (inc if k == 0 else chans[k - 1]), chans[k]
        w = sd[dc + ".weight"]
        if k == 0:                                                             # block 0: quarters padded to a K-tile (23 -> 32 channels), live pairs only
            wp = np.zeros((w.shape[0], 32, 4, 4), np.float32)
            wp[:, :inc] = w
            w = wp
        want = pack_down_live(w)
        assert np.array_equal(take(want.size), want.ravel()), "down %d" % k
        if dbn:
            sc, sh = _fold_bn(sd, dbn)
            assert np.array_equal(take(cout), sc) and np.array_equal(take(cout), sh)
        wt = sd[uc + ".weight"]
        want = pack_last(wt) if k == 0 else pack_up(wt)
        assert np.array_equal(take(want.size), want.ravel()), "up %d" % k
        if k == 0:
            assert np.array_equal(take(12), np.ones(12, np.float32)) and np.array_equal(take(12), np.tile(sd[uc + ".bias"], 4))
            sub = pack_up(wt)                                                  # second form: sub-pixel rows + the bias, for the direct last-layer kernel
            assert np.array_equal(take(sub.size), sub.ravel()) and np.array_equal(take(3), sd[uc + ".bias"])
        else:
            sc, sh = _fold_bn(sd, ubn)
            assert np.array_equal(take(wt.shape[1]), sc) and np.array_equal(take(wt.shape[1]), sh)
    assert off == nb
    # a second pack of the same handle needs the tensors again
    assert lib.lspunet_pack_weights(h, blob.ctypes.data_as(ctypes.c_void_p), nb) == -3
    lib.lspunet_destroy(h)


def test_native_fp16_plan_packs_the_rounded_fp32_rows():
    """dtype 2 (the reference's opt.fp16): conv weights in IEEE binary16, round to nearest even, of exactly the fp32 rows (block 0 padded to 128 channels: a K-tile is 64
    of them); BatchNorm scale / shift and the last layer's sub-pixel weights stay fp32; the plan keeps the masked-K implicit GEMM on every level and needs half the arena."""
    import ctypes
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.unet_small import _fold_bn, block_keys, pack_down, pack_down_live, pack_last, pack_up
    nd, ngf, inc = 5, 64, 23
    sd = synth.make_unet_small_state_dict(inc, 3, nd, ngf, seed=12)
    lib, h, rc = _unet_handle(inc, None, 3, ngf, nd, 64, 2, dtype=2)
    assert rc == 0, lib.lspunet_last_error()
    lib32, h32, _ = _unet_handle(inc, None, 3, ngf, nd, 64, 2)
    assert lib.lspunet_workspace_bytes(h, 2) < lib32.lspunet_workspace_bytes(h32, 2)
    nb = lib.lspunet_packed_bytes(h)
    assert nb < 0.56 * lib32.lspunet_packed_bytes(h32)
    lib32.lspunet_destroy(h32)
    for k, v in sd.items():
        a = np.ascontiguousarray(v, np.float32)
        assert lib.lspunet_set_tensor(h, k.encode(), a.ctypes.data_as(ctypes.c_void_p), a.size) == 0, k
    blob = np.zeros(nb, np.uint8)
    assert lib.lspunet_pack_weights(h, blob.ctypes.data_as(ctypes.c_void_p), nb) == 0
    off = 0

    def take(n, dt):
        nonlocal off
        a = blob[off: off + n * np.dtype(dt).itemsize].view(dt).copy()
        off += (n * np.dtype(dt).itemsize + 255) // 256 * 256
        return a
    chans = [ngf * min(2 ** i, 8) for i in range(nd)]
    for k, (dc, dbn, uc, ubn) in enumerate(block_keys(nd, "model")):
        w = sd[dc + ".weight"]
        want = pack_down(w, 128) if k == 0 else pack_down_live(w)
        assert np.array_equal(take(want.size, np.float16), want.ravel().astype(np.float16)), "down %d" % k
        if dbn:
            sc, sh = _fold_bn(sd, dbn)
            assert np.array_equal(take(chans[k], np.float32), sc) and np.array_equal(take(chans[k], np.float32), sh)
        wt = sd[uc + ".weight"]
        if k == 0:
            assert np.array_equal(take(12 * 9 * 128, np.float32), pack_last(wt).ravel())
            take(12, np.float32); take(12, np.float32)
            assert np.array_equal(take(pack_up(wt).size, np.float32), pack_up(wt).ravel()) and np.array_equal(take(3, np.float32), sd[uc + ".bias"])
        else:
            assert np.array_equal(take(pack_up(wt).size, np.float16), pack_up(wt).ravel().astype(np.float16)), "up %d" % k
            sc, sh = _fold_bn(sd, ubn)
            assert np.array_equal(take(wt.shape[1], np.float32), sc) and np.array_equal(take(wt.shape[1], np.float32), sh)
    assert off == nb
    name, kern = ctypes.c_char_p(), ctypes.c_char_p()
    kerns = []
    for i in range(lib.lspunet_num_launches(h, 1)):
        lib.lspunet_launch_info(h, 1, i, ctypes.byref(name), ctypes.byref(kern), None, None, None)
        kerns.append(kern.value.decode())
    assert not any("unet_tiny" in k or "unet_prepare" in k for k in kerns) and sum("<km>" in k for k in kerns) == nd
    assert any(k.startswith("last_conv") for k in kerns)                      # 32 low-res columns: not a multiple of the row kernel's 64-pixel strips
    lib.lspunet_destroy(h)
    lib, h, rc = _unet_handle(dtype=2)                                        # 512 x 512, ngf 64: the last layer on rowlast128 + the shuffle pass (which adds the bias)
    assert rc == 0
    kerns = []
    for i in range(lib.lspunet_num_launches(h, 1)):
        lib.lspunet_launch_info(h, 1, i, ctypes.byref(name), ctypes.byref(kern), None, None, None)
        kerns.append(kern.value.decode())
    assert kerns[-2].startswith("rowlast128") and kerns[-1].startswith("pixel_shuffle_tanh")
    lib.lspunet_destroy(h)


def test_native_plan_launch_list_workspace_and_errors():
    import ctypes
    lib, h, rc = _unet_handle(max_batch=8)
    assert rc == 0
    name, kern, tm, tn, sk = ctypes.c_char_p(), ctypes.c_char_p(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    for b in (1, 8):
        n = lib.lspunet_num_launches(h, b)
        assert n == 1 + 8 + 7 + 1                                            # input pass, 8 down-convs, 7 transposed convs, the direct last layer (+ tanh)
        rows = []
        for i in range(n):
            assert lib.lspunet_launch_info(h, b, i, ctypes.byref(name), ctypes.byref(kern), ctypes.byref(tm), ctypes.byref(tn), ctypes.byref(sk)) == 0
            rows.append((name.value.decode(), kern.value.decode(), tm.value, tn.value, sk.value))
        assert [r[0] for r in rows] == ["input"] + ["L%d.down" % k for k in range(8)] + ["L%d.up" % k for k in range(7, -1, -1)]
        assert all("lrelu s2d + relu" in r[1] for r in rows[1:8]) and "lrelu" not in rows[8][1]     # every down-conv but the innermost writes the two activated copies
        tiny = [r[0] for r in rows if r[1].startswith("unet_tiny")]
        assert tiny == (["L6.down", "L7.down", "L7.up", "L6.up"] if b == 1 else [])                # <= 16 positions: the weight-streaming kernel
        assert all("<km>" in r[1] for r in rows[1:9] if r[0] not in tiny)
    w1, w8 = lib.lspunet_workspace_bytes(h, 1), lib.lspunet_workspace_bytes(h, 8)
    assert 0 < w1 < w8 < 2 << 30
    assert lib.lspunet_num_launches(h, 9) < 0                                 # beyond max_batch
    lib.lspunet_destroy(h)
    lib, h, rc = _unet_handle(max_batch=1, tune="fused_prepare=0,input_pass=0,graph=0,last_direct=0")
    assert rc == 0 and lib.lspunet_num_launches(h, 1) == 1 + 8 + 7 + 7 + 2   # + one unet_prepare per level but the innermost; GEMM-form last layer + pixel shuffle
    lib.lspunet_destroy(h)
    for bad in (dict(tune="nonsense=1"), dict(tune="graph"), dict(tune="last_tile=7007"), dict(ngf=48), dict(num_downs=4), dict(size=384, num_downs=8),
                dict(output_nc=5), dict(input_nc=64), dict(feat_nc=0), dict(max_batch=0), dict(dtype=1), dict(dtype=3), dict(dtype=2, ngf=32)):
        lib, h, rc = _unet_handle(**bad)
        assert rc < 0 and lib.lspunet_last_error(), bad
    # fp16 plans: the row kernel of the last layer (rowlast128) writes 12 columns per pixel -- output_nc 3 only; any other width keeps the implicit-GEMM form
    # (output_nc 1 / 2 used to be routed to it and wrote 48-byte records into 16 / 32-byte ones)
    for onc, want in ((3, True), (1, False), (2, False), (4, False)):
        lib, h, rc = _unet_handle(output_nc=onc, dtype=2, max_batch=2)
        assert rc == 0
        kerns = []
        for i in range(lib.lspunet_num_launches(h, 1)):
            assert lib.lspunet_launch_info(h, 1, i, ctypes.byref(name), ctypes.byref(kern), ctypes.byref(tm), ctypes.byref(tn), ctypes.byref(sk)) == 0
            kerns.append(kern.value.decode())
        assert any(k.startswith("rowlast128") for k in kerns) == want, (onc, kerns[-3:])
        lib.lspunet_destroy(h)
    import ctypes as C
    from livespeechportraits_amd import _native as N
    cfg = N.UnetConfig(99, 23, 23, 3, 64, 8, 512, 1, 0, 0)
    h = C.c_void_p()
    assert lib.lspunet_create(C.byref(cfg), None, C.byref(h)) == -1           # ABI version mismatch



# ---- GPU ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_live_tap_down_convs_agree_with_the_dense_form(gpu_device):
    """the down-convs with only their 16 live K blocks (default) against the dense 3x3 / space-to-depth form (live_taps=False): the same products in the same order,
    minus exact zeros -- equal up to the split-K boundaries the shorter K moves; both on the reference golden"""
    from livespeechportraits_amd.unet_small import HostSequencedUnetEngine
    meta, sd, x, ref = load_case("small_512")
    outs = []
    for live in (True, False):
        e = HostSequencedUnetEngine(23, 3, meta["num_downs"], meta["ngf"], live_taps=live)
        e.load_state_dict(sd, "model", gpu_device)
        assert any(l["down_live"] for l in e.layers) == live
        outs.append(e.forward(torch.from_numpy(x).to(gpu_device)).cpu().numpy())
        assert np.abs(outs[-1] - ref).max() <= TOL
    print("\nlive vs dense down-convs: max-abs %.2e" % np.abs(outs[0] - outs[1]).max())
    assert np.abs(outs[0] - outs[1]).max() <= 5e-6


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["small_s64_b2", "small_512"])
def test_engine_matches_reference(name):
    from livespeechportraits_amd.unet_small import SmallUnetEngine
    meta, sd, x, ref = load_case(name)
    dev = torch.device("cuda:0")
    e = SmallUnetEngine(23, 3, meta["num_downs"], meta["ngf"])
    e.load_state_dict(sd, "model", dev)
    xd = torch.from_numpy(x).to(dev)
    out = e.forward(xd)
    torch.cuda.synchronize()
    err = np.abs(out.cpu().numpy() - ref).max()
    print("\n[unet %s] max-abs vs reference %.3e (|ref| max %.2f)" % (name, err, np.abs(ref).max()))
    assert out.shape == ref.shape and err <= TOL
    assert torch.equal(e.forward(xd), out)                                     # deterministic
    u8 = e.forward(xd, out_u8=True).cpu().numpy().astype(np.int32)
    want = np.clip((ref.transpose(0, 2, 3, 1) + 1.0) / 2.0 * 255.0, 0, 255).astype(np.uint8).astype(np.int32)
    assert u8.shape == want.shape and np.abs(u8 - want).max() <= 1            # util.tensor2im


@pytest.mark.gpu
def test_drop_in_model_with_size_small(tmp_path):
    """create_model(opt) with opt.size == 'small' -> setup() from a 'module.'-prefixed checkpoint -> inference()."""
    from livespeechportraits_amd.models import create_model
    meta, sd, x, ref = load_case("small_s64_b2")
    ckpt = os.path.join(tmp_path, "Feature2Face.pkl")
    torch.save({"module.netG." + k: torch.from_numpy(v) for k, v in sd.items()}, ckpt)
    opt = argparse.Namespace(model="feature2face", gpu_ids=[0], isTrain=False, size="small", ngf=meta["ngf"], n_downsample_G=meta["num_downs"],
                             fp16=0, checkpoints_dir=str(tmp_path), name="x", load_epoch=ckpt, verbose=False)
    m = create_model(opt)
    m.setup(opt)
    m.eval()
    xd = torch.from_numpy(x).cuda()
    out = m.inference(xd[:, :1], xd[:, 1:])                                   # feature map + 22 candidate channels (cat -> 23)
    assert np.abs(out.cpu().numpy() - ref).max() <= TOL
    out2 = m.inference(xd, None)                                               # cand_image None: feature_map already has 23 channels
    assert torch.equal(out, out2)


@pytest.mark.gpu
def test_prepare_and_shuffle_entry_points():
    import ctypes
    from livespeechportraits_amd import _native as N
    lib, dev = N.load(), torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    stream = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    for nchw in (0, 1):
        src = torch.randn(2, 6, 8, 5, generator=g) if nchw else torch.randn(2, 8, 6, 5, generator=g)   # NCHW [b,c,h,w] / NHWC [b,h,w,c]
        b, c, h, w = (2, 6, 8, 5) if nchw else (2, 5, 8, 6)
        if nchw:
            src = torch.randn(2, 5, 8, 6, generator=g)                        # [b, c=5, h=8, w=6]
        x = (src if nchw else src.permute(0, 3, 1, 2)).contiguous()           # NCHW view of the same data
        d = src.to(dev)
        s2d = torch.full((2, 4, 3, 24), float("nan"), device=dev)
        relu = torch.full((2, 8, 6, 5), float("nan"), device=dev)
        N.check(lib.lspf2f_unet_prepare(p(d), nchw, 2, 8, 6, 5, ctypes.c_float(0.2), p(s2d), 24, p(relu), stream()))
        torch.cuda.synchronize()
        want = torch.zeros(2, 24, 4, 3)
        for dy in range(2):
            for dx in range(2):
                want[:, (dy * 2 + dx) * 5:(dy * 2 + dx + 1) * 5] = F.leaky_relu(x[:, :, dy::2, dx::2], 0.2)
        assert torch.equal(s2d.cpu(), want.permute(0, 2, 3, 1))
        assert torch.equal(relu.cpu(), F.relu(x).permute(0, 2, 3, 1))
    with pytest.raises(N.Lspf2fError):
        N.check(lib.lspf2f_unet_prepare(p(d), 0, 2, 7, 6, 5, ctypes.c_float(0.2), p(s2d), 24, None, stream()))       # odd height
    with pytest.raises(N.Lspf2fError):
        N.check(lib.lspf2f_unet_prepare(p(d), 0, 2, 8, 6, 5, ctypes.c_float(0.2), p(s2d), 16, None, stream()))       # s2d_channels < 4c
    gq = torch.randn(2, 4, 3, 12, generator=g)
    out = torch.empty(2, 3, 8, 6, device=dev)
    N.check(lib.lspf2f_pixel_shuffle(p(gq.to(dev)), 2, 4, 3, 3, 1, p(out), None, stream()))
    want = torch.zeros(2, 3, 8, 6)
    for par in range(4):
        want[:, :, par // 2::2, par % 2::2] = torch.tanh(gq[..., par * 3:(par + 1) * 3]).permute(0, 3, 1, 2)
    assert (out.cpu() - want).abs().max() < 1e-6


@pytest.mark.gpu
def test_graph_replay_equals_host_sequenced_launches(gpu_device):
    """HostSequencedUnetEngine(graph=True) captures its ~40 launches once per (batch, size, output kind) and replays them: same bits as the
    host-sequenced engine, for new inputs, another batch size in between (a larger split-K scratch), and the uint8 output."""
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.unet_small import HostSequencedUnetEngine
    sd = synth.make_unet_small_state_dict(input_nc=5, num_downs=5, ngf=32)
    eager, graph = HostSequencedUnetEngine(5, 3, 5, 32), HostSequencedUnetEngine(5, 3, 5, 32, graph=True)
    eager.load_state_dict(sd, "model", gpu_device); graph.load_state_dict(sd, "model", gpu_device)
    xs = [torch.from_numpy(synth.symmetric(b * 5 * 64 * 64, 0.6, 10 + i).reshape(b, 5, 64, 64)).to(gpu_device) for i, b in enumerate((1, 3, 1, 3))]
    for x in xs:
        assert torch.equal(graph.forward(x), eager.forward(x))
        assert torch.equal(graph.forward(x, out_u8=True), eager.forward(x, out_u8=True))
    assert len(graph._graphs) == 4                    # (1 | 3 frames) x (float | uint8)
    graph.load_state_dict(sd, "model", gpu_device)
    assert not graph._graphs                          # new weights: the captured launches are dropped


@pytest.mark.gpu
def test_masked_k_form_runs_on_every_implicit_gemm_tile_and_on_nothing_else(gpu_device):
    """lspf2f_conv3x3 with k_group = -4 (the 16 live (tap, quarter) blocks of a space-to-depth Conv2d(k4, s2, p1); models/networks.py:680-769) against torch's
    convolution on EVERY implicit-GEMM tile the library instantiates for it (launch_igemm_masked keeps its own instance table: ADVICE r4) -- and refused, before any
    other route can read the masked operand as a dense 9-tap one, with 16-bit storage, a stride, or a tile that selects the full-K / row kernels."""
    import ctypes
    import torch.nn.functional as F
    from livespeechportraits_amd import _native as N
    from livespeechportraits_amd.unet_small import pack_down_live
    lib, dev = N.load(), gpu_device
    g = torch.Generator().manual_seed(5)
    b, ci, co, h = 2, 32, 128, 32                              # space-to-depth image: 16 x 16 x (4 * 32)
    x = torch.rand(b, ci, h, h, generator=g) * 2 - 1
    w = (torch.rand(co, ci, 4, 4, generator=g) * 2 - 1) * 0.05
    ref = F.conv2d(x.double(), w.double(), None, 2, 1).float()                 # [b][co][16][16]
    # space-to-depth with the conv's padding folded in: s2d[y][x][(dy*2+dx)*ci + c] = xpad[2y + dy][2x + dx][c], xpad = x shifted by (+1, +1) so that a 3x3 / p1 conv
    # on the 17 x 17 -> cropped 16 x 16 grid sees the 4x4 / s2 / p1 taps (the mapping include/lspf2f.h documents next to lspf2f_unet_prepare)
    s2d = torch.empty(b, h // 2, h // 2, 4 * ci, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    xd = x.to(dev)
    N.check(lib.lspf2f_unet_prepare(p(xd), 1, b, h, h, ci, ctypes.c_float(1.0), p(s2d), 4 * ci, None, st()))      # slope 1: no activation
    wl = torch.from_numpy(pack_down_live(w.numpy())).to(dev)

    def run(tile, dtype=0, stride=1, kg=-4):
        out = torch.full((b, h // 2, h // 2, co), float("nan"), device=dev)
        sb = lib.lspf2f_conv3x3_scratch_bytes(b, h // 2, h // 2, 4 * ci, 0, co, stride, 0, tile[0], tile[1], 0, kg, dtype)
        scratch = torch.zeros(max(sb, 256), dtype=torch.uint8, device=dev)
        rc = lib.lspf2f_conv3x3(p(s2d), None, p(wl), None, None, None, p(out), b, h // 2, h // 2, 4 * ci, 0, co, stride, 0, 0, tile[0], tile[1], 0, kg, dtype,
                                p(scratch), scratch.numel(), st())
        torch.cuda.synchronize()
        return rc, out.permute(0, 3, 1, 2).cpu()

    for tile in ((0, 0), (128, 128), (128, 64), (64, 128), (64, 64), (32, 128), (32, 64)):
        rc, got = run(tile)
        assert rc == 0, (tile, lib.lspf2f_last_error())
        assert (got - ref).abs().max().item() <= 2e-5, (tile, (got - ref).abs().max().item())
    for bad in (dict(tile=(16, 16)), dict(tile=(1008, 64)), dict(tile=(2000, 32)), dict(tile=(64, 64), dtype=1), dict(tile=(64, 64), stride=2)):
        rc, _ = run(**bad)
        assert rc == -2, (bad, rc)                               # LSPF2F_ERR_UNSUPPORTED, never another kernel's launch


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["small_s64_b2", "small_512"])
def test_native_plan_equals_the_host_sequenced_launches_bit_for_bit(name, gpu_device):
    """The native plan (fused activated copies, two-source input pass, in-launch split-K combines, hipGraph) against the round-3/4 form sequenced from Python out of
    lspf2f_conv3x3 / lspf2f_unet_prepare: the same products summed in the same order -> the same bits; and every A-B arm of the plan against its default."""
    from livespeechportraits_amd.unet_small import HostSequencedUnetEngine, SmallUnetEngine
    meta, sd, x, ref = load_case(name)
    xd = torch.from_numpy(x).to(gpu_device)
    host = HostSequencedUnetEngine(23, 3, meta["num_downs"], meta["ngf"])
    host.load_state_dict(sd, "model", gpu_device)
    want = host.forward(xd)
    # with the last layer in its GEMM form on the host-sequenced form's tile, every arm of the plan repeats its bits; the defaults (direct last-layer kernel) and the
    # in-launch split-K combine (another summation order at 6 splits) are held to the reference golden
    same = "last_direct=0,last_tile=-1,tiny=0,dense0=1"
    for tune, exact in ((same, True), (same + ",fused_prepare=0", True), (same + ",input_pass=0", True), (same + ",graph=0", True),
                        (same + ",fused_prepare=0,input_pass=0,graph=0", True), (None, False), ("last_direct=0", False), ("fused_splitk=1", False), ("graph=0", False), ("tiny=0", False), ("dense0=1", False)):
        e = SmallUnetEngine(23, 3, meta["num_downs"], meta["ngf"], tune=tune)
        e.load_state_dict(sd, "model", gpu_device)
        got = e.forward(xd)
        err = np.abs(got.cpu().numpy() - ref).max()
        print("[%s] %-60s max-abs vs reference %.2e, vs the host-sequenced form %.2e" % (name, tune, err, (got - want).abs().max().item()))
        assert err <= TOL, tune
        if exact:
            assert torch.equal(got, want), (tune, (got - want).abs().max().item())
            assert torch.equal(e.forward(xd, out_u8=True), host.forward(xd, out_u8=True))
        assert torch.equal(e.forward(xd), got)                                 # graph replay
        u8 = e.forward(xd, out_u8=True).cpu().numpy().astype(np.int32)
        assert np.abs(u8 - np.clip((ref.transpose(0, 2, 3, 1) + 1.0) / 2.0 * 255.0, 0, 255).astype(np.uint8).astype(np.int32)).max() <= 1      # util.tensor2im
        e.close()


@pytest.mark.gpu
def test_native_plan_two_sources_broadcast_candidates_and_hazards(gpu_device):
    """(feature_map, cand_image) as two base pointers == the concatenated tensor; a batch-1 candidate stack broadcast over 3 feature maps; alternating distinct
    batches through ONE engine == fresh engines; a workspace poisoned with NaN bytes between forwards changes no bit."""
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.unet_small import SmallUnetEngine
    nd, ngf, S = 6, 32, 128
    sd = synth.make_unet_small_state_dict(23, 3, nd, ngf, seed=5)
    mk = lambda b, seed: torch.from_numpy(synth.symmetric(b * 23 * S * S, 0.6, seed).reshape(b, 23, S, S)).to(gpu_device)
    e = SmallUnetEngine(23, 3, nd, ngf)
    e.load_state_dict(sd, "model", gpu_device)
    x3 = mk(3, 21)
    full = e.forward(x3)
    assert torch.equal(e.render(x3[:, :1].contiguous(), x3[:, 1:].contiguous()), full)
    xb = torch.cat([x3[:, :1], x3[:1, 1:].expand(3, -1, -1, -1)], 1).contiguous()
    assert torch.equal(e.render(x3[:, :1].contiguous(), x3[:1, 1:].contiguous()), e.forward(xb))                  # cand batch 1: broadcast
    assert torch.equal(e.render(x3[:, :5].contiguous(), x3[:, 5:].contiguous(), out_u8=True), e.forward(x3, out_u8=True))
    # alternate distinct batches through one engine; each against a fresh engine
    seq = [(1, 31), (3, 32), (2, 33), (1, 34), (3, 35)]
    for b, seed in seq:
        x = mk(b, seed)
        got = e.forward(x)
        fresh = SmallUnetEngine(23, 3, nd, ngf, max_batch=b)
        fresh.load_state_dict(sd, "model", gpu_device)
        assert torch.equal(got, fresh.forward(x)), (b, seed)
        fresh.close()
    # poison: everything behind the arrival counters (16384 x 4 bytes at the head of the workspace) is scratch
    plan = e._plan(S, 23, 3)
    before = e.forward(x3).clone()
    plan["ws"][16384 * 4:].fill_(0xFF)
    torch.cuda.synchronize()
    assert torch.equal(e.forward(x3), before) and not torch.isnan(before).any()
    assert int(plan["ws"][:16384 * 4].view(torch.int32).ne(0).sum()) == 0      # every last arriver left its counter at zero
    buf = torch.empty_like(before)
    assert e.forward(x3, out=buf) is buf and torch.equal(buf, before)           # caller-owned result tensor
    with pytest.raises(ValueError):
        e.forward(x3, out=torch.empty(1, 3, S, S, device=gpu_device))
    with pytest.raises(ValueError):
        e.forward(x3.cpu())
    with pytest.raises(ValueError):
        e.render(x3[:, :1].contiguous(), x3[:2, 1:].contiguous())               # cand batch neither 1 nor B


@pytest.mark.gpu
def test_fp16_plan_of_the_small_generator_against_the_autocast_oracle(gpu_device):
    """opt.fp16 with size == 'small' (models/feature2face_G.py:28-30 wraps WHICHEVER netG in torch.cuda.amp.autocast): the pin is the oracle's op sequence -- bit-identical
    to the reference module in fp32 (oracle/make_golden_unet.py) -- under torch.autocast(float16) on this device.  Its distance from the fp32 reference output is the error
    the reference accepts when a user sets fp16; the HIP fp16 plan (fp16 storage, fp32 accumulate and epilogue: one rounding per layer) must stay inside it, max and mean,
    per frame, and close to the autocast output itself."""
    from oracle import unet_small_oracle
    from livespeechportraits_amd.unet_small import SmallUnetEngine
    meta, sd, x, ref32 = load_case("small_512")
    sd_d = {k: torch.from_numpy(v).to(gpu_device) for k, v in sd.items()}
    xd = torch.from_numpy(x).to(gpu_device)
    with torch.autocast("cuda", dtype=torch.float16):
        ref16 = unet_small_oracle.generator_forward(sd_d, xd, meta["num_downs"])
    assert ref16.dtype == torch.float16            # what the reference's inference() returns under opt.fp16
    ref16 = ref16.float().cpu().numpy()
    e = SmallUnetEngine(23, 3, meta["num_downs"], meta["ngf"], dtype="f16")
    e.load_state_dict(sd, "model", gpu_device)
    out_t = e.forward(xd)
    assert torch.equal(e.forward(xd), out_t)       # deterministic, graph replay
    out = out_t.cpu().numpy()
    names = [r["kernel"] for r in e.launches(512, xd.shape[0])]
    assert not any("unet_tiny" in n for n in names) and sum("<km>" in n for n in names) == 8        # the fp16 plan: masked-K implicit GEMMs on every level
    B = x.shape[0]
    d_ref, d_got, d_16 = (np.abs(a - b).reshape(B, -1) for a, b in ((ref16, ref32), (out, ref32), (out, ref16)))
    fmt = lambda a: " ".join("%.1e" % v for v in a)
    print("\nsmall generator fp16, per frame:\n  autocast oracle vs fp32 reference  max %s | mean %s\n  HIP fp16 vs fp32 reference         max %s | mean %s\n"
          "  HIP fp16 vs autocast oracle        max %s | mean %s" % (fmt(d_ref.max(1)), fmt(d_ref.mean(1)), fmt(d_got.max(1)), fmt(d_got.mean(1)), fmt(d_16.max(1)), fmt(d_16.mean(1))))
    assert (d_got.max(1) <= 1.25 * d_ref.max(1) + 1e-4).all() and (d_got.mean(1) <= 1.25 * d_ref.mean(1) + 1e-5).all()
    assert (d_16.max(1) <= 2.0 * d_ref.max(1) + 1e-4).all() and (d_16.mean(1) <= 2.0 * d_ref.mean(1) + 1e-5).all()
    assert d_got.max() > 1e-5                      # really the fp16 plan
    # two sources, broadcast candidates, three frames, uint8: the same plan through render()
    x3 = torch.cat([xd, xd.flip(0)[:1] * 0.5, xd[:1] * -0.7])[:3].contiguous() if B < 3 else xd[:3]
    xb = torch.cat([x3[:, :1], x3[:1, 1:].expand(3, -1, -1, -1)], 1).contiguous()
    assert torch.equal(e.render(x3[:, :1].contiguous(), x3[:1, 1:].contiguous()), e.forward(xb))
    u8 = e.forward(xd, out_u8=True).cpu().numpy().astype(np.int32)
    assert np.abs(u8 - np.clip((out.transpose(0, 2, 3, 1) + 1.0) / 2.0 * 255.0, 0, 255).astype(np.uint8).astype(np.int32)).max() <= 1
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("output_nc", [1, 2, 4])
def test_fp16_plan_with_another_output_width_stays_inside_its_buffers(output_nc, gpu_device):
    """fp16 plans with output_nc != 3 were routed to rowlast128, which is hard-wired to 12 columns per pixel -- it wrote past the G buffer
    (output_nc 1, 2) or produced wrong frames (4).  Now they keep the implicit-GEMM last layer: the fp16 frames agree with the fp32 plan of the same weights, at 512 x 512
    (the size whose last level the row kernel would have taken), and a poisoned guard band behind the output stays intact."""
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.unet_small import SmallUnetEngine
    sd = synth.make_unet_small_state_dict(23, output_nc, 8, 64, seed=21)
    x = torch.from_numpy(synth.symmetric(2 * 23 * 512 * 512, 0.6, 5).reshape(2, 23, 512, 512)).to(gpu_device)
    outs = {}
    for dt in ("f32", "f16"):
        e = SmallUnetEngine(23, output_nc, 8, 64, max_batch=2, dtype=dt)
        e.load_state_dict(sd, "model", gpu_device)
        if dt == "f16":
            assert not any("rowlast128" in r["kernel"] for r in e.launches(512, 2))
        y = e.forward(x)
        assert tuple(y.shape) == (2, output_nc, 512, 512) and torch.isfinite(y).all()
        outs[dt] = y.float().cpu().numpy()
        e.close()
    d = np.abs(outs["f16"] - outs["f32"])
    print("\nsmall generator, output_nc %d, fp16 vs fp32 plan: max-abs %.2e mean-abs %.2e" % (output_nc, d.max(), d.mean()))
    assert 1e-6 < d.max() <= 8e-3 and d.mean() <= 8e-4


@pytest.mark.gpu
def test_opt_fp16_with_size_small_selects_the_fp16_plan_and_returns_half(tmp_path):
    """create_model(opt) with size='small', fp16=1 (ngf 64): the fp16 plan, a float16 tensor out like autocast returns; fp16=0 stays fp32; ngf 32 warns and runs fp32."""
    import warnings
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.models import create_model
    outs = {}
    for ngf, fp16 in ((64, 0), (64, 1), (32, 1)):
        sd = synth.make_unet_small_state_dict(23, 3, 6, ngf, seed=9)
        ckpt = os.path.join(tmp_path, "Feature2Face_%d_%d.pkl" % (ngf, fp16))
        torch.save({"module.netG." + k: torch.from_numpy(v) for k, v in sd.items()}, ckpt)
        opt = argparse.Namespace(model="feature2face", gpu_ids=[0], isTrain=False, size="small", ngf=ngf, n_downsample_G=6, fp16=fp16,
                                 checkpoints_dir=str(tmp_path), name="x", load_epoch=ckpt, verbose=False)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            m = create_model(opt)
        m.setup(opt)
        m.eval()
        g = m._g().netG
        want16 = bool(fp16) and ngf % 64 == 0
        assert g.dtype == ("f16" if want16 else "f32")
        assert any("fp16" in str(x.message) for x in w) == (bool(fp16) and not want16)
        x = torch.from_numpy(synth.symmetric(2 * 23 * 128 * 128, 0.6, 77).reshape(2, 23, 128, 128)).cuda()
        y = m.inference(x[:, :1], x[:, 1:])
        assert y.dtype == (torch.float16 if want16 else torch.float32) and tuple(y.shape) == (2, 3, 128, 128)
        assert m.inference_image(x[:, :1], x[:, 1:]).dtype == torch.uint8
        outs[(ngf, fp16)] = y.float().cpu().numpy()
    d = np.abs(outs[(64, 1)] - outs[(64, 0)])
    print("\nsize small, opt.fp16 = 1 at 128x128: max-abs %.2e mean-abs %.2e vs the fp32 plan" % (d.max(), d.mean()))
    assert 1e-6 < d.max() <= 5e-3
