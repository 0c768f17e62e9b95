"""CPU checks of the Winograd F(2x2,3x3) path (csrc/wino.hip): the lane-level data-flow model against a direct convolution, the library's
host packer against the model's statement of the fragment order, and which layers of a plan take the kernel.  No GPU: the kernel's
own parity tests are tests/test_gpu_conv.py::test_conv3x3_winograd* and every network-level golden test (the plans route through it)."""
import os
import sys

import numpy as np

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import wino_model as WM   # noqa: E402


def test_data_flow_model_equals_direct_convolution():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 8, 32, 16))
    w = rng.standard_normal((64, 16, 3, 3)).astype(np.float32)
    sc, sh = rng.standard_normal(64), rng.standard_normal(64)
    res = rng.standard_normal((2, 8, 32, 64))
    got = WM.conv_model(x, WM.pack_u(w), 64, sc, sh, res, relu=True)
    ref = np.maximum(WM.conv_direct(x, w) * sc + sh + res, 0)
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()       # U is rounded to fp32 once; everything else is float64 here


def test_transform_matrices_are_the_published_f2x2_3x3():
    """Y = A^T [(G g G^T) . (B^T d B)] A reproduces a 3x3 correlation on a 4x4 patch exactly (Lavin & Gray 2015, F(2x2, 3x3))."""
    rng = np.random.default_rng(0)
    d, g = rng.standard_normal((4, 4)), rng.standard_normal((3, 3))
    y = WM.AT @ ((WM.G @ g @ WM.G.T) * (WM.BT @ d @ WM.BT.T)) @ WM.AT.T
    ref = np.array([[(d[a:a + 3, b:b + 3] * g).sum() for b in range(2)] for a in range(2)])
    assert np.abs(y - ref).max() < 1e-12


def test_host_packer_writes_the_fragment_order_the_kernel_reads():
    """Plan::pack -> pack_wino_weights: the G g G^T copy of every Winograd layer sits behind the 9-tap copy in the blob and equals the
    numpy statement of the layout (tools/wino_model.pack_u); run through the data-flow model it convolves correctly."""
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.engine import Engine
    topo, sd = synth.synthetic("normal", ngf=32, num_downs=5, size=64)
    e = Engine("normal", ngf=32, num_downs=5, size=64, max_batch=2, tune={"all_forms": 1})      # the 9-tap rows (to identify the layer) AND the G g G^T copy
    e.load_state_dict(sd)
    blob = e.pack().numpy()
    checked = 0
    for i, l in enumerate(e.layers(1)):
        if not l["kernel"].startswith("wino3x3"):
            continue
        cin, cout = l["cin"], l["cout"]
        nine = cout * cin * 9 * 4
        off = e.form_offset(i, "wino")
        assert off > l["w_offset"] >= 0
        got = blob[off: off + cout * cin * 16 * 4].view(np.float32)
        key = [k for k in sd if k.endswith(".weight") and sd[k].shape == (cout, cin, 3, 3)]
        # identify the layer's weight through the 9-tap copy the igemm would read: [co][tap][ci]
        rows = blob[l["w_offset"]: l["w_offset"] + nine].view(np.float32).reshape(cout, 9, cin)
        w = np.ascontiguousarray(rows.transpose(0, 2, 1).reshape(cout, cin, 3, 3))
        assert any(np.array_equal(w, sd[k]) for k in key)
        exp = WM.pack_u(w)
        assert got.shape == exp.shape and np.allclose(got, exp, rtol=3e-7, atol=1e-9)
        assert (got == exp).mean() > 0.99                      # double rounding of G g G^T may differ in the last bit, rarely
        if checked == 0:
            rng = np.random.default_rng(1)
            x = rng.standard_normal((1, 8, 16, cin))
            ref = WM.conv_direct(x, w)
            assert np.abs(WM.conv_model(x, got, cout) - ref).max() <= 1e-5 * np.abs(ref).max()
        checked += 1
    assert checked >= 2
    e.close()


def test_planner_routes_the_stride1_convs_of_the_large_levels_through_the_winograd_kernel():
    from livespeechportraits_amd.engine import Engine
    e = Engine("large", max_batch=8)
    for batch in (1, 8):
        ls = e.layers(batch)
        wino = [l for l in ls if l["kernel"].startswith("wino3x3")]
        # the 32 ResidualBlock convs at 256x256 .. 32x32 (networks.py:650-675): 8 per level, 64 / 128 / 256 / 512 channels; from 4 frames up
        # also the 8 at 16x16 (below that the full-K kernel is as fast)
        big = [l for l in wino if l["h_out"] >= 32]
        assert len(big) == 32 and all(l["stride"] == 1 and not l["upsample"] and l["cin"] == l["cout"] for l in wino)
        assert sorted({(l["cin"], l["h_out"]) for l in big}) == [(64, 256), (128, 128), (256, 64), (512, 32)]
        assert len(wino) - len(big) == (8 if batch >= 4 else 0) and all(l["h_out"] == 16 for l in wino if l not in big)
        assert all(l["exec_flops_per_frame"] * 9 == l["flops_per_frame"] * 4 for l in wino)
        for l in wino:
            groups = l["cout"] // l["tile_n"]
            wgs = batch * (l["h_out"] // 8) * (l["h_out"] // 16) * groups * l["split_k"]
            assert wgs >= 384 or l["h_out"] == 16, (l["name"], wgs)               # the chip has 256 CUs
            assert l["cin"] // 8 // l["split_k"] >= 4
        if batch == 8:
            assert all(l["split_k"] == 1 for l in big)
    assert not any(l["kernel"].startswith("wino3x3") for l in Engine("large", dtype="bf16").layers(1))
    # InstanceNorm plans run the same layers on the same kernel (raw conv output + bias), with the statistics / normalisation passes behind it
    inl = Engine("large", norm="instance").layers(1)
    assert [l["name"] for l in inl if l["kernel"].startswith("wino3x3")] == [l["name"] for l in e.layers(1) if l["kernel"].startswith("wino3x3")]
    assert all(l["kernel"].endswith(("+in_small", "+in_reduce_stats+in_finalize+in_apply", "(stats)+in_finalize+in_apply")) for l in inl if l["kernel"].startswith(("wino3x3", "winoup3x3")))
    # wino3x3 leaves the statistics of its tile-blocks itself (no in_reduce_stats pass, no in_small at 32 x 32); the switch puts the passes back
    big_in = [l for l in inl if l["kernel"].startswith("wino3x3") and l["h_out"] > 32]
    assert big_in and all(l["kernel"].endswith("(stats)+in_finalize+in_apply") for l in inl if l["kernel"].startswith("wino3x3"))
    off = Engine("large", norm="instance", tune={"in_wino_stats": 0})
    assert all(l["kernel"].endswith("+in_reduce_stats+in_finalize+in_apply") for l in off.layers(1) if l["kernel"].startswith(("wino3x3", "winoup3x3")) and l["h_out"] > 32)
    assert all(l["kernel"].endswith("+in_small") for l in off.layers(1) if l["kernel"].startswith(("wino3x3", "winoup3x3")) and l["h_out"] <= 32)
    off.close()
    assert all(l["kernel"].endswith("(stats)+in_finalize+in_apply") for l in inl if l["kernel"].startswith("winoup3x3"))       # the up-conv kernel leaves its statistics too
    e.close()


def test_winograd_can_be_switched_off_per_handle(monkeypatch):
    from livespeechportraits_amd.engine import Engine
    monkeypatch.setenv("LSP_HIP_WINO", "0")
    e = Engine("normal")
    assert not any(l["kernel"].startswith("wino3x3") for l in e.layers(1))
    e.close()


def test_winograd_kernel_instances_do_not_spill(tmp_path):
    """Every wino3x3 instantiation sits at 164-256 registers per lane; one more live value and hipcc spills to scratch, which costs 30 % of the
    forward without failing any parity test (it happened once: a runtime switch between two loop bodies).  Cross-compile the file and read
    the compiler's own resource report: no scratch, and the occupancy the LDS budget is planned for."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    report = ""
    for f in ("wino.hip", "winoup.hip", "wino4.hip"):
        src = os.path.join(ROOT, "livespeechportraits_amd", "csrc", f)
        p = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-Rpass-analysis=kernel-resource-usage",
                            "-c", src, "-o", str(tmp_path / "w.o")], capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        report += p.stderr
    blocks = re.split(r"remark: Function Name: ", report)[1:]
    seen = 0
    for b in blocks:
        name = b.split()[0]
        if "wino3x3" not in name and "winoup3x3" not in name and "wino4_3x3" not in name:
            continue
        seen += 1
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
        occ = int(re.search(r"Occupancy \[waves/SIMD\]: (\d+)", b).group(1))
        if "wino4_3x3" in name:            # one workgroup per CU by design (two ring slots of 56 KB): 400 registers, no scratch
            assert scratch == 0 and occ == 1, (name, scratch, occ)
            continue
        if "wino3x3_chain" in name:        # the probe instance of round 6 (2..4 layers in one launch, nb = 1): same budget as wino3x3<1>
            assert scratch == 0 and occ >= 2, (name, scratch, occ)
            continue
        nb = int(re.search(r"wino(?:up)?3x3ILi(\d)E", name).group(1))
        assert scratch == 0, (name, scratch)
        assert occ >= 2, (name, occ)          # two workgroups per CU share every SIMD (nb = 2: 79 KB of LDS each; nb = 1: up to three)
        assert nb in (1, 2)
    assert seen >= 7


def test_patch_staged_kernel_instances_do_not_spill_and_keep_two_waves_per_simd(tmp_path):
    """conv3x3_patch16 / conv3x3_patch16d / conv3x3_patchup16 (round 6, csrc/patch16.hip) are planned for ONE workgroup of 8 waves per CU = two waves per SIMD, one of
    each wave group: the schedule's overlap (one group's MFMAs under the other's fragment reads) exists only if both fit the register file -- <= 256 registers per lane, no scratch.
    Read it off the compiler's resource report, as for the Winograd kernels."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(ROOT, "livespeechportraits_amd", "csrc", "patch16.hip")
    p = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", str(tmp_path / "p.o")],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    seen = {"patch16I": 0, "patch16d": 0, "patchup16": 0}
    for b in re.split(r"remark: Function Name: ", p.stderr)[1:]:
        name = b.split()[0]
        if "conv3x3_patch" not in name:
            continue
        for k in seen:
            if ("conv3x3_" + k) in name:
                seen[k] += 1
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
        vgpr = int(re.search(r" VGPRs: (\d+)", b).group(1))
        agpr = int(re.search(r"AGPRs: (\d+)", b).group(1))
        occ = int(re.search(r"Occupancy \[waves/SIMD\]: (\d+)", b).group(1))
        assert scratch == 0 and vgpr + agpr <= 256 and occ >= 2, (name, scratch, vgpr, agpr, occ)
    assert seen["patch16I"] == 8 and seen["patch16d"] == 4 and seen["patchup16"] == 10, seen      # {bf16, fp16} x {4x64, 8x32} x {128, 64}; deep: x 64 only; up: + 16x16 x 64


def test_register_form_never_touches_a_u_register_in_flight():
    """The UR form of wino3x3 loads its U fragments by inline asm two K-steps ahead; hipcc does not know those registers are in flight.  A tied asm
    operand once made it copy them BEFORE the counted wait (stale values, caught on the CPU by reading the assembly): nothing but the MFMAs' B
    operand may name one of the 48 registers inside the K loops and their prologues."""
    import shutil
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_ureg_asm as C
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    for f, kernels in C.KERNELS.items():
        asm = C.compile_to_asm(os.path.join(ROOT, "livespeechportraits_amd", "csrc", f), hipcc)
        for prefix, nreg, nmfma, nloads in kernels:
            ureg, nl, nm, bad = C.check(C.kernel_text(asm, prefix))
            assert (len(ureg), nm, nl) == (nreg, nmfma, nloads), (prefix, len(ureg), nm, nl)      # three register sets; every unrolled step found
            assert not bad, (prefix, bad[:5])


def test_the_asm_check_itself_catches_a_copy_of_an_in_flight_register():
    """tools/check_ureg_asm.py on a hand-made listing: a clean loop passes, a v_mov out of a U register inside the loop (what the tied wait operand produced) and a
    spill of one in the prologue are both reported; uses of the same registers behind the loop are not."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_ureg_asm as C
    clean = """
	;;#ASMSTART
	buffer_load_dwordx4 v[10:13], v1, s[4:7], s0 offen offset:0
	;;#ASMEND
	s_branch .LBB0_1
.LBB0_1:                                ; =>This Inner Loop Header: Depth=1
	ds_read_b128 v[40:43], v2
	v_pk_add_f32 v[40:41], v[40:41], v[42:43]
	v_mfma_f32_32x32x2_f32 v[100:115], v40, v10, v[100:115]
	;;#ASMSTART
	buffer_load_dwordx4 v[14:17], v1, s[4:7], s0 offen offset:0x400
	;;#ASMEND
	v_mfma_f32_32x32x2_f32 v[100:115], v41, v11, v[100:115]
	s_cbranch_vccz .LBB0_1
.LBB0_2:
	v_mov_b32_e32 v10, 0
	global_store_dword v[50:51], v11, off
""".splitlines()
    ureg, nl, nm, bad = C.check(clean)
    assert ureg == list(range(10, 18)) and nl == 2 and nm == 2 and not bad
    copied = [l if "v_pk_add_f32" not in l else l + "\n\tv_mov_b64_e32 v[60:61], v[14:15]" for l in clean]
    assert [b[1].strip() for b in C.check("\n".join(copied).splitlines())[3]] == ["v_mov_b64_e32 v[60:61], v[14:15]"]
    spilled = [l if "s_branch .LBB0_1" not in l else "\tscratch_store_dwordx4 off, v[10:13], off offset:16\n" + l for l in clean]
    assert len(C.check("\n".join(spilled).splitlines())[3]) == 1
    wrong_operand = [l.replace("v40, v10, v[100:115]", "v10, v40, v[100:115]") for l in clean]          # a U register as the A operand
    assert len(C.check(wrong_operand)[3]) == 1


# ---- the up-conv form (csrc/winoup.hip): Upsample(x2, nearest) + Conv3x3 with 9 multiplies per 2x2 outputs ------------------------------
def test_upconv_form_is_exact_and_its_data_flow_model_convolves():
    """Row 2 of B^T d B vanishes on an upsampled patch (rows a, b, b, c): 9 of the 16 transformed positions carry everything."""
    rng = np.random.default_rng(5)
    s, g = rng.standard_normal((3, 3)), rng.standard_normal((3, 3))
    m = [0, 1, 1, 2]
    d = np.array([[s[m[a]][m[b]] for b in range(4)] for a in range(4)])           # the upsampled 4x4 patch under an even-aligned output tile
    v = WM.BT @ d @ WM.BT.T
    assert np.abs(v[2]).max() == 0 and np.abs(v[:, 2]).max() == 0
    x0, x1 = rng.standard_normal((2, 4, 16, 8)), rng.standard_normal((2, 4, 16, 8))
    w = rng.standard_normal((64, 16, 3, 3)).astype(np.float32)
    ref = WM.upconv_direct(x0, x1, w)
    assert np.abs(WM.upconv_model(x0, x1, WM.pack_u_up(w), 64) - ref).max() <= 1e-5 * np.abs(ref).max()
    w1 = rng.standard_normal((32, 8, 3, 3)).astype(np.float32)                     # one source
    ref1 = WM.upconv_direct(x0, None, w1)
    assert np.abs(WM.upconv_model(x0, None, WM.pack_u_up(w1), 32) - ref1).max() <= 1e-5 * np.abs(ref1).max()


def test_upconv_packer_and_planner():
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.engine import Engine
    topo, sd = synth.synthetic("normal", ngf=32, num_downs=5, size=128)
    e = Engine("normal", ngf=32, num_downs=5, size=128, max_batch=2)
    e.load_state_dict(sd)
    blob = e.pack().numpy()
    checked = 0
    for i, (l, c) in enumerate(zip(e.layers(1), topo.convs)):
        if not l["kernel"].startswith("winoup3x3"):
            continue
        cin, cout = l["cin"], l["cout"]
        w = sd[c.weight_key]
        exp = WM.pack_u_up(w)
        off = e.form_offset(i, "winoup")
        assert off >= 0 and l["w_offset"] == -1               # every plan of this handle runs the layer on winoup3x3: no sub-pixel rows in its blob
        got = blob[off: off + exp.size * 4].view(np.float32)
        assert np.allclose(got, exp, rtol=3e-7, atol=1e-9) and (got == exp).mean() > 0.99, l["name"]
        assert l["exec_flops_per_frame"] * 4 == l["flops_per_frame"] and l["weight_bytes"] == 9 * cin * cout * 4
        checked += 1
    assert checked >= 1
    big = Engine("large", max_batch=8)
    for batch in (1, 8):
        ups = [l for l in big.layers(batch) if l["kernel"].startswith("winoup3x3")]
        # the sub-pixel up-convs (>= 32x32 outputs); from 2 frames up also L5.up (8x8 -> 16x16), which keeps the full-K kernel at one frame; L6 / L7.up stay 9-tap
        assert [l["name"] for l in ups] == (["L5.up"] if batch >= 2 else []) + ["L4.up", "L3.up", "L2.up", "L1.up"]
        if batch >= 2:
            assert ups[0]["flops_per_frame"] == ups[0]["exec_flops_per_frame"] * 4 and ups[0]["split_k"] == 4
        for l in ups:
            wgs = batch * (l["h_in"] // 4) * (l["h_in"] // 8) * (l["cout"] // l["tile_n"]) * l["split_k"]
            assert wgs >= 384 and l["cin"] // 8 // l["split_k"] >= 8
    assert not any(l["kernel"].startswith("winoup3x3") for l in Engine("large", dtype="bf16").layers(1))
    e.close(); big.close()


# ---- Winograd F(4x4, 3x3) (csrc/wino4.hip): opt-in per handle (LSPF2F_FLAG_WINO4 / Engine(wino4=True)) -------------------------------------
def test_f4x4_transform_matrices_and_data_flow_model():
    """Y = A^T [(G g G^T) . (B^T d B)] A with the points 0, +-1, +-2, inf reproduces a 3x3 correlation on a 6x6 patch (Lavin & Gray 2015,
    F(4x4, 3x3)); the lane-level model of the kernel (chunk map of the 18 x 34 raw patch, 3x3 position blocks per wave, the factored
    transforms, fragment order, accumulator patch, output scatter) convolves correctly, borders and tile-block seams included."""
    rng = np.random.default_rng(0)
    d, g = rng.standard_normal((6, 6)), rng.standard_normal((3, 3))
    y = WM.AT4 @ ((WM.G4 @ g @ WM.G4.T) * (WM.BT4 @ d @ WM.BT4.T)) @ WM.AT4.T
    ref = np.array([[(d[a:a + 3, b:b + 3] * g).sum() for b in range(4)] for a in range(4)])
    assert np.abs(y - ref).max() < 1e-11
    # the chunk map is a bijection of the 18 x 34 x 2 patch onto 0..1223 and the kernel's decode inverts it
    seen = {WM.w4_chunk(py, px, q) for py in range(18) for px in range(34) for q in range(2)}
    assert seen == set(range(1224))
    assert all(WM.w4_chunk(*WM.w4_chunk_decode(ci)) == ci for ci in range(1224))
    x = rng.standard_normal((2, 32, 64, 16))                    # 2 x 2 tile-blocks per frame
    w = rng.standard_normal((64, 16, 3, 3)).astype(np.float32)
    sc, sh = rng.standard_normal(64), rng.standard_normal(64)
    res = rng.standard_normal((2, 32, 64, 64))
    got = WM.conv4_model(x, WM.pack_u4(w), 64, sc, sh, res, relu=True)
    ref = np.maximum(WM.conv_direct(x, w) * sc + sh + res, 0)
    assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()       # U is rounded to fp32 once; everything else is float64 here


def test_f4x4_is_opt_in_and_its_packer_writes_the_order_the_kernel_reads():
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.engine import Engine
    topo, sd = synth.synthetic("normal", ngf=32, num_downs=5, size=128)
    plain = Engine("normal", ngf=32, num_downs=5, size=128, max_batch=2)
    assert not any(l["kernel"].startswith("wino4") for l in plain.layers(1))
    e = Engine("normal", ngf=32, num_downs=5, size=128, max_batch=2, wino4=True)
    assert e.packed_bytes() > plain.packed_bytes()              # the 6x6 transformed copy travels only when asked for
    e.load_state_dict(sd)
    blob = e.pack().numpy()
    checked = 0
    for i, (l, c) in enumerate(zip(e.layers(1), topo.convs)):
        if not l["kernel"].startswith("wino4_3x3"):
            continue
        cin, cout = l["cin"], l["cout"]
        assert l["h_out"] >= 32 and l["h_out"] % 32 == 0 and l["stride"] == 1 and not l["upsample"]
        assert l["exec_flops_per_frame"] * 4 == l["flops_per_frame"] and l["weight_bytes"] == 36 * cin * cout * 4
        w = sd[c.weight_key]
        exp = WM.pack_u4(w)
        off = e.form_offset(i, "wino4")
        assert off >= 0 and l["w_offset"] == -1 and e.form_offset(i, "wino") == -1       # the only form of this layer the handle's plans read
        got = blob[off: off + exp.size * 4].view(np.float32)
        assert np.allclose(got, exp, rtol=3e-7, atol=1e-9) and (got == exp).mean() > 0.99, l["name"]
        checked += 1
    assert checked >= 2
    plain.close(); e.close()
    big = Engine("large", max_batch=8, wino4=True)
    for batch in (1, 8):
        w4 = [l for l in big.layers(batch) if l["kernel"].startswith("wino4_3x3")]
        assert sorted({(l["cin"], l["h_out"]) for l in w4}) == [(64, 256), (128, 128), (256, 64), (512, 32)] and len(w4) == 32
        for l in w4:
            wgs = batch * (l["h_out"] // 16) * (l["h_out"] // 32) * (l["cout"] // 32) * l["split_k"]
            assert wgs >= 256 and l["cin"] // 8 // l["split_k"] >= 4, (l["name"], wgs)     # one workgroup per CU, 256 CUs
    assert not any(l["kernel"].startswith("wino4") for l in Engine("large", dtype="bf16", wino4=True).layers(1))
    big.close()


def test_tune_string_reaches_the_library_and_unknown_keys_are_errors(monkeypatch):
    """The library reads no environment: LSP_HIP_* variables are mapped onto lspf2f_create_tuned's `tune` keys by the Python host."""
    import pytest
    from livespeechportraits_amd import _native as N
    from livespeechportraits_amd.engine import Engine
    assert not any(l["kernel"].startswith("wino") for l in Engine("normal", tune={"wino": 0}).layers(1))
    assert any(l["kernel"].startswith("wino4") for l in Engine("normal", tune="wino4=1").layers(1))
    monkeypatch.setenv("LSP_HIP_WINOUP", "0")
    assert N.tune_string({"wino": 0}) == b"wino=0,winoup=0"
    assert not any(l["kernel"].startswith("winoup") for l in Engine("normal").layers(1))
    monkeypatch.delenv("LSP_HIP_WINOUP")
    with pytest.raises(N.Lspf2fError, match="unknown key"):
        Engine("normal", tune={"no_such_switch": 1})
    # switches of the Python host share the prefix and never reach the library; a typo in the environment is a warning, not a broken run
    monkeypatch.setenv("LSP_HIP_CAND_CACHE", "0")
    assert N.tune_string() == b""
    monkeypatch.setenv("LSP_HIP_WINOO", "0")
    with pytest.warns(UserWarning, match="LSP_HIP_WINOO"):
        assert N.tune_string() == b""
    monkeypatch.delenv("LSP_HIP_WINOO"); monkeypatch.delenv("LSP_HIP_CAND_CACHE")
    # every key the host forwards is one the library accepts
    for k in sorted(N.TUNE_KEYS):
        Engine("normal", tune={k: 1}).close()
    import re
    import subprocess
    out = subprocess.run(["grep", "-c", "getenv", *[os.path.join(ROOT, "livespeechportraits_amd", "csrc", f) for f in sorted(os.listdir(os.path.join(ROOT, "livespeechportraits_amd", "csrc"))) if f.endswith((".hip", ".cpp", ".h"))]],
                         capture_output=True, text=True).stdout
    assert sum(int(m) for m in re.findall(r":(\d+)$", out, re.M)) <= 1      # the one left is inside #ifdef LSPF2F_ABLATE (tools/ablate.sh builds)


def test_stamp_tail_analysis_finds_a_slow_xcd():
    """tools/wino_stamps.py: the placement-aware reading of a stamp build's output (which SIMDs end late, and is it a late start, a whole CU, a whole XCD or
    scattered) on a synthetic launch: 512 workgroups, two per CU, four waves each on the four SIMDs, per-XCD counters with unrelated origins, XCD 3 slower."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("wino_stamps", os.path.join(ROOT, "tools", "wino_stamps.py"))
    ws = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ws)
    rng = np.random.default_rng(0)
    blocks, steps = 512, 16
    raw = np.zeros((blocks, 4, 8), np.int64)
    hw = np.zeros((blocks, 4), np.int64)
    for b in range(blocks):
        xcc, slot = b % 8, b // 8                                 # 64 workgroups per XCD = 32 CUs x 2
        cu_lin, second = slot % 32, slot // 32
        se, cu = cu_lin // 8, cu_lin % 8
        origin = 1_000_000 * (xcc + 1) * 7                        # counters of different XCDs are not comparable
        entry = origin + 300 * second + int(rng.integers(0, 200))
        for w in range(4):
            loop = steps * 2132 + (7000 if xcc == 3 else 0) + int(rng.integers(0, 300))
            raw[b, w, 0] = entry
            raw[b, w, 1] = entry + 1900
            raw[b, w, 2] = entry + 4100
            raw[b, w, 3] = entry + 4100 + loop
            raw[b, w, 4] = raw[b, w, 3] + 1500
            raw[b, w, 5] = raw[b, w, 4] + 3200
            hw[b, w] = -(1 << 63) | (xcc << 32) | (se << 13) | (cu << 8) | (w << 4) | second     # bit 63 set, as WSTAMP_FLUSH leaves it
    assert ws.place(hw)[0].max() == 7 and set(ws.place(hw)[4].ravel()) == {0, 1, 2, 3}
    r = ws.tail_analysis(raw, hw, steps)
    assert (r["xcds"], r["cus"], r["simds"]) == (8, 256, 1024) and r["waves_per_simd"] == {2: 1024}
    assert r["entry_skew"]["max"] < 600                            # taken per XCD: the unrelated origins do not show
    assert set(r["tail"]["slow_simds_per_xcd"]) == {3}             # the slow tenth sits on XCD 3 ...
    hist = r["tail"]["slow_simds_per_cu_hist"]
    assert sum(v for k, v in hist.items() if k >= 3) >= 0.6 * sum(hist.values())      # ... mostly as whole CUs, not scattered SIMDs
    assert r["launch_cycles_per_xcd"][3] > r["launch_cycles_per_xcd"][0] + 6000
    assert abs(r["tail"].get("corr_entry_vs_end", 0.0)) < 0.3      # not explained by a late start
    assert 0.75 < r["simd_window"]["issue_share_median"] < 1.0     # 2 x 16 x 1024 issue cycles inside a ~34.4 k window


def _ur_loop_model(nsteps, ns, epl):
    """The request / wait protocol of wino_loop_ur<ROW, EPL, NS> (csrc/wino.hip) as a queue model: a wave's loads complete in issue order, `wait(n)` = s_waitcnt
    vmcnt(n) returns once at most n are outstanding.  Returns, per step, the loads still in flight when its MFMAs start (none of them may be its own) and the ring
    slot / register set it reads and the one it refills."""
    q, log = [], []
    wait = lambda n: q.__delitem__(slice(0, max(0, len(q) - n)))
    issue = lambda step: q.extend([("s", step)] * 6)              # 2 raw-patch pieces + 4 U fragments
    issue(0)
    if nsteps > 1:
        issue(1)
    if ns == 4 and nsteps > 2:
        issue(2)
    q.extend([("e", -1)] * epl)                                    # epilogue operands, requested behind the first steps
    wait(12 + epl if ns == 4 and nsteps > 2 else 6 + epl if nsteps > 1 else epl)
    ahead = ns - 1
    t = 0
    while t < nsteps:
        for s in range(ns):
            if t + s >= nsteps:
                break
            cur = t + s
            assert cur % ns == s
            inflight = [x for x in q if x == ("s", cur)]
            will_issue = cur + ahead < nsteps
            log.append((cur, len(inflight), s, (s + ahead) % ns if will_issue else None))
            if will_issue:
                issue(cur + ahead)
            if ns == 3:
                wait(6 if will_issue else 0)
            else:
                wait(12 if will_issue else 6 if cur + 2 < nsteps else 0)
            # after the wait (and the barrier): step cur + 1 must have landed
            assert not [x for x in q if x == ("s", cur + 1)], (nsteps, ns, cur, q)
        t += ns
    assert not [x for x in q if x[0] == "s"]                      # nothing of the K loop is left in flight behind it
    return log


def test_register_form_wait_protocol_for_three_and_four_sets():
    """every step's operands have landed when its MFMAs start, the set it refills is the one read one step earlier, nothing is left in flight at the end -- for the
    shipped three-set form and the four-set A-B arm (tile 4004 / wino_ureg=2), any step count (split-K slices of 2..64 steps), any number of epilogue loads"""
    for ns in (3, 4):
        for nsteps in range(1, 40):
            for epl in (0, 6, 12):
                log = _ur_loop_model(nsteps, ns, epl)
                assert [l[0] for l in log] == list(range(nsteps))
                assert all(l[1] == 0 for l in log), (ns, nsteps, epl, log)               # own loads landed
                for cur, _, s, refill in log:
                    if refill is not None:
                        assert refill == (cur + ns - 1) % ns and refill == (cur - 1) % ns    # the set / slot last read in step cur - 1
