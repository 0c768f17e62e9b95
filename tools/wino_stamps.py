#!/usr/bin/env python3
"""Phase stamps of the Winograd kernel (GPU, library built with -DLSPF2F_WINO_STAMPS: tools/sessions/wino_stamps_job.sh):
  python tools/wino_stamps.py c h nb splits [batch]
Prints, per phase, the median / p90 over all waves of the shader cycles since the wave's kernel entry, and the wall time per launch."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livespeechportraits_amd import _native as N   # noqa: E402


def place(hw):
    """(xcc, se, sh, cu, simd, wave slot) of the packed HW_REG_XCC_ID / HW_REG_HW_ID word the stamp build leaves in slot 6"""
    w = hw & 0xffffffff
    return (hw >> 32) & 0xf, (w >> 13) & 7, (w >> 12) & 1, (w >> 8) & 0xf, (w >> 4) & 3, w & 0xf


def tail_analysis(raw, hw, steps):
    """Where the slow tail of a launch sits (pure numpy: tests/test_wino_cpu.py feeds it a synthetic launch).  raw [blocks][4][8] stamps, hw [blocks][4] packed
    places.  Returns a dict: per-wave K-loop time and launch-relative end by SIMD pairing, by CU, by XCD; the start skew inside each XCD; and how much of the
    launch's length the slowest 10 % of SIMDs add over the median SIMD."""
    xcc, se, sh, cu, simd, _ = place(hw)
    cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    simd_key = cu_key * 4 + simd
    entry, first, kend, done = raw[:, :, 0].astype(np.float64), raw[:, :, 2].astype(np.float64), raw[:, :, 3].astype(np.float64), raw[:, :, 5].astype(np.float64)
    out = {"xcds": int(len(np.unique(xcc))), "cus": int(len(np.unique(cu_key))), "simds": int(len(np.unique(simd_key)))}
    # the counter is per XCD: everything launch-relative is taken against the XCD's first entry
    rel_end, rel_entry, rel_kend = np.zeros_like(done), np.zeros_like(done), np.zeros_like(done)
    for x in np.unique(xcc):
        m = xcc == x
        t0 = entry[m].min()
        rel_end[m], rel_entry[m], rel_kend[m] = done[m] - t0, entry[m] - t0, kend[m] - t0
    out["entry_skew"] = {"median": float(np.median(rel_entry)), "p90": float(np.percentile(rel_entry, 90)), "max": float(rel_entry.max())}
    out["launch_cycles_per_xcd"] = {int(x): float(rel_end[xcc == x].max()) for x in np.unique(xcc)}
    # per SIMD: how many waves shared it, when its LAST wave left the K loop (launch-relative) and the K-loop time of each of its waves
    keys, inv = np.unique(simd_key, return_inverse=True)
    inv = inv.reshape(simd_key.shape)
    waves_on = np.bincount(inv.ravel(), minlength=len(keys))
    last_k = np.zeros(len(keys)); np.maximum.at(last_k, inv.ravel(), rel_kend.ravel())
    first_in = np.full(len(keys), np.inf); np.minimum.at(first_in, inv.ravel(), (first - entry + rel_entry).ravel())
    out["waves_per_simd"] = {int(k): int(v) for k, v in zip(*np.unique(waves_on, return_counts=True))}
    busy = last_k - first_in                                    # the SIMD's K-loop window: first wave's first step landed -> last wave's loop done
    mfma = waves_on * steps * 1024.0                            # MFMA issue cycles of the waves that shared it (NB = 1: 1024 per step and wave)
    out["simd_window"] = {"median": float(np.median(busy)), "p90": float(np.percentile(busy, 90)), "max": float(busy.max()),
                          "mfma_issue_median": float(np.median(mfma)), "issue_share_median": float(np.median(mfma / np.maximum(busy, 1)))}
    # tail: by how much the slowest tenth of the SIMDs end after the median one, and where they are
    med = np.median(last_k)
    slow = last_k >= np.percentile(last_k, 90)
    out["tail"] = {"median_simd_kloop_end": float(med), "p90": float(np.percentile(last_k, 90)), "max": float(last_k.max()),
                   "slow_simds_waves": {int(k): int(v) for k, v in zip(*np.unique(waves_on[slow], return_counts=True))}}
    slow_xcc = (keys[slow] // 4 // 16 // 2 // 8)
    out["tail"]["slow_simds_per_xcd"] = {int(k): int(v) for k, v in zip(*np.unique(slow_xcc, return_counts=True))}
    slow_cu = keys[slow] // 4
    n_cu, cnt = np.unique(slow_cu, return_counts=True)
    out["tail"]["slow_simds_per_cu_hist"] = {int(k): int(v) for k, v in zip(*np.unique(cnt, return_counts=True))}      # 4 = whole CUs are slow, 1 = scattered SIMDs
    # does a late start explain a late end?  correlation of a SIMD's first entry with its K-loop end
    simd_entry = np.full(len(keys), np.inf); np.minimum.at(simd_entry, inv.ravel(), rel_entry.ravel())
    if np.std(simd_entry) > 0 and np.std(last_k) > 0:
        out["tail"]["corr_entry_vs_end"] = float(np.corrcoef(simd_entry, last_k)[0, 1])
    # inside a SIMD: the first finisher and the last (unfair arbitration shows as a large gap with an unchanged last end)
    kl = (kend - first)
    fin_first = np.full(len(keys), np.inf); np.minimum.at(fin_first, inv.ravel(), kl.ravel())
    fin_last = np.zeros(len(keys)); np.maximum.at(fin_last, inv.ravel(), kl.ravel())
    two = waves_on == 2
    if two.any():
        out["pair"] = {"first_finisher_kloop_median": float(np.median(fin_first[two])), "last_finisher_kloop_median": float(np.median(fin_last[two]))}
    return out


def tail_report(raw, hw, steps):
    r = tail_analysis(raw, hw, steps)
    print("   placement: %d XCDs, %d CUs, %d SIMDs hold the launch; waves per SIMD %s" % (r["xcds"], r["cus"], r["simds"], r["waves_per_simd"]))
    print("   entry skew inside an XCD (cycles after its first wave): median %.0f  p90 %.0f  max %.0f" % (r["entry_skew"]["median"], r["entry_skew"]["p90"], r["entry_skew"]["max"]))
    print("   launch length per XCD (first entry -> last epilogue store issued): %s" % " ".join("%d:%.0f" % kv for kv in sorted(r["launch_cycles_per_xcd"].items())))
    w = r["simd_window"]
    print("   a SIMD's K-loop window (first landing -> its last wave's loop done): median %.0f  p90 %.0f  max %.0f; MFMA issue of its waves %.0f -> issue share %.2f" % (
        w["median"], w["p90"], w["max"], w["mfma_issue_median"], w["issue_share_median"]))
    t = r["tail"]
    print("   K loop done per SIMD, launch-relative: median %.0f  p90 %.0f  max %.0f; the slowest tenth: waves per SIMD %s, per XCD %s, slow SIMDs per CU %s, corr(entry, end) %s" % (
        t["median_simd_kloop_end"], t["p90"], t["max"], t["slow_simds_waves"], t["slow_simds_per_xcd"], t["slow_simds_per_cu_hist"],
        "%.2f" % t["corr_entry_vs_end"] if "corr_entry_vs_end" in t else "-"))
    if "pair" in r:
        print("   two waves on a SIMD: the first leaves the loop after %.0f cycles, the second after %.0f (median)" % (r["pair"]["first_finisher_kloop_median"], r["pair"]["last_finisher_kloop_median"]))


def main():
    c, h, nb, sp = [int(a) for a in sys.argv[1:5]]
    b = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    tile = 4000 + nb                                         # nb = 3: tile 4003, one channel block per wave with the U fragments in registers; 4: tile 4004, four register sets
    if nb in (3, 4):
        nb = 1
    lib, dev = N.load(), torch.device("cuda:0")
    x = torch.randn(b, h, h, c, device=dev)
    wu = torch.randn(16 * c * c, device=dev) * 0.02
    sc, sh = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    res = torch.randn(b, h, h, c, device=dev)
    out = torch.empty(b, h, h, c, device=dev)
    blocks = b * (h // 8) * (h // 16) * (c // (32 * nb)) * sp
    slab = sp * b * h * h * c * 4 if sp > 1 else 0
    ncnt = b * (h // 8) * (h // 16) * (c // (32 * nb))
    used = (slab + ncnt * 4 + 255) // 256 * 256            # where the library puts the stamps: behind slabs and arrival counters
    scr = torch.zeros(used + blocks * 4 * 8 * 8, dtype=torch.uint8, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    run = lambda: N.check(lib.lspf2f_conv3x3(p(x), None, p(wu), p(sc), p(sh), p(res), p(out), b, h, h, c, 0, c, 1, 0, 1, tile, 0, sp, -1, 0, p(scr), scr.numel(), st))
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    raw = scr[used:].view(torch.int64).view(blocks, 4, 8).cpu().numpy()
    if not raw.any():
        print("no stamps: the library was not built with -DLSPF2F_WINO_STAMPS"); return
    hw = None
    if (raw[:, :, 6] < 0).all():                              # bit 63: slot 6 holds the wave's place instead of a split-K stamp (WSTAMP_FLUSH)
        hw = raw[:, :, 6].copy()
        raw[:, :, 6] = 0
    t = raw.astype(np.float64)
    names = ["entry", "prologue done (descriptors, epilogue operands requested)", "first step landed + barrier", "K loop done", "Z patch written + barrier",
             "epilogue stores issued", "ticket taken (split-K)", "combine done (last arriver)"]
    print("c%d h%d nb%d%s splits %d batch %d: %d workgroups, %.1f us per launch (eager, weights warm)" % (c, h, nb, " (U in registers)" if tile == 4003 else " (U in registers, four sets)" if tile == 4004 else "", sp, b, blocks, us))
    d = t - t[:, :, :1]
    for i, n_ in enumerate(names):
        col = d[:, :, i][t[:, :, i] != 0]
        if col.size:
            print("   %-62s %8.0f %8.0f   (%d waves)" % (n_, np.median(col), np.percentile(col, 90), col.size))
    k = t[:, :, 3] - t[:, :, 2]
    steps = (c // 8 + sp - 1) // sp
    print("   K loop: median %.0f cycles for %d steps = %.0f per step; MFMA issue per step and wave = %d cycles" % (np.median(k), steps, np.median(k) / steps, 64 * 16 * nb))
    if hw is not None:
        tail_report(raw, hw, steps)
        return
    for xcd in range(8):          # older stamp builds without the place word: block b is ASSUMED on XCD b % 8 (each XCD has its own counter)
        sub = t[xcd::8]
        t0 = sub[:, :, 0].min()
        last = np.where(sub[:, :, 7] != 0, sub[:, :, 7], np.where(sub[:, :, 6] != 0, sub[:, :, 6], sub[:, :, 5])).max()
        print("   XCD %d: first wave start -> last wave end %d cycles; spread of starts %d" % (xcd, last - t0, sub[:, :, 0].max() - t0))


if __name__ == "__main__":
    main()
