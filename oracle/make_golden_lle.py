#!/usr/bin/env python3
"""ORACLE tooling -- freeze outputs of the reference's OWN manifold-projection functions into tests/golden/lle_*.npz.
Container only (needs /root/reference).  funcs/utils.py is imported unmodified; the two modules it pulls in that do
not exist here (librosa via funcs/audio_funcs.py:9,18) are stubbed -- none of the functions used touch them.
Asserts oracle/lle_oracle.py is bit-identical to the reference functions."""
import argparse
import json
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)

from livespeechportraits_amd import synth          # noqa: E402
from oracle import lle_oracle                       # noqa: E402

# name -> (m, n, d, intrinsic dim, K, percent, noise)
CASES = {
    "iso_k10": (3000, 96, 512, 0, 10, 1.0, 0.05),         # demo.py defaults: Knear 10, LLE_percent 1 (config/May.yaml:8-10)
    "manifold_k10": (4000, 96, 512, 24, 10, 0.7, 0.05),   # points near a 24-d manifold
    "degenerate_k10": (4000, 96, 512, 6, 10, 1.0, 0.003), # 9 neighbour differences in a ~6-d subspace: near-singular A^T A
    "k1": (500, 16, 64, 0, 1, 1.0, 0.05),                 # K == 1 branch (utils.py:143-145)
    "k16_ragged": (1037, 33, 96, 0, 16, 0.5, 0.05),       # sizes that are not multiples of any tile
}


def main():
    if not os.path.isdir(REF):
        raise SystemExit("make_golden_lle.py needs /root/reference (build container only)")
    for name in ("librosa", "librosa.filters"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["librosa.filters"].mel = None
    sys.path.insert(0, REF)
    from funcs import utils                         # the reference module
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"), help="directory the fixtures are written to")
    out = ap.parse_args().out
    os.makedirs(out, exist_ok=True)
    for name, (m, n, d, intr, K, pct, noise) in CASES.items():
        db, q = synth.make_feature_database(m, n, d, intr, noise=noise)
        ind = utils.KNN_with_torch(q, db, K=K)
        w, fuse = utils.compute_LLE_projection_all_frame(q, db, ind, n)
        res = q * (1 - pct) + fuse * pct            # demo.py:200
        assert np.array_equal(lle_oracle.knn(q, db, K), ind)
        ow, of = lle_oracle.lle_all(q, db, ind)
        assert np.array_equal(ow, w) and np.array_equal(of, fuse) and np.array_equal(lle_oracle.blend(q, of, pct), res)
        # margins that decide how tight a parity test can be
        f64 = q.astype(np.float64); b64 = db.astype(np.float64)
        dd = ((f64 ** 2).sum(1)[:, None] + (b64 ** 2).sum(1)[None, :] - 2 * f64 @ b64.T)
        srt = np.sort(dd, 1)
        gap = ((srt[:, 1:K + 1] - srt[:, :K]) / srt[:, :K + 1].max(1, keepdims=True)).min() if K < m else 0.0
        conds = [np.linalg.cond(((db[i[1:]] - db[i[0]]) @ (db[i[1:]] - db[i[0]]).T).astype(np.float64)) for i in ind] if K > 1 else [1.0]
        # how far the reference's own fp32 solve is from exact arithmetic on the same neighbours: the noise floor of any parity claim
        w64 = np.zeros_like(w); f64fuse = np.zeros((n, d))
        for i in range(n):
            base = b64[ind[i]]
            if K == 1:
                w64[i] = 1; f64fuse[i] = base[0]; continue
            A = (base[1:] - base[0]).T
            w64[i, 1:] = np.linalg.lstsq(A, f64[i] - base[0], rcond=None)[0]; w64[i, 0] = 1 - w64[i, 1:].sum()
            f64fuse[i] = w64[i] @ base
        self_w, self_f = float(np.abs(w - w64).max()), float(np.abs(fuse - f64fuse).max())
        print("%-15s m %5d n %3d d %3d K %2d  min rel. distance gap %.1e  cond(A^T A) median %.1e max %.1e  |w| max %.2f  reference fp32 vs exact: w %.1e fuse %.1e"
              % (name, m, n, d, K, gap, np.median(conds), np.max(conds), np.abs(w).max(), self_w, self_f))
        np.savez_compressed(os.path.join(out, "lle_%s.npz" % name), ind=ind, w=w, fuse=fuse, blend=res.astype(np.float32))
        json.dump({"m": m, "n": n, "d": d, "intrinsic": intr, "K": K, "percent": pct, "noise": noise, "seed": 3,
                   "min_rel_distance_gap": float(gap), "cond_max": float(np.max(conds)),
                   "reference_vs_exact_w": self_w, "reference_vs_exact_fuse": self_f},
                  open(os.path.join(out, "lle_%s.json" % name), "w"), indent=1)


if __name__ == "__main__":
    main()
