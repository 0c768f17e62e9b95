#!/usr/bin/env python3
"""Pins the landmark edge map on REAL OpenCV the moment it is importable (it is not in this image: parity of the rasteriser is
UNPINNED until this script has run somewhere and its fixture is committed).

  python oracle/make_golden_raster.py [--out tests/golden]

Draws the reference's edge maps -- datasets/face_dataset.py:297-322: cv2.line(img, pt1, pt2, 255, 2) over part_list and the two shoulder
chains -- with cv2 itself for the deterministic landmark sets below (the same generator the GPU tests use), checks oracle/raster_oracle.c
against them bit for bit, and writes tests/golden/raster_cv2.npz (inputs + the uint8 images, run-length packed) + a .json with the cv2
version.  tests/test_raster.py::test_edge_map_matches_real_opencv compares the oracle (CPU suite) and the HIP kernel (GPU suite) with
that fixture when it exists and reports "unpinned" (xfail) when it does not.  If cv2 is missing the script says so and exits 2."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = [("s512", 512, 12, 0), ("s256", 256, 6, 1), ("s1024", 1024, 3, 2)]     # name, frame size, landmark sets, seed


def landmark_sets(size, n, seed):
    """deterministic landmark / shoulder sets incl. points outside the frame (clipping) and repeated points (zero-length edges)"""
    rng = np.random.default_rng(1000 + seed)
    lm = size * 0.5 + rng.normal(0, size * 0.17, (n, 73, 2))
    lm[:, ::11] = np.floor(lm[:, ::11])                      # some integer-valued coordinates
    lm[0, :5] = [[-20.5, 10.2], [size + 30.1, 5.0], [3.3, size + 9.9], [-1.0, -1.0], [size - 1, size - 1]]
    sh = np.stack([np.linspace(-10, size + 10, 18), size * 0.9 + rng.normal(0, size * 0.03, 18)], 1)[None].repeat(n, 0)
    return lm, sh


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    a = ap.parse_args()
    try:
        import cv2
    except ImportError:
        print("make_golden_raster: OpenCV (cv2) is not importable here -- nothing written; the rasteriser stays parity-unpinned "
              "(requirements.txt of the reference pins opencv_python==4.4.0.40)", file=sys.stderr)
        return 2
    from oracle import raster_oracle as RO
    arrays, mism = {}, 0
    for name, size, n, seed in CASES:
        lm, sh = landmark_sets(size, n, seed)
        imgs = np.zeros((n, size, size), np.uint8)
        for k in range(n):
            im = np.zeros((size, size), np.uint8)
            for part in RO.PART_LIST:                        # face_dataset.py:311-322
                for e in part:
                    for i in range(len(e) - 1):
                        p1 = tuple(int(v) for v in lm[k, e[i]])
                        p2 = tuple(int(v) for v in lm[k, e[i + 1]])
                        im = cv2.line(im, p1, p2, 255, 2)
            num = sh.shape[1] // 2                           # face_dataset.py:297-305
            for i in range(2):
                for j in range(num - 1):
                    p1 = tuple(int(v) for v in sh[k, i * num + j])
                    p2 = tuple(int(v) for v in sh[k, i * num + j + 1])
                    im = cv2.line(im, p1, p2, 255, 2)
            imgs[k] = im
            ours = RO.get_feature_image(lm[k], (size, size), sh[k], None)
            mism += int((ours != im).sum())
        arrays[name + "_bits"] = np.packbits(imgs > 0)
        arrays[name + "_shape"] = np.array(imgs.shape)
    print("oracle/raster_oracle.c vs cv2 %s: %d differing pixels" % (cv2.__version__, mism))
    os.makedirs(a.out, exist_ok=True)
    np.savez_compressed(os.path.join(a.out, "raster_cv2.npz"), **arrays)
    json.dump({"cv2_version": cv2.__version__, "cases": CASES, "oracle_mismatching_pixels": mism,
               "generator": "oracle/make_golden_raster.py: cv2.line(img, p1, p2, 255, 2), face_dataset.py:297-322"},
              open(os.path.join(a.out, "raster_cv2.json"), "w"), indent=1)
    return 0 if mism == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
