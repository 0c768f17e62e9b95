#!/usr/bin/env python3
"""Failure rate of the patch-staged kernel on one shape, repeated launches (LSP_HIP_DBG picks the ablation arm of a -DLSPF2F_ABLATE build)"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import test_gpu_conv as T

def main():
    b, cin, cout, h, tw, bn = [int(x) for x in sys.argv[1:7]]
    reps = int(sys.argv[7]) if len(sys.argv) > 7 else 10
    dev = torch.device("cuda:0")
    x0 = T.bf16r(T.rnd(b, cin, h, h, seed=171)); w = T.rnd(cout, cin, 3, 3, seed=172) * 0.05
    ref = T.ref_conv(x0, None, T.bf16r(w), None, None, None, 1, False, False)
    nbad = 0; where = set()
    for rep in range(reps):
        got = T.run_conv(dev, x0, None, w, None, None, None, 1, 0, False, (7000 + tw, bn), 0, 0, dtype=1)
        bad = (got - ref).abs() > (ref.abs() * 2.0 ** -8 + 1e-3)
        if bad.any():
            nbad += 1
            idx = bad.nonzero()
            where.add((idx[:, 1].min().item(), idx[:, 1].max().item(), tuple(sorted(set((idx[:, 2] % (256 // tw)).tolist())))))
    print("dbg=%s b%d c%d o%d h%d tw%d bn%d: %d of %d launches wrong %s" % (os.environ.get("LSP_HIP_DBG", "0"), b, cin, cout, h, tw, bn, nbad, reps, sorted(where)))

main()
