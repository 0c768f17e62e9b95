"""ORACLE (test infrastructure) -- numpy restatement of the reference's frame post-processing.

Follows util/util.py:19-42 (tensor2im, normalize=True branch) as demo.py:268 calls it on
``pred_fake[0]``:  CHW float tensor in [-1,1] -> (transpose to HWC + 1) / 2.0 * 255.0 -> clip [0,255]
-> astype(uint8) (truncation).  numpy keeps the float32 dtype through the python-float scalars.
Only tests/ may import this.  Parity pin: the reference ships no vectors; checked against the
reference function itself by oracle/make_golden.py-style import in tests when /root/reference exists.
"""
import numpy as np


def tensor2im(chw: np.ndarray) -> np.ndarray:
    """chw: float32 [3,H,W] -> uint8 [H,W,3]"""
    img = np.asarray(chw, dtype=np.float32)[:3]
    img = (np.transpose(img, (1, 2, 0)) + 1) / 2.0 * 255.0
    img = np.clip(img, 0, 255)
    return img.astype(np.uint8)
