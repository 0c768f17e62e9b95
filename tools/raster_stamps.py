#!/usr/bin/env python3
"""Phase stamps of the edge-map kernel (GPU; library built with -DLSPRASTER_STAMPS by tools/sessions/lastconv_ablate.sh and swapped in by the session script).
Prints, per band workgroup of one frame, the shader cycles spent in: edge setup (points -> quad), planning (outline sides + fill walk), drawing,
the barrier behind it (waiting for the slowest wave), expansion to the output tensor."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livespeechportraits_amd import _native as N
from livespeechportraits_amd.feature_map import FeatureMapRasteriser
dev = torch.device("cuda:0"); r = FeatureMapRasteriser(512, 18, dev)
lib = ctypes.CDLL(N.LIB_PATH)
if not hasattr(lib, "lspraster_debug_stamps"):
    print("no stamps: the library was not built with -DLSPRASTER_STAMPS"); sys.exit(0)
rng = np.random.default_rng(0)
for name, sd in (("face-sized spread (sigma 30 px)", 0.06), ("all edges in one band (sigma 10 px)", 0.02)):
    lm = (256 + rng.normal(0, 512 * sd, (1, 73, 2))).astype(np.float32)
    sh = np.tile(np.stack([np.linspace(0, 512, 18), np.full(18, 460.)], 1)[None], (1, 1, 1)).astype(np.float32)
    pts = torch.from_numpy(np.concatenate([lm, sh], 1)).to(dev).contiguous(); out = torch.empty(1, 1, 512, 512, device=dev)
    for _ in range(3):
        r.rasterise_points(pts, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    r.rasterise_points(pts, out=out)
    e1.record(); torch.cuda.synchronize()
    st = (ctypes.c_ulonglong * 512)()
    assert lib.lspraster_debug_stamps(st) == 0
    t = np.array(st, dtype=np.float64).reshape(64, 8)[:8]
    print("%s: one eager launch %.1f us" % (name, e0.elapsed_time(e1) * 1000))
    print("   band   setup  planning count+scan   drawing   barrier    expand     total   (shader cycles)")
    for b in range(8):
        scan_end = t[b, 6] if t[b, 6] else t[b, 2]           # stamp 6 exists in the dealt-steps form only
        print("   %4d %7.0f %9.0f %10.0f %9.0f %9.0f %9.0f %9.0f" % (b, t[b, 1] - t[b, 0], t[b, 2] - t[b, 1], scan_end - t[b, 2], t[b, 3] - scan_end,
                                                                    t[b, 4] - t[b, 3], t[b, 5] - t[b, 4], t[b, 5] - t[b, 0]))
    print("   first start -> last end over the 8 bands: %.0f cycles" % (t[:, 5].max() - t[:, 0].min()))
