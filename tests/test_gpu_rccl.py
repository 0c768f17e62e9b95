"""RCCL readiness on a 1-GPU box: backend 'nccl' (= RCCL on ROCm) is initialised for real at world size 1
and the packed 511 MB `large` blob goes through dist.broadcast ON THE DEVICE -- this loads librccl, exercises the
set_device-before-init ordering and the uint8 tensor size the 8-GPU run will use -- then a frame is rendered from the
RECEIVED buffer and compared with the reference golden.  Runs in a child process so the process group never leaks into the
other tests.  Replaces models/networks.py:392-401 (nn.DataParallel's per-forward parameter broadcast)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
from conftest import golden_problem
from livespeechportraits_amd import distributed as D
from livespeechportraits_amd.engine import Engine
rank, world, local = D.init_process_group("nccl", force=True)
assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
dev = torch.device("cuda:%%d" %% local)
meta, arrays, topo, sd, feat, cand = golden_problem("large_512")
src = Engine("large", size=512)
src.load_state_dict(sd)
blob = src.pack()                                   # host blob, what rank 0 would hold
eng = Engine("large", size=512)                     # a rank that never sees the state dict
buf = D.broadcast_blob(blob, eng.packed_bytes(), dev)
torch.cuda.synchronize()
assert buf.is_cuda and buf.dtype == torch.uint8 and buf.numel() == blob.numel()
intact = bool(torch.equal(buf.cpu(), blob))
eng.bind(buf)                                       # used in place, no copy
out = eng.forward(torch.from_numpy(feat).to(dev), torch.from_numpy(cand).to(dev)).cpu().numpy()
err = float(np.abs(out - arrays["out"]).max())
# the all-gather of render_sharded(gather=True) through the same backend: with a process group up the collective runs at world
# size 1 too (all_gather_into_tensor on RCCL), and the shared candidate stack goes through broadcast_tensor
cand_d = D.broadcast_tensor(torch.from_numpy(cand), cand.shape, torch.float32, dev)
g = D.render_sharded(eng, torch.from_numpy(feat).to(dev), cand_d, gather=True)
libs = [l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l]
print(json.dumps({"intact": intact, "bytes": int(buf.numel()), "err": err, "gather_equal": bool(np.array_equal(g.cpu().numpy(), out)),
                  "rccl_mapped": sorted(set(libs))[:1]}))
dist.destroy_process_group()
"""


def test_nccl_backend_broadcasts_the_packed_blob_at_world_size_one():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("LSP_DIST_BACKEND", None)
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-1500:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    print(d)
    assert d["intact"] and d["bytes"] > 500e6
    assert d["err"] <= 5e-5
    assert d["gather_equal"]
    assert d["rccl_mapped"], "librccl was not mapped: the collective did not go through RCCL"
