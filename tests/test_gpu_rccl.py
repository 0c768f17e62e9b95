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


CHILD2 = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from livespeechportraits_amd import distributed as D, synth
from livespeechportraits_amd.engine import Engine
rank, world, local = D.init_process_group()          # LSP_DIST_BACKEND=gloo: two ranks share the one GPU of the box (RCCL refuses that)
assert dist.get_backend() == "gloo" and world == 2
dev = torch.device("cuda:0")
T, CHUNK = 687, 8                                    # frames of data/Input/00083.wav at 60 fps (demo.py:260-272); shards of 344 / 343 -> rank 1 ends on a 7-frame chunk
topo, sd = synth.synthetic("normal", ngf=32, num_downs=5, size=64)
eng = Engine("normal", ngf=32, num_downs=5, size=64, max_batch=CHUNK)
D.setup_engine(eng, sd if rank == 0 else None, dev)  # only rank 0 holds the state dict; the blob arrives by broadcast
feat, cand = synth.make_inputs(T, 64, seed=21, cand_batch=1)
feat_d = torch.from_numpy(feat).to(dev)
cand_d = D.broadcast_tensor(torch.from_numpy(cand) if rank == 0 else None, cand.shape, torch.float32, dev)
full = D.render_sharded(eng, feat_d, cand_d, gather=True, chunk=CHUNK)
local_only = D.render_sharded(eng, feat_d, cand_d, gather=False, chunk=CHUNK)
# the single-rank frames: one fresh engine with its own state dict renders every shard's chunks (a plan's tiling depends on the batch it
# is given, so the reference takes the same chunk boundaries)
ref = Engine("normal", ngf=32, num_downs=5, size=64, max_batch=CHUNK)
ref.load_state_dict(sd); ref.bind(ref.pack(), dev)
cand_ref = torch.from_numpy(cand).to(dev)
want = []
for r in range(world):
    lo, hi = D.shard_range(T, r, world)
    want += [ref.forward(feat_d[i:min(i + CHUNK, hi)].contiguous(), cand_ref) for i in range(lo, hi, CHUNK)]
want = torch.cat(want)
lo, hi = D.shard_range(T, rank, world)
distinct = len({float(x) for x in want[:, 0].reshape(T, -1).sum(1).cpu()})
print(json.dumps({"rank": rank, "frames": int(full.shape[0]), "gather_equal": bool(torch.equal(full, want)),
                  "local_equal": bool(torch.equal(local_only, want[lo:hi])), "ragged_last_chunk": (hi - lo) %% CHUNK, "distinct_frames": distinct}))
dist.barrier()
dist.destroy_process_group()
"""


def test_two_gloo_ranks_render_a_687_frame_clip_sharded_and_gathered_bit_for_bit():
    """The N > 1 control flow of distributed.py ON the GPU: weights by one broadcast from the only rank that holds the state dict, the shared
    candidate stack by another, frames sharded 344 / 343 (a ragged 7-frame last chunk on rank 1, a padded all_gather), every gathered frame
    equal bit for bit to a single engine's.  Two ranks share the box's one MI355X through gloo -- RCCL itself needs one device per rank and
    meets N > 1 for the first time in the driver's 8-GPU run.  Replaces models/networks.py:392-401, demo.py:260-272."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   LSP_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", CHILD2 % {"root": ROOT}], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-1500:]
    recs = [json.loads([ln for ln in so.splitlines() if ln.startswith("{")][-1]) for so, _ in outs]
    print(recs)
    assert sorted(r["rank"] for r in recs) == [0, 1]
    for r in recs:
        assert r["frames"] == 687 and r["gather_equal"] and r["local_equal"] and r["distinct_frames"] > 600
    assert [r["ragged_last_chunk"] for r in sorted(recs, key=lambda r: r["rank"])] == [0, 7]
