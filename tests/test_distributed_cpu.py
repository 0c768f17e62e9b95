"""CPU, world_size 2, gloo: the N > 1 path -- one weight broadcast, frame sharding, gather."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_range_partitions_exactly():
    from livespeechportraits_amd.distributed import shard_range
    for total in (0, 1, 7, 8, 64, 687):          # 687 = frames of data/Input/00083.wav at 60 fps
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from livespeechportraits_amd import distributed as D, synth
    from livespeechportraits_amd.engine import Engine
    r, w, _ = D.init_process_group("gloo")
    assert (r, w) == (rank, world)
    topo, sd = synth.synthetic("normal", ngf=32, num_downs=5, size=64)
    eng = Engine("normal", ngf=32, num_downs=5, size=64, max_batch=4)
    # only rank 0 touches the state dict; everyone ends up with the same packed blob
    blob = None
    if rank == 0:
        eng.load_state_dict(sd)
        blob = eng.pack()
    buf = D.broadcast_blob(blob, eng.packed_bytes(), torch.device("cpu"))
    ref = Engine("normal", ngf=32, num_downs=5, size=64, max_batch=4)      # the same configuration: the blob carries the weight forms of the plans of 1 .. max_batch frames
    ref.load_state_dict(sd)
    assert torch.equal(buf, ref.pack()), "rank %d received a different blob" % rank
    # ranks whose engines plan for different batch ranges would enter a size-mismatched broadcast: setup_engine refuses on EVERY rank first
    mis = Engine("normal", ngf=32, num_downs=5, size=64, max_batch=4 if rank == 0 else 8)
    try:
        D.setup_engine(mis, sd if rank == 0 else None, torch.device("cpu"))
        raised = False
    except RuntimeError as ex:
        raised = "ranks disagree" in str(ex)
    assert raised, "rank %d: mismatched max_batch went through" % rank
    # binding needs a device: on CPU this must fail loudly, never fall back
    try:
        eng.bind(buf)
        ok = False
    except Exception:
        ok = True
    assert ok
    # frame sharding + gather bookkeeping with a stand-in renderer (frame id -> constant image)
    class Fake:
        max_batch, output_nc = 3, 3
        def forward(self, feat, cand):
            return feat.expand(-1, 3, -1, -1) * 2.0
    frames = torch.arange(8, dtype=torch.float32).view(8, 1, 1, 1).expand(8, 1, 4, 4).contiguous()
    local = D.render_sharded(Fake(), frames, None)
    lo, hi = D.shard_range(8, rank, world)
    assert local.shape[0] == hi - lo and torch.equal(local[:, 0, 0, 0], 2.0 * torch.arange(lo, hi, dtype=torch.float32))
    full = D.render_sharded(Fake(), frames, None, gather=True)
    assert torch.equal(full[:, 0, 0, 0], 2.0 * torch.arange(8, dtype=torch.float32))
    # a frame count that does not divide the world size: shards of 4 and 3 frames, padded for the collective, padding dropped
    frames7 = frames[:7].contiguous()
    full7 = D.render_sharded(Fake(), frames7, None, gather=True)
    assert full7.shape[0] == 7 and torch.equal(full7[:, 0, 0, 0], 2.0 * torch.arange(7, dtype=torch.float32))
    # one shared tensor from rank 0 (the candidate stack of BASELINE.json configs[3])
    t = D.broadcast_tensor(torch.full((2, 3), 5.0) if rank == 0 else None, (2, 3), torch.float32, torch.device("cpu"))
    assert torch.equal(t, torch.full((2, 3), 5.0))
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, "ok%d" % rank), "w").write("1")


def test_two_rank_broadcast_and_sharding(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok%d" % r)) for r in range(world))


def test_bench_launcher_refuses_to_run_without_devices():
    """bench.py --gpus N with no launcher environment becomes the launcher; with no ROCm device it must exit non-zero with a
    message and print no JSON line (never a silent world-size-1 record)."""
    import subprocess
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "no ROCm device" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    # a launcher that disagrees with --gpus is an error too
    env.update(RANK="0", WORLD_SIZE="3", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999", LSP_DIST_BACKEND="gloo")
    q = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert q.returncode != 0 and not [ln for ln in q.stdout.splitlines() if ln.startswith("{")]
