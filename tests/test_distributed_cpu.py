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
    ref = Engine("normal", ngf=32, num_downs=5, size=64)
    ref.load_state_dict(sd)
    assert torch.equal(buf, ref.pack()), "rank %d received a different blob" % rank
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
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, "ok%d" % rank), "w").write("1")


def test_two_rank_broadcast_and_sharding(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok%d" % r)) for r in range(world))
