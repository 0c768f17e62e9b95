"""One process per GPU, frames sharded, weights broadcast once -- the MI355X replacement for
the reference's ``nn.DataParallel`` (models/networks.py:392-401), which re-broadcasts all
parameters (487 MB for 'large') on EVERY forward and scatters/gathers the batch through one
Python process.

Here: rank 0 reads the checkpoint and packs it on the host (BN fold + layout, C++), the packed
blob goes to every other GPU with ONE ``broadcast`` (backend 'nccl' = RCCL over xGMI), and each
rank then renders its own contiguous slice of the frame range with zero per-frame collectives
(frames are independent: BatchNorm is in eval mode and the generator is stateless --
SURVEY.md 8e).  An optional all_gather returns all frames to every rank.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def env_rank() -> Tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_process_group(backend: Optional[str] = None, force: bool = False) -> Tuple[int, int, int]:
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment (torchrun).  A single process needs no group;
    ``force`` creates one anyway (world size 1), which is how the RCCL path is exercised on a 1-GPU box."""
    rank, world, local = env_rank()
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            # LSP_DIST_BACKEND=gloo lets several ranks share ONE GPU (RCCL refuses that): used by the tests to run the
            # multi-rank control flow on a single-GPU box.  Production: nccl (= RCCL over xGMI).
            backend = os.environ.get("LSP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) slice of ``total`` frames for ``rank`` (first
    ``total % world`` ranks get one extra frame)."""
    if world < 1 or not (0 <= rank < world) or total < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def broadcast_blob(blob: Optional[torch.Tensor], nbytes: int, device: torch.device, src: int = 0) -> torch.Tensor:
    """Every rank returns a uint8 tensor of ``nbytes`` on ``device`` holding rank ``src``'s blob.
    One collective, start-up only."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
    if rank == src:
        if blob is None or blob.numel() != nbytes:
            raise ValueError("source rank must supply the packed blob")
        buf.copy_(blob)
    if dist.is_initialized():          # also at world size 1: the collective then runs through the backend (RCCL) for real
        dist.broadcast(buf, src=src)
    return buf


def broadcast_tensor(t: Optional[torch.Tensor], shape, dtype: torch.dtype, device: torch.device, src: int = 0) -> torch.Tensor:
    """Every rank returns a ``shape`` / ``dtype`` tensor on ``device`` holding rank ``src``'s ``t`` (the one candidate stack all
    ranks render with: BASELINE.json configs[3]).  One collective, start-up only; runs through the backend whenever a group exists."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    buf = torch.empty(tuple(shape), dtype=dtype, device=device)
    if rank == src:
        if t is None or tuple(t.shape) != tuple(shape):
            raise ValueError("source rank must supply the tensor to broadcast")
        buf.copy_(t)
    if dist.is_initialized():
        dist.broadcast(buf, src=src)
    return buf


def setup_engine(engine, state_dict, device: torch.device, src: int = 0) -> None:
    """Pack on ``src`` (the only rank that needs the state dict), broadcast, bind everywhere."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    blob = None
    if rank == src:
        if state_dict is None:
            raise ValueError("rank %d must hold the state dict" % src)
        engine.load_state_dict(state_dict)
        blob = engine.pack()
    nbytes = engine.packed_bytes()
    if dist.is_initialized():
        # The blob layout depends on the handle's configuration AND on the batch range it plans for (the blob carries only the weight forms the plans of batch
        # 1..max_batch read): ranks built with different max_batch would enter a size-mismatched collective.  One tiny all_reduce (min and max of
        # (bytes, max_batch, frame size)) turns that hang / corruption into an error on every rank.
        mine = torch.tensor([nbytes, int(getattr(engine, "max_batch", 0)), int(getattr(engine, "size", 0))], dtype=torch.int64,
                            device=device if dist.get_backend() == "nccl" else "cpu")
        lo, hi = mine.clone(), mine.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if not torch.equal(lo, hi):
            raise RuntimeError("ranks disagree on the packed blob: (bytes, max_batch, frame size) here %s, across ranks min %s max %s -- build every rank's engine "
                               "with the same configuration and max_batch" % (mine.tolist(), lo.tolist(), hi.tolist()))
    engine.bind(broadcast_blob(blob, nbytes, device, src))


def render_sharded(engine, feature_maps: torch.Tensor, cand_image: torch.Tensor, gather: bool = False,
                   chunk: Optional[int] = None) -> torch.Tensor:
    """``feature_maps`` [T,1,H,W] is the GLOBAL frame list (same on every rank, on this
    rank's device); each rank renders frames shard_range(T, rank, world) in chunks of
    ``chunk`` (default engine.max_batch).  Returns the local frames, or all T frames on
    every rank when ``gather``: one all_gather_into_tensor of ceil(T/world) x 3 x H x W floats per rank (shorter shards are padded
    for the collective and the padding dropped afterwards, so T need not divide the world size).  The collective runs whenever a
    process group exists, world size 1 included -- that is how the RCCL path is exercised on a 1-GPU box."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    total = feature_maps.shape[0]
    lo, hi = shard_range(total, rank, world)
    chunk = chunk or engine.max_batch
    outs = [engine.forward(feature_maps[i:min(i + chunk, hi)].contiguous(), cand_image)
            for i in range(lo, hi, chunk)]
    local = torch.cat(outs) if outs else feature_maps.new_empty((0, engine.output_nc) + tuple(feature_maps.shape[2:]))
    if not gather or not dist.is_initialized():
        return local
    per = (total + world - 1) // world                       # the longest shard
    if per == 0:
        return local
    send = local
    if local.shape[0] != per:
        send = local.new_zeros((per,) + tuple(local.shape[1:]))
        send[:local.shape[0]] = local
    full = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(full, send.contiguous())
    if world * per == total:
        return full
    spans = [shard_range(total, r, world) for r in range(world)]
    return torch.cat([full[r * per:r * per + (hi - lo)] for r, (lo, hi) in enumerate(spans)])
