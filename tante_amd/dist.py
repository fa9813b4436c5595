"""Data-parallel plumbing: one process per GPU, torch.distributed ("nccl" = RCCL over xGMI on ROCm; "gloo" on CPU).

The rollout path shards over batch with NO data-path collective (independent samples).  The train step has exactly one
collective: a summed all-reduce of the flat gradient bucket (optim.FlatAdamW.flat_g, 4.2 M fp32 = 16.9 MB for
configs/tante.yaml) before the clip + AdamW launch, which divides by the world size on the fly (grad_scale).
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch
import torch.distributed as dist


def init(backend: Optional[str] = None) -> tuple:
    """(rank, world, local_rank) from the torchrun environment; initialises the process group when world > 1."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_batch(batch: Dict[str, torch.Tensor], rank: int, world: int) -> Dict[str, torch.Tensor]:
    """Contiguous batch shard of this rank (global batch must divide evenly, like drop_last=True in the reference's loaders)."""
    out = {}
    for k, v in batch.items():
        B = v.shape[0]
        if B % world:
            raise ValueError(f"global batch {B} does not divide over {world} ranks")
        per = B // world
        out[k] = v[rank * per:(rank + 1) * per]
    return out


# True: an initialised process group of ONE rank still issues the collective (the identity there).  tests/rccl_world1_child.py uses it
# to run RCCL's all-reduce beside the captured train-step graph on the single GPU a test box has.
FORCE_COLLECTIVE = False


def collective_needed(world: int) -> bool:
    return world > 1 or (FORCE_COLLECTIVE and dist.is_available() and dist.is_initialized())


def allreduce_sum_(flat: torch.Tensor) -> torch.Tensor:
    """In-place summed all-reduce of one flat bucket (a no-op for a single process unless FORCE_COLLECTIVE)."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVE):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def collective_info() -> Optional[str]:
    """What carries the gradient all-reduce in this process: backend, RCCL version, rank count (bench.py prints it)."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    be = dist.get_backend()
    ver = ""
    if be == "nccl":
        try:
            ver = " RCCL " + ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:      # noqa: BLE001
            ver = " RCCL (version unavailable)"
    return f"{be}{ver}, {dist.get_world_size()} rank(s)"


def broadcast_(flat: torch.Tensor, src: int = 0) -> torch.Tensor:
    """In-place broadcast of one flat bucket from rank `src` (a no-op for a single process)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat, src=src)
    return flat


def max_over_ranks(value: float, device=None) -> float:
    """Timing reduction of bench.py: the slowest rank defines the step time."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
