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


# The gradient collective in TWO calls (round 5; SURVEY 8e "overlapped with the tail of backward").  With BPTT every weight is used in every
# rollout call, so no gradient is final before the last call's backward has passed its layer -- but the dense linears of the blocks (85 % of
# the bucket) only RECORD their operands there: their weight gradients run as shared launches when the backward pass ends
# (autograd.flush_deferred_wgrads: ~1.5 ms of the 9.4 ms step at cfg3), followed by the LayerNorm folds.  Everything else -- encoder, decoders,
# propagators, FiLM, embeddings -- is final when that flush STARTS.  So: `early(lo, hi)` (called by the flush, with the element range its
# launches will write) all-reduces flat[:lo] and flat[hi:] on a side stream while the weight-gradient launches run, `finish()` all-reduces
# flat[lo:hi] behind them and joins.  Every element goes through exactly one summed all-reduce; over two ranks the result is bit-identical
# to one call (a + b commutes), over more ranks it is the same sum in RCCL's ring order for that message.
SPLIT_ALLREDUCE = True


class GradAllReduce:
    def __init__(self, flat: torch.Tensor):
        self.flat, self.range, self.side, self.calls = flat, None, None, []

    def active(self) -> bool:
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVE)

    def early(self, lo: int, hi: int):
        """All-reduce what lies OUTSIDE [lo, hi) now, beside whatever the caller's stream does next."""
        if not (self.active() and SPLIT_ALLREDUCE) or self.range is not None:
            return
        n = self.flat.numel()
        lo, hi = max(0, min(lo, n)), max(0, min(hi, n))
        if lo >= hi or (lo == 0 and hi == n):
            return
        self.range = (lo, hi)
        parts = [q for q in (self.flat[:lo], self.flat[hi:]) if q.numel()]
        if self.flat.is_cuda:
            if self.side is None:
                self.side = torch.cuda.Stream(device=self.flat.device)
            self.side.wait_stream(torch.cuda.current_stream(self.flat.device))
            with torch.cuda.stream(self.side):
                for q in parts:
                    dist.all_reduce(q, op=dist.ReduceOp.SUM)
        else:
            for q in parts:
                dist.all_reduce(q, op=dist.ReduceOp.SUM)
        self.calls += [int(q.numel()) for q in parts]

    def finish(self) -> torch.Tensor:
        """The rest (everything when early() did not run), then the caller's stream waits for the side stream."""
        if not self.active():
            return self.flat
        if self.range is None:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.calls = [int(self.flat.numel())]
        else:
            lo, hi = self.range
            dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM)
            self.calls.append(hi - lo)
            if self.side is not None:
                torch.cuda.current_stream(self.flat.device).wait_stream(self.side)
        self.range = None
        return self.flat


LAST_CALLS: list = []      # element counts of the last step's all-reduce calls (bench.py prints them in `collective`)


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
