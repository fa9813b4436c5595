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


# The gradient collective PIPELINED against the end-of-pass weight-gradient flush (round 5: two calls; round 6: segments).  With BPTT every
# weight is used in every rollout call, so no gradient is final before the last call's backward has passed its layer -- but the dense
# linears of the blocks (85 % of the bucket) only RECORD their operands there: their weight gradients run as shared launches when the
# backward pass ends (autograd.flush_run: ~1.5 ms of the step at cfg3, one launch group per block, each followed by the LayerNorm folds it
# completes).  So: what no flush launch writes (encoder, decoders, propagators, FiLM, embeddings) is all-reduced on a side stream when the
# flush STARTS, the spans of flush segment s while segment s + 1 computes, the last segment's behind it.  Every element goes through exactly
# one summed all-reduce; over two ranks the result is bit-identical to one call (a + b commutes), over more ranks it is the same sum in
# RCCL's ring order for that message.
# The call list must be the same on every rank.  It is derived from run-time state (which linears were recorded), so the ranks AGREE on it
# once (agree(): a MIN all-reduce of the plan's signature and of its negation) and fall back to ONE call, all of them, when they differ; a
# rank whose plan later changes raises instead of issuing a different list.
SPLIT_ALLREDUCE = True
FLUSH_SEGMENTS = 3      # the flush in this many segments: the early part + two thirds of the flush's spans travel under compute

_SIDE_STREAMS = {}      # device index -> the one side stream the collectives of this process are issued under
_AGREED = {}            # bucket numel -> (signature, agreed?)


def _side_stream(device):
    st = _SIDE_STREAMS.get(device.index)
    if st is None:
        st = _SIDE_STREAMS[device.index] = torch.cuda.Stream(device=device)
    return st


def plan_signature(plan) -> tuple:
    if plan is None:
        return (-1,)
    early, segs = plan
    flat = [len(early)] + [v for r in early for v in r] + [len(segs)]
    for sg in segs:
        flat += [len(sg)] + [v for r in sg for v in r]
    return tuple(int(v) for v in flat)


def agree(flat: torch.Tensor, plan) -> bool:
    """True when every rank holds this same plan (a collective the first time a bucket is seen; afterwards a local look-up).  A plan that
    differs from the one agreed earlier raises: the other ranks will not be asking again."""
    sig = plan_signature(plan)
    hit = _AGREED.get(flat.numel())
    if hit is not None:
        if hit[0] != sig:
            raise RuntimeError("the gradient all-reduce plan of this rank changed after the ranks agreed on it (a different set of "
                               "recorded weight gradients: ragged local batch or an option flipped on one rank?)")
        return hit[1]
    h = hash(sig)
    v = [len(sig), h & 0x7FFFFFFF, (h >> 31) & 0x7FFFFFFF]
    dev = flat.device if (dist.get_backend() == "nccl") else "cpu"
    t = torch.tensor(v + [-q for q in v], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    t = t.tolist()
    same = all(t[i] == -t[i + 3] for i in range(3))
    _AGREED[flat.numel()] = (sig, same)
    return same


class GradAllReduce:
    def __init__(self, flat: torch.Tensor):
        self.flat, self.done, self.calls, self.used_side = flat, [], [], False

    def active(self) -> bool:
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVE)

    def ranges(self, rs):
        """All-reduce these disjoint (lo, hi) element ranges now -- on the side stream, behind whatever the caller's stream has issued so
        far, beside whatever it issues next."""
        rs = [(int(lo), int(hi)) for lo, hi in rs if hi > lo]
        if not rs or not self.active():
            return
        for lo, hi in rs:
            if any(lo < h and l < hi for l, h in self.done):
                raise RuntimeError("GradAllReduce: an element would pass through two all-reduces")
            self.done.append((lo, hi))
        if self.flat.is_cuda:
            side = _side_stream(self.flat.device)
            side.wait_stream(torch.cuda.current_stream(self.flat.device))
            self.used_side = True
            with torch.cuda.stream(side):
                for lo, hi in rs:
                    dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM)
        else:
            for lo, hi in rs:
                dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM)
        self.calls += [hi - lo for lo, hi in rs]

    def early(self, lo: int, hi: int):
        """All-reduce what lies OUTSIDE [lo, hi) now (the two-call form of round 5)."""
        if not (self.active() and SPLIT_ALLREDUCE) or self.done:
            return
        n = self.flat.numel()
        lo, hi = max(0, min(lo, n)), max(0, min(hi, n))
        if lo >= hi or (lo == 0 and hi == n):
            return
        self.ranges([(0, lo), (hi, n)])

    def finish(self) -> torch.Tensor:
        """Whatever has not travelled yet (everything when nothing has), then the caller's stream waits for the side stream."""
        if not self.active():
            return self.flat
        n, pos, rest = self.flat.numel(), 0, []
        LAST_OVERLAPPED[0] = sum(self.calls) / max(1, n)      # the share that travelled beside compute (before this call)
        for lo, hi in sorted(self.done):
            if lo > pos:
                rest.append((pos, lo))
            pos = max(pos, hi)
        if pos < n:
            rest.append((pos, n))
        if not self.done:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.calls = [n]
        else:
            self.ranges(rest)
        if self.used_side:
            torch.cuda.current_stream(self.flat.device).wait_stream(_side_stream(self.flat.device))
        self.done = []
        return self.flat


LAST_OVERLAPPED = [0.0]  # share of the bucket's elements all-reduced BEFORE finish() in the last step (issued beside the flush)
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
