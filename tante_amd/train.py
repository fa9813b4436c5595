"""One optimisation step of the reference's Trainer.train_one_epoch (trainer/trainer.py:174-207) on the HIP path:

    y_pred, y_ref = rollout_model(model, batch, "train")          # BPTT through n_steps_output re-fed frames
    loss = MSE(y_pred, y_ref).mean()
    loss.backward(); clip_grad_norm_(1.0); AdamW.step(); zero_grad()

Data parallel: the batch is sharded over ranks (one process per GPU); the ONLY collective is a summed all-reduce of the flat
gradient bucket (RCCL over xGMI through torch.distributed), and the division by the world size is folded into the clip + AdamW
launch.  The clip therefore uses the norm of the AVERAGED gradient, exactly as a single process would on the full batch.
"""
from __future__ import annotations

from typing import Dict

import torch

from . import dist as D
from .autograd import MseMeanFn, run_backward
from .optim import FlatAdamW
from .metrics import MSE
from .rollout import rollout_adaptive, rollout_model


def train_step(model, opt: FlatAdamW, batch: Dict[str, torch.Tensor], formatter, n_steps_output: int, world: int = 1,
               lr: float = None, scaler=None) -> torch.Tensor:
    """scaler: a torch.amp.GradScaler -- the reference's fp16 AMP sequence (trainer/trainer.py:191-195):
    scale(loss).backward(); unscale_(opt); [clip, inside step()]; step(opt) -- skipped when a gradient is not finite --; update()."""
    opt.zero_grad()
    y_pred, y_ref = rollout_model(model, batch, formatter, n_steps_output)
    loss = MseMeanFn.apply(y_pred, y_ref)
    if scaler is not None and scaler.is_enabled():
        _backward_and_allreduce(scaler.scale(loss), opt, world)
        scaler.unscale_(opt)
        scaler.step(opt, grad_scale=1.0 / world, lr=lr)
        scaler.update()
        return loss.detach()
    _backward_and_allreduce(loss, opt, world)
    opt.step(grad_scale=1.0 / world, lr=lr)
    return loss.detach()


def _backward_and_allreduce(loss, opt: FlatAdamW, world: int):
    """loss.backward() and the summed all-reduce of the flat gradient bucket, the collective in two calls when the backward pass ends in a
    deferred weight-gradient flush (dist.GradAllReduce: what the flush does not write is reduced on a side stream WHILE it runs)."""
    from . import autograd as A
    if not D.collective_needed(world):
        run_backward(loss)
        return
    ar = D.GradAllReduce(opt.flat_g)

    def driver():      # the end-of-pass flush, run by the autograd engine's callback: segments of it under the all-reduce of the previous ones
        plan = A.flush_plan(opt.flat_g, D.FLUSH_SEGMENTS) if D.SPLIT_ALLREDUCE else None
        if plan is None or not D.agree(opt.flat_g, plan):
            A.flush_run(1)
            return
        early, segs = plan
        ar.ranges(early)
        A.flush_run(len(segs), on_segment=lambda sg: ar.ranges(segs[sg]) if sg + 1 < len(segs) else None)
    A.FLUSH_DRIVER[0] = driver
    try:
        run_backward(loss)
    finally:
        A.FLUSH_DRIVER[0] = None
    ar.finish()
    D.LAST_CALLS[:] = ar.calls


def _splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return x ^ (x >> 31)


_ACTIVE_MIX = {}        # device index -> address of the seed word the library currently reads there (tante_set_seed_mix)


class GraphedTrainStep:
    """train_step with zero_grad + rollout + loss + backward captured ONCE as a HIP graph and replayed: a step costs the host a batch
    copy, an 8-byte seed word and one graph launch (0.15 ms instead of 10 - 20 ms of Python per step -- on a slow host the eager step
    is issue-bound: 15 - 21 ms measured where the GPU needs 12.3).  What a replay cannot change are kernel arguments: shapes (static
    here), and the dropout seeds -- so the fused training kernels XOR a device-resident word into every seed (tante_set_seed_mix) and
    this class rewrites it before each replay: every step draws fresh masks, as the reference's nn.Dropout does.  The gradient
    all-reduce and the clip + AdamW launch stay outside the graph (two launches; the learning rate and step count are their arguments).

    Requirements: the fixed-dt model (no host-synchronising floor(R_t)), every TransformerBlock on the fused one-node path (bf16,
    C = 256; otherwise per-operator dropout kernels would replay frozen masks -- checked at capture), batches of one shape.
    Construction runs two warm-up steps and rolls parameters and optimizer state back afterwards: it leaves no trace in the training
    trajectory."""

    def __init__(self, model, opt: FlatAdamW, example_batch: Dict[str, torch.Tensor], formatter, n_steps_output: int, world: int = 1,
                 seed: int = 0):
        from . import _lib as L
        from . import autograd as A
        from . import train_forward as TF
        if not getattr(model, "deg", True):
            raise RuntimeError("GraphedTrainStep: the adaptive-dt model synchronises with the host every call (floor(R_t[0]))")
        self.model, self.opt, self.fmt, self.n, self.world = model, opt, formatter, n_steps_output, world
        self.dev = opt.flat_p.device
        self.batch = {k: v.to(self.dev).clone() for k, v in example_batch.items()}
        self.mix_dev = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.seed, self.count = int(seed), 0
        with torch.cuda.device(self.dev):      # the library keys the word by the CURRENT device: register it where the kernels will run
            L.check(L.lib().tante_set_seed_mix(self.mix_dev.data_ptr()), "tante_set_seed_mix")
        _ACTIVE_MIX[self.dev.index] = self.mix_dev.data_ptr()
        snap = (opt.flat_p.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.step_count, A._SEED[0])
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            for _ in range(2):          # allocator pools, packs, workspaces of THIS stream, per-device kernel attributes
                train_step(model, opt, self.batch, formatter, n_steps_output, 1)
        torch.cuda.current_stream(self.dev).wait_stream(side)
        opt.flat_p.copy_(snap[0]); opt.exp_avg.copy_(snap[1]); opt.exp_avg_sq.copy_(snap[2]); opt.step_count = snap[3]
        self._bump()
        torch.cuda.synchronize(self.dev)
        self.graph = torch.cuda.CUDAGraph()
        # a data-parallel step is captured as TWO graphs: everything up to the end-of-pass weight-gradient flush, and the flush -- between
        # their replays the part of the bucket the flush does not write is handed to RCCL on a side stream (dist.GradAllReduce)
        self.split = bool(D.collective_needed(world) and D.SPLIT_ALLREDUCE)
        self.graph2, self.plan = None, None
        TF.BLOCK_CALLS[0] = TF.BLOCK_CALLS[1] = 0
        A._SEED[0] = snap[4]            # the captured step draws the seeds an eager step would have drawn here
        ws_before = set(A._AXIS_WS)
        try:
            # thread-local capture mode: RCCL's proxy / watchdog threads of an initialised process group make HIP calls of their own,
            # which a global-mode capture would take for violations
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                opt.zero_grad()
                y_pred, y_ref = rollout_model(model, self.batch, formatter, n_steps_output)
                self.loss = MseMeanFn.apply(y_pred, y_ref)
                if self.split:
                    A.reset_backward_state()
                    A.HOLD_FLUSH[0] = True          # the engine's end-of-pass callback leaves the recorded weight gradients alone
                    try:
                        self.loss.backward()
                    finally:
                        A.HOLD_FLUSH[0] = False
                else:
                    run_backward(self.loss)
            if self.split:
                # the flush as one graph per segment (autograd.flush_plan's partition, agreed between the ranks once): between their
                # replays the spans a segment has finalised go to RCCL on the side stream
                plan = A.flush_plan(opt.flat_g, D.FLUSH_SEGMENTS)
                self.plan = plan if (plan is not None and D.agree(opt.flat_g, plan)) else None
                gen = A.flush_steps(len(plan[1]) if self.plan is not None else 1)
                self.graph2 = []
                for _ in range(len(plan[1]) if self.plan is not None else 1):
                    g2 = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g2, pool=self.graph.pool(), capture_error_mode="thread_local"):
                        next(gen)
                    self.graph2.append(g2)
                next(gen, None)
                A.reset_backward_state(after=True)
            calls, fused = TF.BLOCK_CALLS
            if calls == 0 or fused != calls:
                raise RuntimeError(f"GraphedTrainStep: {calls - fused} of {calls} block calls are not on the fused one-node path "
                                   "(their dropout seeds would be frozen in the graph)")
        except BaseException:
            self.close()            # the kernels must not keep reading a seed word that dies with this object
            A.reset_backward_state(after=True)
            raise
        finally:
            # Whatever the capture cached was only RECORDED, never computed: weight packs (autograd._PACKS and the modules' pack caches
            # hold buffers whose tante_pack_weight launch sits in the graph) and the propagator-gradient workspace keyed to the
            # capture stream (its zero fill is a graph node).  An eager step after a refused capture -- bench.py's and
            # harness.train_one_epoch's fallback -- or between capture and first replay must not find them: drop them all.
            self._bump()
            for k in set(A._AXIS_WS) - ws_before:
                del A._AXIS_WS[k]

    @staticmethod
    def _bump():
        from .attn_backbone import bump_weight_epoch
        from .autograd import clear_pack_cache
        bump_weight_epoch()
        clear_pack_cache()

    def set_seed_word(self, word: int):
        """The word the next step's kernels XOR into their seeds (tests use it to run an eager twin with the same masks)."""
        w = int(word) & 0xFFFFFFFFFFFFFFFF
        # a fill kernel carries the value in its own arguments: a host staging word would be overwritten by the next step's before the
        # queued copy of this one has run (the host issues replays far ahead of the GPU)
        self.mix_dev.fill_(w - (1 << 64) if w >= (1 << 63) else w)

    def __call__(self, batch: Dict[str, torch.Tensor], lr: float = None) -> torch.Tensor:
        for k, v in self.batch.items():
            v.copy_(batch[k], non_blocking=True)
        self.count += 1
        self.set_seed_word(_splitmix64(self.seed * 0x100000001B3 + self.count))
        self.graph.replay()
        if self.split:
            ar = D.GradAllReduce(self.opt.flat_g)
            if self.plan is not None:
                early, segs = self.plan
                ar.ranges(early)                     # encoder / decoders / propagators / FiLM / embeddings: on a side stream ...
                for sg, g2 in enumerate(self.graph2):
                    g2.replay()                      # ... while the blocks' weight gradients and the LayerNorm folds run, segment by segment,
                    if sg + 1 < len(segs):
                        ar.ranges(segs[sg])          # each segment's spans travelling under the next one
            else:
                for g2 in self.graph2:
                    g2.replay()
            ar.finish()
            D.LAST_CALLS[:] = ar.calls
        elif D.collective_needed(self.world):
            D.allreduce_sum_(self.opt.flat_g)
        self.opt.step(grad_scale=1.0 / self.world, lr=lr)
        return self.loss.detach()

    def close(self):
        """Detach the library from this object's seed word (idempotent).  Called by __del__ / the context manager too: a collected
        step object returns `mix_dev` to the allocator, and a kernel that still XORed that word into its seeds would draw a forward
        mask and a backward mask that differ as soon as the memory is reused."""
        if getattr(self, "_closed", False):
            return
        self._closed = True
        if _ACTIVE_MIX.get(self.dev.index) != self.mix_dev.data_ptr():
            return                  # a later step object has registered its own word on this device: leave it alone
        from . import _lib as L
        with torch.cuda.device(self.dev):
            L.check(L.lib().tante_set_seed_mix(None), "tante_set_seed_mix")
        _ACTIVE_MIX.pop(self.dev.index, None)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def train_step_adaptive(model, opt: FlatAdamW, batch: Dict[str, torch.Tensor], formatter, n_steps_output: int, rt_eps: float = 0.5,
                        rt_n: float = 2.0, world: int = 1, lr: float = None, scaler=None):
    """R_Trainer.train_one_epoch's step (trainer/r_trainer.py:135-179) for the adaptive-dt model (deg=False): per-sample rollouts
    with out_T = 1.5, loss = MSE(...).mean() + eval_rt(Rts, eps, n), clip_grad_value_(1.0) instead of a norm clip, AdamW.
    scaler: a torch.amp.GradScaler -- the reference's float16 sequence, as written there (l.152-158): scale(loss).backward();
    clip_grad_value_(1.0) on the gradients AS THEY ARE (there is no unscale_ in front of the clip: the scaled gradients are clipped);
    scaler.step(opt) unscales and skips the step when a gradient is not finite; update()."""
    opt.zero_grad()
    y_pred, y_ref, rts = rollout_adaptive(model, batch, formatter, n_steps_output, 1.5, per_sample=True)
    loss = MseMeanFn.apply(y_pred, y_ref) + MSE.eval_rt(rts, rt_eps, rt_n)
    amp = scaler is not None and scaler.is_enabled()
    run_backward(scaler.scale(loss) if amp else loss)
    if world > 1:
        D.allreduce_sum_(opt.flat_g)
        opt.flat_g.mul_(1.0 / world)
    opt.clip_grad_value_(1.0)
    saved, opt.max_norm = opt.max_norm, 0.0          # value clip replaces the norm clip in this trainer
    try:
        if amp:
            scaler.step(opt, lr=lr)
            scaler.update()
        else:
            opt.step(lr=lr)
    finally:
        opt.max_norm = saved
    return loss.detach(), rts.detach()


def train_step_cvit(model, opt: FlatAdamW, batch: Dict[str, torch.Tensor], formatter, num_query_points: int = 0, world: int = 1,
                    lr: float = None, generator=None) -> torch.Tensor:
    """The Trainer's step for a CViT model (trainer/trainer.py:174-207 with rollout_model / rollout_cvit, l.144-172): one model call
    predicts all out_steps frames; num_query_points > 0 trains on that many random pixels (the `cvit: True` branch), 0 on the full
    grid.  Loss = MSE(...).mean(); clip_grad_norm_(1.0) + AdamW through the flat buckets; one summed all-reduce when world > 1."""
    from .harness import generate_and_extract_coords
    opt.zero_grad()
    device = next(model.parameters()).device
    moving, y_ref = formatter.process_input(batch)
    x = moving[0].to(device)
    y_ref = y_ref.to(device)
    if num_query_points > 0:
        coords, y_pts = generate_and_extract_coords(y_ref, num_query_points, generator)
        y_pred = model(x, coords)                                   # (b, t, n, d)
        loss = MseMeanFn.apply(y_pred.unsqueeze(3), y_pts.unsqueeze(3).contiguous())
    else:
        y_pred = formatter.process_output(model(x))                 # (b, t, h, w, d) view of the prediction
        loss = MseMeanFn.apply(y_pred, y_ref[:, :y_pred.shape[1]].contiguous())
    run_backward(loss)
    if world > 1:
        D.allreduce_sum_(opt.flat_g)
    opt.step(grad_scale=1.0 / world, lr=lr)
    return loss.detach()
