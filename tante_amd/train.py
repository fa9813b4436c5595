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
               lr: float = None) -> torch.Tensor:
    opt.zero_grad()
    y_pred, y_ref = rollout_model(model, batch, formatter, n_steps_output)
    loss = MseMeanFn.apply(y_pred, y_ref)
    run_backward(loss)
    if world > 1:
        D.allreduce_sum_(opt.flat_g)
    opt.step(grad_scale=1.0 / world, lr=lr)
    return loss.detach()


def train_step_adaptive(model, opt: FlatAdamW, batch: Dict[str, torch.Tensor], formatter, n_steps_output: int, rt_eps: float = 0.5,
                        rt_n: float = 2.0, world: int = 1, lr: float = None):
    """R_Trainer.train_one_epoch's step (trainer/r_trainer.py:135-179) for the adaptive-dt model (deg=False): per-sample rollouts
    with out_T = 1.5, loss = MSE(...).mean() + eval_rt(Rts, eps, n), clip_grad_value_(1.0) instead of a norm clip, AdamW."""
    opt.zero_grad()
    y_pred, y_ref, rts = rollout_adaptive(model, batch, formatter, n_steps_output, 1.5, per_sample=True)
    loss = MseMeanFn.apply(y_pred, y_ref) + MSE.eval_rt(rts, rt_eps, rt_n)
    run_backward(loss)
    if world > 1:
        D.allreduce_sum_(opt.flat_g)
        opt.flat_g.mul_(1.0 / world)
    opt.clip_grad_value_(1.0)
    saved, opt.max_norm = opt.max_norm, 0.0          # value clip replaces the norm clip in this trainer
    try:
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
