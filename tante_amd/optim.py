"""Optimiser step of the train harness on flat buckets.

FlatAdamW re-homes a model's parameters into ONE contiguous fp32 bucket (each nn.Parameter becomes a view) with a
matching gradient bucket, so that (i) data-parallel training needs a single all-reduce over xGMI per step
(tante_amd.dist) and (ii) clip_grad_norm_ + AdamW (trainer/trainer.py:193-196, configs/tante.yaml:38-41) are two HIP
launches with no host synchronisation: tante_sumsq (global gradient norm) and tante_adamw_step (clip scale computed on
the device from that sum, decoupled weight decay, bias-corrected moments).  LR schedule: optim.warmup_cosine_lr
(LinearWarmupCosineAnnealingLR, optim/schedulers.py:17-123, stepped per epoch by the reference).
"""
from __future__ import annotations

import math
from typing import Iterable, Optional

import torch

from . import _lib as L


def flat_layout(params):
    """(offsets, total) of the flat fp32 buckets: parameters in order, complex ones as (re, im) pairs, every view 16-byte aligned.
    The data-parallel all-reduce moves exactly this bucket (tante_amd.dist), so the layout is a function of its own."""
    sizes = [p.numel() * (2 if p.is_complex() else 1) for p in params]
    offs, n = [], 0
    for sz in sizes:
        offs.append(n)
        n += (sz + 3) // 4 * 4            # keep every view 16-byte aligned
    return offs, sizes, n


class FlatAdamW:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 5e-5, weight_decay: float = 1e-5,
                 betas=(0.9, 0.999), eps: float = 1e-8, max_norm: float = 1.0):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdamW runs on the GPU only (no CPU fallback)")
        self.lr, self.weight_decay, self.betas, self.eps, self.max_norm = lr, weight_decay, betas, eps, max_norm
        # complex parameters (the spectral layers' weights) live in the buckets as (re, im) pairs, which is also how torch.optim.AdamW
        # treats them (view_as_real)
        offs, sizes, n = flat_layout(self.params)
        self.numel = n
        self.flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self._sumsq = torch.zeros(1, dtype=torch.float64, device=dev)
        with torch.no_grad():
            for p, o, sz in zip(self.params, offs, sizes):
                if p.is_complex():
                    self.flat_p[o:o + sz].copy_(torch.view_as_real(p.detach()).reshape(-1))
                    p.data = torch.view_as_complex(self.flat_p[o:o + sz].view(*p.shape, 2))
                    p.grad = torch.view_as_complex(self.flat_g[o:o + sz].view(*p.shape, 2))
                    continue
                self.flat_p[o:o + sz].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[o:o + sz].view(p.shape)          # the parameter now lives in the bucket
                p.grad = self.flat_g[o:o + sz].view(p.shape)
        self._offs = offs
        self.step_count = 0

    @property
    def param_groups(self):
        """torch.optim's view of this optimiser: ONE group holding the parameters (whose .grad are views of the flat gradient bucket).
        It is what torch.amp.GradScaler walks in unscale_() -- its in-place foreach unscale + inf check act directly on the bucket --
        so the reference's AMP sequence works unchanged:  scaler.scale(loss).backward(); scaler.unscale_(opt); scaler.step(opt);
        scaler.update()   (trainer/trainer.py:191-195; the norm clip lives inside step())."""
        return [{"params": self.params, "lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay}]

    def zero_grad(self, set_to_none: bool = False):
        """Zero the gradient bucket.  The parameters' .grad stay views of it whatever set_to_none says: step() reads only the bucket.
        (The view check runs once per step, in step(); here only a dropped .grad -- model.zero_grad(set_to_none=True) -- is re-bound,
        which a cheap scan of `p.grad is None` finds.)"""
        if any(p.grad is None for p in self.params):
            self._check_views()
        self.flat_g.zero_()

    def _check_views(self):
        """step() reads only the flat buckets, so every p.data / p.grad must still alias them.  model.zero_grad() (set_to_none=True by
        default) leaves .grad None -> re-bound here; a .grad or .data that points elsewhere (model.to(), .float(), a fresh tensor
        written by autograd after the views were dropped) would make the step apply zeros silently -> raise instead."""
        base_p, base_g = self.flat_p.data_ptr(), self.flat_g.data_ptr()
        for p, o in zip(self.params, self._offs):
            if p.data_ptr() != base_p + 4 * o:
                raise RuntimeError("FlatAdamW: a parameter no longer lives in the flat bucket (model.to()/.float() after the optimiser "
                                   "was built?); rebuild the optimiser")
            if p.grad is None:
                sz = p.numel() * (2 if p.is_complex() else 1)
                v = self.flat_g[o:o + sz]
                p.grad = torch.view_as_complex(v.view(*p.shape, 2)) if p.is_complex() else v.view(p.shape)
            elif p.grad.data_ptr() != base_g + 4 * o:
                raise RuntimeError("FlatAdamW: a parameter's .grad is not a view of the flat gradient bucket any more (gradients were "
                                   "written to a fresh tensor, e.g. after model.zero_grad(set_to_none=True) followed by backward); "
                                   "use FlatAdamW.zero_grad()")

    def broadcast_parameters(self, src: int = 0):
        """Data parallel: every rank starts from rank `src`'s weights (one broadcast of the flat parameter bucket)."""
        from . import dist as D
        D.broadcast_(self.flat_p, src)
        from .attn_backbone import bump_weight_epoch
        from .autograd import clear_pack_cache
        bump_weight_epoch()
        clear_pack_cache()

    # ---- checkpoint interchange with torch.optim.AdamW (the layout the reference's checkpoints hold, trainer/trainer.py:116-141) --
    def _views(self, flat: torch.Tensor):
        out, n = [], 0
        for p in self.params:
            sz = p.numel() * (2 if p.is_complex() else 1)
            out.append(torch.view_as_complex(flat[n:n + sz].view(*p.shape, 2)) if p.is_complex() else flat[n:n + sz].view(p.shape))
            n += (sz + 3) // 4 * 4
        return out

    def state_dict(self):
        ea, es = self._views(self.exp_avg), self._views(self.exp_avg_sq)
        state = {}
        if self.step_count > 0:
            state = {i: {"step": torch.tensor(float(self.step_count)), "exp_avg": ea[i].clone(), "exp_avg_sq": es[i].clone()}
                     for i in range(len(self.params))}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps, self.weight_decay = g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]
        ea, es = self._views(self.exp_avg), self._views(self.exp_avg_sq)
        steps = set()
        with torch.no_grad():
            for i in range(len(self.params)):
                st = sd["state"].get(i)
                if st is None:
                    ea[i].zero_()
                    es[i].zero_()
                    continue
                ea[i].copy_(st["exp_avg"])
                es[i].copy_(st["exp_avg_sq"])
                steps.add(int(st["step"]))
        if len(steps) > 1:
            raise ValueError("FlatAdamW keeps one step counter; the checkpoint has per-parameter step counts " + str(sorted(steps)))
        self.step_count = steps.pop() if steps else 0

    def grad_norm(self) -> torch.Tensor:
        """Total 2-norm of the gradient bucket (device tensor; reading it synchronises)."""
        L.check(L.lib().tante_sumsq(self.flat_g.data_ptr(), self.numel, self._sumsq.data_ptr(),
                                    torch.cuda.current_stream().cuda_stream), "tante_sumsq")
        return torch.sqrt(self._sumsq)[0].float()

    def clip_grad_value_(self, clip: float):
        """torch.nn.utils.clip_grad_value_ over the bucket (R_Trainer, trainer/r_trainer.py:155)."""
        L.check(L.lib().tante_clip_value(self.flat_g.data_ptr(), self.numel, float(clip), torch.cuda.current_stream().cuda_stream),
                "tante_clip_value")

    def step(self, grad_scale: float = 1.0, lr: Optional[float] = None):
        """clip_grad_norm_(max_norm) + AdamW.  grad_scale multiplies the gradients first (1/world after a summed all-reduce)."""
        self._check_views()
        self.step_count += 1
        s = torch.cuda.current_stream().cuda_stream
        if self.max_norm and self.max_norm > 0:
            L.check(L.lib().tante_sumsq(self.flat_g.data_ptr(), self.numel, self._sumsq.data_ptr(), s), "tante_sumsq")
        L.check(L.lib().tante_adamw_step(self.flat_p.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                         self.flat_g.data_ptr(), self.numel, self._sumsq.data_ptr(), float(self.max_norm or 0.0),
                                         float(self.lr if lr is None else lr), self.betas[0], self.betas[1], self.eps,
                                         self.weight_decay, self.step_count, float(grad_scale), s), "tante_adamw_step")
        # the kernel wrote through raw pointers: invalidate the packed-weight caches (attn_backbone._PackCache)
        from .attn_backbone import bump_weight_epoch
        from .autograd import clear_pack_cache
        bump_weight_epoch()
        clear_pack_cache()


def warmup_cosine_lr(epoch: int, base_lr: float, warmup_epochs: int, max_epochs: int, warmup_start_lr: float = 0.0,
                     eta_min: float = 0.0) -> float:
    """LinearWarmupCosineAnnealingLR in closed form (optim/schedulers.py:97-123)."""
    if epoch < warmup_epochs:
        return warmup_start_lr + epoch * (base_lr - warmup_start_lr) / max(1, warmup_epochs - 1)
    return eta_min + 0.5 * (base_lr - eta_min) * (1 + math.cos(math.pi * (epoch - warmup_epochs) / (max_epochs - warmup_epochs)))
