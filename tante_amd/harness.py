"""Harness periphery around the hot path (host side only): the checkpoint dictionary of
trainer/trainer.py:116-141, the per-epoch LR scheduler of optim/schedulers.py:17-123 as an object, CViT's query-point
rollouts (trainer/trainer.py:36-69,161-172; trainer/evaler.py:37-76,140-165) and a synthetic stand-in for the datamodule that
yields the `{"input", "output"}` dicts of data/dataset.py:224-227.  No arithmetic of the model lives here."""
from __future__ import annotations

from typing import Dict, Iterator, List, Optional, Tuple

import torch

from .optim import warmup_cosine_lr


# ---- checkpoints: the reference's dictionary, key for key (including its 'optimizer_state_dit' spelling) -------------------------
def save_checkpoint(path: str, model: torch.nn.Module, optimizer, epoch: int, validation_loss: float, best_validation_loss) -> None:
    torch.save({"epoch": epoch, "model_state_dict": model.state_dict(),
                "optimizer_state_dit": optimizer.state_dict() if optimizer is not None else None,
                "validation_loss": validation_loss, "best_validation_loss": best_validation_loss}, path)


def load_checkpoint(path: str, model: Optional[torch.nn.Module] = None, optimizer=None, lr_scheduler=None) -> Dict:
    """-> {"starting_epoch", "starting_val_loss", "best_val_loss"}; replays the per-epoch scheduler like trainer.py:139-141."""
    ck = torch.load(path, weights_only=False)
    if model is not None:
        model.load_state_dict(ck["model_state_dict"])
    if optimizer is not None and ck.get("optimizer_state_dit") is not None:
        optimizer.load_state_dict(ck["optimizer_state_dit"])
    start = ck["epoch"] + 1
    if lr_scheduler is not None:
        for _ in range(start - 1):
            lr_scheduler.step()
    return {"starting_epoch": start, "starting_val_loss": ck["validation_loss"], "best_val_loss": ck["best_validation_loss"]}


class LinearWarmupCosineAnnealingLR:
    """Per-epoch schedule in closed form (optim/schedulers.py:97-123): the same `step()` / `get_last_lr()` surface, driving any
    optimizer that exposes `.lr` (FlatAdamW) or `.param_groups` (torch.optim)."""

    def __init__(self, optimizer, warmup_epochs: int, max_epochs: int, warmup_start_lr: float = 0.0, eta_min: float = 0.0,
                 last_epoch: int = -1):
        self.optimizer, self.warmup_epochs, self.max_epochs = optimizer, warmup_epochs, max_epochs
        self.warmup_start_lr, self.eta_min = warmup_start_lr, eta_min
        groups = getattr(optimizer, "param_groups", None)
        self.base_lrs = [g["lr"] for g in groups] if groups else [optimizer.lr]
        self.last_epoch = last_epoch
        self.step()

    def get_last_lr(self) -> List[float]:
        return [warmup_cosine_lr(self.last_epoch, b, self.warmup_epochs, self.max_epochs, self.warmup_start_lr, self.eta_min)
                for b in self.base_lrs]

    def step(self) -> None:
        self.last_epoch += 1
        lrs = self.get_last_lr()
        groups = getattr(self.optimizer, "param_groups", None)
        if groups:
            for g, lr in zip(groups, lrs):
                g["lr"] = lr
        else:
            self.optimizer.lr = lrs[0]


# ---- CViT query-point harness ---------------------------------------------------------------------------------------------------
def generate_and_extract_coords(y_ref: torch.Tensor, M: int, generator: Optional[torch.Generator] = None):
    """M random pixels of the (B, T, H, W, C) reference: their [0,1]^2 coordinates (M, 2) and values (B, T, M, C)
    (trainer/trainer.py:36-69)."""
    B, T, H, W, C = y_ref.shape
    sel = torch.randperm(H * W, device=y_ref.device, generator=generator)[:M]
    hi, wi = sel // W, sel % W
    coords = torch.stack([hi.float() / (H - 1), wi.float() / (W - 1)], dim=-1)
    return coords, y_ref[:, :, hi, wi, :]


def generate_chunked_coords_with_indices(H: int, W: int, L: int, device="cuda"):
    """All H * W pixels in row-major order, cut into chunks of L: normalised coordinates and integer (h, w) indices
    (trainer/evaler.py:37-61)."""
    idx = torch.arange(H * W, device=device)
    hw = torch.stack([idx // W, idx % W], dim=-1)
    xy = torch.stack([hw[:, 0].float() / (H - 1), hw[:, 1].float() / (W - 1)], dim=-1)
    return list(xy.split(L)), list(hw.split(L))


def reconstruct_full_field(chunks: List[torch.Tensor], indices: List[torch.Tensor], H: int, W: int) -> torch.Tensor:
    """(B, T, n_i, C) predictions at the chunk pixels -> (B, T, C, H, W)  (trainer/evaler.py:63-76)."""
    B, T, _, C = chunks[0].shape
    full = torch.zeros(B, T, C, H, W, device=chunks[0].device, dtype=chunks[0].dtype)
    for y, ij in zip(chunks, indices):
        full[:, :, :, ij[:, 0], ij[:, 1]] = y.permute(0, 1, 3, 2)
    return full


def rollout_cvit_train(model, batch: Dict, formatter, num_query_points: int, device=None, generator=None):
    """Trainer.rollout_cvit (trainer/trainer.py:161-172): one call at random query points."""
    device = device or next(model.parameters()).device
    moving, y_ref = formatter.process_input(batch)
    coords, y_pts = generate_and_extract_coords(y_ref.to(device), num_query_points, generator)
    return model(moving[0].to(device), coords), y_pts


def rollout_cvit_eval(model, batch: Dict, formatter, n_steps: int, num_query_points: int, device=None):
    """Evaler.rollout_cvit (trainer/evaler.py:140-165): the full field in query chunks, re-fed until n_steps frames exist."""
    device = device or next(model.parameters()).device
    moving, y_ref = formatter.process_input(batch)
    moving = moving[0].to(device)
    H, W = y_ref.shape[2], y_ref.shape[3]
    preds, produced = [], 0
    while produced < n_steps:
        cc, ii = generate_chunked_coords_with_indices(H, W, num_query_points, device)
        # (the reference runs the whole model per chunk; the encoder half does not depend on the query points: once per window)
        enc = model.encode(moving) if (hasattr(model, "encode") and not torch.is_grad_enabled()) else None
        y = reconstruct_full_field([model(moving, c, encoded=enc) if enc is not None else model(moving, c) for c in cc], ii, H, W)
        produced += y.shape[1]
        if produced < n_steps:
            moving = torch.cat([moving[:, y.shape[1]:], y], dim=1)
        preds.append(formatter.process_output(y))
    return torch.cat(preds, dim=1)[:, :n_steps], y_ref.to(device)


# ---- synthetic datamodule -------------------------------------------------------------------------------------------------------
class SyntheticDataModule:
    """Stand-in for data.TanteDataModule: standard-normal fields (the datasets are standardised per field, data/dataset.py:206-209)
    in the `{"input": (B, n_in, H, W, C), "output": (B, n_out, H, W, C)}` layout of field_to_tensor (data/dataset.py:224-227),
    sharded like DistributedSampler(num_replicas=world, rank=rank, drop_last=True) (data/datamodule.py:96-119)."""

    def __init__(self, metadata, batch_size: int, n_steps_input: int = 4, n_steps_output: int = 4, n_samples: int = 64, seed: int = 211,
                 world_size: int = 1, rank: int = 0, device="cpu"):
        self.metadata, self.batch_size, self.n_in, self.n_out = metadata, batch_size, n_steps_input, n_steps_output
        self.n_samples, self.seed, self.world, self.rank, self.device = n_samples, seed, world_size, rank, device
        self.epoch = 0

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def __len__(self) -> int:
        return (self.n_samples // self.world) // self.batch_size

    def _sample(self, i: int) -> torch.Tensor:
        H, W = self.metadata.spatial_resolution
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        return torch.randn(self.n_in + self.n_out, H, W, self.metadata.n_fields, generator=g)

    def train_dataloader(self) -> Iterator[Dict[str, torch.Tensor]]:
        g = torch.Generator().manual_seed(self.seed + self.epoch)
        order = torch.randperm(self.n_samples, generator=g).tolist()
        per = self.n_samples // self.world
        mine = order[self.rank::self.world][:per]
        for b in range(len(self)):
            xs = torch.stack([self._sample(i) for i in mine[b * self.batch_size:(b + 1) * self.batch_size]]).to(self.device)
            yield {"input": xs[:, :self.n_in], "output": xs[:, self.n_in:]}

    val_dataloader = train_dataloader


# ---- epoch-level driver: Trainer.train (trainer/trainer.py:234-255) and the Evaler / R_Evaler validation loops ---------------------------
def validation_loop(model, dataloader, formatter, n_steps_rollout: int, device=None) -> Dict:
    """Evaler.validation_loop (trainer/evaler.py:186-230), and R_Evaler's (trainer/r_evaler.py:110-177) for an adaptive-dt model
    (deg=False): per batch a rollout and the four evaluation metrics in the reference's list order [MSE, NNMSE, L2RE, VRMSE]
    (eval_loss_fn1, fn3, fn2, fn4), then their means and `statistics.variance` over the batches and the mean forward time per batch.
    The reference reads the wall clock around the rollout without a device sync (evaler.py:127,134), i.e. it times the launches; here
    the stream is synchronised on both sides, so `forward_time` is completion time.
    deg=False adds the mean step size r_t, the mean number of model calls per rollout and the five-number summaries of the per-batch
    NNMSE and r_t (r_evaler.py:160-177)."""
    import statistics
    import time
    from . import metrics as M
    from .rollout import rollout_adaptive, rollout_model
    device = device or next(model.parameters()).device
    was_training = model.training
    model.eval()
    adaptive = not getattr(model, "deg", True)
    seq = [[], [], [], []]
    times, rt_list, step_list = [], [], []
    with torch.inference_mode():
        for batch in dataloader:
            batch = {"input": batch["input"].to(device), "output": batch["output"][:, :n_steps_rollout].to(device)}
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            if adaptive:
                y_pred, y_ref, rts = rollout_adaptive(model, batch, formatter, n_steps_rollout, float(n_steps_rollout), per_sample=False)
            else:
                y_pred, y_ref = rollout_model(model, batch, formatter, n_steps_rollout)
            torch.cuda.synchronize(device)
            times.append(time.perf_counter() - t0)
            assert y_ref.shape == y_pred.shape, f"Mismatching shapes between reference {y_ref.shape} and prediction {y_pred.shape}"
            y_ref = y_ref.contiguous()
            for lst, fn in zip(seq, (M.MSE, M.NNMSE, M.L2RE, M.VRMSE)):
                lst.append(float(fn.eval(y_pred, y_ref).mean()))
            if adaptive:
                rt_list.append(float(rts.float().mean()))
                step_list.append(int(rts.shape[0]) if rts.dim() > 0 else 1)
    model.train(was_training)
    n = max(1, len(times))
    out = {"validation_loss": [sum(s) / n for s in seq], "metric_names": ["MSE", "NNMSE", "L2RE", "VRMSE"],
           "variance": [statistics.variance(s) if len(s) > 1 else 0.0 for s in seq], "forward_time": sum(times) / n, "n_batches": len(times)}
    if adaptive:
        def five(d):
            t = torch.tensor(d, dtype=torch.float64)
            q = torch.quantile(t, torch.tensor([0.0, 0.25, 0.5, 0.75, 1.0], dtype=torch.float64))
            return dict(zip(("min", "q1", "median", "q3", "max"), (float(v) for v in q)))
        out.update({"RT": sum(rt_list) / n, "Step": sum(step_list) / n, "summary_error": five(seq[1]), "summary_rt": five(rt_list)})
    return out


def train_one_epoch(model, optimizer, dataloader, formatter, n_steps_output: int, world: int = 1, rt_eps: float = 0.5,
                    rt_n: float = 2.0, graph: bool = False, enable_amp: bool = False, amp_type: str = "bfloat16", scaler=None) -> float:
    """Trainer.train_one_epoch (trainer/trainer.py:174-207; R_Trainer's for deg=False, r_trainer.py:135-179): one optimisation step
    per batch, returns the mean training loss of the epoch (ONE host read at the end instead of the reference's loss.item() per batch).
    graph=True (fixed-dt model only): the steps of full-size batches replay one HIP graph (train.GraphedTrainStep, captured on the first
    such batch and kept on the model); a ragged last batch, or a model the capture refuses, takes the eager step."""
    from .train import GraphedTrainStep, train_step, train_step_adaptive
    device = next(model.parameters()).device
    model.train()
    losses = []
    if enable_amp:
        # Trainer.__init__ / train_one_epoch (trainer/trainer.py:86-104, 183-196): autocast(amp_type) around the forward and, for float16
        # only, a GradScaler around backward / step.  Both 16-bit types select the bf16 MFMA kernels (attn_backbone.resolve_compute).
        dt = {"float16": torch.float16, "bfloat16": torch.bfloat16}[amp_type]
        if dt == torch.float16 and not getattr(model, "_tante_fp16_notice", False):
            import warnings
            warnings.warn("tante_amd: amp_type='float16' runs the bf16 MFMA kernels (there is no fp16 compute mode); the GradScaler sequence "
                          "of the reference is kept")
            model._tante_fp16_notice = True
        if not hasattr(model, "set_compute"):
            raise TypeError("tante_amd.harness.train_one_epoch(enable_amp=True) needs a tante_amd model (set_compute selects the bf16 kernels)")
        if scaler is None and dt == torch.float16:
            scaler = getattr(model, "_tante_grad_scaler", None) or torch.amp.GradScaler("cuda", enabled=True)
            model._tante_grad_scaler = scaler
        if graph:
            import warnings
            warnings.warn("tante_amd.harness.train_one_epoch: graph=True is ignored with enable_amp (the captured step resolves its compute mode "
                          "at capture; call model.set_compute('bf16') and pass enable_amp=False to replay the step as a HIP graph)")
        # The reference leaves the autocast context BEFORE scaler.scale(loss).backward() (trainer/trainer.py:183-196).  Here the context only
        # SELECTS the kernels (attn_backbone.resolve_compute reads it during the forward; the backward nodes carry the mode they were built
        # with, and clip + AdamW run on fp32 buckets whatever the context), so the forward alone is what must sit inside it: the model is
        # switched for the step instead of wrapping backward and step in the context.
        mode = "bf16"
        prev = getattr(model, "compute", None)
        for batch in dataloader:
            batch = {"input": batch["input"].to(device), "output": batch["output"][:, :n_steps_output].contiguous().to(device)}
            model.set_compute(mode)
            try:
                if getattr(model, "deg", True):
                    losses.append(train_step(model, optimizer, batch, formatter, n_steps_output, world, scaler=scaler))
                else:
                    losses.append(train_step_adaptive(model, optimizer, batch, formatter, n_steps_output, rt_eps, rt_n, world, scaler=scaler)[0])
            finally:
                model.set_compute(prev)
        return float(torch.stack(losses).mean()) if losses else float("nan")
    for batch in dataloader:
        batch = {"input": batch["input"].to(device), "output": batch["output"][:, :n_steps_output].contiguous().to(device)}
        if graph and getattr(model, "deg", True):
            g = getattr(model, "_tante_graphed_step", None)
            shapes = tuple(tuple(v.shape) for v in batch.values())
            if g is None or g is False:
                if g is None:
                    try:
                        g = GraphedTrainStep(model, optimizer, batch, formatter, n_steps_output, world)
                        g.shapes = shapes
                    except RuntimeError:
                        g = False
                    model._tante_graphed_step = g
            if g and g.shapes == shapes and g.opt is optimizer:
                losses.append(g(batch).clone())
                continue
        if getattr(model, "deg", True):
            losses.append(train_step(model, optimizer, batch, formatter, n_steps_output, world))
        else:
            losses.append(train_step_adaptive(model, optimizer, batch, formatter, n_steps_output, rt_eps, rt_n, world)[0])
    return float(torch.stack(losses).mean()) if losses else float("nan")


def fit(model, optimizer, datamodule, formatter, max_epoch: int, n_steps_output: int, n_steps_rollout: int, checkpoint_folder: str,
        lr_scheduler=None, world: int = 1, log=print, graph: bool = False) -> Dict:
    """Trainer.train (trainer/trainer.py:234-255): resume from <folder>/recent.pt when it exists (utils.set_ckpt, utils.py:36-47), then per
    epoch  sampler.set_epoch -> train_one_epoch -> save recent.pt -> validation_loop -> save best.pt on improvement -> scheduler.step.
    `best` is compared on L2RE (the Trainer's eval_loss_fn, configs/tante.yaml:52-53) and, unlike Trainer (which never updates best_val_loss, trainer.py:254-255, so
    best.pt is rewritten every epoch), IS updated -- R_Trainer's behaviour (r_trainer.py:228-230).  Only rank 0 writes checkpoints."""
    import os
    os.makedirs(checkpoint_folder, exist_ok=True)
    recent, best = os.path.join(checkpoint_folder, "recent.pt"), os.path.join(checkpoint_folder, "best.pt")
    state = {"starting_epoch": 1, "starting_val_loss": None, "best_val_loss": None}
    if os.path.exists(recent):
        state = load_checkpoint(recent, model, optimizer, lr_scheduler)
        log(f"resumed from {recent}: starting at epoch {state['starting_epoch']}")
    val_loss, best_val = state["starting_val_loss"], state["best_val_loss"]
    history = []
    rank0 = getattr(datamodule, "rank", 0) == 0
    for epoch in range(state["starting_epoch"], max_epoch + 1):
        datamodule.set_epoch(epoch)
        train_loss = train_one_epoch(model, optimizer, datamodule.train_dataloader(), formatter, n_steps_output, world, graph=graph)
        if rank0:
            save_checkpoint(recent, model, optimizer, epoch, val_loss, best_val)
        val = validation_loop(model, datamodule.val_dataloader(), formatter, n_steps_rollout)
        val_loss = val["validation_loss"][2]
        if best_val is None or val_loss < best_val:
            best_val = val_loss
            if rank0:
                save_checkpoint(best, model, optimizer, epoch, val_loss, best_val)
        if lr_scheduler is not None:
            lr_scheduler.step()
        history.append({"epoch": epoch, "train_loss": train_loss, "lr": getattr(optimizer, "lr", None), **val})
        log(f"epoch {epoch}/{max_epoch}: train loss {train_loss:.6f}  valid {[round(v, 6) for v in val['validation_loss']]}  "
            f"forward {val['forward_time'] * 1e3:.2f} ms/batch")
    g = getattr(model, "_tante_graphed_step", None)
    if g:                       # training is over: the library must not keep reading the step object's seed word
        g.close()
        model._tante_graphed_step = None
    return {"history": history, "best_val_loss": best_val}
