#!/usr/bin/env python3
"""Epoch-level driver on the HIP path: the reference's train.py / eval.py flow (train.py:22-78, eval.py:21-56) over a yaml in the reference's
schema with a synthetic datamodule standing in for The Well (no datasets on this machine):

    python tools/train_eval.py --config configs/tante_trl.yaml --epochs 2 --samples 32 --out experiments/demo

train_one_epoch -> recent.pt -> validation_loop ([MSE, NNMSE, L2RE, VRMSE] means + variances, synchronised forward time) -> best.pt,
LinearWarmupCosineAnnealingLR stepped per epoch, resume from recent.pt when present.  Under torchrun (WORLD_SIZE > 1) the batch is
sharded over ranks and the gradients all-reduced once per step."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tante_amd  # noqa: E402
from tante_amd import harness as H  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--config", default=os.path.join(ROOT, "configs", "tante_trl.yaml"))
    p.add_argument("--epochs", type=int, default=2)
    p.add_argument("--samples", type=int, default=32)
    p.add_argument("--batch", type=int, default=None)
    p.add_argument("--out", default=os.path.join(ROOT, "experiments", "synthetic"))
    p.add_argument("--eval-only", action="store_true")
    a = p.parse_args()
    rank, world, local = tante_amd.dist.init()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg = tante_amd.load_config(a.config)
    wl = cfg["workload"]
    md = tante_amd.TanteMetadata(n_fields=wl["n_fields"], spatial_resolution=tuple(wl["spatial_resolution"]))
    torch.manual_seed(cfg.get("seed", 211))
    model = tante_amd.build_model(cfg, md).to(dev).set_compute({"bfloat16": "bf16", "float32": "fp32"}[wl.get("amp", "bfloat16")])
    oc = cfg.get("optimizer", {"lr": 5e-5, "weight_decay": 1e-5})
    opt = tante_amd.FlatAdamW(model.parameters(), lr=oc["lr"], weight_decay=oc["weight_decay"], max_norm=1.0)
    opt.broadcast_parameters(0)
    sched = H.LinearWarmupCosineAnnealingLR(opt, warmup_epochs=2, max_epochs=max(a.epochs, 3), warmup_start_lr=0.1 * oc["lr"], eta_min=0.1 * oc["lr"])
    n_out, n_roll = wl.get("n_steps_output", 4), wl.get("n_steps_rollout", 4)
    dm = H.SyntheticDataModule(md, a.batch or wl["batch_size"], wl["n_steps_input"], max(n_out, n_roll), a.samples, cfg.get("seed", 211), world, rank)
    fmt = tante_amd.DefaultChannelsFirstFormatter(md)
    if a.eval_only:
        H.load_checkpoint(os.path.join(a.out, "recent.pt"), model)
        res = H.validation_loop(model, dm.val_dataloader(), fmt, n_roll)
    else:
        res = H.fit(model, opt, dm, fmt, a.epochs, n_out, n_roll, a.out, sched, world, log=print if rank == 0 else (lambda *_: None))
    if rank == 0:
        print(json.dumps(res, default=float))


if __name__ == "__main__":
    main()
