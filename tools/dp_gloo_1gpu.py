#!/usr/bin/env python3
"""The product's data-parallel train step with world = 2 on ONE GPU (reference: data/datamodule.py:96-119 shards the batch,
trainer/trainer.py:193 clips the global gradient norm, l.240-241 re-seeds the sampler; DDP all-reduces the gradients).

    python tools/dp_gloo_1gpu.py            # the parent: spawns the two ranks BEFORE touching the GPU, prints their verdicts

Each rank: dist.init("gloo") -> cuda:0 -> FlatAdamW.broadcast_parameters(0) -> train_step(..., world=2) on its shard of a g14-shaped batch
(C = 256, 8 heads, THWTHWTHW at 64 x 384 x 4) -> GraphedTrainStep(..., world=2) on the next batch.  Rank 0 also runs a single-process twin on
the full batch and compares: all-reduced gradient bucket / 2 == full-batch gradients, updated parameters equal (fp32 compute: 1e-5;
the graphed bf16 step against a bf16 twin: 2e-3 on the gradient bucket -- its weight-gradient partial sums are combined in a different
order).  Both ranks then run the SAME sample with dropout 0.1: their losses must differ (rank-mixed dropout seeds), where two ranks
with equal seeds would agree bit for bit.  What this covers that the CPU gloo test cannot: the HIP train step, the flat-bucket all-reduce
beside a thread-local-captured graph, broadcast_parameters, the seed mixing.  What it does not: RCCL itself (needs two GPUs)."""
import copy
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parent() -> int:
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TANTE_DP_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env))
    rcs = []
    for p in procs:
        try:
            rcs.append(p.wait(timeout=500))
        except subprocess.TimeoutExpired:
            p.kill()
            rcs.append(-9)
    print("dp_gloo_1gpu: rank exit codes", rcs, flush=True)
    return 0 if all(rc == 0 for rc in rcs) else 1


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def child() -> int:
    import torch
    import tante_amd
    from tante_amd import dist as D
    from tante_amd.train import GraphedTrainStep, train_step
    rank, world, _ = D.init("gloo")
    assert world == 2
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    md = tante_amd.TanteMetadata(n_fields=4, spatial_resolution=(64, 384))
    fmt = tante_amd.DefaultChannelsFirstFormatter(md)
    kw = dict(in_T=4, taylor_order=1, attn_axes="THWTHWTHW", n_head=8, embed_dim=256, patch_scale=8)
    report = {"rank": rank}

    def make(seed, dropout, compute):
        torch.manual_seed(seed)
        m = tante_amd.TANTE(dset_metadata=md, dropout=dropout, **kw).to(dev).train().set_compute(compute)
        return m

    gen = torch.Generator().manual_seed(77)
    full = {"input": torch.randn(4, 4, 64, 384, 4, generator=gen), "output": torch.randn(4, 4, 64, 384, 4, generator=gen)}
    full2 = {"input": torch.randn(4, 4, 64, 384, 4, generator=gen), "output": torch.randn(4, 4, 64, 384, 4, generator=gen)}
    shard = {k: v.to(dev) for k, v in D.shard_batch(full, rank, world).items()}
    shard2 = {k: v.to(dev) for k, v in D.shard_batch(full2, rank, world).items()}

    # ---- (1) eager fp32 step: different initial weights per rank, broadcast makes them rank 0's ------------------------------------
    m = make(100 + rank, 0.0, "fp32")
    opt = tante_amd.FlatAdamW(m.parameters(), lr=1e-4, weight_decay=0.01, max_norm=1.0)
    before = opt.flat_p.clone()
    opt.broadcast_parameters(0)
    if rank == 1:
        assert not torch.equal(before, opt.flat_p), "rank 1 kept its own initial weights"
    loss = train_step(m, opt, shard, fmt, 4, world)
    torch.cuda.synchronize()
    if rank == 0:
        ref = make(100, 0.0, "fp32")
        ropt = tante_amd.FlatAdamW(ref.parameters(), lr=1e-4, weight_decay=0.01, max_norm=1.0)
        rloss = train_step(ref, ropt, {k: v.to(dev) for k, v in full.items()}, fmt, 4, 1)
        torch.cuda.synchronize()
        report["eager_fp32"] = {"grad_rel": rel(opt.flat_g * 0.5, ropt.flat_g), "param_rel": rel(opt.flat_p, ropt.flat_p),
                                "loss_shard": float(loss), "loss_full": float(rloss)}
        assert report["eager_fp32"]["grad_rel"] < 1e-5 and report["eager_fp32"]["param_rel"] < 1e-6, report
    D.barrier()

    # ---- (2) graphed bf16 step with a live process group (the all-reduce beside a thread-local-captured graph) ---------------------
    m = make(200 + rank, 0.0, "bf16")
    opt = tante_amd.FlatAdamW(m.parameters(), lr=1e-4, weight_decay=0.01, max_norm=1.0)
    opt.broadcast_parameters(0)
    g = GraphedTrainStep(m, opt, shard, fmt, 4, world, seed=5 + rank)
    try:
        l1 = float(g(shard))
        gsum1 = opt.flat_g.clone()
        p1 = opt.flat_p.clone()
        l2 = float(g(shard2))
        torch.cuda.synchronize()
    finally:
        g.close()
    if rank == 0:
        ref = make(200, 0.0, "bf16")
        ropt = tante_amd.FlatAdamW(ref.parameters(), lr=1e-4, weight_decay=0.01, max_norm=1.0)
        train_step(ref, ropt, {k: v.to(dev) for k, v in full.items()}, fmt, 4, 1)
        torch.cuda.synchronize()
        rg = ropt.flat_g.clone()
        report["graphed_bf16"] = {"grad_rel": rel(gsum1 * 0.5, rg), "param_rel_step1": rel(p1, ropt.flat_p), "loss_1": l1, "loss_2": l2}
        assert report["graphed_bf16"]["grad_rel"] < 2e-3, report
        assert l1 == l1 and l2 == l2
    D.barrier()

    # ---- (3) both ranks, the SAME sample, dropout 0.1: rank-mixed seeds must give different masks ----------------------------------
    m = make(300, 0.1, "bf16")
    opt = tante_amd.FlatAdamW(m.parameters(), lr=1e-4, weight_decay=0.01, max_norm=1.0)
    same = {k: v[:1].to(dev) for k, v in full.items()}
    from tante_amd import autograd as A
    A._SEED[0] = 1234
    ld = train_step(m, opt, same, fmt, 4, world)
    both = torch.stack([ld.detach().float().cpu(), torch.zeros(())]) if rank == 0 else torch.stack([torch.zeros(()), ld.detach().float().cpu()])
    import torch.distributed as dist
    dist.all_reduce(both)
    report["dropout_losses"] = [float(both[0]), float(both[1])]
    assert float(both[0]) != float(both[1]), "both ranks drew the same dropout masks"
    report["ok"] = True
    print("dp_gloo_1gpu " + json.dumps(report), flush=True)
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(child() if os.environ.get("TANTE_DP_CHILD") else parent())
