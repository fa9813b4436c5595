"""Repeat ONE cfg3-shaped train step (dropout 0, fixed weights and batch) N times and compare every gradient bucket with the first
run's: a missing barrier in the tail kernels (csrc/tail_chain.hip) would show as run-to-run differences far above the last-bit noise of
the reduce launches' atomics.   python tools/tail_determinism_soak.py [N]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tante_amd
from tante_amd import autograd as A
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
tcfg = tante_amd.load_config(os.path.join(ROOT, "configs", "tante_trl.yaml")); twl = tcfg["workload"]
tmd = tante_amd.TanteMetadata(n_fields=twl["n_fields"], spatial_resolution=tuple(twl["spatial_resolution"]))
torch.manual_seed(211)
m = tante_amd.build_model(tcfg, tmd, dropout=0.0).to(dev).train().set_compute("bf16")
opt = tante_amd.FlatAdamW(m.parameters(), lr=1e-3)
B, n = twl["batch_size"], twl["n_steps_output"]
g = torch.Generator().manual_seed(1)
batch = {"input": torch.randn(B, twl["n_steps_input"], *twl["spatial_resolution"], twl["n_fields"], generator=g).to(dev),
         "output": torch.randn(B, n, *twl["spatial_resolution"], twl["n_fields"], generator=g).to(dev)}
fmt = tante_amd.DefaultChannelsFirstFormatter(tmd)
names = [k for k, _ in m.named_parameters()]
ref, worst, worst_name = None, 0.0, ""
for s in range(N):
    opt.zero_grad()
    y_pred, y_ref = tante_amd.rollout_model(m, batch, fmt, n)
    loss = A.MseMeanFn.apply(y_pred, y_ref)
    A.run_backward(loss)
    torch.cuda.synchronize()
    grads = [p.grad.detach().clone() for p in m.parameters()]
    if ref is None:
        ref = grads
        continue
    for k, a, b in zip(names, grads, ref):
        e = float((a - b).norm() / (b.norm() + 1e-30))
        if e > worst:
            worst, worst_name = e, k
print(f"{N} runs of one step: worst relative difference of a gradient tensor from the first run {worst:.3e} ({worst_name})")
assert worst < 1e-5, "run-to-run differences above the atomics' last-bit noise: a race?"
