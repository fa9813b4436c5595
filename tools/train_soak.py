"""Overfit one fixed cfg3 batch for N steps (bf16, dropout 0.1, clip, AdamW): the loss must fall monotonically-ish and stay finite."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tante_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
tcfg = tante_amd.load_config(os.path.join(ROOT, "configs", "tante_trl.yaml")); twl = tcfg["workload"]
tmd = tante_amd.TanteMetadata(n_fields=twl["n_fields"], spatial_resolution=tuple(twl["spatial_resolution"]))
torch.manual_seed(211)
m = tante_amd.build_model(tcfg, tmd, dropout=float(tcfg["model"].get("dropout", 0.0))).to(dev).train().set_compute("bf16")
opt = tante_amd.FlatAdamW(m.parameters(), lr=1e-3, weight_decay=1e-5, max_norm=1.0)
B, n = twl["batch_size"], twl["n_steps_output"]
g = torch.Generator().manual_seed(1)
# a smooth, learnable target: the output frames are a slow drift of the inputs (so a Taylor step can fit it)
base = torch.randn(B, 1, *twl["spatial_resolution"], twl["n_fields"], generator=g)
drift = 0.05 * torch.randn(B, 1, *twl["spatial_resolution"], twl["n_fields"], generator=g)
frames = torch.cat([base + k * drift for k in range(twl["n_steps_input"] + n)], dim=1)
batch = {"input": frames[:, :twl["n_steps_input"]].to(dev), "output": frames[:, twl["n_steps_input"]:].to(dev)}
fmt = tante_amd.DefaultChannelsFirstFormatter(tmd)
losses = []
graphed = tante_amd.GraphedTrainStep(m, opt, batch, fmt, n, 1, seed=211) if os.environ.get("SOAK_GRAPH") else None      # SOAK_GRAPH=1: the replayed step
t0 = time.time()
for s in range(N):
    losses.append(float(graphed(batch) if graphed is not None else tante_amd.train_step(m, opt, batch, fmt, n, 1)))
    if s % 10 == 0 or s == N - 1:
        print(f"step {s:3d} loss {losses[-1]:.6f}", flush=True)
print(f"{N} steps in {time.time() - t0:.1f} s; finite: {all(l == l and l < 1e9 for l in losses)}; first {losses[0]:.5f} last {losses[-1]:.5f} min {min(losses):.5f}")
assert all(l == l for l in losses) and losses[-1] < 0.5 * losses[0], "training did not make progress"
