"""Many cfg2 rollouts back to back: the allocator's reserved memory must stop growing, every prediction stay finite and EVERY rollout
give the first one's bits (the fused tail's cross-workgroup hand-off, the split reductions).   python tools/rollout_soak.py [N] [B]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tante_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
cfg = tante_amd.load_config(os.path.join(ROOT, "configs", "tante_am.yaml")); wl = cfg["workload"]
md = tante_amd.TanteMetadata(n_fields=wl["n_fields"], spatial_resolution=tuple(wl["spatial_resolution"]))
torch.manual_seed(211)
m = tante_amd.build_model(cfg, md).to(dev).eval().set_compute("bf16")
B, n = (int(sys.argv[2]) if len(sys.argv) > 2 else wl["batch_size"]), wl["n_steps_rollout"]
g = torch.Generator().manual_seed(1)
batch = {"input": torch.randn(B, wl["n_steps_input"], *wl["spatial_resolution"], wl["n_fields"], generator=g).to(dev),
         "output": torch.randn(B, n, *wl["spatial_resolution"], wl["n_fields"], generator=g).to(dev)}
fmt = tante_amd.DefaultChannelsFirstFormatter(md)
res, first = [], None
for i in range(N):
    with torch.inference_mode():
        y, _ = tante_amd.rollout_model(m, batch, fmt, n, device=dev)
    if first is None:
        first = y.clone()
    assert torch.equal(y, first), f"rollout {i} of the same batch differs from the first"
    if i % 50 == 0 or i == N - 1:
        torch.cuda.synchronize()
        res.append(torch.cuda.memory_reserved())
        assert torch.isfinite(y).all()
        print(f"rollout {i:4d} (B = {B}): reserved {res[-1] / 2**20:.0f} MiB", flush=True)
assert res[-1] == res[1], f"reserved memory still growing: {res}"
print("ok")
