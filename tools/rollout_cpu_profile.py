"""Host-side issue time of one cfg2 rollout (8 model calls) against its GPU time."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tante_amd
dev = torch.device("cuda:0")
cfg = tante_amd.load_config(os.path.join(ROOT, "configs", "tante_am.yaml")); wl = cfg["workload"]
md = tante_amd.TanteMetadata(n_fields=wl["n_fields"], spatial_resolution=tuple(wl["spatial_resolution"]))
torch.manual_seed(211)
m = tante_amd.build_model(cfg, md).to(dev).eval().set_compute("bf16")
B, n = wl["batch_size"], wl["n_steps_rollout"]
g = torch.Generator().manual_seed(1)
batch = {"input": torch.randn(B, wl["n_steps_input"], *wl["spatial_resolution"], wl["n_fields"], generator=g).to(dev),
         "output": torch.randn(B, n, *wl["spatial_resolution"], wl["n_fields"], generator=g).to(dev)}
fmt = tante_amd.DefaultChannelsFirstFormatter(md)
with torch.no_grad():
    for _ in range(3): tante_amd.rollout_model(m, batch, fmt, n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): tante_amd.rollout_model(m, batch, fmt, n)
    t_issue = (time.perf_counter() - t0) / 10
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / 10
print(f"issue {t_issue * 1e3:.2f} ms/rollout, complete {t_all * 1e3:.2f} ms/rollout")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
with torch.no_grad():
    for _ in range(5): tante_amd.rollout_model(m, batch, fmt, n)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
