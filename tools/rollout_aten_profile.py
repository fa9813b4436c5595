"""Which torch (aten) ops a cfg2 rollout still launches, with GPU time and call sites: torch.profiler over 2 rollouts.
    python tools/rollout_aten_profile.py          (on the GPU box)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tante_amd
from torch.profiler import profile, ProfilerActivity
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
def step():
    with torch.inference_mode():
        return tante_amd.rollout_model(m, batch, fmt, n, device=dev)
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(2): step()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.self_device_time_total)
print("count/rollout  self-cuda-us/rollout  name")
for e in rows[:30]:
    print(f"{e.count / 2:9.1f} {e.self_device_time_total / 2:12.1f}   {e.key[:110]}")
print("-- aten ops by call site")
for e in sorted(prof.key_averages(group_by_stack_n=10), key=lambda e: -e.count)[:60]:
    if e.key.startswith("aten::") and e.key not in ("aten::empty", "aten::view", "aten::as_strided", "aten::slice", "aten::select", "aten::detach", "aten::permute", "aten::empty_strided", "aten::empty_like", "aten::reshape", "aten::alias", "aten::_unsafe_view"):
        st = [fr for fr in e.stack if "tante_amd" in fr or "bench" in fr][:3]
        print(f"{e.count / 2:8.1f}  {e.key:22s} {e.self_device_time_total / 2:9.1f} us  {st}")
