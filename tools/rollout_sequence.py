"""Launch sequence of ONE cfg2 rollout (configs/tante_am.yaml, bf16, B = 8) in start order with durations and gaps (torch.profiler).
   python tools/rollout_sequence.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tante_amd
from tante_amd import rollout as R
from torch.profiler import profile, ProfilerActivity
cfg = tante_amd.load_config(os.path.join(ROOT, "configs", "tante_am.yaml")); wl = cfg["workload"]
H, W = wl["spatial_resolution"]
md = tante_amd.TanteMetadata(n_fields=wl["n_fields"], spatial_resolution=(H, W))
torch.manual_seed(211)
m = tante_amd.build_model(cfg, md).cuda().eval().set_compute("bf16")
B, T, n = wl["batch_size"], wl["n_steps_input"], wl["n_steps_rollout"]
batch = {"input": torch.randn(B, T, H, W, wl["n_fields"]).cuda(), "output": torch.randn(B, n, H, W, wl["n_fields"]).cuda()}
fmt = tante_amd.DefaultChannelsFirstFormatter(md)
with torch.no_grad():
    for _ in range(5): R.rollout_model(m, batch, fmt, n)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        R.rollout_model(m, batch, fmt, n)
        torch.cuda.synchronize()
evs = sorted((e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA), key=lambda e: e.time_range.start)
t0 = evs[0].time_range.start
busy = sum(e.time_range.end - e.time_range.start for e in evs)
print(f"one rollout: {len(evs)} device events, {busy:.0f} us of kernels, span {evs[-1].time_range.end - t0:.0f} us")
short = lambda k: k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
prev_end = t0
for e in evs:
    print(f"{e.time_range.start - t0:9.1f} {e.time_range.end - e.time_range.start:7.1f}  gap {e.time_range.start - prev_end:6.1f}  {short(e.name)}")
    prev_end = max(prev_end, e.time_range.end)
