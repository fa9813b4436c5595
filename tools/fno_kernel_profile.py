"""GPU kernels of one cfg5 (configs/tante_fno.yaml) model call in bf16, by total time and in launch order (torch.profiler; a quick stand-in for
rocprofv3 --stats when iterating).   python tools/fno_kernel_profile.py [--sequence]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tante_amd
from torch.profiler import profile, ProfilerActivity
cfg = tante_amd.load_config(os.path.join(ROOT, "configs", "tante_fno.yaml")); wl = cfg["workload"]
H, W = wl["spatial_resolution"]
md = tante_amd.TanteMetadata(n_fields=wl["n_fields"], spatial_resolution=(H, W))
torch.manual_seed(211)
m = tante_amd.build_model(cfg, md).cuda().eval().set_compute("bf16")
x = torch.randn(wl["batch_size"], wl["n_steps_input"], wl["n_fields"], H, W, device="cuda")
N = 4
with torch.no_grad():
    for _ in range(3): m(x)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(N): m(x)
        torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in rows) / N
print(f"total kernel time per model call {tot:.0f} us (B = {wl['batch_size']})")
print("count/call  us/call   avg us   name")
for e in rows[:40]:
    print(f"{e.count / N:9.1f} {e.self_device_time_total / N:9.1f} {e.self_device_time_total / max(1, e.count):8.1f}   {e.key[:150]}")
if "--sequence" in sys.argv:
    evs = sorted((e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA), key=lambda e: e.time_range.start)
    per = len(evs) // N
    step = evs[(N - 1) * per:]
    t0 = step[0].time_range.start
    print(f"--- one call: {len(step)} device events")
    short = lambda k: k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    for e in step:
        print(f"{e.time_range.start - t0:9.1f} {e.time_range.end - e.time_range.start:7.1f}  {short(e.name)}")
