"""GPU kernels of one cfg4 (configs/cvit_rb.yaml) forward in bf16 at B = 1 (or argv[1]), in launch order (torch.profiler).
   python tools/cvit_kernel_profile.py [B]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tante_amd
from torch.profiler import profile, ProfilerActivity
cfg = tante_amd.load_config(os.path.join(ROOT, "configs", "cvit_rb.yaml")); wl = cfg["workload"]
H, W = wl["spatial_resolution"]
md = tante_amd.TanteMetadata(n_fields=wl["n_fields"], spatial_resolution=(H, W))
torch.manual_seed(211)
m = tante_amd.build_model(cfg, md).cuda().eval().set_compute("bf16")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
x = torch.randn(B, wl["n_steps_input"], wl["n_fields"], H, W, device="cuda")
N = 4
with torch.no_grad():
    for _ in range(3): m(x)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(N): m(x)
        torch.cuda.synchronize()
evs = sorted((e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA), key=lambda e: e.time_range.start)
per = len(evs) // N
step = evs[(N - 1) * per:]
t0 = step[0].time_range.start
print(f"--- one forward at B = {B}: {len(step)} device events, {sum(e.time_range.end - e.time_range.start for e in step):.0f} us of kernels, "
      f"{step[-1].time_range.end - t0:.0f} us first start -> last end")
short = lambda k: k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:80]
for e in step:
    print(f"{e.time_range.start - t0:9.1f} {e.time_range.end - e.time_range.start:7.1f}  {short(e.name)}")
