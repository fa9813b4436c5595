"""Which torch (aten) ops and runtime copies one steady-state inference call of a config still launches, by op and INPUT SHAPES.
   python tools/fwd_aten_ops.py configs/cvit_rb.yaml [batch]"""
import collections, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tante_amd
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
cfg = tante_amd.load_config(os.path.join(ROOT, sys.argv[1])); wl = cfg["workload"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else wl["batch_size"]
md = tante_amd.TanteMetadata(n_fields=wl["n_fields"], spatial_resolution=tuple(wl["spatial_resolution"]))
is_cvit = cfg["model"]["_target_"].endswith("CViT")
n_roll = cfg["model"]["out_steps"] if is_cvit else wl["n_steps_rollout"]
torch.manual_seed(211)
m = tante_amd.build_model(cfg, md).to(dev).eval().set_compute("bf16")
g = torch.Generator().manual_seed(1)
batch = {"input": torch.randn(B, wl["n_steps_input"], *wl["spatial_resolution"], wl["n_fields"], generator=g).to(dev),
         "output": torch.randn(B, n_roll, *wl["spatial_resolution"], wl["n_fields"], generator=g).to(dev)}
fmt = tante_amd.DefaultChannelsFirstFormatter(md)
x = fmt.process_input(batch)[0][0].to(dev) if is_cvit else None


def call():
    with torch.no_grad():
        if is_cvit:
            return m(x)
        return tante_amd.rollout_model(m, batch, fmt, n_roll)


for _ in range(3): call()
torch.cuda.synchronize()
N = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(N): call()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
kern = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.name.startswith("aten::") and e.self_device_time_total > 0:
        k = (e.name, str(e.input_shapes)[:150])
        agg[k][0] += 1
        agg[k][1] += e.self_device_time_total
    if e.device_type == torch.autograd.DeviceType.CUDA:
        kern[e.name[:110]][0] += 1
        kern[e.name[:110]][1] += e.device_time_total
print(f"aten ops with GPU time: {sum(v[1] for v in agg.values()) / N:.0f} us per call")
for (name, shp), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{c / N:6.1f}/call {t / N:8.1f} us  {name:22s} {shp}")
print(f"\nGPU kernels / copies: {sum(v[1] for v in kern.values()) / N:.0f} us per call, {sum(v[0] for v in kern.values()) / N:.1f} launches per call")
for name, (c, t) in sorted(kern.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{c / N:6.1f}/call {t / N:8.1f} us  {name}")
