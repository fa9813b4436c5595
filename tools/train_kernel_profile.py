"""GPU kernels of one cfg3 train step (eager), by total time, with counts: torch.profiler over 3 steps (a quick stand-in for
rocprofv3 --stats when iterating).   python tools/train_kernel_profile.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tante_amd
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
tcfg = tante_amd.load_config(os.path.join(ROOT, "configs", "tante_trl.yaml")); twl = tcfg["workload"]
tmd = tante_amd.TanteMetadata(n_fields=twl["n_fields"], spatial_resolution=tuple(twl["spatial_resolution"]))
torch.manual_seed(211)
m = tante_amd.build_model(tcfg, tmd, dropout=float(tcfg["model"].get("dropout", 0.0))).to(dev).train().set_compute("bf16")
oc = tcfg["optimizer"]
opt = tante_amd.FlatAdamW(m.parameters(), lr=oc["lr"], weight_decay=oc["weight_decay"], max_norm=1.0)
B, n = int(os.environ.get("TRAIN_B", twl["batch_size"])), twl["n_steps_output"]
g = torch.Generator().manual_seed(1)
batch = {"input": torch.randn(B, twl["n_steps_input"], *twl["spatial_resolution"], twl["n_fields"], generator=g).to(dev),
         "output": torch.randn(B, n, *twl["spatial_resolution"], twl["n_fields"], generator=g).to(dev)}
fmt = tante_amd.DefaultChannelsFirstFormatter(tmd)
for _ in range(2): tante_amd.train_step(m, opt, batch, fmt, n, 1)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3): tante_amd.train_step(m, opt, batch, fmt, n, 1)
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in rows) / 3
print(f"total kernel time per step {tot:.0f} us")
print("count/step  us/step   avg us   name")
for e in rows[:60]:
    print(f"{e.count / 3:9.1f} {e.self_device_time_total / 3:9.1f} {e.self_device_time_total / max(1, e.count):8.1f}   {e.key[:150]}")
if "--sequence" in sys.argv:      # the launches of the LAST profiled step in start order (what runs between the block kernels)
    evs = sorted((e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA), key=lambda e: e.time_range.start)
    per = len(evs) // 3
    step = evs[2 * per:]
    t0 = step[0].time_range.start
    print(f"--- one step: {len(step)} device events")
    short = lambda k: k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    for e in step:
        print(f"{e.time_range.start - t0:9.1f} {e.time_range.end - e.time_range.start:7.1f}  {short(e.name)}")
