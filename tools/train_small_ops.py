"""Which small torch kernels a cfg3 train step still launches: op, input shapes, GPU time, and the nearest tante_amd frame (or the
autograd engine).   python tools/train_small_ops.py      (on the GPU box)"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tante_amd  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

dev = torch.device("cuda:0")
tcfg = tante_amd.load_config(os.path.join(ROOT, "configs", "tante_trl.yaml"))
twl = tcfg["workload"]
tmd = tante_amd.TanteMetadata(n_fields=twl["n_fields"], spatial_resolution=tuple(twl["spatial_resolution"]))
torch.manual_seed(211)
m = tante_amd.build_model(tcfg, tmd, dropout=float(tcfg["model"].get("dropout", 0.0))).to(dev).train().set_compute("bf16")
oc = tcfg["optimizer"]
opt = tante_amd.FlatAdamW(m.parameters(), lr=oc["lr"], weight_decay=oc["weight_decay"], max_norm=1.0)
B, n = twl["batch_size"], twl["n_steps_output"]
g = torch.Generator().manual_seed(1)
batch = {"input": torch.randn(B, twl["n_steps_input"], *twl["spatial_resolution"], twl["n_fields"], generator=g).to(dev),
         "output": torch.randn(B, n, *twl["spatial_resolution"], twl["n_fields"], generator=g).to(dev)}
fmt = tante_amd.DefaultChannelsFirstFormatter(tmd)
for _ in range(2):
    tante_amd.train_step(m, opt, batch, fmt, n, 1)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tante_amd.train_step(m, opt, batch, fmt, n, 1)
    torch.cuda.synchronize()
WANT = ("aten::add_", "aten::add", "aten::copy_", "aten::fill_", "aten::cat", "aten::mm", "aten::addmm", "aten::sum", "aten::mul", "aten::zero_",
        "aten::nan_to_num", "aten::div", "aten::sub", "aten::neg", "aten::stack", "aten::index_select", "aten::sqrt", "aten::clamp")
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.name in WANT and e.device_time_total > 0:
        fr = [f for f in (e.stack or []) if "tante_amd" in f or "bench" in f]
        site = fr[0].split("tante_amd/")[-1][:70] if fr else "(autograd engine / no python frame)"
        shp = ""
        a = agg[(e.name, shp, site)]
        a[0] += 1
        a[1] += e.device_time_total
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in agg.values())
print(f"total small-op GPU time per step: {tot:.0f} us in {sum(v[0] for v in agg.values())} launches")
for (name, shp, site), (c, t) in rows[:45]:
    print(f"{t:8.1f} us {c:4d}  {name:18s} {shp:62s} {site}")
