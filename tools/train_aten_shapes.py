"""Which torch (aten) ops with GPU time a cfg3 train step still launches, by op and INPUT SHAPES (the backward ops run on autograd's
worker thread and have no Python stack: the shapes say which tensors they are).   python tools/train_aten_shapes.py"""
import collections, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tante_amd
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
tcfg = tante_amd.load_config(os.path.join(ROOT, "configs", "tante_trl.yaml")); twl = tcfg["workload"]
tmd = tante_amd.TanteMetadata(n_fields=twl["n_fields"], spatial_resolution=tuple(twl["spatial_resolution"]))
torch.manual_seed(211)
m = tante_amd.build_model(tcfg, tmd, dropout=0.0).to(dev).train().set_compute("bf16")      # (dropout 0: the profiler cannot record 64-bit seed arguments)
oc = tcfg["optimizer"]
opt = tante_amd.FlatAdamW(m.parameters(), lr=oc["lr"], weight_decay=oc["weight_decay"], max_norm=1.0)
B, n = twl["batch_size"], twl["n_steps_output"]
g = torch.Generator().manual_seed(1)
batch = {"input": torch.randn(B, twl["n_steps_input"], *twl["spatial_resolution"], twl["n_fields"], generator=g).to(dev),
         "output": torch.randn(B, n, *twl["spatial_resolution"], twl["n_fields"], generator=g).to(dev)}
fmt = tante_amd.DefaultChannelsFirstFormatter(tmd)
for _ in range(2): tante_amd.train_step(m, opt, batch, fmt, n, 1)
torch.cuda.synchronize()
N = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(N): tante_amd.train_step(m, opt, batch, fmt, n, 1)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.name.startswith("aten::") and e.self_device_time_total > 0:
        k = (e.name, str(e.input_shapes)[:150])
        agg[k][0] += 1
        agg[k][1] += e.self_device_time_total
tot = sum(v[1] for v in agg.values()) / N
print(f"aten ops with GPU time: {tot:.0f} us per step")
for (name, shp), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:50]:
    print(f"{c / N:6.1f}/step {t / N:8.1f} us  {name:22s} {shp}")
