"""Which torch (aten) ops a cfg3 train step still launches, with their GPU time and call sites: torch.profiler over 3 steps.
    python tools/train_aten_profile.py          (on the GPU box)"""
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
B, n = twl["batch_size"], twl["n_steps_output"]
g = torch.Generator().manual_seed(1)
batch = {"input": torch.randn(B, twl["n_steps_input"], *twl["spatial_resolution"], twl["n_fields"], generator=g).to(dev),
         "output": torch.randn(B, n, *twl["spatial_resolution"], twl["n_fields"], generator=g).to(dev)}
fmt = tante_amd.DefaultChannelsFirstFormatter(tmd)
for _ in range(2): tante_amd.train_step(m, opt, batch, fmt, n, 1)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(3): tante_amd.train_step(m, opt, batch, fmt, n, 1)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.key.startswith("aten::") or "Backward" in e.key or e.key.endswith("Fn")]
rows.sort(key=lambda e: -e.count)
print("count/step  self-cuda-us/step  name")
for e in rows[:45]:
    print(f"{e.count / 3:9.1f} {e.self_device_time_total / 3:12.1f}   {e.key}")
for e in sorted(prof.key_averages(group_by_stack_n=8), key=lambda e: -e.count)[:40]:
    if e.key in ("aten::copy_", "aten::fill_", "aten::add_", "aten::add", "aten::mul", "aten::zeros", "aten::zero_", "aten::clone", "aten::contiguous", "aten::to", "aten::cat", "aten::mm", "aten::sum"):
        st = [fr for fr in e.stack if "tante_amd" in fr][:3]
        print(f"{e.count / 3:8.1f}  {e.key:18s} {st}")
