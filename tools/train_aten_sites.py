"""Call sites of the torch (aten) ops that still launch kernels in a cfg3 train step (eager): python tools/train_aten_sites.py"""
import os, sys, collections, torch
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
# wrap the aten entry points that launch: record the python call site of each call
sites = collections.Counter()
import traceback
def wrap(obj, name, label):
    orig = getattr(obj, name)
    def f(*a, **k):
        st = [fr for fr in traceback.extract_stack()[:-1] if "tante_amd" in fr.filename or "torch/autograd" in fr.filename]
        key = label + " <- " + " | ".join(f"{os.path.basename(fr.filename)}:{fr.lineno}" for fr in st[-3:])
        sites[key] += 1
        return orig(*a, **k)
    setattr(obj, name, f)
for nm in ("zeros", "zeros_like", "cat", "stack", "empty_like"):
    wrap(torch, nm, "torch." + nm)
for nm in ("contiguous", "clone", "zero_", "fill_", "copy_", "add_", "mul_", "float", "to"):
    wrap(torch.Tensor, nm, "Tensor." + nm)
tante_amd.train_step(m, opt, batch, fmt, n, 1)
torch.cuda.synchronize()
for k, v in sorted(sites.items(), key=lambda t: -t[1])[:60]:
    print(f"{v:4d}  {k}")
