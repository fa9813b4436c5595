#!/usr/bin/env python3
"""Does warming the Infinity Cache with the encoder's packed weights on a side stream shorten CViT's B = 1 forward?  (experiment: the touch
is a torch sum per tensor on a second stream)   python tools/cvit_prefetch_probe.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tante_amd
from tante_amd.attn_backbone import _PackCache
dev = torch.device("cuda:0")
cfg = tante_amd.load_config(os.path.join(ROOT, "configs/cvit_rb.yaml")); wl = cfg["workload"]
md = tante_amd.TanteMetadata(n_fields=wl["n_fields"], spatial_resolution=tuple(wl["spatial_resolution"]))
torch.manual_seed(211)
m = tante_amd.build_model(cfg, md).to(dev).eval().set_compute("bf16")
x = torch.randn(1, 4, 512, 128, 4, device=dev).permute(0, 1, 4, 2, 3)


def tensors(o, acc):
    if isinstance(o, torch.Tensor):
        if o.is_cuda and o.numel() * o.element_size() >= 65536: acc.append(o)
    elif isinstance(o, dict):
        for v in o.values(): tensors(v, acc)
    elif isinstance(o, (list, tuple)):
        for v in o: tensors(v, acc)
    elif hasattr(o, "__dict__"):
        for v in vars(o).values(): tensors(v, acc)


with torch.no_grad():
    for _ in range(3): m(x)
    torch.cuda.synchronize()
    ws = []
    for mod in m.Encoder.modules():
        for v in vars(mod).values():
            if isinstance(v, _PackCache):
                for key, val in v._store.values(): tensors(val, ws)
    seen, uniq = set(), []
    for t in ws:
        if t.data_ptr() not in seen:
            seen.add(t.data_ptr()); uniq.append(t)
    print(f"{len(uniq)} packed encoder tensors, {sum(t.numel() * t.element_size() for t in uniq) / 1e6:.1f} MB")
    flat = [t.view(-1).view(torch.int32) if (t.numel() * t.element_size()) % 4 == 0 else t.view(-1) for t in uniq]
    side = torch.cuda.Stream()

    def fwd(pref):
        if pref:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for t in flat: t.sum()
        return m(x)

    for pref in (False, True, False, True):
        for _ in range(5): fwd(pref)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40): fwd(pref)
        e1.record()
        torch.cuda.synchronize()
        print(f"prefetch={pref}: {e0.elapsed_time(e1) / 40 * 1e3:.0f} us per forward")
