#!/usr/bin/env python3
"""Does the cfg2 rollout gain from running the batch as TWO half-batch rollouts on two streams?  Samples are independent (no data-path
collective), and a B = 8 block launch is exactly one resident round in which every workgroup walks the same phase at the same time; two
B = 4 rollouts issued one behind the other put DIFFERENT kernels of the two halves on a CU together (a block kernel beside a propagator
or the tail).  Prints frames/s of: one B = 8 rollout; two B = 4 rollouts on two streams; two B = 4 rollouts on one stream.
    python tools/two_rollout_streams_probe.py"""
import os
import sys
import time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tante_amd

dev = torch.device("cuda:0")
cfg = tante_amd.load_config(os.path.join(ROOT, "configs", "tante_am.yaml"))
wl = cfg["workload"]
B, T_in, res, D, n_steps = wl["batch_size"], wl["n_steps_input"], tuple(wl["spatial_resolution"]), wl["n_fields"], wl["n_steps_rollout"]
md = tante_amd.TanteMetadata(n_fields=D, spatial_resolution=res)
torch.manual_seed(211)
model = tante_amd.build_model(cfg, md).to(dev).eval().set_compute("bf16")
fmt = tante_amd.DefaultChannelsFirstFormatter(md)
gen = torch.Generator().manual_seed(211)
full = {"input": torch.randn(B, T_in, *res, D, generator=gen).to(dev), "output": torch.randn(B, n_steps, *res, D, generator=gen).to(dev)}
halves = [{k: v[i * (B // 2):(i + 1) * (B // 2)].contiguous() for k, v in full.items()} for i in range(2)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def one():
    with torch.inference_mode():
        return tante_amd.rollout_model(model, full, fmt, n_steps, device=dev)[0]


def two(streams):
    outs = []
    cur = torch.cuda.current_stream()
    for h, st in zip(halves, streams):
        st.wait_stream(cur)
        with torch.cuda.stream(st), torch.inference_mode():
            outs.append(tante_amd.rollout_model(model, h, fmt, n_steps, device=dev)[0])
    for st in streams:
        cur.wait_stream(st)
    return outs


def timeit(fn, n=12, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / n)
    return sorted(ts)[1]


y = one()
ya = two([s1, s2])
torch.cuda.synchronize()
print("two half-batch rollouts give the B = 8 rollout's frames:", bool(torch.equal(torch.cat(ya), y)), float((torch.cat(ya) - y).abs().max()))
for r in range(2):
    t8 = timeit(one)
    t44 = timeit(lambda: two([s1, s2]))
    t44s = timeit(lambda: two([s1, s1]))
    f = B * n_steps
    print(f"one B=8 rollout {1e3 * t8:.3f} ms = {f / t8:.0f} frames/s;  two B=4 on two streams {1e3 * t44:.3f} ms = {f / t44:.0f};  two B=4 on one stream {1e3 * t44s:.3f} ms = {f / t44s:.0f}")
