"""Probe: does HIP-graph capture and/or sub-batch interleaving on two streams shorten the cfg2 rollout?
    python tools/graph_probe.py [--config configs/tante_am.yaml] [--iters 20]
Prints ms per rollout of B samples for: eager, one captured graph, eager two streams x B/2, captured two-branch graph x B/2, and x B/4 on 4 streams."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tante_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=os.path.join(os.path.dirname(__file__), "..", "configs", "tante_am.yaml"))
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = tante_amd.load_config(args.config)
    wl = cfg["workload"]
    B, n_steps, T_in, res, D = wl["batch_size"], wl["n_steps_rollout"], wl["n_steps_input"], tuple(wl["spatial_resolution"]), wl["n_fields"]
    md = tante_amd.TanteMetadata(n_fields=D, spatial_resolution=res)
    torch.manual_seed(211)
    model = tante_amd.build_model(cfg, md).to(dev).eval().set_compute("bf16")
    fmt = tante_amd.DefaultChannelsFirstFormatter(md)
    full = {"input": torch.randn(B, T_in, *res, D, device=dev), "output": torch.randn(B, n_steps, *res, D, device=dev)}

    def parts(n):
        k = B // n
        return [{"input": full["input"][i * k:(i + 1) * k].contiguous(), "output": full["output"][i * k:(i + 1) * k].contiguous()} for i in range(n)]

    def run(batch):
        with torch.inference_mode():
            return tante_amd.rollout_model(model, batch, fmt, n_steps, device=dev)[0]

    def timeit(fn, label):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.iters * 1e3
        print(f"{label:46s} {ms:7.3f} ms/rollout  {B * n_steps / ms * 1e3:9.0f} frames/s", flush=True)
        return ms

    ref = run(full).clone()
    timeit(lambda: run(full), "eager, one stream")

    # calibrate torch.cuda._sleep (spin kernel, one workgroup): cycles per microsecond
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); torch.cuda._sleep(2_000_000); e1.record(); torch.cuda.synchronize()
    cyc_per_us = 2_000_000 / (e0.elapsed_time(e1) * 1e3)
    print(f"_sleep: {cyc_per_us:.1f} cycles per us", flush=True)

    def multi_stream(n, stagger_us=0.0):
        ps = parts(n)
        streams = [torch.cuda.Stream() for _ in range(n)]

        def fn():
            cur = torch.cuda.current_stream()
            outs = []
            for i, (s, p) in enumerate(zip(streams, ps)):
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    if stagger_us and i:
                        torch.cuda._sleep(int(stagger_us * i * cyc_per_us))
                    outs.append(run(p))
            for s in streams:
                cur.wait_stream(s)
            return outs
        return fn

    for n in (2, 4):
        f = multi_stream(n)
        o = torch.cat(f(), 0)
        torch.cuda.synchronize()
        print("   max |diff| vs full batch:", float((o - ref).abs().max()))
        timeit(f, f"eager, {n} streams x B/{n}")

    def captured(fn):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = fn()
        return g, out

    try:
        g, out = captured(lambda: run(full))
        g.replay()
        torch.cuda.synchronize()
        print("   graph max |diff|:", float((out - ref).abs().max()))
        timeit(g.replay, "captured graph, one branch")
    except Exception as e:      # noqa: BLE001
        print("capture failed:", repr(e))
    for n, st in ((2, 0), (2, 10), (2, 18), (2, 27), (2, 60), (4, 0), (4, 9), (4, 20)):
        try:
            g, outs = captured(multi_stream(n, st))
            g.replay()
            torch.cuda.synchronize()
            print("   graph max |diff|:", float((torch.cat(outs, 0) - ref).abs().max()))
            timeit(g.replay, f"captured graph, {n} branches x B/{n}, stagger {st} us")
        except Exception as e:      # noqa: BLE001
            print(f"capture ({n} branches) failed:", repr(e))


if __name__ == "__main__":
    main()
