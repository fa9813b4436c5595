#!/usr/bin/env python3
"""Time the propagator backward at cfg3's H / W shapes: the fused MFMA launch (tante_axis_mlp_bwd_fused: dx + the four parameter gradients)
against the three launches it replaces (tante_axis_mlp_bwd + 2 x tante_axis_wgrad_ws).   python tools/axis_bwd_time.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tante_amd import _lib as L  # noqa: E402

if os.environ.get("TANTE_LIB"):
    L.LIB_PATH = os.environ["TANTE_LIB"]
import tante_amd  # noqa: E402,F401


def main():
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    flush = torch.zeros(128 << 20, dtype=torch.float32, device=dev)
    for name, outer, n, inner in (("W", 32 * 16, 48, 256), ("H", 32, 16, 48 * 256), ("T", 8, 4, 16 * 48 * 256)):
        x, dy = torch.randn(outer, n, inner, device=dev), torch.randn(outer, n, inner, device=dev)
        w1, w2, b1 = torch.randn(n, n, device=dev) / n ** 0.5, torch.randn(n, n, device=dev) / n ** 0.5, torch.randn(n, device=dev) * 0.1
        dx, h, dp = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        dW1, dW2, db1, db2 = torch.zeros(n, n, device=dev), torch.zeros(n, n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        ws = torch.zeros(L.lib().tante_axis_wgrad_workspace_bytes() // 4, device=dev)

        def fused():
            L.check(L.lib().tante_axis_mlp_bwd_fused_ws(x.data_ptr(), dy.data_ptr(), outer, n, inner, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                                                        dx.data_ptr(), dW1.data_ptr(), db1.data_ptr(), dW2.data_ptr(), db2.data_ptr(),
                                                        None if os.environ.get("NO_WS") else ws.data_ptr(), ws.numel() * 4, s))

        def three():
            L.check(L.lib().tante_axis_mlp_bwd(x.data_ptr(), dy.data_ptr(), outer, n, inner, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), dx.data_ptr(),
                                               h.data_ptr(), dp.data_ptr(), s))
            L.check(L.lib().tante_axis_wgrad_ws(dy.data_ptr(), h.data_ptr(), outer, n, inner, dW2.data_ptr(), db2.data_ptr(), 1, ws.data_ptr(), ws.numel() * 4, s))
            L.check(L.lib().tante_axis_wgrad_ws(dp.data_ptr(), x.data_ptr(), outer, n, inner, dW1.data_ptr(), db1.data_ptr(), 1, ws.data_ptr(), ws.numel() * 4, s))

        for label, fn in (("fused", fused), ("bwd + 2 wgrad", three)):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            tot = 0.0
            for _ in range(10):
                flush.sum()
                e0.record(); fn(); e1.record()
                torch.cuda.synchronize()
                tot += e0.elapsed_time(e1)
            print(f"axis {name} (outer {outer}, n {n}, inner {inner}): {label:14s} {tot * 100:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
