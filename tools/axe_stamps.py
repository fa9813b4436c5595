#!/usr/bin/env python3
"""Where the whole-tile H+W propagator kernel (axis_hw_exact_kernel, pointwise.hip) spends its time: in-kernel s_memtime stamps.
Needs the diagnostic library (python -m tante_amd.build --ablate).  Run on the GPU box:  python tools/axe_stamps.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tante_amd import _lib as L  # noqa: E402

L.LIB_PATH = os.path.join(ROOT, "tools", "_ab", "libtante_ablate.so")
import tante_amd  # noqa: E402,F401
from tante_amd import kernels as K  # noqa: E402

NAMES = ["issue plane DMA + weight loads", "wait vmcnt(0)", "barrier", "phase H", "W weights + barrier", "phase W (stores issued)", "stores drained"]


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    BT, H, W, Cc = 32, 32, 32, 256
    x = torch.randn(BT, H, W, Cc, device=dev)
    ws = [torch.randn(32, 32, device=dev) / 6, torch.randn(32, device=dev) * 0.1, torch.randn(32, 32, device=dev) / 6, torch.randn(32, device=dev) * 0.1]
    lib = L.lib()
    lib.tante_axe_set_stamps.argtypes = [C.c_void_p]
    nwg = BT * Cc // 16
    stamps = torch.zeros(nwg * 16 * 12, dtype=torch.int64, device=dev)
    for _ in range(3):
        K.axis_hw(x, BT, H, W, Cc, ws, ws, L.BF16)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        K.axis_hw(x, BT, H, W, Cc, ws, ws, L.BF16)
    e1.record()
    torch.cuda.synchronize()
    print(f"un-stamped: {e0.elapsed_time(e1) * 50:.1f} us per launch")
    lib.tante_axe_set_stamps(stamps.data_ptr())
    K.axis_hw(x, BT, H, W, Cc, ws, ws, L.BF16)
    torch.cuda.synchronize()
    lib.tante_axe_set_stamps(None)
    raw = stamps.cpu().numpy().reshape(nwg * 16, 12).astype(np.int64)
    raw = raw[raw[:, 0] != 0]
    d = np.diff(raw[:, :8], axis=1)
    tot = raw[:, 7] - raw[:, 0]
    print(f"per wave: {tot.mean():.0f} shader-clock ticks start -> stores drained (min {tot.min()}, max {tot.max()})")
    for i, n in enumerate(NAMES):
        print(f"  {n:34s} {d[:, i].mean():9.0f}  ({100 * d[:, i].mean() / tot.mean():5.1f} %)   min {d[:, i].min():7d}  max {d[:, i].max():7d}")
    t0, t1 = raw[:, 8], raw[:, 9]       # 100 MHz wall clock
    base = t0.min()
    print(f"wall clock (10 ns ticks): first start 0, last start {t0.max() - base}, first end {t1.min() - base}, last end {t1.max() - base}")
    print(f"ticks per 10 ns: {tot.mean() / (t1 - t0).mean():.2f}  (s_memtime rate in units of 100 MHz)")
    wgs = t0 - base
    print("start-time histogram of waves (10 ns ticks):", np.histogram(wgs, bins=8)[0], np.histogram(wgs, bins=8)[1].astype(int))


if __name__ == "__main__":
    main()
