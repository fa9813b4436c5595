#!/usr/bin/env python3
"""Where the fused derivative head (fused_head_kernel, head_fused.hip) spends its time: in-kernel s_memtime stamps.
Needs the diagnostic library (python -m tante_amd.build --ablate).  Run on the GPU box:  python tools/head_stamps.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tante_amd import _lib as L  # noqa: E402

L.LIB_PATH = os.path.join(ROOT, "tools", "_ab", "libtante_ablate.so")
import tante_amd  # noqa: E402
from tante_amd import kernels as K  # noqa: E402

NAMES = ["W1/W3 DMA, token rows -> fragments, barrier, W2 DMA, RMW operands", "stage 1", "wait vmcnt(0)", "barrier", "sub-pixel 0",
         "sub-pixel 1 (+stores)", "sub-pixel 2", "sub-pixel 3 (+stores)", "stores drained"]


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    B = 8
    md = tante_amd.TanteMetadata(n_fields=11, spatial_resolution=(256, 256))
    cfg = tante_amd.load_config(os.path.join(ROOT, "configs", "tante_am.yaml"))
    m = tante_amd.build_model(cfg, md).to(dev).eval().set_compute("bf16")
    x = torch.randn(B * 4 * 1024, 256, device=dev)
    out = torch.zeros(B, 1, 11, 256, 256, device=dev)
    ph = m.decoders[0].packed_head()

    def run():
        K.head_fused(x, 1024, 4 * 1024 * 256, 256, 3 * 1024 * 256, B, 32, 32, 256, 11, ph, out, out[0].numel(), [1.0], None)
    lib = L.lib()
    lib.tante_head_set_stamps.argtypes = [C.c_void_p]
    nwg = 1024
    stamps = torch.zeros(nwg * 8 * 12, dtype=torch.int64, device=dev)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f"un-stamped: {e0.elapsed_time(e1) * 50:.1f} us per launch")
    lib.tante_head_set_stamps(stamps.data_ptr())
    run()
    torch.cuda.synchronize()
    lib.tante_head_set_stamps(None)
    raw = stamps.cpu().numpy().reshape(nwg * 8, 12).astype(np.int64)
    raw = raw[raw[:, 0] != 0]
    d = np.diff(raw[:, :10], axis=1)
    tot = raw[:, 9] - raw[:, 0]
    print(f"{len(raw)} waves; per wave: {tot.mean():.0f} shader-clock ticks start -> stores drained (min {tot.min()}, max {tot.max()})")
    for i, n in enumerate(NAMES):
        print(f"  {n:40s} {d[:, i].mean():9.0f}  ({100 * d[:, i].mean() / tot.mean():5.1f} %)   min {d[:, i].min():7d}  max {d[:, i].max():7d}")
    print(f"  within the first stage: DMA issued after {(raw[:, 10] - raw[:, 0]).mean():.0f}, token-row loads issued after {(raw[:, 11] - raw[:, 0]).mean():.0f} ticks")
    t0 = raw[:, 0]
    print("span of wave starts (ticks):", t0.max() - t0.min(), " span first start -> last end:", raw[:, 9].max() - t0.min())


if __name__ == "__main__":
    main()
