#!/usr/bin/env python3
"""Time the shared weight-gradient launch of the train step (tante_wgrad_multi_ws: the four BPTT uses of one weight, R = 24 576 rows each,
contracted as one row range) for the shapes of a TransformerBlock at C = 256, with distinct operand tensors per use so that the rows
come from HBM.  Prints us per launch and the operand bytes / time.  Environment switches of wgrad.hip apply (TANTE_WGRAD_TR_WGS, ...).
    python tools/wgrad_multi_time.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tante_amd import _lib as L  # noqa: E402

if os.environ.get("TANTE_LIB"):
    L.LIB_PATH = os.environ["TANTE_LIB"]
import tante_amd  # noqa: E402,F401
from tante_amd.autograd import _rm_linear, _wgrad_workspace  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    R, n = 24576, 4
    for I, J in ((256, 256), (768, 256)):
        dys = [torch.randn(R, I, device=dev).bfloat16() for _ in range(n)]
        acts = [torch.randn(R, J, device=dev).bfloat16() for _ in range(n)]
        gW = torch.zeros(I, J, device=dev)
        gb = torch.zeros(I, device=dev)
        U = (L.RowMat * n)(*[_rm_linear(t) for t in dys])
        V = (L.RowMat * n)(*[_rm_linear(t) for t in acts])
        ws = _wgrad_workspace(dev)
        s = torch.cuda.current_stream().cuda_stream

        def run():
            L.check(L.lib().tante_wgrad_multi_ws(C.byref(U), C.byref(V), n, R, I, J, gW.data_ptr(), gb.data_ptr(), L.W_LINEAR, 0, 0, 0, L.BF16, 1,
                                                 ws.data_ptr(), ws.numel(), s), "wgrad_multi")
        run()
        ref = sum(d.float().T @ a.float() for d, a in zip(dys, acts))
        err = float((gW - ref).norm() / ref.norm())
        flush = torch.zeros(128 << 20, dtype=torch.float32, device=dev)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tot = 0.0
        for _ in range(10):
            if not os.environ.get("NOFLUSH"): flush.sum()                         # operands out of the 256 MB cache (a READ sweep: no dirty lines left to write back)
            e0.record(); run(); e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        us = tot / 10 * 1e3
        nbytes = n * R * (I + J) * 2
        print(f"I={I} J={J}: {us:7.1f} us per launch (incl. the reduce kernel)  {nbytes / us / 1e6:5.2f} TB/s of operand bytes  "
              f"{2.0 * n * R * I * J / us / 1e6:6.1f} TFLOP/s   rel err {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
