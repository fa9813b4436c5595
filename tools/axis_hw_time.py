"""axis_hw_exact_kernel: 16-wave form (one line group at a time) against the 8-wave form (two interleaved), one process, interleaved rounds:
    python tools/axis_hw_time.py      (cfg2 plane: 32 planes x 32 x 32 tokens x 256 channels; also 16 x 48 of cfg3)"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tante_amd
from tante_amd import _lib as L, kernels as K
dev = torch.device("cuda:0")
torch.manual_seed(0)
for BT, H, W in ((32, 32, 32), (32, 16, 48)):
    x0 = torch.randn(BT, H, W, 256, device=dev)
    wh = [torch.randn(H, H, device=dev) / 6, torch.randn(H, device=dev) * 0.1, torch.randn(H, H, device=dev) / 6, torch.randn(H, device=dev) * 0.1]
    ww = [torch.randn(W, W, device=dev) / 6, torch.randn(W, device=dev) * 0.1, torch.randn(W, W, device=dev) / 6, torch.randn(W, device=dev) * 0.1]
    ref = None
    ts = {512: [], 1024: []}
    for r in range(7):
        for nt in (1024, 512):
            L.set_option("TANTE_AXIS_NT", nt)
            x = x0.clone()
            K.axis_hw(x, BT, H, W, 256, wh, ww, L.BF16)
            if ref is None:
                ref = x.clone()
            elif r == 0:
                print(f"{H}x{W} nt={nt}: max |diff| vs 1024-thread form {float((x - ref).abs().max()):.3e}")
            for _ in range(3):
                K.axis_hw(x, BT, H, W, 256, wh, ww, L.BF16)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                K.axis_hw(x, BT, H, W, 256, wh, ww, L.BF16)
            e1.record(); torch.cuda.synchronize()
            ts[nt].append(e0.elapsed_time(e1) * 1e3 / 30)
    for nt in (1024, 512):
        print(f"{H}x{W} threads={nt}: median {statistics.median(ts[nt]):.2f} us  min {min(ts[nt]):.2f} us")
L.set_option("TANTE_AXIS_NT", 0)
