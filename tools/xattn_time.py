"""Time tante_cross_attention (bf16 MFMA kernel) at the CViT cfg4 decoder shape: N = 65536 queries, S = 256 keys, 8 heads x 64."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tante_amd import kernels as K
for B in (1, 4):
    N, S, H, D = 65536, 256, 8, 64
    C = H * D
    q = torch.randn(B * N, C, device="cuda").to(torch.bfloat16)
    kv = torch.randn(B * S, 2 * C, device="cuda").to(torch.bfloat16)
    o = torch.empty(B * N, C, dtype=torch.bfloat16, device="cuda")
    f = lambda: K.cross_attention(q, kv, kv[:, C:], o, B, H, D, N, S, C, 2 * C, C)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    fl = 4.0 * B * N * S * C
    print(f"B={B}: {us:.1f} us  {fl / us / 1e6:.0f} TFLOP/s  ({fl / us / 1e6 / 2500:.1%} of 2.5 PF)")
