"""Time tante_gemm at the training shapes of a TANTE block (M tokens x K -> N, bf16 operands)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tante_amd import kernels as K, _lib as L
M = int(os.environ.get("M", 16384))
for (N, Kd, odt) in ((256, 256, torch.bfloat16), (768, 256, torch.bfloat16), (512, 256, torch.bfloat16), (256, 512, torch.bfloat16),
                     (256, 256, torch.float32), (1024, 256, torch.bfloat16)):
    w = torch.randn(N, Kd, device="cuda") * 0.05
    b = torch.randn(N, device="cuda")
    pw = K.pack_weight(w, b, L.BF16)
    a = torch.randn(M, Kd, device="cuda").to(torch.bfloat16)
    o = torch.empty(M, N, dtype=odt, device="cuda")
    f = lambda: K.linear(a, pw, o, M=M)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    fl = 2.0 * M * N * Kd
    by = M * Kd * 2 + M * N * o.element_size()
    ref = (a.float() @ w.to(torch.bfloat16).float().t() + b)
    err = (o.float() - ref).abs().max().item() / ref.abs().max().item()
    print(f"M={M} N={N} K={Kd} out={str(odt)[6:]}: {us:.1f} us  {fl / us / 1e6:.0f} TFLOP/s  {by / us / 1e3:.0f} GB/s  err {err:.1e}")
