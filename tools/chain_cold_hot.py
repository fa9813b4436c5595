#!/usr/bin/env python3
"""chain512_kernel<0,1> at M = 256 (CViT's encoder at B = 1): the same weights every launch (hot in the XCDs' L2s) against a cycle of 24
different weight sets (36 MB: each launch's weights come from the Infinity Cache, as in the model's forward).  python tools/chain_cold_hot.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tante_amd import kernels as K
dev = torch.device("cuda:0")
torch.manual_seed(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
a = torch.randn(M, 512, device=dev).to(torch.bfloat16)
resid = torch.randn(M, 512, device=dev)
out = torch.empty(M, 512, device=dev)
NW = 24
ws = [(torch.randn(3 * 512 * 512, device=dev) * 0.02).to(torch.bfloat16) for _ in range(NW)]
bs = [torch.zeros(3 * 512, device=dev) for _ in range(NW)]


def run(idx):
    K.cvit_chain512(a, resid, ws[idx], bs[idx], 1e-5, M, out)


def timeit(pick, n=240):
    for i in range(24): run(pick(i))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): run(pick(i))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


print(f"M={M}: same weights every launch {timeit(lambda i: 0):.1f} us; cycling {NW} weight sets {timeit(lambda i: i % NW):.1f} us (back to back, incl. launch gaps)")
