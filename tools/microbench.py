#!/usr/bin/env python3
"""Run single kernels of the path at cfg2 size (for rocprofv3 --pmc / --kernel-trace):
   python tools/microbench.py block|axis|enc|dec [--iters N] [--batch B] [--letter H]"""
import argparse
import sys
import os
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tante_amd
from tante_amd import kernels as K, _lib as L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["block", "axis", "enc", "dec", "head", "model", "wgrad", "attnbwd"])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--letter", default="H")
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    B, T, H, W, C = a.batch, 4, 32, 32, 256
    md = tante_amd.TanteMetadata(n_fields=11, spatial_resolution=(256, 256))
    comp = K.COMPUTE[a.dtype]

    def timeit(fn, name, flops=None, bytes_=None):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / a.iters
        msg = f"{name}: {us:.1f} us"
        if flops:
            msg += f"  {flops / us / 1e6:.1f} TFLOP/s"
        if bytes_:
            msg += f"  {bytes_ / us / 1e6:.2f} TB/s"
        print(msg)

    with torch.no_grad():
        if a.what == "block":
            blk = tante_amd.TransformerBlock(C, 8, mlp_ratio=1.0, dropout=0.0).to(dev).eval()
            x = torch.randn(B * T * H * W, C, device=dev)
            seq = K.make_seq(a.letter, B, T, H, W)
            n_tok = x.shape[0]
            fl = n_tok * (2.0 * C * 3 * C + 4.0 * seq.L * C + 2.0 * C * C + 4.0 * C * C)
            timeit(lambda: blk.forward_tokens(x, seq, a.letter == "T", comp), f"block[{a.letter}] {a.dtype}", fl, 2.0 * x.numel() * 4)
        elif a.what == "axis":
            bb = tante_amd.Attn_Backbone((T, H, W, C), "T", n_head=8).to(dev).eval()
            x = torch.randn(B, T, H, W, C, device=dev)
            vp, hp, tp = bb.vertical_propagator, bb.horizontal_propagator, bb.temporal_propagator
            by = 2.0 * x.numel() * 4
            timeit(lambda: K.axis_hw(x, B * T, H, W, C, (vp[0].weight, vp[0].bias, vp[2].weight, vp[2].bias),
                                     (hp[0].weight, hp[0].bias, hp[2].weight, hp[2].bias), comp), "axis H+W fused", None, by)
            timeit(lambda: K.axis_mlp(x, B * T, H, W * C, vp[0].weight, vp[0].bias, vp[2].weight, vp[2].bias), "axis H", None, by)
            timeit(lambda: K.axis_mlp(x, B * T * H, W, C, hp[0].weight, hp[0].bias, hp[2].weight, hp[2].bias), "axis W", None, by)
            timeit(lambda: K.axis_mlp(x, B, T, H * W * C, tp[0].weight, tp[0].bias, tp[2].weight, tp[2].bias), "axis T", None, by)
        elif a.what == "wgrad":
            from tante_amd.autograd import wgrad, _rm_linear
            R = 24576
            for (I, J) in ((256, 256), (768, 256), (512, 256), (256, 512)):
                U = torch.randn(R, I, device=dev).to(torch.bfloat16)
                V = torch.randn(R, J, device=dev).to(torch.bfloat16)
                timeit(lambda: wgrad(_rm_linear(U), _rm_linear(V), R, I, J, (I, J), L.BF16, device=dev, with_bias=True),
                       f"wgrad bf16 R={R} I={I} J={J}", 2.0 * R * I * J, 2.0 * R * (I + J))
        elif a.what == "attnbwd":
            from tante_amd.autograd import AttentionFn
            for letter, (T_, H_, W_) in (("W", (4, 16, 48)), ("H", (4, 16, 48)), ("T", (4, 16, 48))):
                n = B * T_ * H_ * W_
                qkv = torch.randn(n, 3 * C, device=dev).to(torch.bfloat16)
                do = torch.randn(n, C, device=dev).to(torch.bfloat16)
                dq = torch.empty_like(qkv)
                seq = K.make_seq(letter, B, T_, H_, W_)
                import ctypes as Ct
                timeit(lambda: L.check(L.lib().tante_attention_bwd(qkv.data_ptr(), do.data_ptr(), dq.data_ptr(), L.BF16, C, 8, Ct.byref(seq),
                                                                   int(letter == "T"), float(os.environ.get("PDROP", "0")), 1234, torch.cuda.current_stream().cuda_stream)),
                       f"attn bwd {letter} L={seq.L}")
        elif a.what in ("enc", "dec", "head", "model"):
            m = tante_amd.TANTE(in_T=4, dset_metadata=md, taylor_order=3, attn_axes="THW-THW-THW", n_head=8, embed_dim=256,
                                patch_scale=8).to(dev).eval().set_compute(a.dtype)
            inp = torch.randn(B, 4, 11, 256, 256, device=dev)
            if a.what == "enc":
                fa, fb = m._time_tables()
                film = (fa, fb, m.s_emb.view(1024, 256), 4, 1024)
                timeit(lambda: m.encoder.forward_tokens(inp, comp, film), "encoder (3 stages)", B * 2.517e9)
            elif a.what == "dec":
                x = torch.randn(B * 4 * 1024, 256, device=dev)
                timeit(lambda: m.decoders[0].forward_tokens(x, B, comp, a_n0=1024, a_s1=4 * 1024 * 256, a_s0=256, a_off=3 * 1024 * 256),
                       "head (3 stages)", B * 0.629e9)
            elif a.what == "head":
                x = torch.randn(B * 4 * 1024, 256, device=dev)
                out = torch.zeros(B, 1, 11, 256, 256, device=dev)
                ph = m.decoders[0].packed_head()
                timeit(lambda: K.head_fused(x, 1024, 4 * 1024 * 256, 256, 3 * 1024 * 256, B, 32, 32, 256, 11, ph, out, out[0].numel(), [1.0],
                                            None), "fused head", B * 0.629e9)
            else:
                timeit(lambda: m(inp), "model forward", B * 35.1e9)


if __name__ == "__main__":
    main()
