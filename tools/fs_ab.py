"""One-process A/B of the fused block kernel's launch forms (cdna_hip_programming.md 5.4 rule 24: interleaved rounds, one device):

    python tools/fs_ab.py [--B 8] [--opt TANTE_FS_GROUPS] [--values 1,2] [--rounds 7] [--iters 40]

Runs the T / H / W letters of cfg2 (B x 4 x 32 x 32 tokens, C = 256) on random data through tante_block_fused with each value of the
option, round-robin, timing each batch of launches with HIP events on the launch stream; checks that every form's output is bitwise
equal to the first one's.  Prints median and min per (letter, value) and the algorithmic TFLOP/s (819 200 / 788 992 FLOP per token)."""
import argparse
import json
import statistics
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tante_amd  # noqa: E402
from tante_amd import _lib as L, kernels as K  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--T", type=int, default=4)
    ap.add_argument("--H", type=int, default=32)
    ap.add_argument("--W", type=int, default=32)
    ap.add_argument("--opt", default="TANTE_FS_GROUPS")
    ap.add_argument("--values", default="1,2")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--letters", default="THW")
    ap.add_argument("--json", default="")
    ap.add_argument("--close", type=float, default=0.0, help="forms may differ by this much (different summation order) instead of bitwise")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    vals = [int(v) for v in a.values.split(",")]
    torch.manual_seed(0)
    blk = tante_amd.TransformerBlock(256, 8, mlp_ratio=1.0, dropout=0.0).to(dev).eval()
    with torch.no_grad():
        for ln in (blk.ln1, blk.ln2):
            ln.weight.add_(0.2 * torch.randn_like(ln.weight)); ln.bias.add_(0.2 * torch.randn_like(ln.bias))
    params = [blk.ln1.weight, blk.ln1.bias, blk.attn.in_proj_weight, blk.attn.in_proj_bias, blk.attn.out_proj.weight, blk.attn.out_proj.bias,
              blk.ln2.weight, blk.ln2.bias, blk.mlp[0].weight, blk.mlp[0].bias, blk.mlp[2].weight, blk.mlp[2].bias]
    stream = K.pack_block(params, 256, 256)
    n = a.B * a.T * a.H * a.W
    x0 = (torch.randn(n, 256, device=dev) * 1.5 + 0.3)
    out = {}
    for letter in a.letters:
        seq = K.make_seq(letter, a.B, a.T, a.H, a.W)
        causal = letter == "T"
        flop = n * (819200 if letter != "T" else 788992) if (a.T, a.H, a.W) == (4, 32, 32) else n * (2 * 256 * 1536 + 4 * seq.L * 256)
        ref = None
        for v in vals:
            L.set_option(a.opt, v)
            y = x0.clone()
            K.block_fused(y, stream, 256, 8, 256, seq, causal, 1e-5)
            torch.cuda.synchronize()
            if ref is None:
                ref = y
            else:
                dmax = float((ref - y).abs().max())
                assert torch.equal(ref, y) or dmax <= a.close, f"letter {letter}: {a.opt}={v} differs from {a.opt}={vals[0]} (max {dmax})"
        times = {v: [] for v in vals}
        y = x0.clone()
        for r in range(a.rounds):
            for v in vals:
                L.set_option(a.opt, v)
                y.copy_(x0)
                for _ in range(3):
                    K.block_fused(y, stream, 256, 8, 256, seq, causal, 1e-5)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    K.block_fused(y, stream, 256, 8, 256, seq, causal, 1e-5)
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) * 1e3 / a.iters)
        for v in vals:
            med, mn = statistics.median(times[v]), min(times[v])
            out[f"{letter}:{a.opt}={v}"] = {"median_us": round(med, 2), "min_us": round(mn, 2), "tflops_median": round(flop / med / 1e6, 1),
                                            "frac_of_2.5PF": round(flop / med / 1e6 / 2500, 3)}
            print(f"letter {letter} L={seq.L:3d} {a.opt}={v}: median {med:7.2f} us  min {mn:7.2f} us  -> {flop / med / 1e6:7.1f} TFLOP/s "
                  f"({flop / med / 1e6 / 2500:.3f} of 2.5 PF)", flush=True)
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
