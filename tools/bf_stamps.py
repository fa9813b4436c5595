#!/usr/bin/env python3
"""Where the one-launch block backward (block_bwd_fs.hip) spends its cycles: phase shares from in-kernel s_memtime stamps, at cfg3's
shapes, with the saved tensors rotating through NSETS buffer sets (> the 256 MiB Infinity Cache: the train step's cold-operand regime).

Needs the diagnostic library (python -m tante_amd.build --ablate -> tools/_ab/libtante_ablate.so).  python tools/bf_stamps.py [letter] [p]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tante_amd import _lib as L  # noqa: E402

if os.environ.get("TANTE_LIB"):            # a variant built by tools/build_variant.sh: un-stamped timing only
    L.LIB_PATH = os.environ["TANTE_LIB"]
elif os.environ.get("BF_PRODUCT") != "1":
    L.LIB_PATH = os.path.join(ROOT, "tools", "_ab", "libtante_ablate.so")
import tante_amd  # noqa: E402
from tante_amd import kernels as K, train_forward as TF  # noqa: E402

NAMES = ["setup + DMA issue + prime", "P0 dout in, dy2 out", "vmcnt(0) + barrier 1", "P1 gemm W2^T", "P1 gelu' + stores + barrier 2", "P2 gemm W1^T",
         "P2 LN2 bwd + barriers 3, 4", "P3 gemm Wo^T", "d_o image + barrier 5", "q gemm", "k gemm", "v gemm", "barrier 6 + k image", "attention head 0",
         "attention head 1", "dqkv stores + xh1 loads + barrier 7", "3 closing gemms", "LN1 stats + barrier 8", "LN1 bwd + dx store"]


def main():
    letter = sys.argv[1] if len(sys.argv) > 1 else "W"
    p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
    NSETS = 8
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    B, T, H, W = 8, 4, 16, 48
    blk = tante_amd.TransformerBlock(256, 8, mlp_ratio=1.0, dropout=p).to(dev).train()
    a, m = blk.attn, blk.mlp
    with torch.no_grad(), TF.fold_scope():
        w_in, b_in = TF._folded(a.in_proj_weight, a.in_proj_bias, blk.ln1)
        w1, b1 = TF._folded(m[0].weight, m[0].bias, blk.ln2)
        fs = K.pack_block_train((w_in, b_in, a.out_proj.weight, a.out_proj.bias, w1, b1, m[2].weight, m[2].bias), 256, 256)
        bst = K.pack_block_tail_bwd(m[2].weight, w1, a.out_proj.weight, 256, 256)
        hst = K.pack_block_tail_bwd(w_in[0:256], w_in[256:512], w_in[512:768], 256, 256)
    causal = letter == "T"
    seq = K.make_seq(letter, B, T, H, W)
    n = B * T * H * W
    seeds = (11, 22, 33)
    sets = []
    for i in range(NSETS):
        x = torch.randn(n, 256, device=dev) * 1.3 + 0.2
        t = K.block_fused_train(x, fs, 256, 8, 256, seq, causal, blk.ln1.eps, p, seeds, need_x1=False, need_qkv=False)
        sets.append((torch.randn(n, 256, device=dev), t))
    outs = [None] * NSETS

    def run(i):
        dout, t = sets[i % NSETS]
        outs[i % NSETS] = K.block_bwd_fused(dout, t["xh1"], t["st1"], t["hpre"], t["xh2"], t["st2"], bst, fs, hst, 256, 8, 256, seq, causal, p, seeds)
    for i in range(NSETS):
        run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(4 * NSETS):
        run(i)
    e1.record()
    torch.cuda.synchronize()
    print(f"{letter} L={seq.L} p={p}: un-stamped, operands rotating through {NSETS} sets: {e0.elapsed_time(e1) * 1e3 / (4 * NSETS):.1f} us per launch")
    e0.record()
    for i in range(4 * NSETS):
        run(0)
    e1.record()
    torch.cuda.synchronize()
    print(f"   the same set every launch (cache-warm): {e0.elapsed_time(e1) * 1e3 / (4 * NSETS):.1f} us per launch")
    lib = L.lib()
    try:
        lib.tante_bf_set_stamps
    except AttributeError:
        return
    nwg = (seq.nseq + (48 // seq.L if seq.L <= 48 else 1) - 1) // max(1, (48 // seq.L if seq.L <= 48 else 1))
    stamps = torch.zeros((nwg + 8) * 4 * 24, dtype=torch.int64, device=dev)
    lib.tante_bf_set_stamps.argtypes = [C.c_void_p]
    lib.tante_bf_set_stamps(stamps.data_ptr())
    run(3)
    torch.cuda.synchronize()
    lib.tante_bf_set_stamps(None)
    raw = stamps.cpu().numpy().reshape(-1, 24)
    raw = raw[raw[:, 0] != 0].astype(np.int64)
    s = raw[:, :20]
    d = np.diff(s, axis=1)
    tot = s[:, 19] - s[:, 0]
    clk = tot / np.maximum(1, raw[:, 23] - raw[:, 22]) * 100e6
    print(f"{s.shape[0]} waves; shader clock {np.median(clk) / 1e9:.3f} GHz; stamped span per wave: median {np.median(tot):.0f} cycles "
          f"({np.median(tot) / np.median(clk) * 1e6:.1f} us), max {tot.max()}; grid first entry -> last exit {s[:, 19].max() - s[:, 0].min()} cycles; "
          f"entry spread {s[:, 0].max() - s[:, 0].min()}")
    if (raw[:, 20] != 0).all():      # P0 in three parts (stamps 20, 21)
        for name, a, b in (("  P0a dout rows -> accumulators", raw[:, 20] - s[:, 1]), ("  P0b dropout mask + image A", raw[:, 21] - raw[:, 20]),
                           ("  P0c dy2 row stores + biases", s[:, 2] - raw[:, 21])) if False else ():
            pass
        for name, v in (("P0a dout rows -> accumulators", raw[:, 20] - s[:, 1]), ("P0b dropout mask + image A", raw[:, 21] - raw[:, 20]),
                        ("P0c dy2 row stores + biases", s[:, 2] - raw[:, 21])):
            print(f"    {name:<38s} median {np.median(v):8.0f}   p90 {np.percentile(v, 90):8.0f}")
    for k in range(19):
        print(f"  {NAMES[k]:<40s} median {np.median(d[:, k]):8.0f}   p90 {np.percentile(d[:, k], 90):8.0f}   share {np.median(d[:, k]) / np.median(tot) * 100:5.1f} %")


if __name__ == "__main__":
    main()
