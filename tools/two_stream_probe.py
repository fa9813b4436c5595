#!/usr/bin/env python3
"""Does a block launch cut in two HALF-BATCH launches on two streams beat the one launch?  (cfg3 shapes, operands rotating through buffer
sets larger than the Infinity Cache.)  One 512-workgroup launch runs every workgroup through the same phase at the same time -- memory
phases with idle SIMDs, compute phases with an idle memory system; two 256-workgroup launches that start a few microseconds apart put a
workgroup of each on every CU, out of phase.     python tools/two_stream_probe.py [letter] [p]"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tante_amd
from tante_amd import kernels as K, train_forward as TF

dev = torch.device("cuda:0")
letter = sys.argv[1] if len(sys.argv) > 1 else "H"
p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
T, H, W = 4, 16, 48
NSETS = 8
torch.manual_seed(0)
blk = tante_amd.TransformerBlock(256, 8, mlp_ratio=1.0, dropout=p).to(dev).train()
a, m = blk.attn, blk.mlp
with torch.no_grad(), TF.fold_scope():
    w_in, b_in = TF._folded(a.in_proj_weight, a.in_proj_bias, blk.ln1)
    w1, b1 = TF._folded(m[0].weight, m[0].bias, blk.ln2)
    fs = K.pack_block_train((w_in, b_in, a.out_proj.weight, a.out_proj.bias, w1, b1, m[2].weight, m[2].bias), 256, 256)
    bst = K.pack_block_tail_bwd(m[2].weight, w1, a.out_proj.weight, 256, 256)
    hst = K.pack_block_tail_bwd(w_in[0:256], w_in[256:512], w_in[512:768], 256, 256)
causal = letter == "T"
seeds = (11, 22, 33)


def make(B):
    seq = K.make_seq(letter, B, T, H, W)
    n = B * T * H * W
    sets = []
    for _ in range(NSETS):
        x = torch.randn(n, 256, device=dev) * 1.3 + 0.2
        t = K.block_fused_train(x, fs, 256, 8, 256, seq, causal, blk.ln1.eps, p, seeds, need_x1=False, need_qkv=False)
        sets.append((x, torch.randn(n, 256, device=dev), t))
    return seq, sets


def bwd(seq, s):
    x, dout, t = s
    return K.block_bwd_fused(dout, t["xh1"], t["st1"], t["hpre"], t["xh2"], t["st2"], bst, fs, hst, 256, 8, 256, seq, causal, p, seeds)


def fwd(seq, s):
    return K.block_fused_train(s[0], fs, 256, 8, 256, seq, causal, blk.ln1.eps, p, seeds, need_x1=False, need_qkv=False)


seq8, sets8 = make(8)
seq4, sets4a = make(4)
_, sets4b = make(4)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, n=4 * NSETS):
    for i in range(NSETS):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for name, op in (("backward", bwd), ("training forward", fwd)):
    one = timed(lambda i: op(seq8, sets8[i % NSETS]))

    def pair(i):
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            op(seq4, sets4a[i % NSETS])
        with torch.cuda.stream(s2):
            op(seq4, sets4b[i % NSETS])
        cur.wait_stream(s1); cur.wait_stream(s2)
    two = timed(pair)
    half = timed(lambda i: op(seq4, sets4a[i % NSETS]))
    # captured: the fork / join becomes graph edges (what a captured train step would pay)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(NSETS):
            pair(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(4):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    two_g = 1e3 * e0.elapsed_time(e1) / (4 * NSETS)
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        for i in range(NSETS):
            op(seq8, sets8[i])
    torch.cuda.synchronize()
    g1.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(4):
        g1.replay()
    e1.record()
    torch.cuda.synchronize()
    one_g = 1e3 * e0.elapsed_time(e1) / (4 * NSETS)
    print(f"{letter} L={seq8.L} p={p} {name}: one B=8 launch {one:6.1f} us (captured {one_g:6.1f});  two B=4 launches on two streams {two:6.1f} us "
          f"(captured {two_g:6.1f});  one B=4 launch alone {half:6.1f} us", flush=True)
