"""One TransformerBlock's backward at cfg3's shapes (B = 8, T = 4, 16 x 48 tokens per frame, C = 256): the one-launch kernel
(tante_block_bwd_fused) against the three launches it replaces, per axis letter and dropout, HIP-event time per call over 50 calls.
   python tools/block_bwd_time.py [B]"""
import ctypes as Ct
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tante_amd
from tante_amd import kernels as K, _lib as L, train_forward as TF

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T, H, W = 4, 16, 48
torch.manual_seed(3)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for p in (0.0, 0.1):
    blk = tante_amd.TransformerBlock(256, 8, mlp_ratio=1.0, dropout=p).to(dev).train()
    a, m = blk.attn, blk.mlp
    with torch.no_grad(), TF.fold_scope():
        w_in, b_in = TF._folded(a.in_proj_weight, a.in_proj_bias, blk.ln1)
        w1, b1 = TF._folded(m[0].weight, m[0].bias, blk.ln2)
        fs = K.pack_block_train((w_in, b_in, a.out_proj.weight, a.out_proj.bias, w1, b1, m[2].weight, m[2].bias), 256, 256)
        bst = K.pack_block_tail_bwd(m[2].weight, w1, a.out_proj.weight, 256, 256)
        hst = K.pack_block_tail_bwd(w_in[0:256], w_in[256:512], w_in[512:768], 256, 256)
    for letter in "THW":
        causal = letter == "T"
        seq = K.make_seq(letter, B, T, H, W)
        n = B * T * H * W
        x = torch.randn(n, 256, device=dev) * 1.3 + 0.2
        dout = torch.randn(n, 256, device=dev)
        seeds = (11, 22, 33)
        t = K.block_fused_train(x, fs, 256, 8, 256, seq, causal, blk.ln1.eps, p, seeds, need_x1=False)
        s = torch.cuda.current_stream().cuda_stream
        dqkv = torch.empty_like(t["qkv"])

        def three():
            r = K.block_tail_bwd(dout, t["hpre"], t["xh2"], t["st2"], bst, 256, 256, p, seeds[1], seeds[2])
            L.check(L.lib().tante_attention_bwd(t["qkv"].data_ptr(), r["do"].data_ptr(), dqkv.data_ptr(), L.BF16, 256, 8, Ct.byref(seq), int(causal), p,
                                                seeds[0], s), "attention_bwd")
            return K.block_head_bwd(dqkv, t["xh1"], t["st1"], r["dx1"], hst, 256)

        def one():
            return K.block_bwd_fused(dout, t["xh1"], t["st1"], t["hpre"], t["xh2"], t["st2"], bst, fs, hst, 256, 8, 256, seq, causal, p, seeds)

        def fwd(q):
            return K.block_fused_train(x, fs, 256, 8, 256, seq, causal, blk.ln1.eps, p, seeds, need_x1=False, need_qkv=q)
        print(f"p={p} {letter} L={seq.L:2d} tokens={n}: one launch {timed(one):7.1f} us   three launches {timed(three):7.1f} us   "
              f"training forward with / without the packed projection {timed(lambda: fwd(True)):6.1f} / {timed(lambda: fwd(False)):6.1f} us", flush=True)
