"""Time the fused block kernel's TRAINING forward (block_fs_kernel<..., TRAIN = true>) at a cfg2-like shape (L = 32: <2,4,4,true>) and a
cfg3-like one (L = 16: <1,3,4,true>):  python tools/fs_train_time.py   (TANTE_LIB selects an A/B build)"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tante_amd
from tante_amd import kernels as K
dev = torch.device("cuda:0")
torch.manual_seed(0)
blk = tante_amd.TransformerBlock(256, 8, mlp_ratio=1.0, dropout=0.1).to(dev).train()
for name, (B, T, H, W, letter) in {"cfg2 H letter (L=32)": (8, 4, 32, 32, "H"), "cfg3 H letter (L=16)": (8, 4, 16, 48, "H"), "cfg2 T letter (L=4)": (8, 4, 32, 32, "T")}.items():
    n = B * T * H * W
    x = torch.randn(n, 256, device=dev)
    seq = K.make_seq(letter, B, T, H, W)
    a, m = blk.attn, blk.mlp
    params = [a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias, m[0].weight, m[0].bias, m[2].weight, m[2].bias]
    st = K.pack_block_train(params, 256, 256)
    ts = []
    for r in range(5):
        for _ in range(3):
            K.block_fused_train(x, st, 256, 8, 256, seq, letter == "T", 1e-5, 0.1, (1, 2, 3), need_x1=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            K.block_fused_train(x, st, 256, 8, 256, seq, letter == "T", 1e-5, 0.1, (1, 2, 3), need_x1=False)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 20)
    print(f"{name}: median {statistics.median(ts):.2f} us  min {min(ts):.2f} us")
