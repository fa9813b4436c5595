"""Launch time of the T-letter block kernel with and without the fused temporal propagator (cfg2: B x 4 x 32 x 32 tokens, C = 256), back to back;
TANTE_LIB selects a timing-experiment build (tools/build_variant.sh ... block_sliced.hip "-DFS_EXP_TPROP_NO_STORE").   python tools/tprop_time.py"""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tante_amd
from tante_amd import kernels as K
dev = torch.device("cuda:0")
torch.manual_seed(0)
blk = tante_amd.TransformerBlock(256, 8, mlp_ratio=1.0, dropout=0.0).to(dev).eval()
params = [blk.ln1.weight, blk.ln1.bias, blk.attn.in_proj_weight, blk.attn.in_proj_bias, blk.attn.out_proj.weight, blk.attn.out_proj.bias,
          blk.ln2.weight, blk.ln2.bias, blk.mlp[0].weight, blk.mlp[0].bias, blk.mlp[2].weight, blk.mlp[2].bias]
stream = K.pack_block(params, 256, 256)
B = 8
n = B * 4 * 32 * 32
x0 = torch.randn(n, 256, device=dev)
seq = K.make_seq("T", B, 4, 32, 32)
tp = (0.3 * torch.randn(40, device=dev)).contiguous()
res = {}
for name, kw in (("plain T", {}), ("T + propagator", {"tprop": tp})):
    ts = []
    y = x0.clone()
    for r in range(5):
        y.copy_(x0)
        for _ in range(3):
            K.block_fused(y, stream, 256, 8, 256, seq, True, 1e-5, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            K.block_fused(y, stream, 256, 8, 256, seq, True, 1e-5, **kw)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 30)
    res[name] = statistics.median(ts)
print(os.path.basename(os.environ.get("TANTE_LIB", "product")), {k: round(v, 2) for k, v in res.items()})
