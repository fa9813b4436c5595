import sys, torch
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import tante_amd
from conftest import load_golden, split_prefix, rel_err
dev = torch.device("cuda:0")
for name in ("g15_backbone_grad_LTCAXY", "g15_backbone_grad_C"):
    g = load_golden(name)
    axes = name.split("_")[-1]
    T, H, W, C, E, nh = (int(v) for v in g["meta"])
    for fp32sum in (True, False):
        tante_amd.autograd.FP32_BIAS_SUMS = fp32sum
        bb = tante_amd.Attn_Backbone((T, H, W, C), axes, expanded_channel=E, n_head=nh, mlp_ratio=1.0, dropout=0.0).to(dev).train()
        bb.load_state_dict(split_prefix(g, "w."))
        bb.compute = "bf16"
        x = g["x"].to(dev).requires_grad_(True)
        y = bb(x)
        (y.float() * g["w"].to(dev)).sum().backward()
        errs = sorted(((rel_err(q.grad.cpu(), g["g." + k]), k) for k, q in bb.named_parameters() if g["g." + k].dim() == 1 and float(g["g." + k].abs().max()) > 0), reverse=True)
        print(name, "fp32 bias sums", fp32sum, [(round(e, 4), k) for e, k in errs[:6]])
