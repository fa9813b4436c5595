import sys, torch
sys.path.insert(0, "/root/repo")
import tante_amd
from tante_amd import autograd as A, train_forward as TF
dev = torch.device("cuda:0")
def run(D, order, axes, pix):
    A.PIXEL_WGRAD_IN_KERNEL = pix
    md = tante_amd.TanteMetadata(n_fields=D, spatial_resolution=(32, 128))
    torch.manual_seed(7)
    m = tante_amd.TANTE(in_T=4, dset_metadata=md, taylor_order=order, attn_axes=axes, n_head=8, embed_dim=256, patch_scale=8, dropout=0.0, frame_interval=0.5).to(dev).train().set_compute("bf16")
    gen = torch.Generator().manual_seed(77)
    batch = {"input": torch.randn(2, 4, 32, 128, D, generator=gen).to(dev), "output": torch.randn(2, 3, 32, 128, D, generator=gen).to(dev)}
    opt = tante_amd.FlatAdamW(m.parameters(), lr=1e-4); opt.zero_grad()
    fmt = tante_amd.DefaultChannelsFirstFormatter(md)
    y, yr = tante_amd.rollout_model(m, batch, fmt, 3)
    loss = A.MseMeanFn.apply(y, yr); A.run_backward(loss); torch.cuda.synchronize()
    g = {k: p.grad.detach().clone() for k, p in m.named_parameters() if "enc_conv_1" in k or "dec_conv_3" in k}
    return g
for D, order, axes in ((4, 1, "THW"), (4, 2, "TH-W"), (11, 1, "THW"), (11, 2, "TH-W"), (8, 1, "THW")):
    g1 = run(D, order, axes, True); g0 = run(D, order, axes, False)
    for k in g1:
        print(D, order, k, "max in-kernel %.4g  launches %.4g  rel %.3g" % (float(g1[k].abs().max()), float(g0[k].abs().max()), float((g1[k]-g0[k]).norm()/(g0[k].norm()+1e-30))))
