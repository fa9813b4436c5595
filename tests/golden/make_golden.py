#!/usr/bin/env python3
"""Generate golden input/output vectors by running the REFERENCE (zwu88/TANTE) on CPU.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

Nothing of the reference travels: the fixtures hold tensors only (weights as
name -> array, inputs, outputs, gradients).  The reference modules are imported
by path with two inert stubs (``torchinfo`` and ``h5py`` are not installed and
are not touched by any arithmetic) and a bare ``models`` package so that
``models/__init__.py`` (which eagerly imports third-party baselines) is skipped.

Fixture index:
  g1_tante_tiny      cfg1 TANTE end to end (order 2, "TL-TL")
  g2_encdec_*        enc_CNN / dec_CNN for several patch scales / overlap ratios
  g3_block_*         TransformerBlock causal / non-causal, several L and C
  g4_backbone_*      Attn_Backbone, one fixture per axis letter (Hp != Wp)
  g5_film            film 5-D / 3-D branch, t_series
  g6_interp          interprator incl. both clamps
  g7_taylor_*        Taylor sum order 1/2/3, frame_interval, output_length
  g8_rollout_*       Trainer / Evaler rollout_model window semantics
  g9_trainstep       loss, grads, clip, two AdamW steps
  g10_metrics        MSE / eval_rt / L2RE / NNMSE / VRMSE, LR schedule
  g11_cvit_*         CViT tiny: grid / fourier / mlp coordinate embeddings, full-grid and query-point modes
  g12_spectral_*     SpectralLayer (modes below / above the spectrum size) and TANTE(enc_dec_type='fno') tiny
  g13_deg_false      adaptive-dt forward composed from the reference's own sub-modules
  g15_*              gradients of 'same'-padded encoder / decoder stages (patch_scale 16 / 32 / 64) and of the channel-attention letter 'C'
  g17_*              gradients through TransformerBlock's attn_mask / key_padding_mask forms, and the letter 'C' over 256 channels
  g14_trainstep_wide production-shape train step (C=256, 8 heads x 32, "THWTHWTHW", L in {4, 8, 48}): loss, per-parameter gradient
                     norms, three full gradient tensors.  Weights (4.2 M) and inputs are NOT stored: both come from seeded CPU
                     generators (manual_seed(14) before the constructor; Generator(1414) for the fields) and the fixture holds their
                     checksums, so a test first proves it rebuilt the same tensors.
"""
import os
import sys
import types
import math
import zlib

import numpy as np
import torch

REF = os.environ.get("TANTE_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    if not os.path.isdir(REF):
        raise SystemExit(f"reference not found at {REF}; fixtures can only be generated in the build container")
    for name in ("torchinfo", "h5py", "wandb"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            if name == "torchinfo":
                m.summary = lambda *a, **k: None
            sys.modules[name] = m
    pkg = types.ModuleType("models")
    pkg.__path__ = [os.path.join(REF, "models")]
    sys.modules["models"] = pkg
    tr = types.ModuleType("trainer")
    tr.__path__ = [os.path.join(REF, "trainer")]
    sys.modules["trainer"] = tr
    sys.path.insert(0, REF)


_import_reference()
from models.tante import TANTE, film, interprator, t_series  # noqa: E402
from models.attn_backbone import Attn_Backbone, TransformerBlock  # noqa: E402
from models.enc_dec_cnn import enc_CNN, dec_CNN  # noqa: E402
from data.dataset import TanteMetadata  # noqa: E402
from data.datamodule import DefaultChannelsFirstFormatter  # noqa: E402
from trainer import metrics as ref_metrics  # noqa: E402
from trainer.trainer import Trainer  # noqa: E402
from trainer.evaler import Evaler  # noqa: E402
from optim.schedulers import LinearWarmupCosineAnnealingLR  # noqa: E402

torch.set_num_threads(8)


def md(n_fields, res):
    return TanteMetadata(
        dataset_name="synthetic", n_spatial_dims=2, spatial_resolution=tuple(res),
        field_names={0: [f"f{i}" for i in range(n_fields)]}, boundary_condition_types=["periodic"],
        n_files=1, n_trajectories_per_file=[1], n_steps_per_trajectory=[16], n_fields=n_fields)


def sd_np(module, prefix="w."):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def g1():
    torch.manual_seed(211)
    m = TANTE(in_T=4, dset_metadata=md(1, (64, 64)), taylor_order=2, attn_axes="TL-TL", n_head=4,
              embed_dim=64, patch_scale=8, dropout=0.0).eval()
    x = torch.randn(4, 4, 1, 64, 64)
    with torch.no_grad():
        y = m(x)
    save("g1_tante_tiny", x=x.numpy(), y=y.numpy(), **sd_np(m))


def g2():
    for ps, ov, res, nf, C in [(8, 0.0, (32, 48), 3, 32), (16, 0.0, (32, 64), 2, 32), (32, 0.0, (64, 64), 2, 16),
                               (8, 0.5, (32, 32), 2, 16), (4, 0.0, (16, 24), 5, 16), (64, 0.0, (64, 128), 1, 16)]:
        torch.manual_seed(ps * 7 + int(ov * 10))
        e = enc_CNN(md(nf, res), embed_dim=C, patch_scale=ps, overlap_ratio=ov).eval()
        d = dec_CNN(md(nf, res), embed_dim=C, patch_scale=ps, overlap_ratio=ov).eval()
        x = torch.randn(2, 3, nf, *res)
        with torch.no_grad():
            z = e(x)
            zz = torch.randn(2, 1, *z.shape[2:])
            r = d(zz)
        save(f"g2_encdec_ps{ps}_ov{int(ov * 10)}", x=x.numpy(), z=z.numpy(), zz=zz.numpy(), r=r.numpy(),
             meta=np.array([ps, int(ov * 100), res[0], res[1], nf, C]), **sd_np(e, "enc."), **sd_np(d, "dec."))


def g3():
    for tag, C, nh, L, Bp, causal, ratio in [("c64_L4_causal", 64, 4, 4, 6, True, 1.0),
                                             ("c64_L8", 64, 4, 8, 5, False, 1.0),
                                             ("c64_L32", 64, 8, 32, 3, False, 2.0),
                                             ("c64_L48", 64, 4, 48, 2, False, 1.0),
                                             ("c64_L100", 64, 2, 100, 2, False, 1.0),
                                             ("c64_L7_causal", 64, 4, 7, 3, True, 1.0),
                                             ("c256_L32", 256, 8, 32, 2, False, 1.0)]:
        torch.manual_seed(zlib.crc32(tag.encode()) % 1000)
        blk = TransformerBlock(C, nh, mlp_ratio=ratio, dropout=0.0).eval()
        # non-trivial LayerNorm affine so that gamma/beta handling is pinned
        with torch.no_grad():
            for ln in (blk.ln1, blk.ln2):
                ln.weight.add_(0.1 * torch.randn_like(ln.weight))
                ln.bias.add_(0.1 * torch.randn_like(ln.bias))
            blk.attn.in_proj_bias.add_(0.05 * torch.randn_like(blk.attn.in_proj_bias))
            blk.attn.out_proj.bias.add_(0.05 * torch.randn_like(blk.attn.out_proj.bias))
        x = torch.randn(Bp, L, C)
        with torch.no_grad():
            y = blk(x, causal=causal)
        save(f"g3_block_{tag}", x=x.numpy(), y=y.numpy(), meta=np.array([C, nh, L, int(causal), int(ratio * 100)]),
             **sd_np(blk))


def g4():
    T, H, W, C = 3, 4, 6, 32
    for axes in ["T", "H", "W", "L", "Y", "X", "A", "C", "THW", "LTCAXY"]:
        torch.manual_seed(17 + len(axes) + ord(axes[0]))
        bb = Attn_Backbone((T, H, W, C), axes, expanded_channel=16, n_head=4, mlp_ratio=1.0, dropout=0.0).eval()
        x = torch.randn(2, T, H, W, C)
        with torch.no_grad():
            y = bb(x)
        save(f"g4_backbone_{axes}", x=x.numpy(), y=y.numpy(), meta=np.array([T, H, W, C, 16, 4]), **sd_np(bb))


def g5():
    torch.manual_seed(5)
    f = film(32, in_dim=1).eval()
    x5 = torch.randn(2, 4, 3, 5, 32)
    ts = t_series(4, 1.0)
    ts2 = t_series(5, 0.5)
    x3 = torch.randn(3, 7, 32)
    rt = torch.tensor([1.2, 0.7, 3.4])
    with torch.no_grad():
        y5 = f(x5, ts)
        y3 = f(x3, rt)
    save("g5_film", x5=x5.numpy(), y5=y5.numpy(), x3=x3.numpy(), rt=rt.numpy(), y3=y3.numpy(),
         t_series_4_1=ts.numpy(), t_series_5_05=ts2.numpy(), **sd_np(f))


def g6():
    torch.manual_seed(6)
    it = interprator(32, 12).eval()
    with torch.no_grad():
        it.interprete[4].bias.fill_(0.3)
    x = torch.randn(3, 12, 32) * 4.0
    outs = {}
    with torch.no_grad():
        raw = it.interprete(x).reshape(-1, 12)
        for out_T in (1.5, 8, 1):
            outs[f"rt_{str(out_T).replace('.', 'p')}"] = it(x, out_T).numpy()
    save("g6_interp", x=x.numpy(), raw=raw.numpy(), **outs, **sd_np(it))


def g7():
    for tag, order, axes, fi, ol in [("o1", 1, "TH", 1.0, 1), ("o2", 2, "T-W", 0.5, 3), ("o3", 3, "T-H-W", 2.0, 2)]:
        torch.manual_seed(70 + order)
        m = TANTE(in_T=3, dset_metadata=md(2, (16, 32)), taylor_order=order, frame_interval=fi, output_length=ol,
                  attn_axes=axes, n_head=2, embed_dim=32, patch_scale=8, dropout=0.0).eval()
        x = torch.randn(2, 5, 2, 16, 32)  # T=5 > in_T exercises the window slice
        with torch.no_grad():
            y = m(x)
        save(f"g7_taylor_{tag}", x=x.numpy(), y=y.numpy(),
             meta=np.array([order, int(fi * 100), ol]), **sd_np(m))


class _FakeDS:
    def __init__(self, metadata):
        self.metadata = metadata


class _FakeDM:
    def __init__(self, metadata):
        self.train_dataset = _FakeDS(metadata)
        self.test_dataset = _FakeDS(metadata)
        self.val_dataset = _FakeDS(metadata)


def g8():
    meta = md(2, (16, 16))
    for tag, ol, n_roll in [("ol1_n4", 1, 4), ("ol3_n8", 3, 8), ("ol1_n8", 1, 8)]:
        torch.manual_seed(80 + ol + n_roll)
        m = TANTE(in_T=4, dset_metadata=meta, taylor_order=2, output_length=ol, attn_axes="T-L", n_head=2,
                  embed_dim=32, patch_scale=8, dropout=0.0).eval()
        batch = {"input": torch.randn(2, 4, 16, 16, 2), "output": torch.randn(2, n_roll, 16, 16, 2)}
        batch["input"][0, 0, 0, 0, 0] = float("nan")  # formatter nan_to_num
        tr = Trainer.__new__(Trainer)
        tr.n_steps_output = 4
        tr.n_steps_rollout = n_roll
        tr.device = torch.device("cpu")
        fmt = DefaultChannelsFirstFormatter(meta)
        with torch.no_grad():
            yp, yr = tr.rollout_model(m, batch, fmt, mode="eval")
            ypt, _ = tr.rollout_model(m, {"input": batch["input"], "output": batch["output"][:, :4]}, fmt, mode="train")
        save(f"g8_rollout_{tag}", inp=batch["input"].numpy(), out=batch["output"].numpy(), y_eval=yp.numpy(),
             y_ref=yr.numpy(), y_train=ypt.numpy(), meta=np.array([ol, n_roll]), **sd_np(m))


def g9():
    torch.manual_seed(9)
    meta = md(2, (16, 16))
    m = TANTE(in_T=4, dset_metadata=meta, taylor_order=2, attn_axes="TH-WL", n_head=2, embed_dim=32,
              patch_scale=8, dropout=0.0).train()
    opt = torch.optim.AdamW(m.parameters(), lr=5e-3, weight_decay=1e-2)
    batch = {"input": torch.randn(3, 4, 16, 16, 2), "output": torch.randn(3, 4, 16, 16, 2)}
    tr = Trainer.__new__(Trainer)
    tr.n_steps_output = 4
    tr.n_steps_rollout = 8
    tr.device = torch.device("cpu")
    fmt = DefaultChannelsFirstFormatter(meta)
    w0 = sd_np(m, "w0.")
    loss_fn = ref_metrics.MSE()
    arrs = {}
    for step in range(2):
        y_pred, y_ref = tr.rollout_model(m, batch, fmt, "train")
        loss = loss_fn(y_pred, y_ref, None).mean()
        loss.backward()
        if step == 0:
            arrs["y_pred"] = y_pred.detach().numpy()
            for k, p in m.named_parameters():
                arrs["g0." + k] = p.grad.detach().numpy().copy()
        gn = torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=1.0)
        arrs[f"loss{step}"] = loss.detach().numpy()
        arrs[f"gnorm{step}"] = gn.detach().numpy()
        opt.step()
        opt.zero_grad()
        arrs.update(sd_np(m, f"w{step + 1}."))
    save("g9_trainstep", inp=batch["input"].numpy(), out=batch["output"].numpy(),
         hyper=np.array([5e-3, 1e-2, 0.9, 0.999, 1e-8, 1.0]), **w0, **arrs)


def g10():
    torch.manual_seed(10)
    x = torch.randn(2, 3, 8, 6, 4)
    y = torch.randn(2, 3, 8, 6, 4)
    arrs = dict(x=x.numpy(), y=y.numpy())
    for name in ("MSE", "L2RE", "NNMSE", "VRMSE", "NMSE", "RMSE", "NRMSE", "VMSE"):
        arrs[name] = getattr(ref_metrics, name)()(x, y, None).numpy()
    rts = {"below": torch.tensor([1.1, 1.2, 1.3]), "inside": torch.tensor([2.0, 3.0]), "above": torch.tensor([4.5, 5.5])}
    for k, rt in rts.items():
        arrs["rt_" + k] = rt.numpy()
        arrs["mse_rt_" + k] = np.asarray(float(ref_metrics.MSE()(x, y, rt, 0.5, 2)))
        arrs["eval_rt_" + k] = np.asarray(float(ref_metrics.MSE.eval_rt(rt, 0.5, 2)))
    # LR schedule exactly as train.py builds it (warmup_start = eta_min = 0.1 * lr), stepped per epoch
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=5e-5)
    sch = LinearWarmupCosineAnnealingLR(opt, warmup_epochs=2, max_epochs=34, warmup_start_lr=5e-6, eta_min=5e-6)
    lrs = [opt.param_groups[0]["lr"]]
    for _ in range(40):
        opt.step()
        sch.step()
        lrs.append(opt.param_groups[0]["lr"])
    arrs["lr_schedule"] = np.array(lrs)
    save("g10_metrics", **arrs)


def g11():
    """CViT (models/cvit.py): pins the double layer_norm2 of CrossAttnBlock, the eps-Gaussian grid embedding and both query modes."""
    from models.cvit import CViT
    cases = [
        # name, embedding, ctor overrides, resolution, fields, query points (None = full grid)
        ("grid", "grid", dict(grid_size=(16, 24), eps=1e5), (16, 24), 2, None),          # sharp kernel, queries on the nodes
        ("gridwide", "grid", dict(grid_size=(6, 5), eps=40.0, dec_depth=2, num_mlp_layers=2), (16, 24), 2, 37),  # dense window
        ("fourier", "fourier", dict(), (16, 16), 1, None),
        ("mlp", "mlp", dict(mlp_ratio=2), (16, 24), 3, 50),
    ]
    for name, emb, kw, res, nf, nq in cases:
        torch.manual_seed(zlib.crc32(("g11" + name).encode()))
        base = dict(in_T=4, dset_metadata=md(nf, res), out_steps=3, patch_size=(1, 8, 8), grid_size=(8, 8), latent_dim=24, emb_dim=32,
                    depth=2, num_heads=4, dec_emb_dim=48, dec_num_heads=4, dec_depth=1, num_mlp_layers=1, mlp_ratio=1,
                    embedding_type=emb)
        base.update(kw)
        m = CViT(**base).eval()
        x = torch.randn(2, 4, nf, *res)
        coords = None if nq is None else torch.rand(nq, 2)
        with torch.no_grad():
            y = m(x) if coords is None else m(x, coords)
        extra = {} if coords is None else {"coords": coords.numpy()}
        save(f"g11_cvit_{name}", x=x.numpy(), y=y.numpy(), **extra, **sd_np(m))


def g12():
    """SpectralLayer (models/enc_dec_fno.py:184-222) and the TANTE(enc_dec_type='fno') path."""
    from models.enc_dec_fno import SpectralLayer
    for name, cin, cout, modes, shape in [("low", 3, 5, (4, 3), (2, 3, 16, 12)), ("clip", 2, 4, (40, 40), (3, 2, 8, 8)),
                                          ("odd", 4, 2, (3, 5), (1, 4, 10, 14))]:
        torch.manual_seed(zlib.crc32(("g12" + name).encode()))
        m = SpectralLayer(cin, cout, *modes).eval()
        x = torch.randn(*shape)
        with torch.no_grad():
            y = m(x)
        w = {"w.weight_re": m.weight.detach().real.numpy().copy(), "w.weight_im": m.weight.detach().imag.numpy().copy(),
             "w.w0.weight": m.w0.weight.detach().numpy().copy(), "w.w0.bias": m.w0.bias.detach().numpy().copy()}
        save(f"g12_spectral_{name}", x=x.numpy(), y=y.numpy(), modes=np.array(modes), **w)
    torch.manual_seed(zlib.crc32(b"g12fno"))
    m = TANTE(in_T=4, dset_metadata=md(2, (32, 32)), taylor_order=2, attn_axes="TL-TL", n_head=4, embed_dim=64, patch_scale=8,
              enc_dec_type="fno", modes1=8, modes2=8, dropout=0.0).eval()
    x = torch.randn(2, 4, 2, 32, 32)
    with torch.no_grad():
        y = m(x)
    sd = {}
    for k, v in m.state_dict().items():   # complex weights travel as (re, im) pairs
        if v.is_complex():
            sd["w." + k + "_re"] = v.real.numpy().copy()
            sd["w." + k + "_im"] = v.imag.numpy().copy()
        else:
            sd["w." + k] = v.numpy().copy()
    save("g12_tante_fno", x=x.numpy(), y=y.numpy(), **sd)


def g13():
    """deg=False: TANTE.forward raises in the reference (tante.py:149-152 applies a 3-D einops
    pattern to a 5-D tensor), so the adaptive-dt semantics are pinned by composing the reference's
    own sub-modules in the evidently intended order (comment '# (B, L, C)' at tante.py:151)."""
    from einops import rearrange
    torch.manual_seed(13)
    m = TANTE(in_T=4, dset_metadata=md(1, (32, 32)), taylor_order=2, attn_axes="TH-TW", n_head=2, embed_dim=32,
              patch_scale=8, dropout=0.0, deg=False).eval()
    with torch.no_grad():  # push the interpreter outputs around so that floor(R_t[0]) > 1 for out_T = 6
        for it in m.interprators:
            it.interprete[4].bias.fill_(2.2)
    x = torch.randn(2, 4, 1, 32, 32)
    res = {}
    for out_T in (1.5, 6):
        with torch.no_grad():
            B = x.shape[0]
            h = m.encoder(x)
            _, _, Hp, Wp, C = h.shape
            h = m.t_encode(h, m.t_seq)
            h = h + m.s_emb
            h = rearrange(h, "b t h w c -> (b h w) t c") + m.t_emb
            h = rearrange(h, "(b h w) t c -> b t h w c", b=B, h=Hp, w=Wp)
            ders, rts = [], []
            for i in range(m.taylor_order):
                h = m.blocks[i](h)
                d3 = rearrange(h[:, -1:], "b 1 h w c -> b (h w) c")
                rt = m.interprators[i](d3, out_T)
                rts.append(rt)
                d3 = m.modifiers[i](d3, rt)
                d5 = rearrange(d3, "b (h w) c -> b 1 h w c", h=Hp, w=Wp)
                ders.append(m.decoders[i](d5))
            R_t = torch.stack(rts, 1).mean(1)
            n_out = math.floor(R_t[0])
            outs = []
            for i in range(1, n_out + 1):
                o = 0
                for k in range(1, m.taylor_order + 1):
                    o = o + ders[k - 1] * (i * m.frame_interval) ** k / math.factorial(k)
                outs.append(o + x[:, -1:])
            y = torch.cat(outs, 1)
        tag = str(out_T).replace(".", "p")
        res[f"y_{tag}"] = y.numpy()
        res[f"rt_{tag}"] = R_t.numpy()
    save("g13_deg_false", x=x.numpy(), **res, **sd_np(m))


G14_FULL = ["blocks.0.blocks.4.attn.in_proj_weight", "blocks.0.blocks.8.mlp.0.weight", "encoder.enc_conv_2.conv.weight"]


def g14():
    torch.manual_seed(14)
    meta = md(4, (64, 384))
    m = TANTE(in_T=4, dset_metadata=meta, taylor_order=1, attn_axes="THWTHWTHW", n_head=8, embed_dim=256,
              patch_scale=8, dropout=0.0).train()
    gen = torch.Generator().manual_seed(1414)
    batch = {"input": torch.randn(2, 4, 64, 384, 4, generator=gen), "output": torch.randn(2, 4, 64, 384, 4, generator=gen)}
    tr = Trainer.__new__(Trainer)
    tr.n_steps_output = 4
    tr.n_steps_rollout = 8
    tr.device = torch.device("cpu")
    fmt = DefaultChannelsFirstFormatter(meta)
    y_pred, y_ref = tr.rollout_model(m, batch, fmt, "train")
    loss = ref_metrics.MSE()(y_pred, y_ref, None).mean()
    loss.backward()
    names = [k for k, _ in m.named_parameters()]
    arrs = {"loss": loss.detach().numpy(), "param_names": np.array(names),
            "w_norm": np.array([float(p.detach().double().norm()) for _, p in m.named_parameters()]),
            "g_norm": np.array([float(p.grad.double().norm()) for _, p in m.named_parameters()]),
            "in_sum": np.array([float(batch["input"].double().sum()), float(batch["input"].double().pow(2).sum()),
                                float(batch["output"].double().sum()), float(batch["output"].double().pow(2).sum())]),
            "y_pred_slice": y_pred.detach()[:, :, ::8, ::8, :].numpy().copy(),
            "y_pred_norm": np.array(float(y_pred.detach().double().norm()))}
    sdict = dict(m.named_parameters())
    for k in G14_FULL:
        arrs["g." + k] = sdict[k].grad.detach().numpy().copy()
    arrs["gnorm"] = np.array(float(torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=1.0)))
    save("g14_trainstep_wide", **arrs)


def g15():
    """Round 5: GRADIENTS of the training-surface cases the HIP path did not cover before -- 'same'-padded conv / deconv stages
    (patch_scale 16 / 32 / 64: kernel-4 stages; 32 is the constructor default) and the channel-attention letter 'C' -- from the
    reference's own backward(): loss = sum(output * w) with a seeded w, gradients of the input and of every parameter."""
    for ps, res, nf, C in [(16, (32, 64), 2, 32), (32, (64, 64), 2, 16), (64, (64, 128), 1, 16)]:
        torch.manual_seed(1500 + ps)
        e = enc_CNN(md(nf, res), embed_dim=C, patch_scale=ps, overlap_ratio=0.0).train()
        d = dec_CNN(md(nf, res), embed_dim=C, patch_scale=ps, overlap_ratio=0.0).train()
        x = torch.randn(2, 3, nf, *res, requires_grad=True)
        z = e(x)
        wz = torch.randn_like(z)
        (z * wz).sum().backward()
        zz = torch.randn(2, 1, *z.shape[2:], requires_grad=True)
        r = d(zz)
        wr = torch.randn_like(r)
        (r * wr).sum().backward()
        arrs = {"x": x.detach().numpy(), "z": z.detach().numpy(), "wz": wz.numpy(), "dx": x.grad.numpy(), "zz": zz.detach().numpy(),
                "r": r.detach().numpy(), "wr": wr.numpy(), "dzz": zz.grad.numpy(), "meta": np.array([ps, 0, res[0], res[1], nf, C])}
        for k, q in e.named_parameters():
            arrs["genc." + k] = q.grad.numpy().copy()
        for k, q in d.named_parameters():
            arrs["gdec." + k] = q.grad.numpy().copy()
        save(f"g15_encdec_grad_ps{ps}", **arrs, **sd_np(e, "enc."), **sd_np(d, "dec."))
    T, H, W, C = 3, 4, 6, 32
    for axes in ["C", "LTCAXY"]:
        torch.manual_seed(1515 + len(axes))
        bb = Attn_Backbone((T, H, W, C), axes, expanded_channel=16, n_head=4, mlp_ratio=1.0, dropout=0.0).train()
        x = torch.randn(2, T, H, W, C, requires_grad=True)
        y = bb(x)
        w = torch.randn_like(y)
        (y * w).sum().backward()
        arrs = {"x": x.detach().numpy(), "y": y.detach().numpy(), "w": w.numpy(), "dx": x.grad.numpy(), "meta": np.array([T, H, W, C, 16, 4])}
        for k, q in bb.named_parameters():
            arrs["g." + k] = q.grad.numpy().copy()
        save(f"g15_backbone_grad_{axes}", **arrs, **sd_np(bb))


def g16():
    """Round 5: GRADIENTS through OVERLAPPING stages (overlap_ratio > 0; 0.5 is the constructor default, enc_dec_cnn.py:39-46): the conv
    has stride round(P (1 - overlap)) < P and its output is adaptive-average-pooled back to H / P x W / P (enc_dec_cnn.py:97-110); the
    transposed conv's overlapping taps are summed and the result resized (bilinear) to H P x W P (enc_dec_cnn.py:164-184).  Same recipe as
    g15: loss = sum(output * w), gradients of the input and of every parameter from the reference's backward().  0.25 gives stride 3 under
    kernel 4: uneven pooling windows."""
    for ps, ov, res, nf, C in [(8, 0.5, (16, 32), 2, 32), (32, 0.5, (64, 64), 2, 16), (16, 0.25, (32, 48), 1, 16)]:
        torch.manual_seed(1600 + ps)
        e = enc_CNN(md(nf, res), embed_dim=C, patch_scale=ps, overlap_ratio=ov).train()
        d = dec_CNN(md(nf, res), embed_dim=C, patch_scale=ps, overlap_ratio=ov).train()
        x = torch.randn(2, 3, nf, *res, requires_grad=True)
        z = e(x)
        wz = torch.randn_like(z)
        (z * wz).sum().backward()
        zz = torch.randn(2, 1, *z.shape[2:], requires_grad=True)
        r = d(zz)
        wr = torch.randn_like(r)
        (r * wr).sum().backward()
        arrs = {"x": x.detach().numpy(), "z": z.detach().numpy(), "wz": wz.numpy(), "dx": x.grad.numpy(), "zz": zz.detach().numpy(),
                "r": r.detach().numpy(), "wr": wr.numpy(), "dzz": zz.grad.numpy(), "meta": np.array([ps, int(round(ov * 100)), res[0], res[1], nf, C])}
        for k, q in e.named_parameters():
            arrs["genc." + k] = q.grad.numpy().copy()
        for k, q in d.named_parameters():
            arrs["gdec." + k] = q.grad.numpy().copy()
        save(f"g16_encdec_grad_ps{ps}_ov{int(round(ov * 100))}", **arrs, **sd_np(e, "enc."), **sd_np(d, "dec."))


def g17():
    """Round 6: GRADIENTS through TransformerBlock.forward(x, key_padding_mask, attn_mask, causal) (attn_backbone.py:59-83) with masks --
    a bool (L, L) attn_mask, a float additive one, a per-(batch, head) one, a bool key_padding_mask, and bool mask | causal -- and through
    the channel letter 'C' over 256 channels (sequences of 256: past the length the block backward kernels take).  Recipe of g15."""
    L_, C = 12, 64
    cases = {}
    gen = torch.Generator().manual_seed(1717)
    am_bool = torch.rand(L_, L_, generator=gen) < 0.3
    am_bool[torch.arange(L_), torch.arange(L_)] = False                      # every query keeps its own key: no all-blocked row
    am_float = torch.randn(L_, L_, generator=gen)
    am_bh = torch.randn(3 * 4, L_, L_, generator=gen)
    am_bh[torch.rand(3 * 4, L_, L_, generator=gen) < 0.2] = float("-inf")
    am_bh[:, torch.arange(L_), torch.arange(L_)] = 0.0
    kp = torch.zeros(3, L_, dtype=torch.bool)
    kp[0, -3:] = True
    kp[2, 1] = True
    cases["ambool"] = dict(attn_mask=am_bool)
    cases["amfloat"] = dict(attn_mask=am_float)
    cases["ambh_kp"] = dict(attn_mask=am_bh, key_padding_mask=kp)
    cases["kp"] = dict(key_padding_mask=kp)
    cases["ambool_causal"] = dict(attn_mask=am_bool, causal=True)
    for name, kw in cases.items():
        torch.manual_seed(1700 + len(name))
        blk = TransformerBlock(C, 4, mlp_ratio=2.0, dropout=0.0).train()
        x = torch.randn(3, L_, C, requires_grad=True)
        y = blk(x, **kw)
        w = torch.randn_like(y)
        (y * w).sum().backward()
        arrs = {"x": x.detach().numpy(), "y": y.detach().numpy(), "w": w.numpy(), "dx": x.grad.numpy(), "meta": np.array([L_, C, 4, int(kw.get("causal", False))])}
        for k, v in kw.items():
            if k != "causal":
                arrs[k] = v.numpy()
        for k, q in blk.named_parameters():
            arrs["g." + k] = q.grad.numpy().copy()
        save(f"g17_block_masked_grad_{name}", **arrs, **sd_np(blk))
    T, H, W, C = 2, 2, 3, 256
    torch.manual_seed(1790)
    bb = Attn_Backbone((T, H, W, C), "C", expanded_channel=16, n_head=4, mlp_ratio=1.0, dropout=0.0).train()
    x = torch.randn(1, T, H, W, C, requires_grad=True)
    y = bb(x)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    arrs = {"x": x.detach().numpy(), "y": y.detach().numpy(), "w": w.numpy(), "dx": x.grad.numpy(), "meta": np.array([T, H, W, C, 16, 4])}
    for k, q in bb.named_parameters():
        arrs["g." + k] = q.grad.numpy().copy()
    save("g17_backbone_grad_C256", **arrs, **sd_np(bb))


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14", "g15", "g16", "g17"]
    for w in which:
        globals()[w]()
