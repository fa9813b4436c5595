"""CPU: the oracle (oracle/tante_oracle.py) against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  This is what PINS the oracle."""
import glob
import math
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden, split_prefix, rel_err, max_rel
from oracle import tante_oracle as O

TOL = 2e-6   # fp32 re-association noise between two CPU evaluation orders (hard part 3)


def names(pattern):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, pattern + ".npz")))


def test_g1_tiny_end_to_end():
    g = load_golden("g1_tante_tiny")
    cfg = O.TanteCfg(4, 1, (64, 64), taylor_order=2, attn_axes="TL-TL", n_head=4, embed_dim=64, patch_scale=8)
    y = O.tante_forward(split_prefix(g, "w."), cfg, g["x"])
    assert y.shape == g["y"].shape
    assert max_rel(y, g["y"]) < TOL


@pytest.mark.parametrize("name", names("g2_encdec_*"))
def test_g2_encdec(name):
    g = load_golden(name)
    ps, ov = int(g["meta"][0]), float(g["meta"][1]) / 100
    z = O.enc_cnn(split_prefix(g, "enc."), g["x"], ps, ov)
    assert z.shape == g["z"].shape and max_rel(z, g["z"]) < TOL
    r = O.dec_cnn(split_prefix(g, "dec."), g["zz"], ps, ov)
    assert r.shape == g["r"].shape and max_rel(r, g["r"]) < TOL


@pytest.mark.parametrize("name", names("g3_block_*"))
def test_g3_block(name):
    g = load_golden(name)
    C, nh, L, causal, _ = (int(v) for v in g["meta"])
    y = O.transformer_block(split_prefix(g, "w."), g["x"], nh, bool(causal))
    assert max_rel(y, g["y"]) < TOL


@pytest.mark.parametrize("name", names("g4_backbone_*"))
def test_g4_backbone(name):
    g = load_golden(name)
    axes = name.split("_")[-1]
    y = O.attn_backbone(split_prefix(g, "w."), g["x"], axes, int(g["meta"][5]))
    assert max_rel(y, g["y"]) < TOL


def test_g5_film_and_t_series():
    g = load_golden("g5_film")
    w = split_prefix(g, "w.")
    assert torch.equal(O.t_series(4, 1.0), g["t_series_4_1"])
    assert torch.equal(O.t_series(5, 0.5), g["t_series_5_05"])
    assert O.t_series(4, 1.0).tolist() == [-2.0, -1.0, 0.0, 0.0]   # the duplicated-zero quirk
    assert max_rel(O.film(w, g["x5"], g["t_series_4_1"]), g["y5"]) < TOL
    assert max_rel(O.film(w, g["x3"], g["rt"]), g["y3"]) < TOL


def test_g6_interprator():
    g = load_golden("g6_interp")
    w = split_prefix(g, "w.")
    raw = g["raw"]
    assert (raw < 0).any() and (raw > 0.5).any()    # both clamps are exercised
    for out_T, key in ((1.5, "rt_1p5"), (8, "rt_8"), (1, "rt_1")):
        assert max_rel(O.interprator(w, g["x"], out_T), g[key]) < TOL


@pytest.mark.parametrize("tag,axes,nf", [("o1", "TH", 2), ("o2", "T-W", 2), ("o3", "T-H-W", 2)])
def test_g7_taylor(tag, axes, nf):
    g = load_golden("g7_taylor_" + tag)
    order, fi, ol = int(g["meta"][0]), float(g["meta"][1]) / 100, int(g["meta"][2])
    cfg = O.TanteCfg(3, nf, (16, 32), taylor_order=order, frame_interval=fi, output_length=ol, attn_axes=axes,
                     n_head=2, embed_dim=32, patch_scale=8)
    y = O.tante_forward(split_prefix(g, "w."), cfg, g["x"])
    assert y.shape == g["y"].shape and max_rel(y, g["y"]) < TOL


@pytest.mark.parametrize("name", names("g8_rollout_*"))
def test_g8_rollout(name):
    g = load_golden(name)
    ol, n_roll = int(g["meta"][0]), int(g["meta"][1])
    cfg = O.TanteCfg(4, 2, (16, 16), taylor_order=2, output_length=ol, attn_axes="T-L", n_head=2, embed_dim=32,
                     patch_scale=8)
    w = split_prefix(g, "w.")
    assert torch.isnan(g["inp"]).any()
    y, y_ref = O.rollout(w, cfg, {"input": g["inp"], "output": g["out"]}, n_roll)
    assert y.shape == g["y_eval"].shape and max_rel(y, g["y_eval"]) < 5 * TOL
    assert torch.equal(y_ref, g["y_ref"])
    yt, _ = O.rollout(w, cfg, {"input": g["inp"], "output": g["out"][:, :4]}, 4)
    assert yt.shape == g["y_train"].shape and max_rel(yt, g["y_train"]) < 5 * TOL


def test_g9_train_step():
    g = load_golden("g9_trainstep")
    lr, wd, b1, b2, eps, max_norm = (float(v) for v in g["hyper"])
    cfg = O.TanteCfg(4, 2, (16, 16), taylor_order=2, attn_axes="TH-WL", n_head=2, embed_dim=32, patch_scale=8)
    w = {k: v.clone().requires_grad_(True) for k, v in split_prefix(g, "w0.").items()}
    m = {k: torch.zeros_like(v) for k, v in w.items()}
    v2 = {k: torch.zeros_like(v) for k, v in w.items()}
    batch = {"input": g["inp"], "output": g["out"]}
    for step in range(2):
        y, y_ref = O.rollout(w, cfg, batch, 4)
        loss = O.mse(y, y_ref).mean()
        ks = list(w.keys())
        grads = torch.autograd.grad(loss, [w[k] for k in ks])
        assert abs(float(loss.detach()) - float(g[f"loss{step}"])) < 1e-5 * abs(float(g[f"loss{step}"]))
        if step == 0:
            assert max_rel(y, g["y_pred"]) < 5 * TOL
            for k, gr in zip(ks, grads):
                assert max_rel(gr, g["g0." + k]) < 2e-4, k
        clipped, total = O.clip_grad_norm(grads, max_norm)
        assert abs(float(total) - float(g[f"gnorm{step}"])) < 1e-4 * float(g[f"gnorm{step}"])
        with torch.no_grad():
            for k, gr in zip(ks, clipped):
                p, m[k], v2[k] = O.adamw_step(w[k].detach(), gr, m[k], v2[k], step + 1, lr, wd, b1, b2, eps)
                w[k] = p.requires_grad_(True)
        for k in ks:
            # Adam's first steps are sign-like (|update| ~ lr): compare on the update scale
            assert float((w[k].detach() - g[f"w{step + 1}." + k]).abs().max()) < 2e-2 * lr, k


def test_fast_form_of_the_oracle_is_pinned_too():
    """bench.py's cpu_baseline leg times the oracle with torch's fused CPU ops switched in (O.set_fast: F.layer_norm, F.gelu, F.linear,
    scaled_dot_product_attention, the always-run adaptive pool) so that it costs what the reference's forward costs; that spelling is
    held to the same golden vectors: cfg1 end to end, causal and long blocks, the encoder/decoder pair."""
    old = O.set_fast(True)
    try:
        g = load_golden("g1_tante_tiny")
        cfg = O.TanteCfg(4, 1, (64, 64), taylor_order=2, attn_axes="TL-TL", n_head=4, embed_dim=64, patch_scale=8)
        assert max_rel(O.tante_forward(split_prefix(g, "w."), cfg, g["x"]), g["y"]) < TOL
        for name, nh, causal in (("g3_block_c64_L4_causal", 4, True), ("g3_block_c256_L32", 8, False), ("g3_block_c64_L48", 4, False)):
            g = load_golden(name)
            assert max_rel(O.transformer_block(split_prefix(g, "w."), g["x"], nh, causal), g["y"]) < TOL, name
        g = load_golden("g8_rollout_ol1_n8")
        cfg = O.TanteCfg(4, 2, (16, 16), taylor_order=2, output_length=1, attn_axes="T-L", n_head=2, embed_dim=32, patch_scale=8)
        y, _ = O.rollout(split_prefix(g, "w."), cfg, {"input": g["inp"], "output": g["out"]}, 8)
        assert max_rel(y, g["y_eval"]) < 5 * TOL
    finally:
        O.set_fast(old)


def test_g14_wide_train_step():
    """Production shape (C = 256, 8 heads x 32, THWTHWTHW, L in {4, 8, 48}, 4-step BPTT): the oracle's loss, every parameter's gradient
    norm and three full gradient tensors against the reference's."""
    from conftest import g14_setup, G14_KW, G14_FIELDS, G14_RES
    m, batch, g, names = g14_setup()
    cfg = O.TanteCfg(G14_KW["in_T"], G14_FIELDS, G14_RES, taylor_order=1, attn_axes=G14_KW["attn_axes"], n_head=8, embed_dim=256,
                     patch_scale=8)
    w = {k: v.detach().clone() for k, v in m.state_dict().items()}
    for n in names:
        w[n].requires_grad_(True)
    y, y_ref = O.rollout(w, cfg, batch, 4)
    loss = O.mse(y, y_ref).mean()
    grads = torch.autograd.grad(loss, [w[n] for n in names])
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    assert max_rel(y.detach()[:, :, ::8, ::8, :], torch.from_numpy(g["y_pred_slice"])) < 5 * TOL
    gn = np.array([float(x.double().norm()) for x in grads])
    bad = [(n, a, b) for n, a, b in zip(names, gn, g["g_norm"]) if abs(a - b) > 2e-4 * b + 1e-12]
    assert not bad, bad[:5]
    for n, gr in zip(names, grads):
        if "g." + n in g:
            assert max_rel(gr, torch.from_numpy(g["g." + n])) < 2e-4, n
    total = float(np.sqrt((gn ** 2).sum()))
    assert abs(total - float(g["gnorm"])) < 1e-4 * float(g["gnorm"])


def test_g10_metrics_and_lr():
    g = load_golden("g10_metrics")
    x, y = g["x"], g["y"]
    assert max_rel(O.mse(x, y), g["MSE"]) < TOL
    assert max_rel(O.l2re(x, y), g["L2RE"]) < TOL
    assert max_rel(O.nnmse(x, y), g["NNMSE"]) < TOL
    assert max_rel(O.vrmse(x, y), g["VRMSE"]) < TOL
    assert max_rel(O.nmse(x, y), g["NMSE"]) < TOL
    for k in ("below", "inside", "above"):
        assert abs(float(O.eval_rt(g["rt_" + k])) - float(g["eval_rt_" + k])) < 1e-7
        assert abs(float(O.mse_with_rt(x, y, g["rt_" + k])) - float(g["mse_rt_" + k])) < 1e-5
    assert float(g["eval_rt_below"]) > 0 and float(g["eval_rt_inside"]) == 0 and float(g["eval_rt_above"]) > 0
    lrs = g["lr_schedule"].numpy()
    for e in range(35):
        mine = O.warmup_cosine_lr(e, 5e-5, 2, 34, 5e-6, 5e-6)
        assert abs(mine - lrs[e]) < 1e-10, (e, mine, lrs[e])


def test_g13_adaptive_dt():
    g = load_golden("g13_deg_false")
    cfg = O.TanteCfg(4, 1, (32, 32), taylor_order=2, attn_axes="TH-TW", n_head=2, embed_dim=32, patch_scale=8,
                     deg=False)
    w = split_prefix(g, "w.")
    for out_T, tag in ((1.5, "1p5"), (6, "6")):
        y, rt = O.tante_forward(w, cfg, g["x"], out_T)
        assert y.shape == g["y_" + tag].shape
        assert max_rel(rt, g["rt_" + tag]) < TOL and max_rel(y, g["y_" + tag]) < TOL
    assert g["y_6"].shape[1] > 1      # the multi-frame branch is exercised


# ---- CViT (g11) and the spectral operator path (g12) ----------------------------------------------------------------------------
from oracle import cvit_oracle as OC          # noqa: E402
from oracle import spectral_oracle as OS      # noqa: E402

CVIT_CASES = {   # name -> (CvitCfg kwargs, n_fields, resolution): the ctor arguments tests/golden/make_golden.py g11 used
    "grid": (dict(grid_size=(16, 24), eps=1e5), 2, (16, 24)),
    "gridwide": (dict(grid_size=(6, 5), eps=40.0, dec_depth=2, num_mlp_layers=2), 2, (16, 24)),
    "fourier": (dict(embedding_type="fourier"), 1, (16, 16)),
    "mlp": (dict(embedding_type="mlp", mlp_ratio=2), 3, (16, 24)),
}


def cvit_cfg(name, cls=OC.CvitCfg):
    kw, nf, res = CVIT_CASES[name]
    base = dict(out_steps=3, patch_size=(1, 8, 8), grid_size=(8, 8), latent_dim=24, emb_dim=32, depth=2, num_heads=4, dec_emb_dim=48,
                dec_num_heads=4, dec_depth=1, num_mlp_layers=1, mlp_ratio=1)
    base.update(kw)
    return cls(4, nf, res, **base)


@pytest.mark.parametrize("name", sorted(CVIT_CASES))
def test_g11_cvit(name):
    g = load_golden("g11_cvit_" + name)
    y = OC.cvit_forward(split_prefix(g, "w."), cvit_cfg(name), g["x"], g.get("coords"))
    assert y.shape == g["y"].shape and max_rel(y, g["y"]) < 5 * TOL


@pytest.mark.parametrize("name", ["low", "clip", "odd"])
def test_g12_spectral_layer(name):
    g = load_golden("g12_spectral_" + name)
    m1, m2 = (int(v) for v in g["modes"])
    y = OS.spectral_layer(split_prefix(g, "w."), g["x"], m1, m2)
    assert y.shape == g["y"].shape and max_rel(y, g["y"]) < TOL


def test_g12_tante_fno():
    g = load_golden("g12_tante_fno")
    cfg = O.TanteCfg(4, 2, (32, 32), taylor_order=2, attn_axes="TL-TL", n_head=4, embed_dim=64, patch_scale=8, enc_dec_type="fno",
                     modes1=8, modes2=8)
    y = O.tante_forward(split_prefix(g, "w."), cfg, g["x"])
    assert y.shape == g["y"].shape and max_rel(y, g["y"]) < 5 * TOL


# ---- g15 (round 5): the reference's GRADIENTS through padded conv / deconv stages and the channel-attention letter ------------------
@pytest.mark.parametrize("name", names("g15_encdec_grad_*"))
def test_g15_encdec_gradients(name):
    """torch autograd through the oracle's enc_cnn / dec_cnn ('same'-padded kernel-4 stages: patch_scale 16 / 32 / 64) reproduces the
    reference's backward(): input gradient and every parameter's gradient."""
    g = load_golden(name)
    ps = int(g["meta"][0])
    we = {k: v.clone().requires_grad_(True) for k, v in split_prefix(g, "enc.").items()}
    x = g["x"].clone().requires_grad_(True)
    z = O.enc_cnn(we, x, ps, 0.0)
    assert max_rel(z.detach(), g["z"]) < TOL
    (z * g["wz"]).sum().backward()
    assert max_rel(x.grad, g["dx"]) < 1e-5
    for k, v in we.items():
        assert max_rel(v.grad, g["genc." + k]) < 1e-5, k
    wd = {k: v.clone().requires_grad_(True) for k, v in split_prefix(g, "dec.").items()}
    zz = g["zz"].clone().requires_grad_(True)
    r = O.dec_cnn(wd, zz, ps, 0.0)
    assert max_rel(r.detach(), g["r"]) < TOL
    (r * g["wr"]).sum().backward()
    assert max_rel(zz.grad, g["dzz"]) < 1e-5
    for k, v in wd.items():
        assert max_rel(v.grad, g["gdec." + k]) < 1e-5, k


@pytest.mark.parametrize("name", names("g16_encdec_grad_*"))
def test_g16_overlapping_stage_gradients(name):
    """The same through OVERLAPPING stages (overlap_ratio 0.5 -- the constructor default -- and 0.25: stride < kernel, adaptive average
    pool after the conv, summed taps + bilinear resize after the transposed conv; enc_dec_cnn.py:97-110, 164-184)."""
    g = load_golden(name)
    ps, ov = int(g["meta"][0]), int(g["meta"][1]) / 100.0
    we = {k: v.clone().requires_grad_(True) for k, v in split_prefix(g, "enc.").items()}
    x = g["x"].clone().requires_grad_(True)
    z = O.enc_cnn(we, x, ps, ov)
    assert max_rel(z.detach(), g["z"]) < TOL
    (z * g["wz"]).sum().backward()
    assert max_rel(x.grad, g["dx"]) < 1e-5
    for k, v in we.items():
        assert max_rel(v.grad, g["genc." + k]) < 1e-5, k
    wd = {k: v.clone().requires_grad_(True) for k, v in split_prefix(g, "dec.").items()}
    zz = g["zz"].clone().requires_grad_(True)
    r = O.dec_cnn(wd, zz, ps, ov)
    assert max_rel(r.detach(), g["r"]) < TOL
    (r * g["wr"]).sum().backward()
    assert max_rel(zz.grad, g["dzz"]) < 1e-5
    for k, v in wd.items():
        assert max_rel(v.grad, g["gdec." + k]) < 1e-5, k


@pytest.mark.parametrize("name", names("g15_backbone_grad_*"))
def test_g15_backbone_gradients(name):
    g = load_golden(name)
    axes = name.split("_")[-1]
    w = {k: v.clone().requires_grad_(True) for k, v in split_prefix(g, "w.").items()}
    x = g["x"].clone().requires_grad_(True)
    y = O.attn_backbone(w, x, axes, int(g["meta"][5]))
    assert max_rel(y.detach(), g["y"]) < TOL
    (y * g["w"]).sum().backward()
    assert max_rel(x.grad, g["dx"]) < 1e-5
    for k, v in w.items():      # (six blocks deep in "LTCAXY": fp32 re-association between two CPU evaluation orders reaches 2.2e-5)
        assert max_rel(v.grad, g["g." + k]) < 5e-5, k
