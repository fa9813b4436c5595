"""GPU parity, round 6.

* csrc/tail_chain.hip -- the training form of a rollout call's token-local tail (dec_CNN stages, enc_dec_cnn.py:263-277 -> Taylor sum,
  tante.py:165-171 -> enc_CNN stages on the predicted frame, enc_dec_cnn.py:217-229) as one launch forward and one backward: against the
  per-operator nodes it replaces (same bf16 compute mode; values and every gradient), at the production shape (fixture g14: D = 4) and
  on a two-order model with D = 11 (three row tiles at the pixel level, two decoders).  Against the REFERENCE's gradients the fused
  tail is what test_g14_wide_train_step[bf16-...] runs since this round (tests/test_hip_round2.py).

Bars: fp32 compute 1e-5 / gradients 2e-4, bf16 compute 1e-2 / gradients 4e-2, relative to the reference's fp32 CPU result.
"""
import numpy as np
import pytest
import torch

from conftest import rel_err, max_rel, record_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _train_grads(dev, monkeypatch, model, batch, md, fused_tail, n_steps=4, n_loss=None):
    import tante_amd
    from tante_amd import autograd as A
    from tante_amd import train_forward as TF
    monkeypatch.setattr(TF, "FUSED_TAIL", fused_tail)
    m = model.to(dev).train().set_compute("bf16")
    opt = tante_amd.FlatAdamW(m.parameters(), lr=1e-4)
    opt.zero_grad()
    fmt = tante_amd.DefaultChannelsFirstFormatter(md)
    b = {k: v.to(dev) for k, v in batch.items()}
    y_pred, y_ref = tante_amd.rollout_model(m, b, fmt, n_steps)
    n_loss = n_steps if n_loss is None else n_loss
    loss = A.MseMeanFn.apply(y_pred[:, :n_loss].contiguous(), y_ref[:, :n_loss].contiguous())
    A.run_backward(loss)
    torch.cuda.synchronize()
    return y_pred.detach().float().cpu(), {k: p.grad.detach().clone().cpu() for k, p in m.named_parameters()}, float(loss)


def _compare(ya, ga, yb, gb, what, tol_y=2e-3, tol_g=2e-2):
    e = rel_err(ya, yb)
    record_parity(e, max_rel(ya, yb), tol_y, "bf16", what + ": predicted frames, fused tail vs per-operator nodes")
    assert e < tol_y, e
    worst, wk = 0.0, None
    for k in gb:
        if float(gb[k].abs().max()) == 0.0:
            assert float(ga[k].abs().max()) == 0.0, k
            continue
        e = rel_err(ga[k], gb[k])
        if e > worst:
            worst, wk = e, k
        assert e < tol_g, (k, e)
    record_parity(worst, worst, tol_g, "bf16", what + ": every parameter gradient, worst " + str(wk))
    return worst


def test_training_tail_in_one_launch_production_shape(dev, monkeypatch):
    """g14's model (C = 256, THWTHWTHW, 64 x 384 x 4 fields, B = 2, 4-step BPTT): the fused tail (three of its four calls re-encode the
    predicted frame, the last does not) against the per-operator path in the same compute mode.  Both are bf16 paths that round at
    slightly different places (the fused forward's GELU is the polynomial of the inference kernels): 2e-3 on the frames, 2e-2 on every
    gradient tensor (relative L2) -- the bar against the reference is test_g14_wide_train_step's 4e-2."""
    import tante_amd
    from conftest import g14_setup, G14_FIELDS, G14_RES
    md = tante_amd.TanteMetadata(n_fields=G14_FIELDS, spatial_resolution=G14_RES)
    m, batch, _, _ = g14_setup()
    yf, gf, lf = _train_grads(dev, monkeypatch, m, batch, md, True)
    m2, batch2, _, _ = g14_setup()
    yu, gu, lu = _train_grads(dev, monkeypatch, m2, batch2, md, False)
    assert abs(lf - lu) < 2e-3 * abs(lu)
    _compare(yf, gf, yu, gu, "g14")
    enc = [k for k in gf if k.startswith("encoder.")]
    dec = [k for k in gf if k.startswith("decoders.")]
    assert enc and dec and all(float(gf[k].abs().max()) > 0.0 for k in enc + dec)


@pytest.mark.parametrize("n_loss", [3, 1])
def test_training_tail_two_orders_eleven_fields(dev, monkeypatch, n_loss):
    """Two Taylor orders (two decoders, two residual streams), D = 11 (4 D = 44: three row tiles, zero-padded k), Hp x Wp = 4 x 16, a loss
    on the first `n_loss` of three frames (n_loss = 1: the later calls' nodes never run; the first call's frame still receives the
    gradient of its encoding's consumers only where they are part of the graph)."""
    import tante_amd
    md = tante_amd.TanteMetadata(n_fields=11, spatial_resolution=(32, 128))
    kw = dict(in_T=4, taylor_order=2, attn_axes="TH-W", n_head=8, embed_dim=256, patch_scale=8, dropout=0.0, frame_interval=0.5)

    def build():
        torch.manual_seed(7)
        m = tante_amd.TANTE(dset_metadata=md, **kw)
        gen = torch.Generator().manual_seed(77)
        batch = {"input": torch.randn(2, 4, 32, 128, 11, generator=gen), "output": torch.randn(2, 3, 32, 128, 11, generator=gen)}
        return m, batch
    m, batch = build()
    yf, gf, lf = _train_grads(dev, monkeypatch, m, batch, md, True, n_steps=3, n_loss=n_loss)
    m2, batch2 = build()
    yu, gu, lu = _train_grads(dev, monkeypatch, m2, batch2, md, False, n_steps=3, n_loss=n_loss)
    assert abs(lf - lu) < 2e-3 * abs(lu)
    _compare(yf, gf, yu, gu, f"two orders, D = 11, loss on {n_loss} of 3 frames")


def test_training_tail_is_taken_and_counts_launches(dev, monkeypatch):
    """The fused tail is the path the shipped training configuration takes (tail_train_cfg returns a configuration for cfg3's model), and a
    train step with it issues fewer launches than without (the point of the fusion: ~45 fewer per rollout call)."""
    import tante_amd
    from tante_amd import train_forward as TF
    from conftest import g14_setup, G14_FIELDS, G14_RES
    md = tante_amd.TanteMetadata(n_fields=G14_FIELDS, spatial_resolution=G14_RES)
    counts = {}
    for fused in (True, False):
        monkeypatch.setattr(TF, "FUSED_TAIL", fused)
        m, batch, _, _ = g14_setup()
        m = m.to(dev).train().set_compute("bf16")
        opt = tante_amd.FlatAdamW(m.parameters(), lr=1e-4)
        fmt = tante_amd.DefaultChannelsFirstFormatter(md)
        b = {k: v.to(dev) for k, v in batch.items()}
        tante_amd.train_step(m, opt, b, fmt, 4)      # warm-up: packs, workspaces
        torch.cuda.synchronize()
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
            tante_amd.train_step(m, opt, b, fmt, 4)
            torch.cuda.synchronize()
        counts[fused] = sum(1 for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA)
    assert counts[True] < counts[False] - 100, counts
    record_parity(0.0, 0.0, 1.0, "bf16", f"launches per train step: {counts[True]} with the fused tail, {counts[False]} without")


def test_adaptive_trainer_step_with_the_float16_grad_scaler(dev):
    """R_Trainer.train_one_epoch's float16 sequence (trainer/r_trainer.py:152-158) through train_step_adaptive(scaler=): scale(loss).backward(),
    clip_grad_value_(1.0) on the gradients AS THEY ARE -- the reference has no unscale_ in front of the clip, so the SCALED gradients are
    clipped --, scaler.step (unscales, skips on a non-finite gradient), update.  (1) scale 1: the same step as without a scaler; (2) scale 2^10: after the step the bucket holds clip(2^10 g, 1) / 2^10, i.e. nothing above 2^-10; (3) a non-finite gradient:
    the step is skipped, the scale halves."""
    import math
    import tante_amd
    from conftest import load_golden, split_prefix
    g = load_golden("g13_deg_false")
    md = tante_amd.TanteMetadata(n_fields=1, spatial_resolution=(32, 32))
    fmt = tante_amd.DefaultChannelsFirstFormatter(md)
    gen = torch.Generator().manual_seed(0)
    batch = {"input": torch.randn(2, 4, 32, 32, 1, generator=gen).to(dev), "output": torch.randn(2, 3, 32, 32, 1, generator=gen).to(dev)}

    def build():
        m = tante_amd.TANTE(in_T=4, dset_metadata=md, taylor_order=2, attn_axes="TH-TW", n_head=2, embed_dim=32, patch_scale=8,
                            dropout=0.0, deg=False).to(dev).train()
        m.load_state_dict(split_prefix(g, "w."))
        return m, tante_amd.FlatAdamW(m.parameters(), lr=1e-3)
    m0, o0 = build()
    l0, _ = tante_amd.train_step_adaptive(m0, o0, batch, fmt, 3)
    m1, o1 = build()
    s1 = torch.amp.GradScaler("cuda", init_scale=1.0, enabled=True)
    l1, _ = tante_amd.train_step_adaptive(m1, o1, batch, fmt, 3, scaler=s1)
    # (two runs of the same step differ in the last bits: the loss reduction and some parameter gradients add with atomics)
    assert abs(float(l0) - float(l1)) < 1e-5 * abs(float(l0)) and float((o0.flat_p - o1.flat_p).abs().max()) < 1e-6, "scale 1 must be the plain step"
    m2, o2 = build()
    s2 = torch.amp.GradScaler("cuda", init_scale=2.0 ** 10, enabled=True)
    w_before = o2.flat_p.clone()
    l2, _ = tante_amd.train_step_adaptive(m2, o2, batch, fmt, 3, scaler=s2)
    assert math.isfinite(float(l2)) and float((o2.flat_p - w_before).abs().max()) > 0
    assert float(o2.flat_g.abs().max()) <= 2.0 ** -10 * (1 + 1e-6)          # the clip acted on the scaled gradients
    assert float(o0.flat_g.abs().max()) > 2.0 ** -10                         # ... which the unscaled step's gradients exceed
    m3, o3 = build()
    s3 = torch.amp.GradScaler("cuda", init_scale=2.0 ** 10, enabled=True)
    with torch.no_grad():
        next(m3.parameters()).view(-1)[0] = float("inf")                      # a non-finite weight -> non-finite loss and gradients
    w_before = o3.flat_p.clone()
    tante_amd.train_step_adaptive(m3, o3, batch, fmt, 3, scaler=s3)
    fin = torch.isfinite(w_before)
    assert torch.equal(o3.flat_p[fin], w_before[fin]) and o3.step_count == 0 and s3.get_scale() == 2.0 ** 9


@pytest.mark.parametrize("mode", ["bf16", "fp32"])
def test_train_mode_dropout_without_autograd(dev, mode):
    """A backbone in train() mode with dropout 0.1 under torch.no_grad(): nn.Dropout and the attention's dropout stay active whatever the
    grad mode (attn_backbone.py:47,56,81-82); round 5 raised NotImplementedError here.  (1) it runs and differs from eval(); (2) with the
    same seeds it is the differentiable path's forward (same block kernels, same masks); (3) the dropped activations average out: over 24
    seeds the mean output is within 3 standard errors of eval()'s (dropout is unbiased up to the non-linearity, at p = 0.1 and small
    random weights)."""
    import tante_amd
    from tante_amd import autograd as A
    torch.manual_seed(5)
    bb = tante_amd.Attn_Backbone((4, 8, 16, 256), "TH", expanded_channel=64, n_head=8, mlp_ratio=1.0, dropout=0.1).to(dev)
    bb.compute = mode
    x = torch.randn(2, 4, 8, 16, 256, device=dev)
    with torch.no_grad():
        y_eval = bb.eval()(x)
        bb.train()
        A._SEED[0] = 1234
        y_drop = bb(x)
    assert torch.isfinite(y_drop).all() and not torch.equal(y_drop, y_eval)
    A._SEED[0] = 1234
    xg = x.clone().requires_grad_(True)
    y_grad = bb(xg)
    # (the same dropout masks: the blocks draw their seeds in the same order; the propagators between them run the inference kernels
    # here and the training ones there -- different roundings in bf16 mode, none of the size a different mask would make)
    e = rel_err(y_grad.detach().float().cpu(), y_drop.float().cpu())
    record_parity(e, e, 1e-2 if mode == "bf16" else 1e-5, mode, "train() + dropout without autograd vs the differentiable path's forward, same seeds")
    assert e < (1e-2 if mode == "bf16" else 1e-5), e
    assert rel_err(y_drop.float().cpu(), y_eval.float().cpu()) > 10 * e
    acc = torch.zeros_like(y_eval, dtype=torch.float64)
    n = 24
    with torch.no_grad():
        for s in range(n):
            A._SEED[0] = 77 * s + 1
            acc += bb(x).double()
    mean = (acc / n).float()
    spread = float((y_drop - y_eval).float().std())
    bias = float((mean - y_eval).abs().mean())
    assert bias < 3.0 * spread / n ** 0.5 + 1e-3 * float(y_eval.abs().mean()), (bias, spread)


def _golden_names(pattern):
    import glob
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(here, pattern + ".npz")))


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("name", _golden_names("g17_block_masked_grad_*"))
def test_g17_masks_under_autograd(dev, name, mode):
    """TransformerBlock.forward(x, key_padding_mask, attn_mask, causal) (attn_backbone.py:59-83) with masks AND autograd: bool / additive
    float / per-(batch, head) attn_mask, bool key_padding_mask, bool mask | causal.  Output, input gradient and every parameter's gradient
    against the reference's backward() (round 5 raised here: masks were inference-only)."""
    import tante_amd
    from conftest import load_golden, split_prefix
    g = load_golden(name)
    Lq, C, nh, causal = (int(v) for v in g["meta"])
    blk = tante_amd.TransformerBlock(C, nh, mlp_ratio=2.0, dropout=0.0).to(dev).train()
    blk.load_state_dict(split_prefix(g, "w."))
    blk.compute = mode
    ft, gt = (1e-5, 2e-4) if mode == "fp32" else (1e-2, 4e-2)
    kw = {k: g[k].to(dev) for k in ("attn_mask", "key_padding_mask") if k in g}
    x = g["x"].to(dev).requires_grad_(True)
    y = blk(x, causal=bool(causal), **kw)
    (y.float() * g["w"].to(dev)).sum().backward()
    assert max_rel(y.detach().float().cpu(), g["y"]) < ft
    worst = (max_rel if mode == "fp32" else rel_err)(x.grad.cpu(), g["dx"])
    assert worst < gt, ("dx", worst)
    for k, q in blk.named_parameters():
        ref = g["g." + k]
        err = max_rel(q.grad.cpu(), ref) if mode == "fp32" else rel_err(q.grad.cpu(), ref)
        assert err < (gt if (mode == "fp32" or ref.dim() > 1) else 1.25 * gt), (k, err)
        worst = max(worst, err)
    # the inference kernels under the same masks give the same block output.  With `causal` and no key_padding_mask the fixture holds what the
    # reference computes in train() mode / under autograd: PLAIN causal attention (torch drops the mask tensor next to is_causal=True);
    # eval() without autograd is torch's fused path, which applies mask | causal -- so there the outputs differ, as the reference's do.
    with torch.no_grad():
        y2 = blk(g["x"].to(dev), causal=bool(causal), **kw)
        assert max_rel(y2.float().cpu(), g["y"]) < ft
        y3 = blk.eval()(g["x"].to(dev), causal=bool(causal), **kw)
        if causal:
            assert max_rel(y3.float().cpu(), g["y"]) > 1e-2
            y4 = blk(g["x"].to(dev), causal=True)
            assert torch.equal(y2, y4)
        else:
            assert max_rel(y3.float().cpu(), g["y"]) < ft
    record_parity(worst, worst, gt if mode == "fp32" else 1.25 * gt, mode, f"{name}: masked TransformerBlock gradients vs the reference")


def test_masks_under_autograd_refuse_dropout(dev):
    import tante_amd
    blk = tante_amd.TransformerBlock(64, 4, mlp_ratio=1.0, dropout=0.1).to(dev).train()
    x = torch.randn(2, 8, 64, device=dev, requires_grad=True)
    with pytest.raises(NotImplementedError, match="dropout"):
        blk(x, attn_mask=torch.zeros(8, 8, dtype=torch.bool, device=dev))


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_g17_channel_letter_over_256_channels_trains(dev, mode):
    """Letter 'C' attends over the channel axis (attn_backbone.py:184-189): at the production width that is sequences of 256, past the MFMA
    attention backward's 128 -- the recomputing lane-per-row backward (tante_attention_masked_bwd, no masks) takes them.  Against the
    reference's backward()."""
    import tante_amd
    from conftest import load_golden, split_prefix
    g = load_golden("g17_backbone_grad_C256")
    T, H, W, C, E, nh = (int(v) for v in g["meta"])
    bb = tante_amd.Attn_Backbone((T, H, W, C), "C", expanded_channel=E, n_head=nh, mlp_ratio=1.0, dropout=0.0).to(dev).train()
    bb.load_state_dict(split_prefix(g, "w."))
    bb.compute = mode
    ft, gt = (1e-5, 2e-4) if mode == "fp32" else (1e-2, 4e-2)
    x = g["x"].to(dev).requires_grad_(True)
    y = bb(x)
    (y.float() * g["w"].to(dev)).sum().backward()
    assert max_rel(y.detach().float().cpu(), g["y"]) < ft
    worst = (max_rel if mode == "fp32" else rel_err)(x.grad.cpu(), g["dx"])
    assert worst < gt, ("dx", worst)
    ours, refs = [], []
    for k, q in bb.named_parameters():
        ref = g["g." + k]
        if float(ref.abs().max()) == 0.0:
            assert q.grad is None or float(q.grad.abs().max()) == 0.0, k
            continue
        ours.append(q.grad.cpu().reshape(-1))
        refs.append(ref.reshape(-1))
        err = max_rel(q.grad.cpu(), ref) if mode == "fp32" else rel_err(q.grad.cpu(), ref)
        assert err < (gt if (mode == "fp32" or ref.dim() > 1) else 1.25 * gt), (k, err)
        worst = max(worst, err)
    err = rel_err(torch.cat(ours), torch.cat(refs))
    assert err < gt, ("all parameters", err)
    record_parity(max(worst, err), max(worst, err), gt if mode == "fp32" else 1.25 * gt, mode,
                  "g17_backbone_grad_C256: letter 'C' over 256 channels, gradients vs the reference")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("D", [4, 8, 16, 32, 64])
def test_masked_attention_backward_against_torch(dev, D, dtype):
    """tante_attention_masked / _bwd alone, every head dim, against torch's scaled_dot_product_attention under autograd (fp32), with a
    per-(batch, head) additive mask holding -inf entries + a key-padding mask + causal."""
    from tante_amd import autograd as A
    nh, Bp, Lq = 3, 2, 37
    Cc = nh * D
    gen = torch.Generator().manual_seed(60 + D)
    qkv = torch.randn(Bp * Lq, 3 * Cc, generator=gen)
    am = torch.randn(Bp * nh, Lq, Lq, generator=gen)
    am[torch.rand(Bp * nh, Lq, Lq, generator=gen) < 0.25] = float("-inf")
    am[:, torch.arange(Lq), torch.arange(Lq)] = 0.0
    kp = torch.zeros(Bp, Lq)
    kp[1, 30:] = float("-inf")
    do = torch.randn(Bp * Lq, Cc, generator=gen)
    for causal in (False, True):
        kpc = torch.zeros_like(kp) if causal else kp      # causal + padded tail would block whole rows of batch 1 (its own key is padded)
        q32 = qkv.to(dtype).float().detach().requires_grad_(True)
        q, k, v = (t.view(Bp, Lq, nh, D).transpose(1, 2) for t in q32.split(Cc, dim=1))
        bias = am.view(Bp, nh, Lq, Lq) + kpc.view(Bp, 1, 1, Lq)
        if causal:
            bias = bias + torch.full((Lq, Lq), float("-inf")).triu(1)
        o_ref = torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=bias).transpose(1, 2).reshape(Bp * Lq, Cc)
        (o_ref * do.to(dtype).float()).sum().backward()
        x = qkv.to(dev).to(dtype).detach().requires_grad_(True)
        o = A.MaskedAttentionFn.apply(x, Cc, nh, Bp, Lq, causal, am.to(dev), kpc.to(dev))
        (o.float() * do.to(dev).to(dtype).float()).sum().backward()
        tol = 2e-5 if dtype == torch.float32 else 1.5e-2
        assert rel_err(o.detach().float().cpu(), o_ref.detach()) < tol
        assert rel_err(x.grad.float().cpu(), q32.grad) < tol, (D, causal)


@pytest.mark.parametrize("M", [16, 200, 256, 1024])
@pytest.mark.parametrize("case", ["ln_qkv", "out_res", "ln_fc1_gelu", "dense_gelu_res"])
def test_small_row_gemm_against_float64(dev, M, case):
    """gemm.hip gemm_small_kernel (bf16 compute, K = 512, at most TANTE_GEMM_SMALLM rows: one wave per (16 rows, 32 features), CViT's
    encoder blocks at B = 1, cvit.py:112-139): every epilogue it serves against float64 on bf16-rounded operands, and against the
    token-stationary kernel it replaces (TANTE_GEMM_SMALLM = 0) within rounding.  M = 200: a ragged last row tile."""
    import tante_amd
    from tante_amd import kernels as K, _lib as L
    gen = torch.Generator().manual_seed(600 + M)
    Kd = 512
    N = 1536 if case == "ln_qkv" else 512
    w = (torch.randn(N, Kd, generator=gen) / Kd ** 0.5).to(dev)
    b = (0.1 * torch.randn(N, generator=gen)).to(dev)
    ln = case.startswith("ln_")
    gamma = (1 + 0.1 * torch.randn(Kd, generator=gen)).to(dev) if ln else None
    beta = (0.1 * torch.randn(Kd, generator=gen)).to(dev) if ln else None
    a_bf16 = case == "out_res"
    x = torch.randn(M, Kd, generator=gen).to(dev) * 1.5 + 0.3
    xa = x.to(torch.bfloat16) if a_bf16 else x
    res = torch.randn(M, N, generator=gen).to(dev) if case in ("out_res", "dense_gelu_res") else None
    act = L.ACT_GELU_ERF if "gelu" in case else L.ACT_NONE
    odt = torch.bfloat16 if case in ("ln_qkv", "ln_fc1_gelu") else torch.float32
    pk = K.pack_weight(w, b, L.BF16, gamma=gamma, beta=beta)

    def run():
        out = torch.empty(M, N, dtype=odt, device=dev)
        K.linear(xa, pk, out, M=M, ln=ln, ln_eps=1e-5, act=act, residual=res)
        return out.float()
    y_small = run()
    tante_amd.set_option("TANTE_GEMM_SMALLM", 0)
    try:
        y_old = run()
    finally:
        tante_amd.set_option("TANTE_GEMM_SMALLM", 1024)
    xr = xa.double()
    if ln:
        xr = torch.nn.functional.layer_norm(xr, (Kd,), gamma.double(), beta.double(), 1e-5)
    ref = xr @ w.double().t() + b.double()
    if act == L.ACT_GELU_ERF:
        ref = torch.nn.functional.gelu(ref)
    if res is not None:
        ref = ref + res.double()
    tol = 1.2e-2
    assert rel_err(y_small.cpu(), ref.float().cpu()) < tol
    assert rel_err(y_small.cpu(), y_old.cpu()) < 6e-3
    if M in (200, 256) and case == "ln_qkv":
        record_parity(rel_err(y_small.cpu(), ref.float().cpu()), max_rel(y_small.cpu(), ref.float().cpu()), tol, "bf16",
                      f"gemm_small_kernel {case} M={M} vs float64")


def test_spectral_layer_reads_strided_frames_and_writes_rows(dev):
    """tante_spectral_layer_x (round 6): the spectral layer on one frame of every item of a rollout buffer read IN PLACE (images a batch
    stride apart; enc_FNO's first layer, enc_dec_fno.py:224-273) and its output as channels-last rows for the row GEMM behind dec_FNO's first
    layer (enc_dec_fno.py:276-323) -- the same kernels with a pointer stride / the MFMA operands exchanged: bit-identical to the dense
    call and to a layout copy of the image."""
    import tante_amd
    from tante_amd import _lib as L
    torch.manual_seed(66)
    # (a) strided frames: cfg5's first encoder layer, 8 -> 32 channels at 512 x 512, 20 x 20 modes, bf16 and fp32 output
    lay = tante_amd.SpectralLayer(8, 32, 20, 20).to(dev)
    buf = torch.randn(2, 3, 8, 512, 512, device=dev)
    frame = buf[:, 1]                                     # (2, 8, 512, 512), batch stride 3 frames
    assert not frame.is_contiguous()
    assert L.lib().tante_spectral_layer_x_supported(2, 8, 32, 512, 512, 20, 20, 1, 1)
    for bf16_out in (True, False):
        y_str = lay.run(frame, L.ACT_GELU_ERF, L.BF16, bf16_out=bf16_out)
        y_den = lay.run(frame.contiguous(), L.ACT_GELU_ERF, L.BF16, bf16_out=bf16_out)
        assert y_str.dtype == y_den.dtype and torch.equal(y_str, y_den)
    # (b) rows out: cfg5's first decoder layer, 128 -> 64 channels at 128 x 128, 5 x 5 modes
    lay2 = tante_amd.SpectralLayer(128, 64, 5, 5).to(dev)
    x = torch.randn(2, 128, 128, 128, device=dev)
    assert L.lib().tante_spectral_layer_x_supported(2, 128, 64, 128, 128, 5, 5, 0, 2)
    rows = lay2.run(x, L.ACT_GELU_ERF, L.BF16, nhwc_out=True)
    img = lay2.run(x, L.ACT_GELU_ERF, L.BF16)
    assert rows.shape == (2 * 128 * 128, 64) and torch.equal(rows.view(2, 128, 128, 64), img.permute(0, 2, 3, 1))
    # the fp32 mode and unsupported shapes take the copy route with the same result
    rows32 = lay2.run(x, L.ACT_NONE, L.F32, nhwc_out=True)
    assert torch.equal(rows32.view(2, 128, 128, 64), lay2.run(x, L.ACT_NONE, L.F32).permute(0, 2, 3, 1))
    small = tante_amd.SpectralLayer(4, 8, 3, 3).to(dev)
    xs = torch.randn(3, 2, 4, 32, 32, device=dev)[:, 0]
    assert torch.equal(small.run(xs, L.ACT_NONE, L.BF16), small.run(xs.contiguous(), L.ACT_NONE, L.BF16))


def test_vertical_propagator_applies_film_while_loading(dev):
    """tante_axis_mlp_film (round 6): FiLM(t) + positional terms (tante.py:136-141) applied by the vertical propagator's tile load
    (attn_backbone.py:140-141) to a window of cached frame encodings -- against the two launches it replaces (tante_film_pos_fwd_frames,
    then tante_axis_mlp_c in place): the same expression per token and the same kernel behind it, bit for bit.  Frames a batch stride
    apart and in any order (a sliding window of a frame-major cache)."""
    import ctypes as C
    from tante_amd import _lib as L, kernels as K
    torch.manual_seed(67)
    B, T, Hp, Wp, Cc = 2, 4, 64, 64, 256
    HW = Hp * Wp
    assert L.lib().tante_axis_mlp_film_supported(B, T, Hp, Wp * Cc, Cc)
    cache = torch.randn(T + 2, B, HW, Cc, device=dev)
    fa, fb = torch.randn(T, Cc, device=dev), torch.randn(T, Cc, device=dev)
    se = torch.randn(HW, Cc, device=dev)
    w1, w2 = torch.randn(Hp, Hp, device=dev) / 8, torch.randn(Hp, Hp, device=dev) / 8
    b1, b2 = torch.randn(Hp, device=dev), torch.randn(Hp, device=dev)
    fr = L.Frames()
    for t in range(T):
        f = cache[t + 1]
        fr.f[t], fr.bstride[t] = f.data_ptr(), f.stride(0)
    st = torch.cuda.current_stream().cuda_stream
    x2 = torch.empty(B * T * HW, Cc, device=dev)
    L.check(L.lib().tante_film_pos_fwd_frames(C.byref(fr), fa.data_ptr(), fb.data_ptr(), se.data_ptr(), B, T, HW, Cc, x2.data_ptr(), st))
    K.axis_mlp(x2, B * T, Hp, Wp * Cc, w1, b1, w2, b2, L.BF16)
    x1 = torch.full((B * T * HW, Cc), float("nan"), device=dev)
    L.check(L.lib().tante_axis_mlp_film(x1.data_ptr(), C.byref(fr), fa.data_ptr(), fb.data_ptr(), se.data_ptr(), B, T, Hp, Wp * Cc, Cc,
                                        w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), st))
    assert torch.equal(x1, x2)
    assert not L.lib().tante_axis_mlp_film_supported(B, T, Hp, Wp * 128, 128)
