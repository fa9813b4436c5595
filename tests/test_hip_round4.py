"""GPU parity, round 4: the re-cut launch list of a rollout call.

* csrc/head_enc.hip -- every Taylor order's derivative head + the Taylor sum + the RE-ENCODING of the predicted frame in one launch
  (enc_dec_cnn.py:263-277, tante.py:165-171, enc_dec_cnn.py:217-229): frames bit-identical to the launch it replaces, the encoding
  against the encoder launches it replaces and against the oracle's encoder, deterministic across arrival orders.

Bars: fp32 compute 1e-5, bf16 compute 1e-2 relative (L2 and max-norm) to the oracle's fp32 CPU result, as `north_star` states.
"""
import os

import pytest
import torch

from conftest import rel_err, max_rel, record_parity

pytestmark = pytest.mark.gpu

TOL = {"fp32": 1e-5, "bf16": 1e-2}


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def close(a, b, mode, note="", scale=1.0):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.isfinite(a).all()
    r, m = rel_err(a, b), max_rel(a, b)
    record_parity(r, m, TOL[mode] * scale, mode, note)
    assert r < TOL[mode] * scale and m < TOL[mode] * scale * 2, f"{note}: rel={r:.3e} max={m:.3e} (tol {TOL[mode] * scale:.1e})"
    return r


def _model(dev, D, res, order, axes, seed=0):
    import tante_amd
    torch.manual_seed(100 + seed)
    md = tante_amd.TanteMetadata(n_fields=D, spatial_resolution=res)
    return tante_amd.TANTE(in_T=4, dset_metadata=md, taylor_order=order, attn_axes=axes, n_head=8, embed_dim=256, patch_scale=8, frame_interval=0.5,
                           dropout=0.0).to(dev).eval().set_compute("bf16")


@pytest.mark.parametrize("D,res,B,order,axes", [
    (11, (128, 128), 3, 3, "T-H-W"),        # 3 stage-3 tiles; 768 rows: 64-token groups
    (11, (64, 96), 3, 1, "THW"),            # order 1; 288 rows: the last 64-token group half empty (dead tiles)
    (4, (128, 384), 2, 1, "THW"),           # TRL-2D shape, order 1, ONE stage-3 tile; 1 536 rows
    (8, (128, 128), 5, 2, "T-W"),           # two tiles, order 2
    (16, (128, 128), 1, 4, "T-T-T-T"),      # four tiles, four orders
    (1, (32, 32), 2, 1, "HW"),              # one field, 16 tokens per image: one tile per image, a single group
    (11, (256, 256), 8, 3, "T-H-W"),        # cfg2's shape: 8 192 rows, 128-token groups, one workgroup per CU
])
def test_fused_tail_frames_and_encoding(dev, D, res, B, order, axes):
    """tante_head_enc_fused against the launches it replaces.
    (1) the predicted frame is BIT-IDENTICAL to tante_head_fused_multi_streams' (same head arithmetic, same sum order);
    (2) the encoding z of that frame against the encoder launches (stage-1 GEMM + enc23_kernel) on the stored frame: the same bf16
        operands and fp32 accumulation, but the k orders of stages 1 and 3 differ (stage 3 runs tap by tap over K = 512) -- fp32
        rounding of the sum order, and a bf16 re-rounding of an intermediate where that flips it: 2e-4;
    (3) z against the ORACLE's encoder (fp32 CPU, enc_dec_cnn.py:217-229) on the same frame at the bf16 bar;
    (4) two launches give the same bits (stage 3 is one accumulation chain in a fixed tap order whatever the arrival order of the
        four pixel workgroups that hand their stage-2 outputs to the last one)."""
    import tante_amd
    from tante_amd import tante as TT
    from oracle import tante_oracle as O
    m = _model(dev, D, res, order, axes, seed=D)
    assert m.tail_fused_supported()
    HW = m.H_p * m.W_p
    x = torch.randn(B, 4, D, *res, generator=torch.Generator().manual_seed(D + 1)).to(dev)
    with torch.no_grad():
        z_new = torch.full((B, HW, 256), float("nan"), device=dev)
        y = m(x, enc_next=z_new)
        z_again = torch.full((B, HW, 256), float("nan"), device=dev)
        y_again = m(x, enc_next=z_again)
        y_noenc = m(x)                                   # the same launch without the encoding half
        saved = TT.HEAD_ENC
        try:
            TT.HEAD_ENC = False
            y_old = m(x)                                 # round 3's launch (order >= 2: tante_head_fused_multi_streams; order 1: tante_head_fused)
        finally:
            TT.HEAD_ENC = saved
        z_old = torch.empty(B, HW, 256, device=dev)
        m.encode_frame(y_old, z_old)
    assert torch.isfinite(y).all() and torch.isfinite(z_new).all()
    if order >= 2:
        assert torch.equal(y, y_old), "fused tail: the predicted frame differs from the head launch it replaces"
    else:      # order 1 ran tante_head_fused, whose epilogue is ONE fma (last + c d); the fused tail rounds c d first: fp32 rounding apart
        d, d_old = (y - x[:, -1:]).cpu(), (y_old - x[:, -1:]).cpu()
        r = rel_err(d, d_old)
        record_parity(r, max_rel(d, d_old), 1e-6, "fp32", f"fused tail vs tante_head_fused, order 1, derivative part, D={D}")
        assert r < 1e-6, r
    assert torch.equal(y, y_noenc) and torch.equal(y, y_again)
    assert torch.equal(z_new, z_again), "fused tail: the encoding is not reproducible"
    r, mx = rel_err(z_new, z_old), max_rel(z_new, z_old)
    record_parity(r, mx, 2e-4, "bf16", f"fused tail encoding vs encoder launches, D={D} order {order} B={B}")
    assert r < 2e-4 and mx < 2e-3, (r, mx)
    w = {k[len("encoder."):]: v.detach().cpu() for k, v in m.state_dict().items() if k.startswith("encoder.")}
    ref = O.enc_cnn(w, y.cpu(), 8)                       # (B, 1, Hp, Wp, C)
    close(z_new.view(B, 1, m.H_p, m.W_p, 256), ref, "bf16", f"fused tail encoding vs oracle encoder, D={D}")


def test_cfg2_rollout_fused_tail_against_plain_loop(dev, monkeypatch):
    """bench.py's rollout with the fused tail (default) against the same rollout with the predicted frames re-encoded by the encoder
    launches (TANTE_NO_TAIL_ENC): step 1 is bit-identical (its window holds input frames only); later steps see encodings that differ by
    fp32 sum order, and every bf16 rounding inside the nine blocks that such a difference flips is a 4e-3 step of that element: the two
    rollouts are two bf16 evaluations of the same function, 1.6e-3 apart on the DERIVATIVE part (prediction minus its last input frame)
    where each is 3.3e-3 from the oracle -- held to the bf16 bar, 1e-2."""
    import tante_amd
    torch.manual_seed(211)
    md = tante_amd.TanteMetadata(n_fields=11, spatial_resolution=(256, 256))
    m = tante_amd.TANTE(in_T=4, dset_metadata=md, n_head=8, mlp_ratio=1.0, dropout=0.1, embed_dim=256, patch_scale=8, taylor_order=3,
                        attn_axes="THW-THW-THW").to(dev).eval().set_compute("bf16")
    assert m.tail_fused_supported() and m.enc_cache_supported()
    fmt = tante_amd.DefaultChannelsFirstFormatter(md)
    g = torch.Generator().manual_seed(5)
    batch = {"input": torch.randn(2, 4, 256, 256, 11, generator=g).to(dev), "output": torch.randn(2, 5, 256, 256, 11, generator=g).to(dev)}
    with torch.no_grad():
        y_tail, _ = tante_amd.rollout_model(m, batch, fmt, 5)
        y_tail2, _ = tante_amd.rollout_model(m, batch, fmt, 5)
        monkeypatch.setattr(tante_amd.rollout, "NO_TAIL_ENC", True)
        y_plain, _ = tante_amd.rollout_model(m, batch, fmt, 5)
    assert torch.isfinite(y_tail).all()
    assert torch.equal(y_tail, y_tail2), "the fused-tail rollout is not reproducible"
    assert torch.equal(y_tail[:, 0], y_plain[:, 0])
    prev = torch.cat([batch["input"][:, -1:], y_plain[:, :-1]], dim=1)
    for t in range(1, 5):
        d, dref = (y_tail[:, t] - prev[:, t]).cpu(), (y_plain[:, t] - prev[:, t]).cpu()
        r = rel_err(d, dref)
        record_parity(r, max_rel(d, dref), 1e-2, "bf16", f"fused-tail rollout vs re-encoded rollout, step {t + 1}, derivative part")
        assert r < 1e-2, (t, r)


# ---- parity hardening (round-3 verdict, item 4) ---------------------------------------------------------------------------------------
def test_cfg4_shipped_model_query_subset_against_oracle(dev):
    """configs/cvit_rb.yaml AS SHIPPED (width 512, depth 10, eps 1e5, 128 x 128 latent grid, 512 x 128 x 4 fields), B = 1, 2 048 random
    query points, against oracle.cvit_forward (models/cvit.py:427-466 restated; its (2 048, 16 384, 2) fp32 temporary is 268 MB) in
    both compute modes.  The full 65 536-query grid stays with the size-independent properties of test_cvit_cfg4_full_size_properties."""
    import tante_amd
    from oracle import cvit_oracle as OC
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = tante_amd.load_config(os.path.join(root, "configs", "cvit_rb.yaml"))
    wl = cfg["workload"]
    H, W = wl["spatial_resolution"]
    md = tante_amd.TanteMetadata(n_fields=wl["n_fields"], spatial_resolution=(H, W))
    torch.manual_seed(211)
    m = tante_amd.build_model(cfg, md).to(dev).eval()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 4, wl["n_fields"], H, W, generator=g)
    coords = torch.rand(2048, 2, generator=g)
    mk = {k: v for k, v in cfg["model"].items() if k not in ("_target_", "in_T")}
    mk["grid_size"] = tuple(mk["grid_size"])
    ocfg = OC.CvitCfg(cfg["model"]["in_T"], wl["n_fields"], (H, W), **mk)
    w = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        ref = OC.cvit_forward(w, ocfg, x, coords)                    # (1, 4, 2048, 4)
        for mode in ("fp32", "bf16"):
            y = m.set_compute(mode)(x.to(dev), coords.to(dev))
            assert y.shape == ref.shape
            close(y, ref, mode, f"cfg4 as shipped, 2 048 query points, {mode}")
            # the formatter's channels-first VIEW of a channels-last batch is gathered in place (no input copy): the same bits
            x_cl = x.permute(0, 1, 3, 4, 2).contiguous().to(dev).permute(0, 1, 4, 2, 3)
            assert not x_cl.is_contiguous()
            assert torch.equal(m(x_cl, coords.to(dev)), y), f"cfg4 {mode}: channels-last view input differs from the contiguous input"


def test_cfg5_one_sample_full_size_against_oracle(dev):
    """configs/tante_fno.yaml at its full size (512 x 512 x 8 fields, modes 20 x 20), ONE sample, one model call, both compute modes,
    against oracle.tante_forward with the spectral encoder / decoder (models/enc_dec_fno.py:184-323 restated)."""
    import tante_amd
    from oracle import tante_oracle as O
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = tante_amd.load_config(os.path.join(root, "configs", "tante_fno.yaml"))
    wl, mk = cfg["workload"], cfg["model"]
    res = tuple(wl["spatial_resolution"])
    md = tante_amd.TanteMetadata(n_fields=wl["n_fields"], spatial_resolution=res)
    torch.manual_seed(211)
    m = tante_amd.build_model(cfg, md).to(dev).eval()
    ocfg = O.TanteCfg(mk["in_T"], wl["n_fields"], res, taylor_order=mk.get("taylor_order", 1), frame_interval=mk.get("frame_interval", 1.0),
                      attn_axes=mk.get("attn_axes", "THWTHWTHW"), n_head=mk.get("n_head", 8), mlp_ratio=mk.get("mlp_ratio", 1.0),
                      embed_dim=mk.get("embed_dim", 256), patch_scale=mk.get("patch_scale", 32), enc_dec_type="fno",
                      modes1=mk.get("modes1", 32), modes2=mk.get("modes2", 32))
    w = {k: (v.detach().cpu() if v.is_complex() else v.detach().float().cpu()) for k, v in m.state_dict().items()}
    x = torch.randn(1, mk["in_T"], wl["n_fields"], *res, generator=torch.Generator().manual_seed(55))
    O.set_fast(True)
    try:
        with torch.no_grad():
            ref = O.tante_forward(w, ocfg, x)
    finally:
        O.set_fast(False)
    last = x[:, -1:]
    for mode in ("fp32", "bf16"):
        with torch.no_grad():
            y = m.set_compute(mode)(x.to(dev)).cpu()
        close(y, ref, mode, f"cfg5 full size, one sample, {mode}")
        d, dref = y - last, ref - last
        r = rel_err(d, dref)
        bar = 5e-5 if mode == "fp32" else 1e-2
        record_parity(r, max_rel(d, dref), bar, mode, f"cfg5 full size, derivative part, {mode}")
        assert r < bar, (mode, r)


@pytest.mark.parametrize("B,T,H,W", [(2, 4, 8, 8), (1, 4, 6, 10), (8, 4, 32, 32)])
def test_temporal_propagator_fused_against_oracle(dev, B, T, H, W):
    """The T-letter launch that carries the temporal propagator (tante_block_fused_tprop) against the ORACLE -- not against the
    library's own separate launch: Attn_Backbone("THW") with the temporal propagator's weights scaled x 3 (the default initialisation
    is small), bf16 fused path vs oracle.attn_backbone (attn_backbone.py:134-191 restated) on the whole stream and on the UPDATE
    x_out - x_in, both at the bf16 bar."""
    import tante_amd
    from tante_amd import _lib as L
    from oracle import tante_oracle as O
    torch.manual_seed(B * 100 + H + 7)
    bb = tante_amd.Attn_Backbone(tensor_shape=(T, H, W, 256), attn_axes="THW", n_head=8, mlp_ratio=1.0, dropout=0.0).to(dev).eval()
    with torch.no_grad():
        for p in bb.temporal_propagator.parameters():
            p.mul_(3.0)
    assert bb.blocks[0].takes_tprop(T, L.BF16)
    x0 = torch.randn(B, T, H, W, 256, generator=torch.Generator().manual_seed(3))
    w = {k: v.detach().cpu() for k, v in bb.state_dict().items()}
    with torch.no_grad():
        ref = O.attn_backbone(w, x0, "THW", 8)
        x = x0.to(dev).clone()
        bb.forward_tokens(x, B, L.BF16)
    close(x.view_as(ref), ref, "bf16", "fused temporal propagator + THW blocks vs oracle, stream")
    close(x.view_as(ref).cpu() - x0, ref - x0, "bf16", "fused temporal propagator + THW blocks vs oracle, update x_out - x_in")


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_cfg2_rollout_per_step_derivative_parts(dev, mode):
    """bench.py's workload, ONE sample, 8 re-fed steps: for EVERY step the derivative part -- the frame a call adds to its own last
    input frame, y_t - y_(t-1), the quantity the network computes -- against the oracle's ref_t - ref_(t-1): fp32 5e-5 (the frame's
    fp32 rounding is ~1e-2 of a derivative's), bf16 1e-2.  (The whole-frame bars are test_cfg2_rollout_eight_steps_against_oracle's.)"""
    import tante_amd
    from oracle import tante_oracle as O
    torch.manual_seed(211)
    md = tante_amd.TanteMetadata(n_fields=11, spatial_resolution=(256, 256))
    m = tante_amd.TANTE(in_T=4, dset_metadata=md, n_head=8, mlp_ratio=1.0, dropout=0.1, embed_dim=256, patch_scale=8, taylor_order=3,
                        attn_axes="THW-THW-THW").to(dev).eval()
    w = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    cfg = O.TanteCfg(4, 11, (256, 256), taylor_order=3, attn_axes="THW-THW-THW", n_head=8, embed_dim=256, patch_scale=8)
    g = torch.Generator().manual_seed(2110)
    batch = {"input": torch.randn(1, 4, 256, 256, 11, generator=g), "output": torch.randn(1, 8, 256, 256, 11, generator=g)}
    O.set_fast(True)
    try:
        with torch.no_grad():
            ref, _ = O.rollout(w, cfg, batch, 8)
    finally:
        O.set_fast(False)
    fmt = tante_amd.DefaultChannelsFirstFormatter(md)
    with torch.no_grad():
        y, _ = tante_amd.rollout_model(m.set_compute(mode), {k: v.to(dev) for k, v in batch.items()}, fmt, 8)
    y = y.cpu()
    prev_y = torch.cat([batch["input"][:, -1:], y[:, :-1]], dim=1)
    prev_r = torch.cat([batch["input"][:, -1:], ref[:, :-1]], dim=1)
    bar = 5e-5 if mode == "fp32" else 1e-2
    for t in range(8):
        d, dref = y[:, t] - prev_y[:, t], ref[:, t] - prev_r[:, t]
        r = rel_err(d, dref)
        record_parity(r, max_rel(d, dref), bar, mode, f"cfg2 rollout step {t + 1}, derivative part")
        assert r < bar, (mode, t, r)


def test_rccl_world1_all_reduce_beside_the_captured_graph(request):
    """RCCL on the one GPU of a test box (tests/rccl_world1_child.py, started by conftest.py before this process touched the GPU):
    init_process_group("nccl", world_size=1), the bucket all-reduce (bit-exact identity over one rank), train_step with the collective
    forced on, and GraphedTrainStep with the all-reduce beside the thread-local-captured graph -- what an 8-GPU driver run would
    otherwise be the first to exercise (data/datamodule.py:96-119, trainer/trainer.py:193)."""
    import json
    child = getattr(request.config, "_rccl_child", None)
    if child is None:
        pytest.skip("the RCCL child is only started for `-m gpu` sessions on a GPU box")
    proc, log = child
    rc = proc.wait(timeout=900)
    text = open(log).read()
    verdict = [ln for ln in text.splitlines() if ln.startswith("VERDICT ")]
    assert rc == 0 and verdict, f"RCCL child failed (rc {rc}):\n{text[-3000:]}"
    v = json.loads(verdict[-1][len("VERDICT "):])
    assert v["ok"], v
    assert "RCCL" in (v["collective"] or ""), v
    record_parity(v["checks"]["graph_replay_plus_all_reduce_vs_eager_twin"]["worst_rel"], 0.0, 1e-3, "bf16",
                  "graph replay + RCCL all-reduce (world 1) vs eager twin: loss and flat gradient")


# ---- boundary completeness (round-3 verdict, item 9 / "what's missing" 4 and 5) -------------------------------------------------------
def _grads_against_oracle(mod, oracle_fn, inputs, mode, note, grad_tol=None):
    """Run mod(*inputs) under autograd on the GPU and oracle_fn(w, *inputs) under torch autograd on the CPU with the same weights and
    the same random cotangent; compare the output, every parameter's gradient and the first input's gradient."""
    import contextlib
    dev = next(mod.parameters()).device
    w = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in mod.state_dict().items() if v.is_floating_point()}
    xc = [t.detach().clone().requires_grad_(True) if (torch.is_tensor(t) and t.is_floating_point()) else t for t in inputs]
    ref = oracle_fn(w, *xc)
    r = torch.randn(ref.shape, generator=torch.Generator().manual_seed(99))
    (ref * r).sum().backward()
    xg = [t.detach().to(dev).clone().requires_grad_(True) if (torch.is_tensor(t) and t.is_floating_point()) else t for t in inputs]
    ctx = torch.autocast("cuda", dtype=torch.bfloat16) if mode == "bf16" else contextlib.nullcontext()
    for p in mod.parameters():
        p.grad = None
    with ctx:
        y = mod(*xg)
    (y.float() * r.to(dev)).sum().backward()
    close(y, ref, mode, f"{note}: output")
    gt = grad_tol or (2e-4 if mode == "fp32" else 4e-2)
    worst = 0.0
    for name, p in mod.named_parameters():
        if name not in w or w[name].grad is None:
            continue
        assert p.grad is not None, f"{note}: no gradient for {name}"
        e = rel_err(p.grad.cpu(), w[name].grad)
        worst = max(worst, e)
        assert e < gt, f"{note}: grad {name}: {e:.3e} (tol {gt:.0e})"
    e = rel_err(xg[0].grad.cpu(), xc[0].grad)
    worst = max(worst, e)
    assert e < gt, f"{note}: input gradient: {e:.3e}"
    record_parity(worst, worst, gt, mode, f"{note}: worst gradient (parameters and input)")


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_submodules_are_differentiable_standalone(dev, mode):
    """The reference's sub-modules are differentiable wherever they are called (attn_backbone.py:59-83, 134-191;
    enc_dec_cnn.py:217-229, 263-277; tante.py:191-201, 218-230).  Round 3 raised under autograd for everything but TANTE / CViT / FNO
    .forward: now TransformerBlock, Attn_Backbone, enc_CNN, dec_CNN, film (5-D and 3-D) and interprator take the differentiable HIP
    path on their own -- output, every parameter's gradient and the input's gradient against torch autograd through the CPU oracle."""
    import tante_amd
    from oracle import tante_oracle as O
    g = torch.Generator().manual_seed(31)
    torch.manual_seed(31)
    blk = tante_amd.TransformerBlock(256, 8, mlp_ratio=1.0, dropout=0.0).to(dev)
    x = torch.randn(6, 16, 256, generator=g)
    _grads_against_oracle(blk, lambda w, a: O.transformer_block(w, a, 8, False), [x], mode, "TransformerBlock")
    bb = tante_amd.Attn_Backbone(tensor_shape=(4, 16, 16, 256), attn_axes="THW", n_head=8, mlp_ratio=1.0, dropout=0.0).to(dev)
    xb = torch.randn(2, 4, 16, 16, 256, generator=g)
    _grads_against_oracle(bb, lambda w, a: O.attn_backbone(w, a, "THW", 8), [xb], mode, "Attn_Backbone THW")
    md = tante_amd.TanteMetadata(n_fields=3, spatial_resolution=(32, 48))
    enc = tante_amd.enc_CNN(dset_metadata=md, embed_dim=64, patch_scale=8, overlap_ratio=0.0).to(dev)
    xi = torch.randn(2, 3, 3, 32, 48, generator=g)
    _grads_against_oracle(enc, lambda w, a: O.enc_cnn(w, a, 8), [xi], mode, "enc_CNN")
    dec = tante_amd.dec_CNN(dset_metadata=md, embed_dim=64, patch_scale=8, overlap_ratio=0.0).to(dev)
    xt = torch.randn(2, 2, 4, 6, 64, generator=g)
    _grads_against_oracle(dec, lambda w, a: O.dec_cnn(w, a, 8), [xt], mode, "dec_CNN")
    if mode == "fp32":
        fl = tante_amd.film(64, in_dim=1).to(dev)
        _grads_against_oracle(fl, O.film, [torch.randn(2, 4, 3, 5, 64, generator=g), torch.tensor([-2.0, -1.0, 0.0, 0.0])], mode, "film 5-D")
        _grads_against_oracle(fl, O.film, [torch.randn(3, 10, 64, generator=g), torch.tensor([1.1, 1.4, 1.25])], mode, "film 3-D")
        it = tante_amd.interprator(64, sp_dim=12).to(dev)
        _grads_against_oracle(it, lambda w, a, oT: O.interprator(w, a, oT), [torch.randn(3, 12, 64, generator=g), 6.0], mode, "interprator")


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_transformer_block_masks(dev, mode):
    """TransformerBlock.forward(x, key_padding_mask, attn_mask, causal) -- the reference's documented signature (attn_backbone.py:59-72;
    the TANTE path passes `causal` only) -- against torch's own nn.MultiheadAttention on the CPU with the same parameters: bool and
    additive float attn_mask, (L, L) and per-(batch x head) masks, bool key_padding_mask, and causal combined with padding."""
    import copy
    import tante_amd
    torch.manual_seed(17)
    blk = tante_amd.TransformerBlock(64, 4, mlp_ratio=2.0, dropout=0.0).to(dev).eval()
    blk.compute = mode
    ref_blk = copy.deepcopy(blk).cpu()
    Bp, Lq = 3, 10
    g = torch.Generator().manual_seed(5)
    x = torch.randn(Bp, Lq, 64, generator=g)

    def reference(kp, am, causal):
        with torch.no_grad():
            h = ref_blk.ln1(x)
            mask = am
            if causal:
                tri = torch.triu(torch.ones(Lq, Lq, dtype=torch.bool), diagonal=1)
                mask = tri if am is None else (am | tri if am.dtype == torch.bool else am.masked_fill(tri, float("-inf")))
            y, _ = ref_blk.attn(h, h, h, key_padding_mask=kp, attn_mask=mask, need_weights=False)
            x1 = x + y
            return x1 + ref_blk.mlp(ref_blk.ln2(x1))
    am_bool = torch.rand(Lq, Lq, generator=g) < 0.3
    am_bool.fill_diagonal_(False)                          # every query keeps itself: no fully blocked row
    am_float = torch.randn(Lq, Lq, generator=g)
    am_heads = torch.randn(Bp * 4, Lq, Lq, generator=g)
    kp = torch.zeros(Bp, Lq, dtype=torch.bool)
    kp[0, 7:] = True
    kp[2, 0] = True
    cases = [("bool attn_mask", None, am_bool, False), ("float attn_mask", None, am_float, False), ("per-head float attn_mask", None, am_heads, False),
             ("key_padding_mask", kp, None, False), ("key_padding_mask + bool attn_mask", kp, am_bool, False)]
    kp_c = torch.zeros(Bp, Lq, dtype=torch.bool)
    kp_c[1, 6:] = True
    cases.append(("causal + key_padding_mask", kp_c, None, True))
    for name, kpm, am, causal in cases:
        with torch.no_grad():
            y = blk(x.to(dev), None if kpm is None else kpm.to(dev), None if am is None else am.to(dev), causal)
        close(y, reference(kpm, am, causal), mode, f"TransformerBlock masks: {name}")


def test_fp16_autocast_and_grad_scaler(dev):
    """The reference Trainer's DEFAULT mixed-precision setting: amp_type "float16" with a GradScaler (trainer/trainer.py:86-104, 183-196).
    torch.autocast(float16) selects the 16-bit MFMA path (bf16 operands: tante_amd/attn_backbone.py) and torch.amp.GradScaler drives
    FlatAdamW through its param_groups: (1) two scaled steps land where two unscaled bf16-autocast steps land -- the scale is a power of
    two; what differs is torch's own FiLM-table MLPs, which autocast runs in fp16 here and in bf16 there: the bf16 gradient bar 4e-2;
    (2) a non-finite gradient makes the scaler skip the step and halve its scale, the parameters untouched."""
    import copy
    import warnings
    import tante_amd
    from tante_amd import autograd as A
    from tante_amd.train import train_step
    torch.manual_seed(3)
    md = tante_amd.TanteMetadata(n_fields=2, spatial_resolution=(64, 64))
    m1 = tante_amd.TANTE(in_T=4, dset_metadata=md, taylor_order=1, attn_axes="THW", n_head=8, embed_dim=256, patch_scale=8, dropout=0.0).to(dev).train()
    m2 = copy.deepcopy(m1)
    fmt = tante_amd.DefaultChannelsFirstFormatter(md)
    g = torch.Generator().manual_seed(8)
    b = {"input": torch.randn(2, 4, 64, 64, 2, generator=g).to(dev), "output": torch.randn(2, 2, 64, 64, 2, generator=g).to(dev)}
    o1 = tante_amd.FlatAdamW(m1.parameters(), lr=1e-4, weight_decay=0.01, max_norm=1.0)
    o2 = tante_amd.FlatAdamW(m2.parameters(), lr=1e-4, weight_decay=0.01, max_norm=1.0)
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 12, enabled=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for step in range(2):
            A._SEED[0] = 50 + step
            with torch.autocast("cuda", dtype=torch.float16):
                l1 = float(train_step(m1, o1, b, fmt, 2, 1, scaler=scaler))
            A._SEED[0] = 50 + step
            with torch.autocast("cuda", dtype=torch.bfloat16):
                l2 = float(train_step(m2, o2, b, fmt, 2, 1))
            assert abs(l1 - l2) < 2e-3 * abs(l2), (step, l1, l2)
            eg = float((o1.flat_g - o2.flat_g).norm() / o2.flat_g.norm())      # o1's bucket is unscaled in place by scaler.unscale_
            record_parity(eg, eg, 4e-2, "bf16", f"fp16 autocast + GradScaler vs bf16 autocast, step {step + 1}: flat gradient")
            assert eg < 4e-2, (step, eg)
        assert o1.step_count == 2 and scaler.get_scale() == 2.0 ** 12
        # (2) a non-finite gradient: the step is skipped, the scale backs off
        p_before = o1.flat_p.clone()
        o1.zero_grad()
        with torch.autocast("cuda", dtype=torch.float16):
            y_pred, y_ref = tante_amd.rollout_model(m1, b, fmt, 2)
            loss = tante_amd.autograd.MseMeanFn.apply(y_pred, y_ref)
        A.run_backward(scaler.scale(loss))
        o1.flat_g[123] = float("inf")
        scaler.unscale_(o1)
        scaler.step(o1)
        scaler.update()
        assert torch.equal(o1.flat_p, p_before) and o1.step_count == 2
        assert scaler.get_scale() == 2.0 ** 11


@pytest.mark.parametrize("grad", [False, True])
def test_adaptive_rollout_batched_with_per_sample_frame_counts(dev, grad):
    """R_Trainer's per-sample loop for an out_T where the samples advance at DIFFERENT rates (floor(R_t[i]) frames per call,
    r_trainer.py:112-133, tante.py:163) against the batched form that keeps them in one batch (rollout._rollout_adaptive_batched):
    the same frames, the same R_t in the same order -- without autograd and, with it, the same parameter gradients."""
    import tante_amd
    from test_hip_parity import _tante_from
    from conftest import load_golden
    g = load_golden("g13_deg_false")
    md = tante_amd.TanteMetadata(n_fields=1, spatial_resolution=(32, 32))
    m = _tante_from(g, dev, in_T=4, dset_metadata=md, taylor_order=2, attn_axes="TH-TW", n_head=2, embed_dim=32, patch_scale=8, dropout=0.0,
                    deg=False)
    with torch.no_grad():      # step sizes that depend on the sample: a steep last layer around the middle of the clamp range
        for it in m.interprators:
            it.interprete[4].weight.mul_(60.0)
            it.interprete[4].bias.add_(2.2)
    fmt = tante_amd.DefaultChannelsFirstFormatter(md)
    gen = torch.Generator().manual_seed(11)
    amp = torch.tensor([0.05, 0.4, 1.0, 2.0, 5.0, 12.0]).view(6, 1, 1, 1, 1)
    batch = {"input": (torch.randn(6, 4, 32, 32, 1, generator=gen) * amp).to(dev), "output": torch.randn(6, 7, 32, 32, 1, generator=gen).to(dev)}
    if grad:
        m.train()
        for p in m.parameters():
            p.grad = None
        y_b, _, rt_b = tante_amd.rollout_adaptive(m, batch, fmt, 7, 6.0, per_sample=True)
        (y_b.square().mean() + rt_b.mean()).backward()
        gb = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
        for p in m.parameters():
            p.grad = None
        y_s, _, rt_s = tante_amd.rollout_adaptive(m, batch, fmt, 7, 6.0, per_sample=True, batch_when_equivalent=False)
        (y_s.square().mean() + rt_s.mean()).backward()
        worst = 0.0
        for n, p in m.named_parameters():
            if p.grad is None:
                continue
            e = rel_err(gb[n], p.grad)
            worst = max(worst, e)
            assert e < 2e-4, (n, e)
        record_parity(worst, worst, 2e-4, "fp32", "batched adaptive rollout with per-sample counts vs serial loop: worst parameter gradient")
    else:
        m.eval()
        with torch.no_grad():
            y_b, _, rt_b = tante_amd.rollout_adaptive(m, batch, fmt, 7, 6.0, per_sample=True)
            y_s, _, rt_s = tante_amd.rollout_adaptive(m, batch, fmt, 7, 6.0, per_sample=True, batch_when_equivalent=False)
    assert y_b.shape == y_s.shape == (6, 7, 32, 32, 1)
    assert rt_b.shape == rt_s.shape
    counts = torch.floor(rt_s.detach()).cpu()
    assert counts.min() >= 1 and len(set(counts.tolist())) > 1, f"the fixture must give the samples different frame counts: {counts.tolist()}"
    close(y_b, y_s.detach().cpu(), "fp32", "batched adaptive rollout (per-sample counts) vs serial loop: frames")
    close(rt_b, rt_s.detach().cpu(), "fp32", "batched adaptive rollout (per-sample counts) vs serial loop: R_t, in order")


@pytest.mark.parametrize("T,C_", [(4, 256), (8, 64), (1, 32)])
def test_film_table_fn_gradients(dev, T, C_):
    """autograd.FilmTableFn (the FiLM tables of a rollout and their backward as two HIP launches, tante.py:203-230) against torch autograd
    through the same two MLPs: the tables, and every parameter's gradient from accumulated table gradients."""
    import tante_amd
    from tante_amd.autograd import FilmTableFn
    torch.manual_seed(T * 10 + C_)
    fl = tante_amd.film(C_, in_dim=1).to(dev)
    sc, sh = fl.condition_to_scale, fl.condition_to_shift
    t = torch.linspace(-2.0, 0.5, T, device=dev)
    add = torch.randn(T, C_, device=dev, requires_grad=True)
    ga, gb = torch.randn(T, C_, device=dev), torch.randn(T, C_, device=dev)
    a_ref = 1.0 + sc(t[:, None])
    b_ref = sh(t[:, None]) + add
    (a_ref * ga).sum().backward(retain_graph=True)
    (b_ref * gb).sum().backward()
    ref = {n: p.grad.clone() for n, p in fl.named_parameters()}
    ref_add = add.grad.clone()
    for p in fl.parameters():
        p.grad = None
    add.grad = None
    a, b, acc = FilmTableFn.apply(t, add, sc[0].weight, sc[0].bias, sc[2].weight, sc[2].bias, sh[0].weight, sh[0].bias, sh[2].weight, sh[2].bias)
    close(a, a_ref, "fp32", "FilmTableFn: scale table")
    close(b, b_ref, "fp32", "FilmTableFn: shift table")
    with torch.no_grad():      # the uses' contributions, as FilmPos*Fn adds them: two halves
        acc[0] += 0.25 * ga; acc[0] += 0.75 * ga
        acc[1] += gb
    (a.sum() * 0.0 + b.sum() * 0.0).backward()      # nothing arrives through autograd but zeros: the accumulators carry the gradient
    for n, p in fl.named_parameters():
        e = rel_err(p.grad, ref[n])
        assert e < 1e-5, (n, e)
    assert rel_err(add.grad, ref_add) < 1e-5


@pytest.mark.parametrize("B,res,axes", [(1, (256, 256), "THW"), (2, (256, 256), "THW"), (3, (128, 256), "TW"), (1, (128, 128), "HT")])
def test_half_size_block_workgroups_bit_identical(dev, B, res, axes):
    """Small batches run the fused block kernel on 32-token workgroups (block_fs_kernel<TPS, 2, 4>: twice the workgroups, half the serial
    chain each) where that needs fewer resident rounds; TANTE_FS_HALF = 0 keeps the 64-token form.  A token's arithmetic does not depend
    on the tiling: the two must agree bit for bit (both are held to the oracle by test_fused_block / the G fixtures)."""
    from tante_amd import _lib as L
    m = _model(dev, 11, res, 1, axes, seed=7)
    x = torch.randn(B, 4, 11, *res, generator=torch.Generator().manual_seed(B)).to(dev)
    try:
        with torch.no_grad():
            L.set_option("TANTE_FS_HALF", 0)
            y_full = m(x).clone()
            L.set_option("TANTE_FS_HALF", 1)
            y_half = m(x).clone()
    finally:
        L.set_option("TANTE_FS_HALF", 1)
    assert torch.isfinite(y_half).all()
    assert torch.equal(y_full, y_half), "half-size block workgroups changed the result"


def test_graphed_rollout_equals_eager_and_follows_weight_updates(dev):
    """tante_amd.GraphedRollout: the rollout as one captured HIP graph gives the eager rollout's bits, on the batch it was captured with
    and on a new one; after a parameter changed the graph is re-captured (its packed weights are referenced by address)."""
    import tante_amd
    torch.manual_seed(3)
    md = tante_amd.TanteMetadata(n_fields=4, spatial_resolution=(64, 96))
    m = tante_amd.TANTE(in_T=4, dset_metadata=md, n_head=8, embed_dim=256, patch_scale=8, taylor_order=2, attn_axes="TH-W",
                        dropout=0.0).to(dev).eval().set_compute("bf16")
    fmt = tante_amd.DefaultChannelsFirstFormatter(md)
    g = torch.Generator().manual_seed(9)

    def mk():
        return {"input": torch.randn(2, 4, 64, 96, 4, generator=g).to(dev), "output": torch.randn(2, 5, 64, 96, 4, generator=g).to(dev)}
    b1, b2 = mk(), mk()
    b2["output"][0, 0, 0, 0, 0] = float("nan")          # the formatter's nan_to_num runs inside the graph too
    roll = tante_amd.GraphedRollout(m, b1, fmt, 5)
    with torch.no_grad():
        for b in (b1, b2, b1):
            y, ref = roll(b)
            y, ref = y.clone(), ref.clone()
            ye, refe = tante_amd.rollout_model(m, b, fmt, 5)
            assert torch.equal(y, ye) and torch.equal(ref, refe)
        with torch.no_grad():
            m.decoders[0].dec_conv_3.deconv.bias.add_(0.25)
        y = roll(b2)[0].clone()
        ye = tante_amd.rollout_model(m, b2, fmt, 5)[0]
        assert torch.equal(y, ye), "GraphedRollout did not follow a parameter update"
    with pytest.raises(ValueError):
        roll({"input": b1["input"][:1], "output": b1["output"][:1]})


def test_spectral_bf16_output_feeds_the_patch_gather_bit_identically(dev):
    """enc_FNO in bf16 mode: the first spectral layer writes its (GELU'd) image as bf16 and tante_im2col gathers the conv's patches from
    the bf16 image (tante_spectral_layer_bf16out) -- the rounding the gather did anyway, done once at the store: the encoder's tokens
    must be the same bits as with the fp32 image in between (TANTE_SPECTRAL_BF16OUT = 0), at half the traffic both ways."""
    import tante_amd
    from tante_amd import _lib as L
    torch.manual_seed(17)
    md = tante_amd.TanteMetadata(n_fields=8, spatial_resolution=(256, 128))
    m = tante_amd.TANTE(in_T=4, dset_metadata=md, n_head=8, embed_dim=256, patch_scale=8, taylor_order=1, attn_axes="THW", enc_dec_type="fno",
                        modes1=10, modes2=10, overlap_ratio=0.0, dropout=0.0).to(dev).eval().set_compute("bf16")
    x = torch.randn(2, 4, 8, 256, 128, generator=torch.Generator().manual_seed(2)).to(dev)
    assert L.lib().tante_spectral_bf16out_supported(8, 8, 32, 256, 128, 10, 10)
    try:
        with torch.no_grad():
            L.set_option("TANTE_SPECTRAL_BF16OUT", 0)
            y0 = m(x).clone()
            L.set_option("TANTE_SPECTRAL_BF16OUT", 1)
            y1 = m(x).clone()
    finally:
        L.set_option("TANTE_SPECTRAL_BF16OUT", 1)
    assert torch.isfinite(y1).all()
    assert torch.equal(y0, y1), "the bf16 image changed the encoder's result"
