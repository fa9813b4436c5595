"""GPU parity, round 3: the headline metric's own computation -- an 8-step re-fed rollout of the cfg2 model in bf16 AND fp32 -- against
the CPU oracle per step and at rollout end; the `models.FNO` surface under the rollout harness; the capture-refusal path of the graphed
train step.

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


def test_cfg2_rollout_eight_steps_against_oracle(dev):
    """bench.py's workload (configs/tante_am.yaml: order 3, THW-THW-THW, 256 x 256 x 11, 8 rollout steps), ONE sample, through
    rollout_model (fused kernels + frame-encoding cache in bf16; parity kernels in fp32) against oracle.rollout
    (trainer/evaler.py:121-138 restated): every frame and the whole rollout at the stated bars; the per-step errors land in
    parity_report.json.  The derivative part of step 1 (prediction minus last input frame) is held to the same bars."""
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
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    for mode in ("fp32", "bf16"):
        with torch.no_grad():
            y, _ = tante_amd.rollout_model(m.set_compute(mode), dbatch, fmt, 8)
        y = y.cpu()
        assert y.shape == ref.shape == (1, 8, 256, 256, 11)
        for t in range(8):
            close(y[:, t], ref[:, t], mode, f"cfg2 rollout step {t + 1}")
        close(y, ref, mode, "cfg2 rollout, all 8 frames")
        last = batch["input"][:, -1]
        d, dref = y[:, 0] - last, ref[:, 0] - last
        r = rel_err(d, dref)
        record_parity(r, max_rel(d, dref), 5e-5 if mode == "fp32" else 1e-2, mode, "cfg2 rollout step 1, derivative part")
        assert r < (5e-5 if mode == "fp32" else 1e-2), (mode, r)


# ---- models.FNO (the wrapper surface of models/fno.py:63-106; arithmetic inside parity-unpinned, see tante_amd/fno.py) ----
def _fno_small(dev):
    import tante_amd
    torch.manual_seed(5)
    md = tante_amd.TanteMetadata(n_fields=3, spatial_resolution=(40, 48))
    m = tante_amd.FNO(in_T=4, dset_metadata=md, modes1=6, modes2=5, hidden_channels=24).to(dev)
    w = {}
    for k, v in m.state_dict().items():
        w[k] = v.detach().cpu()
    return m, md, w


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_fno_wrapper_forward_and_rollout(dev, mode):
    """I/O contract b t c h w -> b 1 c h w, and a 4-step re-fed rollout through rollout_model, against the oracle's restatement."""
    import tante_amd
    from oracle import spectral_oracle as OS
    m, md, w = _fno_small(dev)
    m.eval()
    g = torch.Generator().manual_seed(6)
    batch = {"input": torch.randn(2, 4, 40, 48, 3, generator=g), "output": torch.randn(2, 4, 40, 48, 3, generator=g)}
    x = batch["input"].permute(0, 1, 4, 2, 3).contiguous()
    import contextlib
    amp = torch.autocast("cuda", dtype=torch.bfloat16) if mode == "bf16" else contextlib.nullcontext()
    with torch.no_grad(), amp:
        y = m(x.to(dev))
        assert y.shape == (2, 1, 3, 40, 48)
        close(y, OS.fno_wrapper(w, x, 6, 5), mode, "FNO one call")
        fmt = tante_amd.DefaultChannelsFirstFormatter(md)
        yr, y_ref = tante_amd.rollout_model(m, {k: v.to(dev) for k, v in batch.items()}, fmt, 4)
    assert yr.shape == (2, 4, 40, 48, 3)
    mv, preds = x, []
    with torch.no_grad():
        for _ in range(4):
            yo = OS.fno_wrapper(w, mv, 6, 5)
            preds.append(yo.permute(0, 1, 3, 4, 2))
            mv = torch.cat([mv[:, 1:], yo], dim=1)
    close(yr, torch.cat(preds, dim=1), mode, "FNO 4-step rollout", scale=2.0 if mode == "bf16" else 1.0)


def test_fno_wrapper_gradients_against_oracle_autograd(dev):
    from oracle import spectral_oracle as OS
    m, md, w = _fno_small(dev)
    m.train()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 4, 3, 40, 48, generator=g)
    proj = torch.randn(2, 1, 3, 40, 48, generator=g)
    y = m(x.to(dev))
    (y * proj.to(dev)).sum().backward()
    wr = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    (OS.fno_wrapper(wr, x, 6, 5) * proj).sum().backward()
    for k, p in m.named_parameters():
        gr = wr[k].grad
        gm = p.grad.detach().cpu()
        if gr.is_complex():
            gr, gm = torch.view_as_real(gr), torch.view_as_real(gm)
        r = rel_err(gm, gr)
        record_parity(r, max_rel(gm, gr), 2e-4, "fp32", "FNO grad " + k)
        assert r < 2e-4, (k, r)


def test_fno_reference_yaml_builds_and_rolls_out(dev):
    """configs/fno_vf.yaml carries the reference's configs/fno.yaml model block (l.21-26): build through the `_target_: models.FNO`
    alias and run a 4-step rollout at a reduced resolution (the full 512 x 512 case is bench.py --config fno_vf)."""
    import tante_amd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = tante_amd.load_config(os.path.join(root, "configs", "fno_vf.yaml"))
    md = tante_amd.TanteMetadata(n_fields=8, spatial_resolution=(64, 64))
    m = tante_amd.build_model(cfg, md).to(dev).eval()
    assert type(m).__name__ == "FNO" and m.n_modes == (20, 20) and m.hidden_channels == 48 and m.dim_in == 32 and m.dim_out == 8
    g = torch.Generator().manual_seed(8)
    batch = {"input": torch.randn(2, 4, 64, 64, 8, generator=g).to(dev), "output": torch.randn(2, 4, 64, 64, 8, generator=g).to(dev)}
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        y, y_ref = tante_amd.rollout_model(m, batch, tante_amd.DefaultChannelsFirstFormatter(md), 4)
    assert y.shape == (2, 4, 64, 64, 8) and torch.isfinite(y).all()


def test_refused_capture_leaves_no_uncomputed_packs(dev):
    """A GraphedTrainStep capture that is refused AFTER recording (blocks off the fused one-node path) has cached weight packs whose
    pack kernels were only recorded, never run.  The eager fallback step that follows must not read them: its loss, gradients and
    updated parameters must equal those of a twin that never attempted a capture (dropout 0: the same arithmetic either way)."""
    import copy
    import tante_amd
    from tante_amd import harness as H
    md = tante_amd.TanteMetadata(n_fields=2, spatial_resolution=(32, 32))
    torch.manual_seed(3)
    a = tante_amd.TANTE(in_T=4, dset_metadata=md, taylor_order=1, attn_axes="THW", n_head=2, embed_dim=32, patch_scale=8,
                        dropout=0.0).to(dev).train().set_compute("bf16")
    b = copy.deepcopy(a)
    oa, ob = tante_amd.FlatAdamW(a.parameters(), lr=1e-3), tante_amd.FlatAdamW(b.parameters(), lr=1e-3)
    gen = torch.Generator().manual_seed(5)
    loader = [{"input": torch.randn(2, 4, 32, 32, 2, generator=gen), "output": torch.randn(2, 2, 32, 32, 2, generator=gen)}]
    fmt = tante_amd.DefaultChannelsFirstFormatter(md)
    # poison the allocator's free memory so that a never-written pack buffer reads as NaN rather than as lucky zeros
    junk = torch.full((64 << 20,), float("nan"), device=dev)
    del junk
    la = H.train_one_epoch(a, oa, loader, fmt, 2, graph=True)
    assert a._tante_graphed_step is False                      # the capture was refused
    lb = H.train_one_epoch(b, ob, loader, fmt, 2, graph=False)
    assert la == la and abs(la - lb) <= 1e-6 * abs(lb), (la, lb)
    eg = float((oa.flat_g - ob.flat_g).norm() / ob.flat_g.norm())
    ep = float((oa.flat_p - ob.flat_p).norm() / ob.flat_p.norm())
    assert eg < 1e-5 and ep < 1e-6, (eg, ep)


@pytest.mark.parametrize("n", [1, 3, 4, 1023, 4096 * 11 + 2, 8 * 8 * 256 * 256 * 11 // 16])
def test_nan_to_num_kernel_equals_torch(dev, n):
    """tante_nan_to_num = torch.nan_to_num (data/datamodule.py:187) bit for bit: NaN -> 0, +-inf -> +-FLT_MAX, everything else untouched."""
    from tante_amd import rollout as R
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, generator=g)
    x[::7] = float("nan"); x[1::13] = float("inf"); x[2::17] = float("-inf"); x[3::19] = -0.0
    xd = x.to(dev)
    y = R._nan_to_num(xd)
    assert y.data_ptr() != xd.data_ptr()
    assert torch.equal(y.cpu(), torch.nan_to_num(x))
    assert torch.equal(y.cpu().view(torch.int32), torch.nan_to_num(x).view(torch.int32))      # signed zeros too


@pytest.mark.parametrize("letter,B,T,H,W", [("T", 8, 4, 32, 32), ("H", 8, 4, 32, 32), ("W", 2, 4, 16, 48), ("T", 1, 4, 5, 3), ("H", 3, 2, 16, 7),
                                            ("W", 1, 3, 7, 64), ("L", 1, 1, 6, 6)])
def test_block_kernel_paired_form_is_bit_identical(dev, letter, B, T, H, W):
    """The paired form of the fused block kernel (two 4-wave groups per workgroup, one barrier segment apart: TANTE_FS_GROUPS = 2) against
    the default unpaired form: the same instruction stream on the same tokens, so the outputs must be bitwise equal -- on full grids,
    on ragged ones (an odd number of 64-token groups: the last workgroup's second group has no live sequence) and for every layout
    class (several sequences per tile, tile-aligned, element-masked)."""
    import tante_amd
    from tante_amd import _lib as L, kernels as Kk
    torch.manual_seed(B * 100 + H + W)
    blk = tante_amd.TransformerBlock(256, 8, mlp_ratio=1.0, dropout=0.0).to(dev).eval()
    st = blk._packed_fused()
    seq = Kk.make_seq(letter, B, T, H, W)
    assert Kk.block_fused_supported(256, 8, 256, seq.L)
    x0 = torch.randn(B * T * H * W, 256, device=dev) * 1.3 + 0.2
    outs = []
    try:
        for groups in (1, 2):
            L.set_option("TANTE_FS_GROUPS", groups)
            y = x0.clone()
            Kk.block_fused(y, st, 256, 8, 256, seq, letter == "T", 1e-5)
            torch.cuda.synchronize()
            outs.append(y)
    finally:
        L.set_option("TANTE_FS_GROUPS", 0)
    assert torch.isfinite(outs[0]).all() and not torch.equal(outs[0], x0)
    assert torch.equal(outs[0], outs[1])


def test_set_option_rejects_bad_names(dev):
    from tante_amd import _lib as L
    assert L.lib().tante_set_option(b"FS_GROUPS", 1) != 0           # names start with TANTE_
    assert L.lib().tante_set_option(b"", 1) != 0
    L.set_option("TANTE_TEST_OPTION", 7)
    assert L.get_option("TANTE_TEST_OPTION", 0) == 7 and L.get_option("TANTE_NEVER_SET", 5) == 5


@pytest.mark.parametrize("n,cin,cout,H,W,m1,m2,act", [(2, 3, 5, 32, 64, 4, 5, 0), (1, 8, 32, 64, 96, 8, 20, 1), (3, 6, 40, 48, 32, 2, 3, 1),
                                                      (1, 64, 128, 128, 128, 5, 5, 1), (2, 8, 32, 512, 512, 20, 20, 1), (1, 32, 8, 512, 512, 20, 20, 0)])
def test_spectral_layer_truncated_dft_against_oracle_and_fft_path(dev, n, cin, cout, H, W, m1, m2, act):
    """tante_spectral_layer's truncated-DFT path (spectral_dft.hip: both transforms as skinny fp32 MFMA products over the kept modes, the
    1x1 conv in the output pass) against (a) the CPU oracle's rfft2 / irfft2 restatement of SpectralLayer.forward
    (models/enc_dec_fno.py:184-222) where that is affordable and (b) this library's own hipFFT path (TANTE_SPECTRAL_DFT = 0), at the fp32
    bar 1e-5 -- incl. the cfg5 shapes (512 x 512, modes 20 x 20, 8 -> 32 and 32 -> 8 channels; 128 x 128, modes 5 x 5, 64 -> 128)."""
    from oracle import spectral_oracle as OS
    from tante_amd import _lib as L, kernels as Kk
    g = torch.Generator().manual_seed(n * 1000 + H + W)
    x = torch.randn(n, cin, H, W, generator=g)
    wre, wim = torch.randn(cin, cout, m1, m2, generator=g) / (cin * cout) ** 0.5, torch.randn(cin, cout, m1, m2, generator=g) / (cin * cout) ** 0.5
    w0, b0 = torch.randn(cout, cin, generator=g) / cin ** 0.5, torch.randn(cout, generator=g)
    assert L.lib().tante_get_option(b"TANTE_SPECTRAL_DFT", 1) == 1
    args = (x.to(dev), wre.to(dev), wim.to(dev), m1, m2, w0.to(dev), b0.to(dev), act)
    y_dft = Kk.spectral_layer(*args)
    try:
        L.set_option("TANTE_SPECTRAL_DFT", 0)
        y_fft = Kk.spectral_layer(*args)
    finally:
        L.set_option("TANTE_SPECTRAL_DFT", 1)
    torch.cuda.synchronize()
    close(y_dft, y_fft, "fp32", "truncated DFT vs hipFFT path")
    if n * cout * H * W <= 4 << 20:
        w = {"weight": torch.complex(wre, wim), "w0.weight": w0.view(cout, cin, 1, 1), "w0.bias": b0}
        ref = OS.spectral_layer(w, x, m1, m2)
        if act == 1:
            ref = torch.nn.functional.gelu(ref)
        close(y_dft, ref, "fp32", "truncated DFT vs oracle")


@pytest.mark.parametrize("order,axes", [(2, "T-W"), (3, "T-H-W"), (3, "THW-THW-THW")])
def test_multi_order_head_launch_equals_per_order_launches(dev, order, axes):
    """tante_head_fused_multi (every Taylor order's derivative head in one launch, the frame read and written once) against one
    tante_head_fused launch per order (tante.py:145-154, 165-171): same heads, same coefficients, the sum over the orders taken in
    registers instead of through the frame -- agreement to fp32 rounding of the sum order; and against the oracle at the bf16 bar."""
    import tante_amd
    from tante_amd import tante as TT
    from oracle import tante_oracle as O
    torch.manual_seed(order * 7)
    md = tante_amd.TanteMetadata(n_fields=5, spatial_resolution=(64, 96))
    m = tante_amd.TANTE(in_T=4, dset_metadata=md, taylor_order=order, attn_axes=axes, n_head=8, embed_dim=256, patch_scale=8, frame_interval=0.5,
                        dropout=0.0).to(dev).eval().set_compute("bf16")
    x = torch.randn(3, 4, 5, 64, 96, generator=torch.Generator().manual_seed(1))
    outs = []
    saved = TT.HEAD_MULTI
    try:
        for multi in (True, False):
            TT.HEAD_MULTI = multi
            with torch.no_grad():
                outs.append(m(x.to(dev)).cpu())
    finally:
        TT.HEAD_MULTI = saved
    d0, d1 = outs[0] - x[:, -1:], outs[1] - x[:, -1:]
    r = rel_err(d0, d1)
    record_parity(r, max_rel(d0, d1), 1e-5, "fp32", f"multi-order head vs per-order launches, derivative part, order {order}")
    assert r < 1e-5, r
    cfg = O.TanteCfg(4, 5, (64, 96), taylor_order=order, attn_axes=axes, n_head=8, embed_dim=256, patch_scale=8, frame_interval=0.5)
    ref = O.tante_forward({k: v.detach().cpu() for k, v in m.state_dict().items()}, cfg, x[:1])
    close(outs[0][:1], ref, "bf16", f"multi-order head, order {order}, vs oracle")
    # every order's stream in a buffer of its own (tante_axis_hw_oop + tante_head_fused_multi_streams) against the row copies: bit for bit
    saved_s = TT.HEAD_STREAMS
    try:
        TT.HEAD_STREAMS = False
        with torch.no_grad():
            copies = m(x.to(dev)).cpu()
    finally:
        TT.HEAD_STREAMS = saved_s
    assert torch.equal(outs[0], copies)


@pytest.mark.parametrize("n,C,H,W,P,pad,dt", [(2, 32, 64, 96, 4, 1, "bf16"), (1, 128, 32, 32, 2, 0, "bf16"), (3, 5, 20, 36, 4, 1, "fp32"), (1, 8, 512, 512, 4, 1, "bf16"),
                                              (2, 7, 18, 10, 2, 1, "fp32"), (1, 300, 16, 16, 4, 3, "bf16")])
def test_im2col_tiled_nchw_equals_scalar_kernel(dev, n, C, H, W, P, pad, dt):
    """The tiled channels-first im2col (stride = kernel, padded: the spectral encoder's RealConv2d stages, enc_dec_cnn.py:49-110 as used by
    enc_dec_fno.py:224-273) against the scalar kernel it replaces (TANTE_IM2COL_TILED = 0): the same patch matrix, bit for bit -- incl. a
    width that is not a multiple of the tile, many channels (narrower tiles) and zero padding on every side."""
    from tante_amd import _lib as L, kernels as Kk
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(n, C, H, W, generator=g).to(dev)
    odt = torch.bfloat16 if dt == "bf16" else torch.float32
    outs = []
    try:
        for tiled in (1, 0):
            L.set_option("TANTE_IM2COL_TILED", tiled)
            outs.append(Kk.im2col(x, True, n, C, H, W, P, P, P, P, pad, pad, 0, odt))
    finally:
        L.set_option("TANTE_IM2COL_TILED", 1)
    torch.cuda.synchronize()
    assert outs[0].shape == outs[1].shape and torch.equal(outs[0], outs[1])
    ref = torch.nn.functional.unfold(x.float().cpu(), kernel_size=P, stride=P, padding=pad).transpose(1, 2).reshape(outs[0].shape)
    assert torch.equal(outs[0].float().cpu(), ref.to(odt).float())


# ---- CViT at width 512: the one-launch block tail / model tail (cvit_fused.hip) --------------------------------------------------------
def _chain_reference(a, resid, blk, tail=None):
    """float64 restatement of what tante_cvit_chain512 fuses (models/cvit.py:133-139 behind the attention; 459-466 + Mlp 213-242)."""
    import torch.nn.functional as F
    d = torch.float64
    A, R = a.to(d), resid.to(d)
    M = A.shape[0]
    R = R.repeat(M // R.shape[0], 1)
    p = {k: v.detach().to(d).cpu() for k, v in blk.state_dict().items()}
    x1 = A @ p["attn.out_proj.weight"].T + p["attn.out_proj.bias"] + R
    h = F.gelu(F.layer_norm(x1, (512,), p["layer_norm2.weight"], p["layer_norm2.bias"], blk.eps) @ p["mlp.fc1.weight"].T + p["mlp.fc1.bias"])
    x2 = x1 + h @ p["mlp.fc2.weight"].T + p["mlp.fc2.bias"]
    if tail is None:
        return x2
    norm2, mlp = tail
    z = F.layer_norm(x2, (512,), norm2.weight.detach().to(d).cpu(), norm2.bias.detach().to(d).cpu(), norm2.eps)
    q = {k: v.detach().to(d).cpu() for k, v in mlp.state_dict().items()}
    y = z + F.gelu(z @ q["dense_layers.0.weight"].T + q["dense_layers.0.bias"])
    y = F.layer_norm(y, (512,), q["layer_norms.0.weight"], q["layer_norms.0.bias"], mlp.layer_norms[0].eps)
    return y @ q["output_layer.weight"].T + q["output_layer.bias"]


@pytest.mark.parametrize("M,period,out_dim,tokens", [(64, 64, 0, 64), (320, 64, 0, 64), (256, 256, 16, 64), (384, 128, 16, 64), (128, 128, 7, 64),
                                                     (48, 48, 0, 16), (256, 256, 0, 16), (80, 16, 16, 16), (1024, 256, 5, 16), (33280, 33280, 16, 0)])
def test_cvit_chain512_against_float64(dev, M, period, out_dim, tokens, monkeypatch):
    """tante_cvit_chain512 (mode 0: out_dim = 0; mode 1 otherwise) against a float64 restatement fed the same bf16 attention rows: random
    (non-trivial) LayerNorm affines everywhere, residual rows shared with a period, an output layer narrower than the 16-row tile; both
    workgroup shapes (64 / 16 tokens, forced through TANTE_CVIT_CHAIN_TOKENS; 0 = the launcher's own choice)."""
    from tante_amd import cvit as CV, kernels as Kk, _lib as L
    monkeypatch.setattr(CV, "CVIT_SMALL_ROWS", 0)      # (round 6: short tails run as three wave-per-tile GEMMs by default; this is the chain's test)
    torch.manual_seed(M + out_dim)
    blk = CV.SelfAttnBlock(8, 512, 1).to(dev)
    norm2 = torch.nn.LayerNorm(512).to(dev)
    mlp = CV.Mlp(512, 1, 512, max(out_dim, 1)).to(dev)
    with torch.no_grad():
        for ln in (blk.layer_norm2, norm2, mlp.layer_norms[0]):
            ln.weight.copy_(1.0 + 0.3 * torch.randn(512))
            ln.bias.copy_(0.2 * torch.randn(512))
    a = torch.randn(M, 512).to(dev, torch.bfloat16)
    resid = (torch.randn(period, 512) * 1.5 + 0.3).to(dev)
    tail = (norm2, mlp) if out_dim else None
    with torch.no_grad():
        assert blk._chain_ok(L.BF16, M, period)
        try:
            L.set_option("TANTE_CVIT_CHAIN_TOKENS", tokens)
            got = blk._tail(a, resid, None, L.BF16, model_tail=tail)
        finally:
            L.set_option("TANTE_CVIT_CHAIN_TOKENS", 0)
    ref = _chain_reference(a.float().cpu(), resid.cpu(), blk, tail)
    assert got.shape == ref.shape
    close(got, ref.float(), "bf16", f"chain512 M={M} out_dim={out_dim}")


def test_cvit_width512_fused_against_oracle_and_unfused(dev):
    """A cfg4-width CViT (emb 512, 8 x 64 heads, depth 2, 64 x 64 fields, 16 x 16 latent grid) at B = 4: (a) bf16 with the fused tails and
    the once-projected shared queries against the ORACLE (cvit.py:427-466 restated) at the bf16 bar; (b) the same with
    TANTE_CVIT_FUSED = 0 (per-op launches); (c) query-point mode (777 points: not a multiple of 64, so the decoder takes the per-op
    route) against the full grid at those pixels."""
    import tante_amd
    from tante_amd import _lib as L
    from oracle import cvit_oracle as OC
    torch.manual_seed(12)
    kw = dict(out_steps=4, patch_size=(1, 16, 16), grid_size=(16, 16), latent_dim=512, emb_dim=512, depth=2, num_heads=8, dec_emb_dim=512,
              dec_num_heads=8, dec_depth=1, num_mlp_layers=1, mlp_ratio=1, eps=300.0)
    m = tante_amd.CViT(4, tante_amd.TanteMetadata(n_fields=4, spatial_resolution=(64, 64)), **kw).to(dev).eval().set_compute("bf16")
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.LayerNorm):
                mod.weight.add_(0.2 * torch.randn_like(mod.weight))
                mod.bias.add_(0.1 * torch.randn_like(mod.bias))
    xs = torch.randn(4, 4, 4, 64, 64)
    w = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    ref = OC.cvit_forward(w, OC.CvitCfg(4, 4, (64, 64), **kw), xs)
    with torch.no_grad():
        y = m(xs.to(dev))
        close(y, ref, "bf16", "width-512 CViT, fused tails")
        try:
            L.set_option("TANTE_CVIT_FUSED", 0)
            y0 = m(xs.to(dev))
        finally:
            L.set_option("TANTE_CVIT_FUSED", 1)
        close(y0, ref, "bf16", "width-512 CViT, per-op launches")
        try:      # the opt-in form whose block launch also projects the next block's q | k | v (tante_cvit_chain512_qkv)
            L.set_option("TANTE_CVIT_CHAIN_QKV", 1)
            y1 = m(xs.to(dev))
        finally:
            L.set_option("TANTE_CVIT_CHAIN_QKV", 0)
        close(y1, ref, "bf16", "width-512 CViT, next block's projection inside the block launch")
        idx = torch.randint(0, 64 * 64, (777,), device=dev)
        yq = m(xs.to(dev), tante_amd.cvit.generate_coords(64, 64, dev)[idx])
    close(yq, y.permute(0, 1, 3, 4, 2).reshape(4, 4, 64 * 64, -1)[:, :, idx], "bf16", "query points vs full grid")


@pytest.mark.parametrize("n,C,H,W,p,idt,odt", [(2, 32, 64, 96, 1, "bf16", "fp32"), (1, 8, 130, 70, 1, "fp32", "fp32"), (3, 40, 20, 200, 2, "bf16", "bf16"),
                                               (1, 32, 512, 512, 1, "bf16", "fp32"), (2, 5, 33, 65, 0, "fp32", "fp32")])
def test_resize_channels_last_to_first_equals_generic_kernel(dev, n, C, H, W, p, idt, odt):
    """The tiled crop + bilinear resize of a channels-last stage into a channels-first tensor (the last padded decoder stage,
    enc_dec_cnn.py:164-184: crop p pixels per side, resize back to (H, W)) against the generic kernel (TANTE_RESIZE_TILED = 0): bit for
    bit; and against torch's interpolate of the cropped window (fp32 bar)."""
    from tante_amd import _lib as L, kernels as Kk
    g = torch.Generator().manual_seed(C * H)
    dt = {"bf16": torch.bfloat16, "fp32": torch.float32}
    full = torch.randn(n, H, W, C, generator=g).to(dev, dt[idt])
    outs = []
    try:
        for tiled in (1, 0):
            L.set_option("TANTE_RESIZE_TILED", tiled)
            out = torch.empty(n, C, H, W, dtype=dt[odt], device=dev)
            Kk.resize_bilinear(full, n, C, H - 2 * p, W - 2 * p, (p, p), (H * W * C, 1, W * C, C), H, W, out, (C * H * W, H * W, W, 1), L.ACT_GELU_ERF)
            outs.append(out)
    finally:
        L.set_option("TANTE_RESIZE_TILED", 1)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    win = full.float().cpu().permute(0, 3, 1, 2)[:, :, p:H - p, p:W - p]
    ref = torch.nn.functional.gelu(torch.nn.functional.interpolate(win, size=(H, W), mode="bilinear", align_corners=False))
    close(outs[0], ref if odt == "fp32" else ref.to(torch.bfloat16).float(), "fp32" if odt == "fp32" else "bf16", "tiled resize vs torch")


@pytest.mark.parametrize("n,Cin,Cout,H,W,m1,m2", [(2, 8, 32, 64, 128, 5, 5), (1, 32, 32, 32, 256, 8, 20), (3, 40, 16, 48, 128, 4, 3), (1, 8, 8, 512, 512, 20, 20),
                                                   (2, 64, 24, 16, 128, 2, 32)])
def test_spectral_layer_split_bf16_inverse_rows(dev, n, Cin, Cout, H, W, m1, m2):
    """tante_spectral_layer_c in the bf16 mode (inverse row transform + 1x1 conv as three split-operand products on the bf16 matrix pipe,
    enc_dec_fno.py:184-222) against the fp32 mode of the same entry and against the oracle's rfft2 / irfft2 restatement: the split keeps
    16 mantissa bits per operand, so the bar here is 1e-4 of the output's scale -- a hundred times tighter than the mode's own 1e-2."""
    from tante_amd import _lib as L, kernels as Kk
    from oracle import spectral_oracle as SO
    torch.manual_seed(Cin * H + m2)
    x = torch.randn(n, Cin, H, W)
    wre = torch.randn(Cin, Cout, m1, m2) / (Cin * Cout) ** 0.5
    wim = torch.randn(Cin, Cout, m1, m2) / (Cin * Cout) ** 0.5
    w0 = torch.randn(Cout, Cin) / Cin ** 0.5
    b0 = torch.randn(Cout)
    args = [t.to(dev) for t in (x, wre, wim)]
    # without the activation: the split products alone (bar 1e-4); with GELU the bf16 mode's polynomial (|error| <= 8.3e-5) joins in
    for act, bar in ((L.ACT_NONE, 1e-4), (L.ACT_GELU_ERF, 3e-4)):
        y32 = Kk.spectral_layer(*args, m1, m2, w0.to(dev), b0.to(dev), act, L.F32)
        y16 = Kk.spectral_layer(*args, m1, m2, w0.to(dev), b0.to(dev), act, L.BF16)
        torch.cuda.synchronize()
        r, m = rel_err(y16.cpu(), y32.cpu()), max_rel(y16.cpu(), y32.cpu())
        record_parity(r, m, bar, "bf16", f"split-bf16 inverse rows vs fp32 kernel, act {act} ({n},{Cin},{Cout},{H},{W},{m1},{m2})")
        assert r < bar and m < 2 * bar, (act, r, m)
    ref = torch.nn.functional.gelu(SO.spectral_layer({"weight": torch.complex(wre, wim), "w0.weight": w0.view(Cout, Cin, 1, 1), "w0.bias": b0}, x, m1, m2))
    close(y16, ref, "bf16", "split-bf16 spectral layer vs oracle", scale=1e-2)


@pytest.mark.parametrize("outer,n,inner", [(3, 64, 512), (8, 64, 256 * 64), (5, 16, 256), (2, 48, 768), (7, 32, 256)])
def test_axis_propagator_on_the_matrix_pipe(dev, outer, n, inner):
    """tante_axis_mlp_c in the bf16 mode for long axes (n = 16 .. 64, inner a multiple of 256: cfg5's 64 x 64 planes) -- the two n x n
    products of nn.Sequential(Linear(n, n), GELU, Linear(n, n)) (attn_backbone.py:100-106, 140-143) on bf16 MFMAs, the residual in fp32 --
    against a float64 restatement at the bf16 bar, and against the fp32 vector kernel it replaces (TANTE_AXIS_MFMA = 0) at the same bar."""
    from tante_amd import _lib as L, kernels as Kk
    g = torch.Generator().manual_seed(outer * n)
    x = torch.randn(outer, n, inner, generator=g)
    w1, w2 = torch.randn(n, n, generator=g) / n ** 0.5, torch.randn(n, n, generator=g) / n ** 0.5
    b1, b2 = torch.randn(n, generator=g) * 0.1, torch.randn(n, generator=g) * 0.1
    xd = x.double()
    ref = xd + torch.einsum("pq,oqi->opi", w2.double(), torch.nn.functional.gelu(torch.einsum("pq,oqi->opi", w1.double(), xd) + b1.double()[None, :, None])) \
        + b2.double()[None, :, None]
    outs = []
    try:
        for mf in (1, 0):
            L.set_option("TANTE_AXIS_MFMA", mf)
            y = x.clone().to(dev)
            Kk.axis_mlp(y, outer, n, inner, w1.to(dev), b1.to(dev), w2.to(dev), b2.to(dev), L.BF16)
            outs.append(y.cpu())
    finally:
        L.set_option("TANTE_AXIS_MFMA", 1)
    # the propagator's own contribution (output minus the residual input) carries the bf16 rounding
    close(outs[0] - x, (ref - xd).float(), "bf16", f"matrix-pipe propagator n={n} vs float64")
    close(outs[1] - x, (ref - xd).float(), "fp32", f"vector propagator n={n} vs float64", scale=10)


def test_weight_gradient_jobs_share_a_launch(dev):
    """tante_wgrad_jobs_ws: the four weight gradients of a block (768 x 256 and three 256 x 256, four BPTT uses of 1 536 rows each, bias
    gradients on two of them) as the jobs of ONE launch against float64 and against the per-weight launches (tante_wgrad_multi_ws);
    accumulation into non-zero gradient slots; a job outside the shape rules sends every job down the per-weight path."""
    import ctypes as C
    from tante_amd import _lib as L
    from tante_amd.autograd import _rm_linear, _wgrad_workspace
    torch.manual_seed(3)
    R, n = 1536, 4
    shapes = [(768, 256, True), (256, 256, False), (256, 256, True), (256, 256, False)]
    ws = _wgrad_workspace(dev)
    s = torch.cuda.current_stream().cuda_stream

    def build(shapes):
        ops, jobs, keep = [], (L.WgradJob * len(shapes))(), []
        for jb, (I, J, bias) in zip(jobs, shapes):
            dys = [torch.randn(R, I, device=dev).bfloat16() for _ in range(n)]
            acts = [torch.randn(R, J, device=dev).bfloat16() for _ in range(n)]
            gW = torch.randn(I, J, device=dev)
            gb = torch.randn(I, device=dev) if bias else None
            U = (L.RowMat * n)(*[_rm_linear(t) for t in dys])
            V = (L.RowMat * n)(*[_rm_linear(t) for t in acts])
            keep.append((U, V))
            jb.U, jb.V, jb.n_seg, jb.R, jb.I, jb.J = U, V, n, R, I, J
            jb.dW, jb.dbias = gW.data_ptr(), None if gb is None else gb.data_ptr()
            jb.layout, jb.P, jb.C_other, jb.swap = L.W_LINEAR, 0, 0, 0
            ops.append((dys, acts, gW, gb, gW.clone(), None if gb is None else gb.clone()))
        return ops, jobs, keep

    ops, jobs, keep = build(shapes)
    L.check(L.lib().tante_wgrad_jobs_ws(jobs, len(shapes), L.BF16, ws.data_ptr(), ws.numel(), s), "tante_wgrad_jobs")
    torch.cuda.synchronize()
    for dys, acts, gW, gb, gW0, gb0 in ops:
        ref = gW0.double() + sum(d.double().T @ a.double() for d, a in zip(dys, acts))
        close(gW, ref.float(), "fp32", "wgrad jobs: dW", scale=10)
        if gb is not None:
            close(gb, (gb0.double() + sum(d.double().sum(0) for d in dys)).float(), "fp32", "wgrad jobs: dbias", scale=10)
        # the per-weight launch on the same operands (from the same starting slots): the same sums, another split of the rows
        g2, b2 = gW0.clone(), None if gb0 is None else gb0.clone()
        U = (L.RowMat * n)(*[_rm_linear(t) for t in dys])
        V = (L.RowMat * n)(*[_rm_linear(t) for t in acts])
        L.check(L.lib().tante_wgrad_multi_ws(C.byref(U), C.byref(V), n, R, gW.shape[0], gW.shape[1], g2.data_ptr(), None if b2 is None else b2.data_ptr(),
                                             L.W_LINEAR, 0, 0, 0, L.BF16, 1, ws.data_ptr(), ws.numel(), s), "tante_wgrad_multi")
        close(gW, g2, "fp32", "wgrad jobs vs per-weight launch", scale=10)
    # a 64-column job is outside the shared kernel's tiles: the call still returns the right sums (job by job)
    ops, jobs, keep = build([(256, 256, True), (128, 64, False)])
    L.check(L.lib().tante_wgrad_jobs_ws(jobs, 2, L.BF16, ws.data_ptr(), ws.numel(), s), "tante_wgrad_jobs")
    torch.cuda.synchronize()
    for dys, acts, gW, gb, gW0, gb0 in ops:
        ref = gW0.double() + sum(d.double().T @ a.double() for d, a in zip(dys, acts))
        close(gW, ref.float(), "fp32", "wgrad jobs (fallback): dW", scale=10)
