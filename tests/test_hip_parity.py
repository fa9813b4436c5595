"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the committed golden vectors.

Bars (BASELINE.json north_star): fp32 compute within 1e-5 relative, bf16 compute within 1e-2 relative,
of the reference's fp32 CPU path.  `rel` below is ||a-b||_2 / ||b||_2 and `mx` is max|a-b| / max|b|.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, split_prefix, rel_err, max_rel, record_parity

pytestmark = pytest.mark.gpu

TOL = {"fp32": 1e-5, "bf16": 1e-2}


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _ops():
    from tante_amd import kernels as K, _lib as L
    return K, L


def close(a, b, mode, scale=1.0):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.isfinite(a).all()
    r, m = rel_err(a, b), max_rel(a, b)
    record_parity(r, m, TOL[mode] * scale, mode)
    assert r < TOL[mode] * scale and m < TOL[mode] * scale * 2, f"rel={r:.3e} max={m:.3e} (tol {TOL[mode] * scale:.1e})"


def bf16_round(t):
    return t.to(torch.bfloat16).float()


# ---------------------------------------------------------------------------------------------------
# GEMM core: layouts, K/N/M tails, epilogues
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("M,N,K", [(200, 768, 256), (64, 64, 64), (37, 20, 44), (130, 256, 128), (70, 12, 512),
                                   (129, 1, 16), (16, 96, 32), (300, 130, 100)])
def test_gemm_linear(dev, mode, M, N, K):
    Kk, L = _ops()
    comp = Kk.COMPUTE[mode]
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    ref = a @ w.t() + b
    pw = Kk.pack_weight(w.to(dev), b.to(dev), comp)
    out = torch.full((M, N), float("nan"), device=dev)
    Kk.linear(a.to(dev), pw, out, M=M)
    close(out, ref, mode)


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_gemm_layout_asymmetric(dev, mode):
    """A = identity-like rows against an asymmetric integer W: catches transposed / permuted fragments exactly."""
    Kk, L = _ops()
    comp = Kk.COMPUTE[mode]
    M = N = K = 64
    a = torch.eye(M, K)
    w = (torch.arange(N)[:, None] * 3 + torch.arange(K)[None, :] * 1).float() % 61      # exact in bf16
    pw = Kk.pack_weight(w.to(dev), None, comp)
    out = torch.empty(M, N, device=dev)
    Kk.linear(a.to(dev), pw, out, M=M)
    assert torch.equal(out.cpu(), w.t().contiguous())


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("act", ["none", "gelu_erf", "gelu_tanh", "relu"])
def test_gemm_ln_act_residual(dev, mode, act):
    from oracle import tante_oracle as O
    Kk, L = _ops()
    comp = Kk.COMPUTE[mode]
    M, N, K = 150, 128, 256
    g = torch.Generator().manual_seed(5)
    a = torch.randn(M, K, generator=g) * 2 + 0.5
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    gamma = 1 + 0.2 * torch.randn(K, generator=g)
    beta = 0.2 * torch.randn(K, generator=g)
    res = torch.randn(M, N, generator=g)
    f = {"none": lambda x: x, "gelu_erf": O.gelu_erf, "gelu_tanh": O.gelu_tanh, "relu": torch.relu}[act]
    code = {"none": L.ACT_NONE, "gelu_erf": L.ACT_GELU_ERF, "gelu_tanh": L.ACT_GELU_TANH, "relu": L.ACT_RELU}[act]
    ref = f(O.layer_norm(a, gamma, beta) @ w.t() + b) + res
    pw = Kk.pack_weight(w.to(dev), b.to(dev), comp, gamma=gamma.to(dev), beta=beta.to(dev))
    out = res.to(dev).clone()
    Kk.linear(a.to(dev), pw, out, M=M, ln=True, act=code, residual=out)     # in place on the residual stream
    close(out, ref, mode)
    # bf16 activations as source and destination
    if mode == "bf16":
        a16 = a.to(torch.bfloat16)
        ref16 = f(a16.float() @ w.t() + b)
        pw2 = Kk.pack_weight(w.to(dev), b.to(dev), comp)
        o16 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        Kk.linear(a16.to(dev), pw2, o16, M=M, act=code)
        close(o16, ref16, mode)


@pytest.mark.parametrize("act", ["none", "gelu_erf", "gelu_tanh", "relu"])
@pytest.mark.parametrize("M,N,K,odt,res", [(4096, 256, 256, "bf16", False), (5000, 200, 128, "f32", True), (4100, 768, 256, "bf16", False),
                                           (4224, 96, 512, "f32", False), (8192, 512, 256, "bf16", True)])
def test_gemm_training_shapes(dev, act, M, N, K, odt, res):
    """bf16 rows x packed bf16 weight at training sizes (M >= 4096 tokens): the resident-round kernel (gemm_lite_kernel) with ragged M,
    N that is not a whole tile, every activation, fp32 / bf16 destinations and the fp32 residual operand."""
    from oracle import tante_oracle as O
    Kk, L = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g) if res else None
    f = {"none": lambda x: x, "gelu_erf": O.gelu_erf, "gelu_tanh": O.gelu_tanh, "relu": torch.relu}[act]
    code = {"none": L.ACT_NONE, "gelu_erf": L.ACT_GELU_ERF, "gelu_tanh": L.ACT_GELU_TANH, "relu": L.ACT_RELU}[act]
    ref = f(a.float() @ w.to(torch.bfloat16).float().t() + b)
    if res:
        ref = ref + r
    pw = Kk.pack_weight(w.to(dev), b.to(dev), L.BF16)
    out = torch.full((M + 1, N), float("nan"), dtype=torch.bfloat16 if odt == "bf16" else torch.float32, device=dev)
    Kk.linear(a.to(dev), pw, out, M=M, act=code, residual=None if r is None else r.to(dev))
    assert torch.isnan(out[M].float()).all()                      # nothing written past row M
    err = (out[:M].float().cpu() - ref).abs().max() / ref.abs().max()
    assert err < (1e-2 if odt == "bf16" else 2e-3), err          # weights rounded identically on both sides: what is left is the output rounding (+ polynomial GELU)


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("nchw,Cin,Cout,P,H,W", [(True, 11, 64, 2, 16, 32), (False, 64, 128, 2, 8, 12), (True, 1, 16, 2, 8, 8),
                                                 (False, 6, 20, 2, 4, 6), (True, 3, 8, 1, 4, 4)])
def test_gemm_patch_embed(dev, mode, nchw, Cin, Cout, P, H, W):
    from oracle import tante_oracle as O
    Kk, L = _ops()
    comp = Kk.COMPUTE[mode]
    g = torch.Generator().manual_seed(Cin * 11 + Cout)
    n_img = 3
    x = torch.randn(n_img, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, P, P, generator=g) / math.sqrt(Cin * P * P)
    b = torch.randn(Cout, generator=g)
    ref = O.gelu_erf(O.real_conv2d(x, w, b, P, 0.0)).permute(0, 2, 3, 1).contiguous()      # channels-last
    if nchw:
        pw = Kk.pack_weight(w.to(dev), b.to(dev), comp, L.W_LINEAR, N=Cout, K=Cin * P * P)
        src = x.to(dev)
    else:
        pw = Kk.pack_weight(w.to(dev), b.to(dev), comp, L.W_CONV_NHWC, N=Cout, K=Cin * P * P, P=P, C_other=Cin)
        src = x.permute(0, 2, 3, 1).contiguous().to(dev)
    out = torch.empty(n_img, H // P, W // P, Cout, device=dev)
    Kk.patch_embed(src, pw, out, n_img=n_img, Hin=H, Win=W, Cin=Cin, P=P, nchw=nchw, act=L.ACT_GELU_ERF)
    close(out, ref, mode)


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("nchw_out,Cin,Cout,P,H,W", [(False, 64, 32, 2, 4, 6), (True, 16, 11, 2, 8, 8), (True, 8, 3, 1, 4, 4),
                                                      (False, 12, 6, 2, 3, 5), (True, 32, 1, 2, 8, 8)])
def test_gemm_deconv(dev, mode, nchw_out, Cin, Cout, P, H, W):
    from oracle import tante_oracle as O
    Kk, L = _ops()
    comp = Kk.COMPUTE[mode]
    g = torch.Generator().manual_seed(Cin * 5 + Cout)
    n_img = 2
    x = torch.randn(n_img, Cin, H, W, generator=g)
    w = torch.randn(Cin, Cout, P, P, generator=g) / math.sqrt(Cin)
    b = torch.randn(Cout, generator=g)
    ref = O.real_transconv2d(x, w, b, P, 0.0)
    src = x.permute(0, 2, 3, 1).contiguous().to(dev)
    lay = L.W_DECONV_NCHW if nchw_out else L.W_DECONV_NHWC
    pw = Kk.pack_weight(w.to(dev), b.to(dev), comp, lay, N=Cout * P * P, K=Cin, P=P, C_other=Cout)
    if nchw_out:
        out = torch.empty(n_img, Cout, H * P, W * P, device=dev)
    else:
        out = torch.empty(n_img, H * P, W * P, Cout, device=dev)
        ref = ref.permute(0, 2, 3, 1).contiguous()
    Kk.deconv(src, pw, out, n_img=n_img, Hi=H, Wi=W, P=P, Cout=Cout, nchw_out=nchw_out, act=L.ACT_NONE)
    close(out, ref, mode)


# ---------------------------------------------------------------------------------------------------
# attention core per axis letter
# ---------------------------------------------------------------------------------------------------
def _attn_ref(qkv, n_head, causal):
    Bp, Lq, C3 = qkv.shape
    C_ = C3 // 3
    d = C_ // n_head
    q, k, v = (t.reshape(Bp, Lq, n_head, d).transpose(1, 2) for t in qkv.split(C_, dim=-1))
    s = q @ k.transpose(-1, -2) / math.sqrt(d)
    if causal:
        s = s.masked_fill(torch.triu(torch.ones(Lq, Lq, dtype=torch.bool), 1), float("-inf"))
    return (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(Bp, Lq, C_)


_PERM = {  # (permutation to (batch', L, C) view of (B,T,H,W,C), inverse)
    "T": ((0, 2, 3, 1, 4), lambda B, T, H, W: (B * H * W, T)),
    "H": ((0, 1, 3, 2, 4), lambda B, T, H, W: (B * T * W, H)),
    "W": ((0, 1, 2, 3, 4), lambda B, T, H, W: (B * T * H, W)),
    "L": ((0, 1, 2, 3, 4), lambda B, T, H, W: (B * T, H * W)),
    "Y": ((0, 3, 1, 2, 4), lambda B, T, H, W: (B * W, T * H)),
    "X": ((0, 2, 1, 3, 4), lambda B, T, H, W: (B * H, T * W)),
    "A": ((0, 1, 2, 3, 4), lambda B, T, H, W: (B, T * H * W)),
}


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("letter,B,T,H,W,C,nh", [("T", 2, 4, 3, 5, 32, 4), ("H", 2, 3, 32, 4, 64, 2), ("W", 1, 2, 3, 48, 64, 8),
                                                  ("L", 2, 2, 6, 7, 32, 2), ("Y", 2, 3, 5, 4, 16, 2), ("X", 2, 3, 4, 6, 32, 4),
                                                  ("A", 2, 4, 9, 10, 32, 1), ("L", 1, 1, 20, 30, 64, 8), ("T", 3, 7, 2, 2, 128, 2)])
def test_attention_letters(dev, dtype, letter, B, T, H, W, C, nh):
    Kk, L = _ops()
    g = torch.Generator().manual_seed(ord(letter) + T * H * W)
    qkv = torch.randn(B, T, H, W, 3 * C, generator=g)
    if dtype == torch.bfloat16:
        qkv = bf16_round(qkv)
    perm, shp = _PERM[letter]
    Bp, Lq = shp(B, T, H, W)
    causal = letter == "T"
    ref_seq = _attn_ref(qkv.permute(*perm).reshape(Bp, Lq, 3 * C), nh, causal)
    inv = [perm.index(i) for i in range(5)]
    if letter == "T":
        ref = ref_seq.reshape(B, H, W, T, C).permute(*inv)
    elif letter == "H":
        ref = ref_seq.reshape(B, T, W, H, C).permute(*inv)
    elif letter == "Y":
        ref = ref_seq.reshape(B, W, T, H, C).permute(*inv)
    elif letter == "X":
        ref = ref_seq.reshape(B, H, T, W, C).permute(*inv)
    else:
        ref = ref_seq.reshape(B, T, H, W, C)
    o = torch.full((B * T * H * W, C), float("nan"), dtype=dtype, device=dev)
    Kk.attention(qkv.reshape(-1, 3 * C).to(dev, dtype), o, C, nh, Kk.make_seq(letter, B, T, H, W), causal)
    close(o.view(B, T, H, W, C), ref.contiguous(), "fp32" if dtype == torch.float32 else "bf16")


# ---------------------------------------------------------------------------------------------------
# pointwise stages
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("outer,n,inner", [(6, 32, 64), (3, 4, 1000), (2, 48, 33), (5, 7, 16), (2, 100, 40), (4, 64, 70)])
def test_axis_mlp(dev, outer, n, inner):
    from oracle import tante_oracle as O
    Kk, L = _ops()
    g = torch.Generator().manual_seed(n)
    x = torch.randn(outer, n, inner, generator=g)
    w = {"p.0.weight": torch.randn(n, n, generator=g) / math.sqrt(n), "p.0.bias": torch.randn(n, generator=g),
         "p.2.weight": torch.randn(n, n, generator=g) / math.sqrt(n), "p.2.bias": torch.randn(n, generator=g)}
    ref = x + O._axis_mlp(w, "p", x.transpose(1, 2)).transpose(1, 2)
    xd = x.to(dev)
    Kk.axis_mlp(xd, outer, n, inner, *(w[k].to(dev) for k in ("p.0.weight", "p.0.bias", "p.2.weight", "p.2.bias")))
    close(xd, ref, "fp32")


def test_film_taylor_rt(dev):
    from oracle import tante_oracle as O
    import tante_amd
    Kk, L = _ops()
    g = load_golden("g5_film")
    f = tante_amd.film(32).to(dev)
    f.load_state_dict(split_prefix(g, "w."))
    with torch.no_grad():
        close(f(g["x5"].to(dev), g["t_series_4_1"].to(dev)), g["y5"], "fp32")
        close(f(g["x3"].to(dev), g["rt"].to(dev)), g["y3"], "fp32")
    assert torch.equal(tante_amd.t_series(4, 1.0), g["t_series_4_1"]) and torch.equal(tante_amd.t_series(5, 0.5), g["t_series_5_05"])
    # Taylor sum against the literal loop of tante.py:165-171
    gen = torch.Generator().manual_seed(1)
    B, T, frame = 3, 4, 2 * 8 * 12
    inp = torch.randn(B, T, frame, generator=gen)
    ders = [torch.randn(B, frame, generator=gen) for _ in range(3)]
    for dt, n_out in ((1.0, 1), (0.5, 3), (2.0, 8), (0.1, 2)):
        ref = torch.stack([inp[:, -1] + sum(ders[k - 1] * (i * dt) ** k / math.factorial(k) for k in (1, 2, 3))
                           for i in range(1, n_out + 1)], 1)
        out = torch.empty(B, n_out, frame, device=dev)
        Kk.taylor(inp.to(dev), (T - 1) * frame, T * frame, [d.to(dev) for d in ders], dt, n_out, out, B, frame)
        close(out, ref, "fp32")
    # interprator (both clamps exercised by the fixture)
    g6 = load_golden("g6_interp")
    it = tante_amd.interprator(32, 12).to(dev)
    it.load_state_dict(split_prefix(g6, "w."))
    with torch.no_grad():
        for out_T, key in ((1.5, "rt_1p5"), (8, "rt_8"), (1, "rt_1")):
            close(it(g6["x"].to(dev), out_T), g6[key], "fp32")


# ---------------------------------------------------------------------------------------------------
# modules against the golden vectors (weights loaded from the reference's state_dict)
# ---------------------------------------------------------------------------------------------------
import glob
import os
from conftest import GOLDEN


def names(pattern):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, pattern + ".npz")))


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("name", names("g3_block_*"))
def test_g3_block(dev, name, mode):
    import tante_amd
    g = load_golden(name)
    C, nh, Lq, causal, ratio = (int(v) for v in g["meta"])
    blk = tante_amd.TransformerBlock(C, nh, mlp_ratio=ratio / 100, dropout=0.0).to(dev).eval()
    blk.load_state_dict(split_prefix(g, "w."))
    blk.compute = mode
    with torch.no_grad():
        y = blk(g["x"].to(dev), causal=bool(causal))
    close(y, g["y"], mode)


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("name", names("g4_backbone_*"))
def test_g4_backbone(dev, name, mode):
    import tante_amd
    g = load_golden(name)
    axes = name.split("_")[-1]
    T, H, W, C, E, nh = (int(v) for v in g["meta"])
    bb = tante_amd.Attn_Backbone((T, H, W, C), axes, expanded_channel=E, n_head=nh, mlp_ratio=1.0, dropout=0.0).to(dev).eval()
    bb.load_state_dict(split_prefix(g, "w."))
    bb.compute = mode
    with torch.no_grad():
        y = bb(g["x"].to(dev))
    close(y, g["y"], mode)


@pytest.mark.parametrize("name", names("g2_encdec_*"))
def test_g2_encdec(dev, name):
    import tante_amd
    g = load_golden(name)
    ps, ov, H, W, nf, C = (int(v) for v in g["meta"])
    md = tante_amd.TanteMetadata(n_fields=nf, spatial_resolution=(H, W))
    ov = ov / 100
    e = tante_amd.enc_CNN(md, embed_dim=C, patch_scale=ps, overlap_ratio=ov).to(dev)
    d = tante_amd.dec_CNN(md, embed_dim=C, patch_scale=ps, overlap_ratio=ov).to(dev)
    e.load_state_dict(split_prefix(g, "enc."))
    d.load_state_dict(split_prefix(g, "dec."))
    with torch.no_grad():
        close(e(g["x"].to(dev)), g["z"], "fp32")     # every patch scale, 'same' padding and overlap (im2col + pool route)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            close(e(g["x"].to(dev)), g["z"], "bf16")
        close(d(g["zz"].to(dev)), g["r"], "fp32")      # padded stages: crop + resize; overlapping taps: tap GEMM + gather-sum + resize
        with torch.autocast("cuda", dtype=torch.bfloat16):
            close(d(g["zz"].to(dev)), g["r"], "bf16")


def _tante_from(g, dev, **kw):
    import tante_amd
    m = tante_amd.TANTE(**kw).to(dev).eval()
    missing = m.load_state_dict(split_prefix(g, "w."), strict=True)
    return m


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_g1_tiny_end_to_end(dev, mode):
    import tante_amd
    g = load_golden("g1_tante_tiny")
    m = _tante_from(g, dev, in_T=4, dset_metadata=tante_amd.TanteMetadata(n_fields=1, spatial_resolution=(64, 64)),
                    taylor_order=2, attn_axes="TL-TL", n_head=4, embed_dim=64, patch_scale=8, dropout=0.0)
    m.set_compute(mode)
    with torch.no_grad():
        y = m(g["x"].to(dev))
    close(y, g["y"], mode)
    # the derivative part alone (output minus the last input frame) must also agree
    d, dref = y.cpu() - g["x"][:, -1:], g["y"] - g["x"][:, -1:]
    assert rel_err(d, dref) < (2e-5 if mode == "fp32" else 1e-2)


@pytest.mark.parametrize("tag,axes", [("o1", "TH"), ("o2", "T-W"), ("o3", "T-H-W")])
def test_g7_taylor_orders(dev, tag, axes):
    import tante_amd
    g = load_golden("g7_taylor_" + tag)
    order, fi, ol = int(g["meta"][0]), float(g["meta"][1]) / 100, int(g["meta"][2])
    m = _tante_from(g, dev, in_T=3, dset_metadata=tante_amd.TanteMetadata(n_fields=2, spatial_resolution=(16, 32)),
                    taylor_order=order, frame_interval=fi, output_length=ol, attn_axes=axes, n_head=2, embed_dim=32,
                    patch_scale=8, dropout=0.0)
    with torch.no_grad():
        y = m(g["x"].to(dev))          # T=5 > in_T: window slice
    close(y, g["y"], "fp32")


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("name", names("g8_rollout_*"))
def test_g8_rollout(dev, name, mode):
    """The reference's Trainer / Evaler.rollout_model outputs (trainer/trainer.py:144-159, evaler.py:121-138), re-fed 4 and 8 steps, in
    BOTH compute modes: every frame of the rollout (per step) and the whole rollout (end) under the stated 1e-5 / 1e-2."""
    import tante_amd
    g = load_golden(name)
    ol, n_roll = int(g["meta"][0]), int(g["meta"][1])
    md = tante_amd.TanteMetadata(n_fields=2, spatial_resolution=(16, 16))
    m = _tante_from(g, dev, in_T=4, dset_metadata=md, taylor_order=2, output_length=ol, attn_axes="T-L", n_head=2,
                    embed_dim=32, patch_scale=8, dropout=0.0)
    m.set_compute(mode)
    fmt = tante_amd.DefaultChannelsFirstFormatter(md)
    with torch.no_grad():
        y, y_ref = tante_amd.rollout_model(m, {"input": g["inp"], "output": g["out"]}, fmt, n_roll)
        yt, _ = tante_amd.rollout_model(m, {"input": g["inp"], "output": g["out"][:, :4]}, fmt, 4)
    close(y, g["y_eval"], mode)
    close(yt, g["y_train"], mode)
    for t in range(n_roll):
        close(y[:, t], g["y_eval"][:, t], mode)
    # the derivative part of every model call (its frames minus the call's own last input frame) against the reference's
    yc, rc = y.cpu(), g["y_eval"]
    prev_y = torch.cat([g["inp"][:, -1:], yc], dim=1)
    prev_r = torch.cat([g["inp"][:, -1:], rc], dim=1)
    bar = 5e-5 if mode == "fp32" else 1e-2
    for t0 in range(0, n_roll, ol):
        d = yc[:, t0:t0 + ol] - prev_y[:, t0:t0 + 1]
        dref = rc[:, t0:t0 + ol] - prev_r[:, t0:t0 + 1]
        r = rel_err(d, dref)
        record_parity(r, max_rel(d, dref), bar, mode, f"g8 rollout, call at step {t0 + 1}, derivative part")
        assert r < bar, (name, mode, t0, r)
    assert torch.equal(y_ref.cpu(), g["y_ref"])


def test_g13_adaptive_dt(dev):
    import tante_amd
    g = load_golden("g13_deg_false")
    m = _tante_from(g, dev, in_T=4, dset_metadata=tante_amd.TanteMetadata(n_fields=1, spatial_resolution=(32, 32)),
                    taylor_order=2, attn_axes="TH-TW", n_head=2, embed_dim=32, patch_scale=8, dropout=0.0, deg=False)
    with torch.no_grad():
        for out_T, tag in ((1.5, "1p5"), (6, "6")):
            y, rt = m(g["x"].to(dev), out_T)
            close(rt, g["rt_" + tag], "fp32")
            close(y, g["y_" + tag], "fp32")


# ---------------------------------------------------------------------------------------------------
# full-size configuration (BASELINE cfg2): oracle on one sample + size-independent properties
# ---------------------------------------------------------------------------------------------------
def _cfg2_model(dev, order3=True):
    import tante_amd
    torch.manual_seed(211)
    md = tante_amd.TanteMetadata(n_fields=11, spatial_resolution=(256, 256))
    kw = dict(taylor_order=3, attn_axes="THW-THW-THW") if order3 else dict(taylor_order=1, attn_axes="THWTHWTHW")
    m = tante_amd.TANTE(in_T=4, dset_metadata=md, n_head=8, mlp_ratio=1.0, dropout=0.1, embed_dim=256, patch_scale=8, **kw)
    return m.to(dev).eval()


def test_cfg2_full_size_against_oracle(dev):
    from oracle import tante_oracle as O
    m = _cfg2_model(dev)
    w = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    cfg = O.TanteCfg(4, 11, (256, 256), taylor_order=3, attn_axes="THW-THW-THW", n_head=8, embed_dim=256, patch_scale=8)
    x = torch.randn(2, 4, 11, 256, 256, generator=torch.Generator().manual_seed(211))
    ref = O.tante_forward(w, cfg, x[:1])
    with torch.no_grad():
        y32 = m.set_compute("fp32")(x.to(dev))
        y16 = m.set_compute("bf16")(x.to(dev))
        y32_b = m.set_compute("fp32")(x[1:].to(dev))
    close(y32[:1], ref, "fp32")
    close(y16[:1], ref, "bf16")
    # derivative part (what the network actually computes) under the same bars, slightly widened
    d32, d16, dref = y32[:1].cpu() - x[:1, -1:], y16[:1].cpu() - x[:1, -1:], ref - x[:1, -1:]
    assert rel_err(d32, dref) < 5e-5 and rel_err(d16, dref) < 1e-2
    # batch independence: sample 1 computed alone == computed inside the batch (bitwise: same kernels, same tiles)
    assert torch.equal(y32[1:], y32_b)


def test_enc23_stage3_split_is_bit_identical(dev):
    """The fused stage 2 + 3 encoder kernel spreads the stage-3 tiles of a token group over 4 / 2 / 1 workgroups depending on the
    input size (enc23_kernel<CB, SPLIT>): the same images encoded alone (4-way split), in a mid-sized batch (2-way) and inside a large
    batch (no split) must come out bitwise equal."""
    m = _cfg2_model(dev).set_compute("bf16")
    g = torch.Generator().manual_seed(9)
    x = torch.randn(10, 4, 11, 256, 256, generator=g).to(dev)          # 40 images: 320 token groups -> SPLIT 1
    with torch.no_grad():
        fa, fb = m._time_tables()
        film = (fa, fb, m.s_emb.view(1024, 256), 4, 1024)
        big = m.encoder.forward_tokens(x, 1, film).view(10, 4, 1024, 256)          # compute code 1 = bf16
        mid = m.encoder.forward_tokens(x[:3], 1, film).view(3, 4, 1024, 256)       # 12 images: 96 groups -> SPLIT 2
        one = m.encoder.forward_tokens(x[:1], 1, film).view(1, 4, 1024, 256)       # 4 images: 32 groups -> SPLIT 4
    assert torch.isfinite(big).all()
    assert torch.equal(big[:3], mid) and torch.equal(big[:1], one)


@pytest.mark.parametrize("shape", [(20, 13, 11), (20, 13, 4), (20, 12, 4), (64, 96, 4)])
def test_format_input_kernel(dev, shape):
    """tante_format_input = DefaultChannelsFirstFormatter.process_input's permute + nan_to_num, written straight into a rollout buffer.
    D = 4 with H W % 4 == 0 takes the register-transpose kernel (four pixels per lane), everything else the generic LDS one."""
    from tante_amd import _lib as L
    B, T, extra = 2, 3, 2
    H, W, D = shape
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, T, H, W, D, generator=g)
    x[0, 1, 3, 4, D // 2] = float("nan"); x[1, 0, 0, 0, 0] = float("inf"); x[1, 2, H - 1, W - 1, D - 1] = float("-inf")
    ref = torch.nan_to_num(x.permute(0, 1, 4, 2, 3))
    buf = torch.full((B, T + extra, D, H, W), 7.0, device=dev)
    xd = x.to(dev)
    L.check(L.lib().tante_format_input(xd.data_ptr(), B * T, T, H * W, D, buf.data_ptr(), buf.stride(0), torch.cuda.current_stream().cuda_stream))
    assert torch.equal(buf[:, :T].cpu(), ref) and bool((buf[:, T:] == 7.0).all())


def test_cfg2_rollout_frame_cache_is_bit_identical(dev, monkeypatch):
    """The rollout loop encodes every frame once (pre-FiLM cache + FiLM applied by the first propagator kernel while it loads) instead
    of once per window: the same arithmetic per token, so the frames must equal the window-by-window encoder's BITWISE, and both must
    equal the plain `model(window)` loop of the reference's rollout_model."""
    import tante_amd
    monkeypatch.setattr(tante_amd.rollout, "NO_TAIL_ENC", True)      # (round 4's fused tail re-encodes predicted frames with another sum order: test_hip_round4.py)
    m = _cfg2_model(dev).set_compute("bf16")
    assert m.enc_cache_supported()
    md = tante_amd.TanteMetadata(n_fields=11, spatial_resolution=(256, 256))
    fmt = tante_amd.DefaultChannelsFirstFormatter(md)
    g = torch.Generator().manual_seed(5)
    batch = {"input": torch.randn(2, 4, 256, 256, 11, generator=g).to(dev), "output": torch.randn(2, 3, 256, 256, 11, generator=g).to(dev)}
    with torch.no_grad():
        y_cache, _ = tante_amd.rollout_model(m, batch, fmt, 3)
        monkeypatch.setattr(tante_amd.rollout, "NO_ENC_CACHE", True)
        y_plain, _ = tante_amd.rollout_model(m, batch, fmt, 3)
        moving = fmt.process_input(batch)[0][0]
        frames = []
        for _ in range(3):
            y = m(moving)
            frames.append(y)
            moving = torch.cat([moving[:, y.shape[1]:], y], dim=1)
        y_loop = fmt.process_output(torch.cat(frames, dim=1))
    assert torch.isfinite(y_cache).all()
    assert torch.equal(y_cache, y_plain)
    assert torch.equal(y_cache, y_loop)


# ---------------------------------------------------------------------------------------------------
# fused block kernels (bf16): every supported shape against the oracle and against the unfused path
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("C,nh,Lq,Bp,causal,ratio", [(256, 8, 32, 5, False, 1.0), (256, 8, 4, 37, True, 1.0), (256, 8, 16, 3, False, 1.0),
                                                      (256, 8, 8, 9, True, 1.0), (256, 8, 1, 70, False, 1.0), (256, 8, 2, 33, True, 1.0),
                                                      (128, 4, 32, 2, False, 2.0), (64, 2, 4, 50, True, 1.0), (64, 2, 32, 3, False, 2.0),
                                                      (128, 4, 8, 20, False, 1.0), (256, 8, 32, 130, False, 1.0),
                                                      # shapes only the feature-sliced kernel (block_sliced.hip) takes: sequences of 3 / 4 /
                                                      # 8 token tiles, lengths that are no multiple or divisor of a tile (element masks),
                                                      # causal long sequences, ragged last workgroups, the 96-token workgroup form
                                                      (256, 8, 48, 5, False, 1.0), (256, 8, 48, 512, False, 1.0), (256, 8, 64, 3, False, 1.0),
                                                      (256, 8, 128, 2, False, 1.0), (256, 8, 12, 23, False, 1.0), (256, 8, 20, 7, True, 1.0),
                                                      (256, 8, 100, 3, False, 1.0), (256, 8, 32, 7, True, 1.0), (256, 8, 48, 3, True, 1.0),
                                                      (256, 8, 16, 1536, False, 1.0), (256, 8, 4, 6144, True, 1.0), (256, 8, 32, 771, False, 1.0),
                                                      (256, 8, 24, 1, False, 1.0), (256, 8, 5, 300, True, 1.0)])
def test_fused_block(dev, C, nh, Lq, Bp, causal, ratio):
    import tante_amd
    from oracle import tante_oracle as O
    from tante_amd import kernels as Kk
    torch.manual_seed(C + Lq + Bp)
    blk = tante_amd.TransformerBlock(C, nh, mlp_ratio=ratio, dropout=0.0).to(dev).eval()
    with torch.no_grad():
        for ln in (blk.ln1, blk.ln2):
            ln.weight.add_(0.2 * torch.randn_like(ln.weight))
            ln.bias.add_(0.2 * torch.randn_like(ln.bias))
        blk.attn.in_proj_bias.add_(0.1 * torch.randn_like(blk.attn.in_proj_bias))
        blk.attn.out_proj.bias.add_(0.1 * torch.randn_like(blk.attn.out_proj.bias))
    assert Kk.block_fused_supported(C, nh, blk.hidden, Lq)
    x = torch.randn(Bp, Lq, C) * 1.5 + 0.3
    ref = O.transformer_block({k: v.detach().cpu() for k, v in blk.state_dict().items()}, x, nh, causal)
    blk.compute = "bf16"
    with torch.no_grad():
        blk.fused = True
        y_f = blk(x.to(dev), causal=causal)
        blk.fused = False
        y_u = blk(x.to(dev), causal=causal)
    close(y_f, ref, "bf16")
    close(y_u, ref, "bf16")
    # the block's own contribution (output minus the residual input) must also hold the bar
    d_f, d_ref = y_f.cpu() - x, ref - x
    assert rel_err(d_f, d_ref) < 2e-2, rel_err(d_f, d_ref)


def test_fused_backbone_strided_letters(dev):
    """T/H/W letters through the fused kernels on a (B,T,H,W,C) grid: the strided gather inside the kernel."""
    import tante_amd
    from oracle import tante_oracle as O
    torch.manual_seed(3)
    T, H, W, C = 4, 8, 16, 64
    bb = tante_amd.Attn_Backbone((T, H, W, C), "THW", n_head=2, mlp_ratio=1.0, dropout=0.0).to(dev).eval()
    x = torch.randn(3, T, H, W, C)
    ref = O.attn_backbone({k: v.detach().cpu() for k, v in bb.state_dict().items()}, x, "THW", 2)
    bb.compute = "bf16"
    with torch.no_grad():
        y = bb(x.to(dev))
    close(y, ref, "bf16")


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("BT,nH,nW,C", [(3, 32, 32, 64), (2, 16, 48, 32), (5, 4, 6, 16), (1, 64, 16, 48), (2, 33, 7, 16), (1, 8, 64, 16),
                                          (2, 48, 32, 32), (3, 16, 16, 16), (1, 32, 64, 16), (1, 48, 48, 32), (2, 64, 32, 16),
                                          (24, 32, 32, 256), (12, 16, 48, 512)])      # the last two: 32-channel tiles (16 / 8 waves)
def test_axis_hw_fused(dev, mode, BT, nH, nW, C):
    """Fused vertical+horizontal propagators (MFMA) against the oracle's two sequential axis MLPs."""
    from oracle import tante_oracle as O
    Kk, L = _ops()
    if not Kk.axis_hw_supported(nH, nW, C, Kk.COMPUTE[mode]):
        assert nH * nW >= 32 * 48, "only the large planes may be refused"
        with pytest.raises(RuntimeError, match="tante_axis_hw"):        # the C side must refuse, not fault
            z = torch.zeros(BT, nH, nW, C, device=dev)
            w = [torch.zeros(nH, nH, device=dev), torch.zeros(nH, device=dev)] * 2
            w2 = [torch.zeros(nW, nW, device=dev), torch.zeros(nW, device=dev)] * 2
            Kk.axis_hw(z, BT, nH, nW, C, w, w2, Kk.COMPUTE[mode])
        return
    g = torch.Generator().manual_seed(nH * 100 + nW)
    x = torch.randn(BT, nH, nW, C, generator=g)

    def mk(n):
        return {"p.0.weight": torch.randn(n, n, generator=g) / math.sqrt(n), "p.0.bias": 0.3 * torch.randn(n, generator=g),
                "p.2.weight": torch.randn(n, n, generator=g) / math.sqrt(n), "p.2.bias": 0.3 * torch.randn(n, generator=g)}
    wh, ww = mk(nH), mk(nW)
    ref = x + O._axis_mlp(wh, "p", x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)        # along h
    ref = ref + O._axis_mlp(ww, "p", ref.permute(0, 1, 3, 2)).permute(0, 1, 3, 2)    # along w
    xd = x.to(dev)
    keys = ("p.0.weight", "p.0.bias", "p.2.weight", "p.2.bias")
    Kk.axis_hw(xd, BT, nH, nW, C, [wh[k].to(dev) for k in keys], [ww[k].to(dev) for k in keys], Kk.COMPUTE[mode])
    close(xd, ref, mode)


# ---------------------------------------------------------------------------------------------------
# harness: metrics, loss gradient, clip + AdamW on the flat bucket
# ---------------------------------------------------------------------------------------------------
def test_metrics_golden_and_strided(dev):
    import tante_amd
    from oracle import tante_oracle as O
    g = load_golden("g10_metrics")
    x, y = g["x"].to(dev), g["y"].to(dev)
    for name in ("MSE", "L2RE", "NNMSE", "VRMSE", "NMSE", "RMSE", "NRMSE", "VMSE"):
        out = getattr(tante_amd, name)()(x, y, None)
        close(out, g[name], "fp32")
    for k in ("below", "inside", "above"):
        v = tante_amd.MSE()(x, y, g["rt_" + k].to(dev), 0.5, 2)
        assert abs(float(v) - float(g["mse_rt_" + k])) < 1e-5
    # prediction given as the channels-last VIEW of a channels-first rollout buffer (no copy), larger and ragged sizes
    gen = torch.Generator().manual_seed(4)
    buf = torch.randn(3, 5, 7, 33, 20, generator=gen)                 # (B, T, C, H, W)
    ref = torch.randn(3, 5, 33, 20, 7, generator=gen)
    pred_view = buf.to(dev).permute(0, 1, 3, 4, 2)
    assert not pred_view.is_contiguous()
    close(tante_amd.MSE.eval(pred_view, ref.to(dev)), O.mse(buf.permute(0, 1, 3, 4, 2), ref), "fp32")
    close(tante_amd.L2RE.eval(pred_view, ref.to(dev)), O.l2re(buf.permute(0, 1, 3, 4, 2), ref), "fp32")
    close(tante_amd.VRMSE.eval(pred_view, ref.to(dev)), O.vrmse(buf.permute(0, 1, 3, 4, 2), ref), "fp32", scale=3.0)
    # gradient of the train loss MSE(...).mean()
    xx = buf.permute(0, 1, 3, 4, 2).clone().requires_grad_(True)
    O.mse(xx, ref).mean().backward()
    close(tante_amd.metrics.mse_mean_grad(pred_view, ref.to(dev)), xx.grad, "fp32")


def test_clip_adamw_against_reference_step(dev):
    """g9: the reference's gradients of step 0 -> clip_grad_norm_(1.0) + AdamW -> its weights after step 1."""
    import tante_amd
    g = load_golden("g9_trainstep")
    lr, wd, b1, b2, eps, max_norm = (float(v) for v in g["hyper"])
    md = tante_amd.TanteMetadata(n_fields=2, spatial_resolution=(16, 16))
    m = tante_amd.TANTE(in_T=4, dset_metadata=md, taylor_order=2, attn_axes="TH-WL", n_head=2, embed_dim=32, patch_scale=8,
                        dropout=0.0).to(dev)
    m.load_state_dict(split_prefix(g, "w0."))
    opt = tante_amd.FlatAdamW(m.parameters(), lr=lr, weight_decay=wd, betas=(b1, b2), eps=eps, max_norm=max_norm)
    for k, p in m.named_parameters():                                 # parameters now live in the flat bucket
        assert p.data_ptr() >= opt.flat_p.data_ptr() and torch.equal(p.detach().cpu(), g["w0." + k])
        p.grad.copy_(g["g0." + k].to(dev))
    gn = opt.grad_norm()
    assert abs(float(gn) - float(g["gnorm0"])) < 1e-5 * float(g["gnorm0"])
    opt.step()
    for k, p in m.named_parameters():
        assert float((p.detach().cpu() - g["w1." + k]).abs().max()) < 2e-3 * lr, k      # vs |update| ~ lr
    # the packed-weight caches notice the raw-pointer update
    from tante_amd.attn_backbone import _WEIGHT_EPOCH
    assert _WEIGHT_EPOCH[0] >= 1


# ---------------------------------------------------------------------------------------------------
# train path: HIP forward + HIP backward against the reference's full gradient fixture (g9)
# ---------------------------------------------------------------------------------------------------
def _g9_model(dev):
    import tante_amd
    g = load_golden("g9_trainstep")
    md = tante_amd.TanteMetadata(n_fields=2, spatial_resolution=(16, 16))
    m = tante_amd.TANTE(in_T=4, dset_metadata=md, taylor_order=2, attn_axes="TH-WL", n_head=2, embed_dim=32, patch_scale=8,
                        dropout=0.0).to(dev).train()
    m.load_state_dict(split_prefix(g, "w0."))
    return g, m, md


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_g9_train_step_gradients(dev, mode):
    import tante_amd
    from tante_amd.autograd import MseMeanFn
    g, m, md = _g9_model(dev)
    m.set_compute(mode)
    fmt = tante_amd.DefaultChannelsFirstFormatter(md)
    batch = {"input": g["inp"].to(dev), "output": g["out"].to(dev)}
    y_pred, y_ref = tante_amd.rollout_model(m, batch, fmt, 4)            # BPTT through 4 re-fed steps
    loss = MseMeanFn.apply(y_pred, y_ref)
    loss.backward()
    tol = 2e-4 if mode == "fp32" else 4e-2
    assert abs(float(loss) - float(g["loss0"])) < (1e-5 if mode == "fp32" else 2e-3) * float(g["loss0"])
    close(y_pred, g["y_pred"], mode)
    worst = 0.0
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        e = max_rel(p.grad.detach().cpu(), g["g0." + k])
        worst = max(worst, e)
        assert e < tol, (k, e)
    # global gradient norm as clip_grad_norm_ sees it
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters()))
    assert abs(float(gn) - float(g["gnorm0"])) < (1e-4 if mode == "fp32" else 2e-2) * float(g["gnorm0"])


def test_g9_two_full_train_steps(dev):
    """loss -> backward -> clip -> AdamW, twice, against the reference's weights after each step (fp32)."""
    import tante_amd
    from tante_amd.autograd import MseMeanFn
    g, m, md = _g9_model(dev)
    lr, wd, b1, b2, eps, max_norm = (float(v) for v in g["hyper"])
    opt = tante_amd.FlatAdamW(m.parameters(), lr=lr, weight_decay=wd, betas=(b1, b2), eps=eps, max_norm=max_norm)
    fmt = tante_amd.DefaultChannelsFirstFormatter(md)
    batch = {"input": g["inp"].to(dev), "output": g["out"].to(dev)}
    for step in range(2):
        opt.zero_grad()
        y_pred, y_ref = tante_amd.rollout_model(m, batch, fmt, 4)
        loss = MseMeanFn.apply(y_pred, y_ref)
        loss.backward()
        assert abs(float(loss) - float(g[f"loss{step}"])) < 2e-4 * float(g[f"loss{step}"])
        opt.step()
        for k, p in m.named_parameters():
            assert float((p.detach().cpu() - g[f"w{step + 1}." + k]).abs().max()) < 3e-2 * lr, (step, k)


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("R,I,J", [(1000, 96, 64), (300, 768, 256), (77, 20, 44), (4096, 130, 70), (513, 8, 8)])
def test_wgrad_linear_with_bias(dev, mode, R, I, J):
    """dW = U^T V and db = column sums of U, dense row matrices (odd sizes exercise the generic loader and tile edges)."""
    from tante_amd.autograd import wgrad, _rm_linear
    from tante_amd import kernels as Kk
    g = torch.Generator().manual_seed(R + I)
    dt = torch.float32 if mode == "fp32" else torch.bfloat16
    U = torch.randn(R, I, generator=g).to(dt)
    V = torch.randn(R, J, generator=g).to(dt)
    ref, refb = U.float().t() @ V.float(), U.float().sum(0)
    Ud, Vd = U.to(dev), V.to(dev)
    dW, db = wgrad(_rm_linear(Ud), _rm_linear(Vd), R, I, J, (I, J), Kk.COMPUTE[mode], device=dev, with_bias=True)
    close(dW, ref, mode)
    close(db, refb, mode)


def test_wgrad_lines_and_patches(dev):
    """the other operand shapes: axis lines (element stride = inner) and k = s patches, against autograd of torch ops"""
    from tante_amd.autograd import wgrad, _rm_linear, _rm_patch
    from tante_amd import _lib as L
    g = torch.Generator().manual_seed(0)
    outer, n, inner = 3, 24, 40
    dy, h = torch.randn(outer, n, inner, generator=g), torch.randn(outer, n, inner, generator=g)
    ref = torch.einsum("oai,obi->ab", dy, h)
    dyd, hd = dy.to(dev), h.to(dev)

    def lines(t):
        return _rm_linear(t, n0=inner, s1=n * inner, s0=1, es=inner, cols=n)
    close(wgrad(lines(dyd), lines(hd), outer * inner, n, n, (n, n), L.F32, device=dev), ref, "fp32")
    # conv weight gradient: patches of a channels-last / channels-first image
    n_img, Cin, Cout, P, H, W = 2, 6, 10, 2, 8, 12
    x = torch.randn(n_img, Cin, H, W, generator=g, requires_grad=True)
    w = torch.randn(Cout, Cin, P, P, generator=g, requires_grad=True)
    y = torch.nn.functional.conv2d(x, w, stride=P)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    d = gy.permute(0, 2, 3, 1).reshape(-1, Cout).contiguous().to(dev)
    M = d.shape[0]
    xn = x.detach().to(dev)
    dW = wgrad(_rm_linear(d), _rm_patch(xn, False, n_img, H, W, Cin, P), M, Cout, Cin * P * P, tuple(w.shape), L.F32, layout=L.W_LINEAR,
               P=P, C_other=Cin, device=dev)
    close(dW, w.grad, "fp32")
    xl = x.detach().permute(0, 2, 3, 1).contiguous().to(dev)
    dW = wgrad(_rm_linear(d), _rm_patch(xl, True, n_img, H, W, Cin, P), M, Cout, Cin * P * P, tuple(w.shape), L.F32, layout=L.W_CONV_NHWC,
               P=P, C_other=Cin, device=dev)
    close(dW, w.grad, "fp32")


# ---------------------------------------------------------------------------------------------------
# dropout (train mode): statistics of the mask and exact consistency of forward / backward masks
# ---------------------------------------------------------------------------------------------------
def test_dropout_add_statistics_and_backward(dev):
    from tante_amd import _lib as L
    n, p, seed = 1 << 20, 0.1, 12345
    y = torch.ones(n, device=dev)
    res = torch.zeros(n, device=dev)
    out = torch.empty(n, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    L.check(L.lib().tante_dropout_add(y.data_ptr(), L.F32, res.data_ptr(), p, seed, n, out.data_ptr(), s))
    kept = out != 0
    rate = float(kept.float().mean())
    assert abs(rate - (1 - p)) < 3e-3, rate                              # keep probability 1 - p (3 sigma ~ 9e-4)
    assert torch.allclose(out[kept], torch.full_like(out[kept], 1 / (1 - p)))   # inverted-dropout scaling
    assert abs(float(out.mean()) - 1.0) < 5e-3                            # expectation preserved
    dy = torch.empty(n, device=dev)
    L.check(L.lib().tante_dropout_bwd(torch.ones(n, device=dev).data_ptr(), p, seed, n, dy.data_ptr(), L.F32, s))
    assert torch.equal(dy != 0, kept)                                     # backward regenerates exactly the forward's mask
    out2 = torch.empty(n, device=dev)
    L.check(L.lib().tante_dropout_add(y.data_ptr(), L.F32, res.data_ptr(), p, seed + 1, n, out2.data_ptr(), s))
    assert float(((out2 != 0) != kept).float().mean()) > 0.1             # another seed, another mask


@pytest.mark.parametrize("causal", [False, True])
def test_attention_dropout_gradient_matches_finite_differences(dev, causal):
    """fp32, fixed seed: d(sum(o * w)) / d(qkv) from the HIP backward against central differences of the HIP forward."""
    import ctypes as Ct
    from tante_amd import _lib as L, kernels as Kk
    torch.manual_seed(0)
    nseq, Lq, C, nh, p, seed = 3, 6, 16, 2, 0.3, 777
    seq = Kk.dense_seq(nseq, Lq)
    qkv = torch.randn(nseq * Lq, 3 * C, device=dev)
    w = torch.randn(nseq * Lq, C, device=dev)
    s = torch.cuda.current_stream().cuda_stream

    def fwd(t):
        o = torch.empty(nseq * Lq, C, device=dev)
        L.check(L.lib().tante_attention_dropout(t.data_ptr(), o.data_ptr(), L.F32, C, nh, Ct.byref(seq), int(causal), p, seed, s))
        return o
    o = fwd(qkv)
    # dropout really happened: o differs from the deterministic attention
    o0 = torch.empty_like(o)
    Kk.attention(qkv, o0, C, nh, seq, causal)
    assert float((o - o0).abs().max()) > 1e-3
    dq = torch.empty_like(qkv)
    L.check(L.lib().tante_attention_bwd(qkv.data_ptr(), w.data_ptr(), dq.data_ptr(), L.F32, C, nh, Ct.byref(seq), int(causal), p, seed, s))
    g = torch.Generator().manual_seed(1)
    eps = 1e-2
    for _ in range(12):
        i, j = int(torch.randint(0, nseq * Lq, (1,), generator=g)), int(torch.randint(0, 3 * C, (1,), generator=g))
        tp, tm = qkv.clone(), qkv.clone()
        tp[i, j] += eps
        tm[i, j] -= eps
        num = float(((fwd(tp) - fwd(tm)) * w).sum()) / (2 * eps)
        assert abs(num - float(dq[i, j])) < 2e-2 * max(1.0, abs(num)), (i, j, num, float(dq[i, j]))


@pytest.mark.parametrize("p", [0.0, 0.25])
@pytest.mark.parametrize("letter,B,T,H,W,causal", [("T", 2, 4, 6, 5, True), ("T", 1, 3, 4, 7, False), ("H", 2, 2, 16, 3, False),
                                                    ("W", 1, 2, 3, 48, False), ("H", 1, 2, 20, 3, True), ("W", 2, 1, 2, 64, False),
                                                    ("L", 3, 1, 6, 6, False), ("W", 1, 3, 5, 32, True), ("T", 3, 9, 2, 2, True)])
def test_attention_bwd_mfma_matches_fp32_kernel(dev, letter, B, T, H, W, causal, p):
    """bf16 attention backward with head dim 32 and L <= 64 runs on the matrix cores (attn_bwd_mfma_kernel): every axis letter's token
    stride pattern, sequence lengths below / at / between / above 16-slot tiles, causal masks, 5 heads (a ragged group of four), with and
    without attention dropout, against the fp32 lane-per-token kernel on the same bf16-rounded operands and the same seed."""
    import ctypes as Ct
    from tante_amd import _lib as L, kernels as Kk
    nh, C = 5, 160
    seq = Kk.make_seq(letter, B, T, H, W)
    n = B * T * H * W
    g = torch.Generator().manual_seed(n + seq.L)
    qkv = torch.randn(n, 3 * C, generator=g).to(torch.bfloat16).to(dev)
    do = torch.randn(n, C, generator=g).to(torch.bfloat16).to(dev)
    s = torch.cuda.current_stream().cuda_stream
    d16 = torch.full((n + 1, 3 * C), float("nan"), dtype=torch.bfloat16, device=dev)
    L.check(L.lib().tante_attention_bwd(qkv.data_ptr(), do.data_ptr(), d16.data_ptr(), L.BF16, C, nh, Ct.byref(seq), int(causal), p, 99, s))
    d32 = torch.empty(n, 3 * C, device=dev)
    q32, g32 = qkv.float(), do.float()
    L.check(L.lib().tante_attention_bwd(q32.data_ptr(), g32.data_ptr(), d32.data_ptr(), L.F32, C, nh, Ct.byref(seq), int(causal), p, 99, s))
    assert torch.isnan(d16[n].float()).all() and torch.isfinite(d16[:n].float()).all()
    for m, name in enumerate(("dq", "dk", "dv")):
        a, b = d16[:n, m * C:(m + 1) * C].float(), d32[:, m * C:(m + 1) * C]
        err = float((a - b).abs().max() / b.abs().max())
        assert err < 1e-2, (name, err)                  # bf16 bar of the round: 1e-2 relative


@pytest.mark.parametrize("p", [0.0, 0.25])
@pytest.mark.parametrize("letter,B,T,H,W,causal", [("T", 2, 4, 6, 5, True), ("T", 1, 3, 4, 7, False), ("H", 2, 2, 16, 3, False),
                                                    ("W", 1, 2, 3, 48, False), ("H", 1, 2, 20, 3, True), ("W", 2, 1, 2, 64, False),
                                                    ("L", 3, 1, 6, 6, False), ("W", 1, 3, 5, 32, True), ("T", 3, 9, 2, 2, True)])
def test_attention_fwd_mfma_matches_fp32_kernel(dev, letter, B, T, H, W, causal, p):
    """The forward counterpart (attn_fwd_mfma_kernel): bf16, head dim 32, L <= 64 on the matrix cores against the fp32 lane-per-token
    kernel on the same bf16-rounded operands, same dropout seed."""
    import ctypes as Ct
    from tante_amd import _lib as L, kernels as Kk
    nh, C = 5, 160
    seq = Kk.make_seq(letter, B, T, H, W)
    n = B * T * H * W
    g = torch.Generator().manual_seed(n + seq.L + 1)
    qkv = torch.randn(n, 3 * C, generator=g).to(torch.bfloat16).to(dev)
    s = torch.cuda.current_stream().cuda_stream
    o16 = torch.full((n + 1, C), float("nan"), dtype=torch.bfloat16, device=dev)
    L.check(L.lib().tante_attention_dropout(qkv.data_ptr(), o16.data_ptr(), L.BF16, C, nh, Ct.byref(seq), int(causal), p, 99, s))
    q32 = qkv.float()
    o32 = torch.empty(n, C, device=dev)
    L.check(L.lib().tante_attention_dropout(q32.data_ptr(), o32.data_ptr(), L.F32, C, nh, Ct.byref(seq), int(causal), p, 99, s))
    assert torch.isnan(o16[n].float()).all() and torch.isfinite(o16[:n].float()).all()
    err = float((o16[:n].float() - o32).abs().max() / o32.abs().max())
    assert err < 1e-2, err


@pytest.mark.parametrize("outer,n,inner", [(3, 4, 1024), (5, 16, 48), (7, 48, 32), (2, 33, 16), (1, 64, 80), (300, 6, 16)])
def test_axis_wgrad_kernel(dev, outer, n, inner):
    """tante_axis_wgrad: dW[a][j] = sum U[o][a][i] V[o][j][i], db[a] = sum U[o][a][i] on (outer, n, inner) fp32 tensors, fresh and
    accumulating, against float64 einsums (fp32 MFMA: 1e-5 of the largest entry)."""
    from tante_amd import _lib as L
    g = torch.Generator().manual_seed(outer * 100 + n)
    U = torch.randn(outer, n, inner, generator=g)
    V = torch.randn(outer, n, inner, generator=g)
    ref_w = torch.einsum("oai,oji->aj", U.double(), V.double())
    ref_b = U.double().sum((0, 2))
    Ud, Vd = U.to(dev), V.to(dev)
    dW = torch.full((n, n), float("nan"), device=dev)
    db = torch.full((n,), float("nan"), device=dev)
    s = torch.cuda.current_stream().cuda_stream
    L.check(L.lib().tante_axis_wgrad(Ud.data_ptr(), Vd.data_ptr(), outer, n, inner, dW.data_ptr(), db.data_ptr(), 0, s))
    assert float((dW.cpu().double() - ref_w).abs().max()) < 1e-5 * float(ref_w.abs().max())
    assert float((db.cpu().double() - ref_b).abs().max()) < 1e-5 * float(ref_b.abs().max()) + 1e-5
    L.check(L.lib().tante_axis_wgrad(Ud.data_ptr(), Vd.data_ptr(), outer, n, inner, dW.data_ptr(), db.data_ptr(), 1, s))
    assert float((dW.cpu().double() - 2 * ref_w).abs().max()) < 2e-5 * float(ref_w.abs().max())
    L.check(L.lib().tante_axis_wgrad(Ud.data_ptr(), Vd.data_ptr(), outer, n, inner, dW.data_ptr(), None, 0, s))   # without the bias sum
    assert float((dW.cpu().double() - ref_w).abs().max()) < 1e-5 * float(ref_w.abs().max())


@pytest.mark.parametrize("M,N,K", [(4096, 256, 256), (4500, 200, 512), (8192, 128, 128)])
def test_gemm_training_epilogues(dev, M, N, K):
    """TanteGemm.drop_p (dropout of the product before the residual add) and .dact (times act'(pre)) against the separate kernels they
    replace on the train path: the same mask (seed, element index), the same derivative."""
    from tante_amd import _lib as L, kernels as Kk
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    pre = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev)
    pw = Kk.pack_weight(w, b, L.BF16)
    s = torch.cuda.current_stream().cuda_stream
    p, seed = 0.3, 4242
    # dropout + residual
    fused = torch.full((M + 1, N), float("nan"), device=dev)
    Kk.linear(a, pw, fused, M=M, residual=res, drop_p=p, drop_seed=seed)
    y = torch.empty(M, N, device=dev)
    Kk.linear(a, pw, y, M=M)
    ref = torch.empty(M, N, device=dev)
    L.check(L.lib().tante_dropout_add(y.data_ptr(), L.F32, res.data_ptr(), p, seed, y.numel(), ref.data_ptr(), s))
    assert torch.isnan(fused[M]).all()
    assert torch.equal(fused[:M] == res, ref == res)                       # the very same elements were dropped
    assert float((fused[:M] - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    # activation derivative
    for act in (L.ACT_GELU_TANH, L.ACT_GELU_ERF, L.ACT_RELU):
        d16 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        Kk.linear(a, pw, d16, M=M, dact=pre, dact_kind=act)
        y32 = torch.empty(M, N, device=dev)
        Kk.linear(a, pw, y32, M=M)
        ref2 = torch.empty(M, N, device=dev)
        L.check(L.lib().tante_act_bwd(y32.data_ptr(), L.F32, pre.data_ptr(), L.BF16, ref2.data_ptr(), L.F32, y32.numel(), act, s))
        assert float((d16.float() - ref2).abs().max()) <= 1e-2 * float(ref2.abs().max())
    # shapes the epilogues do not cover are refused, not silently computed without them
    small = torch.empty(64, N, device=dev)
    with pytest.raises(RuntimeError):
        Kk.linear(a[:64], pw, small, M=64, residual=res[:64], drop_p=p, drop_seed=seed)


@pytest.mark.parametrize("N,K,bias", [(768, 256, True), (50, 36, True), (256, 512, False)])
def test_fold_kernels(dev, N, K, bias):
    """tante_fold_fwd / tante_fold_bwd (LayerNorm affine folded into the consumer weight on the train path) against the torch expressions
    W * gamma, b + W @ beta and their autograd."""
    from tante_amd import _lib as L
    g = torch.Generator().manual_seed(N + K)
    W = torch.randn(N, K, generator=g).to(dev).requires_grad_()
    b = torch.randn(N, generator=g).to(dev).requires_grad_() if bias else None
    ga = (1 + 0.3 * torch.randn(K, generator=g)).to(dev).requires_grad_()
    be = (0.3 * torch.randn(K, generator=g)).to(dev).requires_grad_()
    GW, Gb = torch.randn(N, K, generator=g).to(dev), torch.randn(N, generator=g).to(dev)
    We_ref = W * ga[None, :]
    be_ref = (b if bias else 0) + W @ be
    ((We_ref * GW).sum() + (be_ref * Gb).sum()).backward()
    s = torch.cuda.current_stream().cuda_stream
    We, bo = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
    L.check(L.lib().tante_fold_fwd(W.data_ptr(), b.data_ptr() if bias else None, ga.data_ptr(), be.data_ptr(), N, K, We.data_ptr(), bo.data_ptr(), s))
    assert torch.allclose(We, We_ref.detach(), rtol=1e-6, atol=1e-6) and torch.allclose(bo, be_ref.detach(), rtol=1e-5, atol=1e-5)
    dW, db, dg, dbt = torch.ones(N, K, device=dev), torch.ones(N, device=dev), torch.ones(K, device=dev), torch.ones(K, device=dev)
    L.check(L.lib().tante_fold_bwd(GW.data_ptr(), Gb.data_ptr(), W.data_ptr(), ga.data_ptr(), be.data_ptr(), N, K, dW.data_ptr(),
                                   db.data_ptr() if bias else None, dg.data_ptr(), dbt.data_ptr(), s))
    assert torch.allclose(dW - 1, W.grad, rtol=1e-5, atol=1e-5)                      # the kernel ADDS onto what is there
    assert torch.allclose(dg - 1, ga.grad, rtol=1e-4, atol=1e-3) and torch.allclose(dbt - 1, be.grad, rtol=1e-4, atol=1e-3)
    if bias:
        assert torch.allclose(db - 1, b.grad, rtol=1e-6, atol=1e-6)


def test_adaptive_rollout_batched_equals_per_sample(dev):
    """R_Trainer's per-sample loop (out_T = 1.5: one frame per call for every sample) against the same rollout run as one batch."""
    import tante_amd
    g = load_golden("g13_deg_false")
    md = tante_amd.TanteMetadata(n_fields=1, spatial_resolution=(32, 32))
    m = _tante_from(g, dev, in_T=4, dset_metadata=md, taylor_order=2, attn_axes="TH-TW", n_head=2, embed_dim=32, patch_scale=8, dropout=0.0,
                    deg=False)
    fmt = tante_amd.DefaultChannelsFirstFormatter(md)
    gen = torch.Generator().manual_seed(3)
    batch = {"input": torch.randn(3, 4, 32, 32, 1, generator=gen).to(dev), "output": torch.randn(3, 3, 32, 32, 1, generator=gen).to(dev)}
    with torch.no_grad():
        y_b, _, rt_b = tante_amd.rollout_adaptive(m, batch, fmt, 3, 1.5, per_sample=True)
        y_s, _, rt_s = tante_amd.rollout_adaptive(m, batch, fmt, 3, 1.5, per_sample=True, batch_when_equivalent=False)
    assert y_b.shape == y_s.shape == (3, 3, 32, 32, 1)
    close(y_b, y_s.cpu(), "fp32")
    assert abs(float(rt_b.mean()) - float(rt_s.mean())) < 1e-5 and rt_b.numel() == rt_s.numel()
    assert float(rt_s.min()) >= 1.0 and float(rt_s.max()) < 2.0


def test_train_step_with_dropout_runs(dev):
    import tante_amd
    g, m, md = _g9_model(dev)
    for blk in (b for bb in m.blocks for b in bb.blocks):
        blk.p_drop = 0.1
    opt = tante_amd.FlatAdamW(m.parameters(), lr=1e-3)
    fmt = tante_amd.DefaultChannelsFirstFormatter(md)
    batch = {"input": g["inp"].to(dev), "output": g["out"].to(dev)}
    l0 = float(tante_amd.train_step(m, opt, batch, fmt, 4))
    l1 = float(tante_amd.train_step(m, opt, batch, fmt, 4))
    assert math.isfinite(l0) and math.isfinite(l1)
    assert abs(l0 - float(g["loss0"])) < 0.2 * float(g["loss0"])       # dropout perturbs, it does not wreck, the loss
    assert all(torch.isfinite(p).all() for p in m.parameters())


# ---------------------------------------------------------------------------------------------------
# adaptive-dt (deg=False) on the train path: gradients against autograd of the (g13-pinned) oracle
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("out_T", [1.5, 6])
def test_adaptive_dt_gradients_against_oracle(dev, out_T):
    import tante_amd
    from oracle import tante_oracle as O
    g = load_golden("g13_deg_false")
    md = tante_amd.TanteMetadata(n_fields=1, spatial_resolution=(32, 32))
    m = tante_amd.TANTE(in_T=4, dset_metadata=md, taylor_order=2, attn_axes="TH-TW", n_head=2, embed_dim=32, patch_scale=8,
                        dropout=0.0, deg=False).to(dev).train()
    m.load_state_dict(split_prefix(g, "w."))
    cfg = O.TanteCfg(4, 1, (32, 32), taylor_order=2, attn_axes="TH-TW", n_head=2, embed_dim=32, patch_scale=8, deg=False)
    w = {k: v.clone().requires_grad_(True) for k, v in split_prefix(g, "w.").items()}
    x = g["x"]
    y_o, rt_o = O.tante_forward(w, cfg, x, out_T)
    tgt = torch.randn(y_o.shape, generator=torch.Generator().manual_seed(3))
    loss_o = ((y_o - tgt) ** 2).mean() + 0.3 * (rt_o ** 2).sum()
    ks = list(w)
    grads_o = torch.autograd.grad(loss_o, [w[k] for k in ks], allow_unused=True)
    y, rt = m(x.to(dev), out_T)
    assert y.shape == y_o.shape
    close(y, y_o, "fp32")
    close(rt, rt_o, "fp32")
    loss = ((y - tgt.to(dev)) ** 2).mean() + 0.3 * (rt ** 2).sum()
    loss.backward()
    named = dict(m.named_parameters())
    for k, go in zip(ks, grads_o):
        if go is None:
            continue
        assert named[k].grad is not None, k
        assert max_rel(named[k].grad.cpu(), go) < 3e-4, (k, max_rel(named[k].grad.cpu(), go))


def test_adaptive_train_step_runs(dev):
    """R_Trainer-style step: per-sample rollouts with out_T = 1.5, MSE + eval_rt, clip_grad_value_, AdamW."""
    import tante_amd
    g = load_golden("g13_deg_false")
    md = tante_amd.TanteMetadata(n_fields=1, spatial_resolution=(32, 32))
    m = tante_amd.TANTE(in_T=4, dset_metadata=md, taylor_order=2, attn_axes="TH-TW", n_head=2, embed_dim=32, patch_scale=8,
                        dropout=0.0, deg=False).to(dev).train()
    m.load_state_dict(split_prefix(g, "w."))
    opt = tante_amd.FlatAdamW(m.parameters(), lr=1e-3)
    gen = torch.Generator().manual_seed(0)
    batch = {"input": torch.randn(2, 4, 32, 32, 1, generator=gen).to(dev), "output": torch.randn(2, 3, 32, 32, 1, generator=gen).to(dev)}
    fmt = tante_amd.DefaultChannelsFirstFormatter(md)
    w_before = opt.flat_p.clone()
    loss, rts = tante_amd.train_step_adaptive(m, opt, batch, fmt, 3)
    assert math.isfinite(float(loss)) and rts.numel() >= 2 and bool(((rts >= 1.0) & (rts <= 1.502)).all())
    assert float((opt.flat_p - w_before).abs().max()) > 0          # the step moved the weights
    assert float(opt.flat_g.abs().max()) <= 1.0 + 1e-6              # clip_grad_value_(1.0) was applied


# ---------------------------------------------------------------------------------------------------
# fused derivative head + Taylor accumulation (bf16)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("C,D,B,Hp,Wp,n_out", [(256, 11, 2, 4, 6, 1), (128, 3, 3, 5, 3, 3), (256, 16, 1, 3, 7, 2), (128, 1, 2, 8, 8, 8),
                                                (256, 4, 7, 32, 32, 1), (128, 2, 8, 30, 30, 2)])     # the last two: >= 7168 rows = the 8-wave form, one ragged
def test_fused_head_against_oracle(dev, C, D, B, Hp, Wp, n_out):
    import tante_amd
    from oracle import tante_oracle as O
    from tante_amd import kernels as Kk
    torch.manual_seed(C + D)
    md = tante_amd.TanteMetadata(n_fields=D, spatial_resolution=(Hp * 8, Wp * 8))
    decs = [tante_amd.dec_CNN(md, embed_dim=C, patch_scale=8, overlap_ratio=0.0).to(dev) for _ in range(2)]
    T, HW = 3, Hp * Wp
    x = torch.randn(B, T, Hp, Wp, C)
    last = torch.randn(B, 1, D, Hp * 8, Wp * 8)
    dt = 0.5
    ref = last.clone().repeat(1, n_out, 1, 1, 1)
    for k, dec in enumerate(decs):
        d = O.dec_cnn({n: v.detach().cpu() for n, v in dec.state_dict().items()}, x[:, -1:], 8, 0.0)
        for i in range(n_out):
            ref[:, i:i + 1] += d * O.taylor_coeff(i + 1, dt, k + 1)
    xd, lastd = x.to(dev).contiguous(), last.to(dev).contiguous()
    out = torch.full((B, n_out, D, Hp * 8, Wp * 8), float("nan"), device=dev)
    frame = D * Hp * 8 * Wp * 8
    for k, dec in enumerate(decs):
        coefs = [O.taylor_coeff(i + 1, dt, k + 1) for i in range(n_out)]
        Kk.head_fused(xd, HW, T * HW * C, C, (T - 1) * HW * C, B, Hp, Wp, C, D, dec.packed_head(), out, out.stride(0), coefs,
                      lastd if k == 0 else None, 0, frame)
    close(out, ref, "bf16")
    # the derivative part alone (what the head computes) also holds the bf16 bar
    assert rel_err((out.cpu() - last), (ref - last)) < 2e-2


# ---- spectral operator path (fixtures g12) -----------------------------------------------------------------
@pytest.mark.parametrize("name", ["low", "clip", "odd"])
def test_g12_spectral_layer(dev, name):
    from tante_amd.spectral import SpectralLayer
    g = load_golden("g12_spectral_" + name)
    m1, m2 = (int(v) for v in g["modes"])
    cin, cout = g["w.weight_re"].shape[:2]
    m = SpectralLayer(cin, cout, m1, m2).to(dev)
    with torch.no_grad():
        m.weight.copy_(torch.complex(g["w.weight_re"], g["w.weight_im"]))
        m.w0.weight.copy_(g["w.w0.weight"])
        m.w0.bias.copy_(g["w.w0.bias"])
        y = m(g["x"].to(dev))
    close(y, g["y"], "fp32")
    # against the oracle on a fresh, larger random case (hipFFT vs the CPU FFT)
    from oracle import spectral_oracle as OS
    torch.manual_seed(5)
    x = torch.randn(3, cin, 48, 40)
    w = {"weight": m.weight.detach().cpu(), "w0.weight": m.w0.weight.detach().cpu(), "w0.bias": m.w0.bias.detach().cpu()}
    with torch.no_grad():
        close(m(x.to(dev)), OS.spectral_layer(w, x, m1, m2), "fp32")


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_g12_tante_fno(dev, mode):
    import tante_amd
    g = load_golden("g12_tante_fno")
    md = tante_amd.TanteMetadata(n_fields=2, spatial_resolution=(32, 32))
    m = tante_amd.TANTE(in_T=4, dset_metadata=md, taylor_order=2, attn_axes="TL-TL", n_head=4, embed_dim=64, patch_scale=8,
                        enc_dec_type="fno", modes1=8, modes2=8).to(dev).eval()
    sd = {}
    for k, v in split_prefix(g, "w.").items():
        if k.endswith("_re"):
            sd[k[:-3]] = torch.complex(v, g["w." + k[:-3] + "_im"])
        elif not k.endswith("_im"):
            sd[k] = v
    m.load_state_dict(sd, strict=True)
    m.set_compute(mode)
    with torch.no_grad():
        close(m(g["x"].to(dev)), g["y"], mode)


# ---- CViT (fixtures g11) --------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ["grid", "gridwide", "fourier", "mlp"])
def test_g11_cvit(dev, name, mode):
    import tante_amd
    from test_oracle_golden import CVIT_CASES
    g = load_golden("g11_cvit_" + name)
    kw, nf, res = CVIT_CASES[name]
    base = dict(out_steps=3, patch_size=(1, 8, 8), grid_size=(8, 8), latent_dim=24, emb_dim=32, depth=2, num_heads=4, dec_emb_dim=48,
                dec_num_heads=4, dec_depth=1, num_mlp_layers=1, mlp_ratio=1)
    base.update(kw)
    m = tante_amd.CViT(4, tante_amd.TanteMetadata(n_fields=nf, spatial_resolution=res), **base).to(dev).eval()
    m.load_state_dict(split_prefix(g, "w."), strict=True)
    m.set_compute(mode)
    coords = g.get("coords")
    with torch.no_grad():
        y = m(g["x"].to(dev)) if coords is None else m(g["x"].to(dev), coords.to(dev))
    close(y, g["y"], mode)


def test_cvit_grid_embed_sparse_equals_dense(dev):
    """The ordered-compaction grid embedding against the dense definition (oracle) at the shipped sharpness eps = 1e5 on a
    64 x 64 latent grid with off-node queries, and with a wide kernel that overflows the compaction list."""
    from tante_amd import kernels as K
    torch.manual_seed(3)
    n_x = n_y = 64
    xs, ys = torch.meshgrid(torch.linspace(0, 1, n_x), torch.linspace(0, 1, n_y), indexing="ij")
    grid = torch.stack([xs.flatten(), ys.flatten()], -1)
    lat = torch.randn(n_x * n_y, 40)
    coords = torch.rand(500, 2)
    for eps in (1e5, 3.0):
        d2 = ((coords[:, None, :] - grid[None]) ** 2).sum(2)
        e = torch.exp(-eps * d2)
        ref = (e / e.sum(1, keepdim=True)) @ lat
        out = K.grid_embed(coords.to(dev), grid.to(dev), lat.to(dev), eps)
        close(out, ref, "fp32")


def test_cvit_cfg4_full_size_properties(dev):
    """cfg4 at its full size (configs/cvit_rb.yaml: 512 x 128 queries, 128 x 128 latent grid, width 512, depth 10): too large for
    the oracle in seconds, so check size-independent properties: (a) query-point mode at a subset of the grid nodes reproduces the
    full-grid prediction at those pixels; (b) a smaller slice of the same model agrees with the oracle end to end."""
    import tante_amd
    from oracle import cvit_oracle as OC
    cfg = tante_amd.load_config(os.path.join(os.path.dirname(GOLDEN), "..", "configs", "cvit_rb.yaml"))
    wl = cfg["workload"]
    H, W = wl["spatial_resolution"]
    md = tante_amd.TanteMetadata(n_fields=wl["n_fields"], spatial_resolution=(H, W))
    torch.manual_seed(211)
    m = tante_amd.build_model(cfg, md).to(dev).eval().set_compute("bf16")
    x = torch.randn(1, 4, wl["n_fields"], H, W, device=dev)
    with torch.no_grad():
        y = m(x)                                                     # (1, 4, 4, 512, 128)
        assert y.shape == (1, 4, wl["n_fields"], H, W) and torch.isfinite(y).all()
        idx = torch.randint(0, H * W, (777,), device=dev)
        coords = tante_amd.cvit.generate_coords(H, W, dev)[idx]
        yq = m(x, coords)                                            # (1, 4, 777, 4)
    full = y.permute(0, 1, 3, 4, 2).reshape(1, 4, H * W, -1)[:, :, idx]
    close(yq, full, "bf16")
    # (b) fp32 against the oracle on a model of the same width at 64 x 32 with an 16 x 16 latent grid
    torch.manual_seed(5)
    kw = dict(out_steps=2, patch_size=(1, 16, 16), grid_size=(16, 16), latent_dim=64, emb_dim=128, depth=2, num_heads=8, dec_emb_dim=128,
              dec_num_heads=8, dec_depth=1, num_mlp_layers=1, mlp_ratio=1, eps=300.0)   # (the default 1e5 underflows off the nodes)
    ms = tante_amd.CViT(4, tante_amd.TanteMetadata(n_fields=4, spatial_resolution=(64, 32)), **kw).to(dev).eval().set_compute("fp32")
    xs = torch.randn(2, 4, 4, 64, 32)
    w = {k: v.detach().cpu() for k, v in ms.state_dict().items()}
    with torch.no_grad():
        close(ms(xs.to(dev)), OC.cvit_forward(w, OC.CvitCfg(4, 4, (64, 32), **kw), xs), "fp32")


@pytest.mark.parametrize("nb,nh,Lq,Lk", [(2, 3, 777, 256), (1, 2, 40, 130), (1, 1, 1000, 512), (3, 2, 5, 4), (1, 8, 2048, 256)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_cross_attention_kernels(dev, nb, nh, Lq, Lk, dtype):
    """tante_cross_attention, head dim 64: the exact VALU kernel (fp32) and the MFMA flash kernel (bf16), incl. key tails that are
    not a multiple of the 128-key chunk, packed (k | v) buffers and query tails."""
    from tante_amd import kernels as K
    torch.manual_seed(Lq + Lk)
    D = 64
    C_ = nh * D
    q = torch.randn(nb * Lq, C_)
    kv = torch.randn(nb * Lk, 2 * C_)
    qd, kvd = q.to(dev, dtype), kv.to(dev, dtype)
    o = torch.empty(nb * Lq, C_, dtype=dtype, device=dev)
    K.cross_attention(qd, kvd, kvd[:, C_:], o, nb, nh, D, Lq, Lk, C_, 2 * C_, C_)
    qf, kvf = qd.float().cpu(), kvd.float().cpu()
    qh = qf.view(nb, Lq, nh, D).transpose(1, 2)
    kh = kvf[:, :C_].reshape(nb, Lk, nh, D).transpose(1, 2)
    vh = kvf[:, C_:].reshape(nb, Lk, nh, D).transpose(1, 2)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(D), -1) @ vh).transpose(1, 2).reshape(nb * Lq, C_)
    close(o, ref, "fp32" if dtype == torch.float32 else "bf16")


def test_flat_adamw_state_dict_interchanges_with_torch_adamw(dev):
    """A checkpoint written after FlatAdamW steps resumes in torch.optim.AdamW (and back) with identical subsequent updates."""
    import tante_amd
    torch.manual_seed(1)
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3)).to(dev)
    ref = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3)).to(dev)
    ref.load_state_dict(net.state_dict())
    opt = tante_amd.FlatAdamW(net.parameters(), lr=1e-2, weight_decay=1e-2, max_norm=0.0)
    x = torch.randn(16, 7, device=dev)
    for _ in range(2):
        opt.zero_grad()
        net(x).square().mean().backward()
        opt.step()
    topt = torch.optim.AdamW(ref.parameters(), lr=1e-2, weight_decay=1e-2)
    ref.load_state_dict({k: v.clone() for k, v in net.state_dict().items()})
    topt.load_state_dict(opt.state_dict())
    opt2 = tante_amd.FlatAdamW(net.parameters(), lr=1e-2, weight_decay=1e-2, max_norm=0.0)
    opt2.load_state_dict(topt.state_dict())
    for o, mdl in ((topt, ref), (opt2, net)):
        o.zero_grad()
        mdl(x).square().mean().backward()
        o.step()
    for a, b in zip(net.parameters(), ref.parameters()):
        close(a, b, "fp32")


def test_cvit_chunked_query_rollout_equals_full_grid(dev):
    """Evaler.rollout_cvit semantics (query chunks + reassembly + re-feed) against the model's own full-grid forward."""
    import tante_amd
    from tante_amd import harness as Hn
    torch.manual_seed(2)
    md = tante_amd.TanteMetadata(n_fields=2, spatial_resolution=(16, 24))
    m = tante_amd.CViT(4, md, out_steps=4, patch_size=(1, 8, 8), grid_size=(16, 24), latent_dim=24, emb_dim=32, depth=1, num_heads=4,
                       dec_emb_dim=32, dec_num_heads=4).to(dev).eval()
    fmt = tante_amd.DefaultChannelsFirstFormatter(md)
    batch = {"input": torch.randn(2, 4, 16, 24, 2), "output": torch.randn(2, 6, 16, 24, 2)}
    with torch.no_grad():
        y, y_ref = Hn.rollout_cvit_eval(m, batch, fmt, n_steps=6, num_query_points=100, device=dev)
        x0 = fmt.process_input(batch)[0][0].to(dev)
        f1 = m(x0)
        f2 = m(torch.cat([x0[:, 4:], f1], 1)[:, -4:])
    want = fmt.process_output(torch.cat([f1, f2], 1))[:, :6]
    assert y.shape == (2, 6, 16, 24, 2) and y_ref.shape == (2, 6, 16, 24, 2)
    close(y, want, "fp32")


def test_fused_encoder_stages_against_oracle_and_unfused(dev):
    """enc23_kernel (stages 2 + 3 + FiLM / positional epilogue in one launch) against the oracle and against the three-GEMM route,
    on a non-square token grid with a ragged last workgroup."""
    import tante_amd
    from tante_amd import _lib as L
    from oracle import tante_oracle as O
    torch.manual_seed(11)
    md = tante_amd.TanteMetadata(n_fields=3, spatial_resolution=(40, 72))          # Hp, Wp = 5, 9 -> 45 tokens per frame
    m = tante_amd.TANTE(in_T=3, dset_metadata=md, taylor_order=1, attn_axes="T", n_head=8, embed_dim=256, patch_scale=8).to(dev).eval()
    x = torch.randn(2, 3, 3, 40, 72, device=dev)
    fa, fb = m._time_tables()
    film = (fa, fb, m.s_emb.view(45, 256), 3, 45)
    with torch.no_grad():
        fused = m.encoder.forward_tokens(x, L.BF16, film)
        m.encoder.fused = False
        plain = m.encoder.forward_tokens(x, L.BF16, film)
    w = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    cfg = O.TanteCfg(3, 3, (40, 72), taylor_order=1, attn_axes="T", n_head=8, embed_dim=256, patch_scale=8)
    want = O.tante_embed(w, cfg, x.cpu()).reshape(-1, 256)
    close(fused, want, "bf16")
    close(fused, plain, "bf16")


@pytest.mark.parametrize("R,I,J", [(4096, 128, 128), (2080, 384, 256), (24576, 256, 768), (96, 128, 256)])
def test_wgrad_dma_transposed_read_path(dev, R, I, J):
    """wgrad_tr_kernel (LDS-DMA ring + ds_read_b64_tr_b16 operand reads): bf16 dense rows, I and J multiples of 128, R a multiple of 32;
    incl. a row count that leaves a short last split, fused bias gradient, and accumulation onto an existing gradient."""
    from tante_amd.autograd import wgrad, _rm_linear
    from tante_amd import _lib as L
    g = torch.Generator().manual_seed(R + I + J)
    U = torch.randn(R, I, generator=g).to(torch.bfloat16)
    V = torch.randn(R, J, generator=g).to(torch.bfloat16)
    ref, refb = U.float().t() @ V.float(), U.float().sum(0)
    Ud, Vd = U.to(dev), V.to(dev)
    dW, db = wgrad(_rm_linear(Ud), _rm_linear(Vd), R, I, J, (I, J), L.BF16, device=dev, with_bias=True)
    close(dW, ref, "bf16", scale=0.2)       # the products are exact in fp32: only the summation order differs
    close(db, refb, "bf16", scale=0.2)
    base, baseb = torch.randn(I, J, generator=g), torch.randn(I, generator=g)
    accW, accb = base.to(dev).clone(), baseb.to(dev).clone()
    wgrad(_rm_linear(Ud), _rm_linear(Vd), R, I, J, (I, J), L.BF16, device=dev, with_bias=True, into=accW, db_into=accb)
    close(accW, ref + base, "bf16", scale=0.2)
    close(accb, refb + baseb, "bf16", scale=0.2)


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ["grid", "gridwide", "fourier", "mlp"])
def test_cvit_gradients_against_oracle_autograd(dev, name, mode):
    """CViT training path: every parameter's gradient of a random linear functional of the output, HIP backward kernels
    (cross-attention, LayerNorm-affine, grid embedding, dgrad / wgrad GEMMs) against torch.autograd through the CPU oracle."""
    import tante_amd
    from test_oracle_golden import CVIT_CASES
    from oracle import cvit_oracle as OC
    g = load_golden("g11_cvit_" + name)
    kw, nf, res = CVIT_CASES[name]
    base = dict(out_steps=3, patch_size=(1, 8, 8), grid_size=(8, 8), latent_dim=24, emb_dim=32, depth=2, num_heads=4, dec_emb_dim=48,
                dec_num_heads=4, dec_depth=1, num_mlp_layers=1, mlp_ratio=1)
    base.update(kw)
    m = tante_amd.CViT(4, tante_amd.TanteMetadata(n_fields=nf, spatial_resolution=res), **base).to(dev).train()
    m.load_state_dict(split_prefix(g, "w."), strict=True)
    m.set_compute(mode)
    coords = g.get("coords")
    x = g["x"]
    torch.manual_seed(7)
    proj = torch.randn(g["y"].shape)
    y = m(x.to(dev)) if coords is None else m(x.to(dev), coords.to(dev))
    close(y, g["y"], mode)
    (y * proj.to(dev)).sum().backward()
    w = {k: v.clone().requires_grad_(True) for k, v in split_prefix(g, "w.").items()}
    yo = OC.cvit_forward(w, OC.CvitCfg(4, nf, res, **base), x, coords)
    (yo * proj).sum().backward()
    # fp32: every parameter to 2e-4.  bf16: matrices to 4e-2; vectors (biases, LayerNorm affine: sums with heavy cancellation over
    # bf16-rounded activations) to 0.25 each, and the whole gradient (all parameters concatenated) to 4e-2.
    tol = 2e-4 if mode == "fp32" else 4e-2
    mine, refs = [], []
    for k, p in m.named_parameters():
        ref = w[k].grad
        assert p.grad is not None and ref is not None, k
        mine.append(p.grad.detach().cpu().flatten())
        refs.append(ref.detach().flatten())
        if float(ref.norm()) < 1e-6 * max(1.0, float(w[k].detach().norm())):   # identically ~0 in the reference (e.g. far grid nodes)
            assert float(p.grad.norm()) < 1e-4, k
            continue
        r = rel_err(p.grad.detach().cpu(), ref.detach())
        if mode == "fp32" or ref.dim() >= 2:
            assert r < tol, f"{k}: rel {r:.3e}"
        else:   # small vectors: relative to their own norm or to a sliver of the whole gradient's norm, whichever is larger
            gn = float(torch.cat([t.flatten() for t in (v.grad for v in w.values()) if t is not None]).norm())
            err = float((p.grad.detach().cpu() - ref.detach()).norm())
            assert err < 0.25 * float(ref.norm()) + 2e-3 * gn, f"{k}: err {err:.3e} ref {float(ref.norm()):.3e} total {gn:.3e}"
    assert rel_err(torch.cat(mine), torch.cat(refs)) < tol


def test_linear_fn_long_contraction(dev):
    """LinearFn with K > 512 (CViT's 16 x 16 patch embed at full size): forward chunks and gradients against torch."""
    from tante_amd.autograd import LinearFn
    from tante_amd import _lib as L
    torch.manual_seed(9)
    a = torch.randn(300, 1024)
    W = (torch.randn(96, 1024) * 0.05).requires_grad_(True)
    b = torch.randn(96).requires_grad_(True)
    ref = a @ W.t() + b
    ref.square().sum().backward()
    Wd, bd = W.detach().to(dev).requires_grad_(True), b.detach().to(dev).requires_grad_(True)
    out = LinearFn.apply(a.to(dev), Wd, bd, None, L.F32, torch.float32)
    close(out, ref, "fp32")
    out.square().sum().backward()
    close(Wd.grad, W.grad, "fp32", scale=20.0)
    close(bd.grad, b.grad, "fp32", scale=20.0)


@pytest.mark.parametrize("nq", [0, 64])
def test_cvit_train_steps_reduce_the_loss_and_match_torch_adamw(dev, nq):
    """Two CViT optimisation steps on the HIP path against the same two steps through the oracle + torch.optim.AdamW (full grid and
    random query points)."""
    import tante_amd
    from oracle import cvit_oracle as OC
    kw = dict(out_steps=2, patch_size=(1, 8, 8), grid_size=(16, 16), latent_dim=24, emb_dim=32, depth=1, num_heads=4, dec_emb_dim=32,
              dec_num_heads=4, dec_depth=1, num_mlp_layers=1, mlp_ratio=1, eps=200.0)
    md = tante_amd.TanteMetadata(n_fields=2, spatial_resolution=(16, 16))
    torch.manual_seed(21)
    m = tante_amd.CViT(4, md, **kw).to(dev).train().set_compute("fp32")
    w = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    fmt = tante_amd.DefaultChannelsFirstFormatter(md)
    batch = {"input": torch.randn(2, 4, 16, 16, 2), "output": torch.randn(2, 2, 16, 16, 2)}
    opt = tante_amd.FlatAdamW(m.parameters(), lr=1e-3, weight_decay=1e-5, max_norm=1.0)
    ropt = torch.optim.AdamW(list(w.values()), lr=1e-3, weight_decay=1e-5)
    cfg = OC.CvitCfg(4, 2, (16, 16), **kw)
    losses = []
    for step in range(2):
        gen = torch.Generator(device=dev).manual_seed(100 + step) if nq else None
        losses.append(float(tante_amd.train_step_cvit(m, opt, {k: v.to(dev) for k, v in batch.items()}, fmt, nq, generator=gen)))
        # reference step
        ropt.zero_grad()
        x, y_ref = fmt.process_input(batch)
        if nq:
            gen2 = torch.Generator(device=dev).manual_seed(100 + step)
            coords, y_pts = tante_amd.harness.generate_and_extract_coords(y_ref.to(dev), nq, gen2)
            yo = OC.cvit_forward(w, cfg, x[0], coords.cpu())
            lo = ((yo - y_pts.cpu()) ** 2).mean()
        else:
            yo = OC.cvit_forward(w, cfg, x[0])
            lo = ((fmt.process_output(yo) - y_ref) ** 2).mean()
        lo.backward()
        torch.nn.utils.clip_grad_norm_(list(w.values()), 1.0)
        ropt.step()
        assert abs(losses[-1] - float(lo.detach())) < 1e-4 * max(1.0, float(lo.detach()))
    # AdamW normalises each entry's step to ~lr whatever the gradient's size, so entries whose gradient is ~0 (latent-grid nodes no query
    # reaches: exactly 0 here, 1e-40 through torch) may differ by a full step; everything else must agree closely
    for k, p in m.named_parameters():
        d = (p.detach().cpu() - w[k].detach()).abs()
        assert float(d.max()) <= 2.2 * 1e-3 * 2, k
        assert rel_err(p.detach().cpu(), w[k].detach()) < 1e-2, k
    assert losses[1] < losses[0]


def test_cfg5_spectral_path_full_size_properties(dev):
    """cfg5 (configs/tante_fno.yaml, 512 x 512 x 8 fields, modes 20 x 20) at full size: too large for the oracle in seconds, so
    size-independent properties -- (a) the spectral layer is linear: L(a x + b y) = a L(x) + b L(y) - (a + b - 1) L(0);
    (b) a field whose spectrum lies outside the kept modes passes through the spectral branch as zero (only the 1x1 conv acts);
    (c) the whole model runs a rollout step and agrees between fp32 and bf16 compute to the bf16 bar."""
    import tante_amd
    from tante_amd.spectral import SpectralLayer
    torch.manual_seed(3)
    lay = SpectralLayer(8, 32, 20, 20).to(dev)
    x, y = torch.randn(2, 8, 512, 512, device=dev), torch.randn(2, 8, 512, 512, device=dev)
    with torch.no_grad():
        l0 = lay(torch.zeros_like(x))
        close(lay(0.7 * x - 1.3 * y), 0.7 * lay(x) - 1.3 * lay(y) + 1.6 * l0, "fp32", scale=20.0)
        ii, jj = torch.meshgrid(torch.arange(512, device=dev), torch.arange(512, device=dev), indexing="ij")
        hf = torch.cos(2 * math.pi * (100 * ii + 37 * jj) / 512).expand(2, 8, 512, 512).contiguous()     # mode (100, 37): outside 20 x 20
        conv_only = torch.nn.functional.conv2d(hf, lay.w0.weight, lay.w0.bias)
        close(lay(hf), conv_only, "fp32", scale=20.0)
    cfg = tante_amd.load_config(os.path.join(os.path.dirname(GOLDEN), "..", "configs", "tante_fno.yaml"))
    wl = cfg["workload"]
    md = tante_amd.TanteMetadata(n_fields=wl["n_fields"], spatial_resolution=tuple(wl["spatial_resolution"]))
    m = tante_amd.build_model(cfg, md).to(dev).eval()
    xin = torch.randn(1, 4, 8, 512, 512, device=dev)
    with torch.no_grad():
        y32 = m.set_compute("fp32")(xin)
        y16 = m.set_compute("bf16")(xin)
    assert y32.shape == (1, 1, 8, 512, 512) and torch.isfinite(y32).all()
    close(y16, y32, "bf16")


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_g12_tante_fno_gradients_against_oracle_autograd(dev, mode):
    """Training path of the spectral encoder / decoder (SpectralLayer backward through hipFFT, padded conv via im2col / col2im, padded
    transposed conv via crop + bilinear-resize backward): every parameter's gradient -- incl. the complex spectral weights -- against
    torch.autograd through the CPU oracle."""
    import tante_amd
    from oracle import tante_oracle as O
    g = load_golden("g12_tante_fno")
    md = tante_amd.TanteMetadata(n_fields=2, spatial_resolution=(32, 32))
    m = tante_amd.TANTE(in_T=4, dset_metadata=md, taylor_order=2, attn_axes="TL-TL", n_head=4, embed_dim=64, patch_scale=8,
                        enc_dec_type="fno", modes1=8, modes2=8).to(dev).train()
    sd = {}
    for k, v in split_prefix(g, "w.").items():
        if k.endswith("_re"):
            sd[k[:-3]] = torch.complex(v, g["w." + k[:-3] + "_im"])
        elif not k.endswith("_im"):
            sd[k] = v
    m.load_state_dict(sd, strict=True)
    m.set_compute(mode)
    torch.manual_seed(17)
    proj = torch.randn(g["y"].shape)
    y = m(g["x"].to(dev))
    close(y, g["y"], mode)
    (y * proj.to(dev)).sum().backward()
    w = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    cfg = O.TanteCfg(4, 2, (32, 32), taylor_order=2, attn_axes="TL-TL", n_head=4, embed_dim=64, patch_scale=8, enc_dec_type="fno",
                     modes1=8, modes2=8)
    yo = O.tante_forward(w, cfg, g["x"])
    (yo * proj).sum().backward()
    tol = 2e-4 if mode == "fp32" else 4e-2
    mine, refs = [], []
    for k, p in m.named_parameters():
        ref = w[k].grad
        assert p.grad is not None and ref is not None, k
        a, b = p.grad.detach().cpu(), ref.detach()
        if a.is_complex():
            a, b = torch.view_as_real(a), torch.view_as_real(b)
        mine.append(a.flatten())
        refs.append(b.flatten())
        if mode == "fp32" or b.dim() >= 2:
            assert rel_err(a, b) < tol, f"{k}: rel {rel_err(a, b):.3e}"
    assert rel_err(torch.cat(mine), torch.cat(refs)) < tol


def test_tante_fno_two_train_steps_match_torch_adamw(dev):
    """Two full optimisation steps of TANTE with the spectral encoder / decoder (BPTT over 2 re-fed frames, MSE, global-norm clip, AdamW
    with the complex spectral weights in the flat buckets as (re, im) pairs) against the oracle + torch.optim.AdamW."""
    import tante_amd
    from oracle import tante_oracle as O
    md = tante_amd.TanteMetadata(n_fields=2, spatial_resolution=(32, 32))
    torch.manual_seed(33)
    kw = dict(in_T=4, taylor_order=1, attn_axes="TL", n_head=4, embed_dim=64, patch_scale=8, enc_dec_type="fno", modes1=6, modes2=6)
    m = tante_amd.TANTE(dset_metadata=md, **kw).to(dev).train().set_compute("fp32")
    w = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    fmt = tante_amd.DefaultChannelsFirstFormatter(md)
    batch = {"input": torch.randn(2, 4, 32, 32, 2), "output": torch.randn(2, 2, 32, 32, 2)}
    opt = tante_amd.FlatAdamW(m.parameters(), lr=1e-3, weight_decay=1e-5, max_norm=1.0)
    ropt = torch.optim.AdamW(list(w.values()), lr=1e-3, weight_decay=1e-5)
    cfg = O.TanteCfg(4, 2, (32, 32), taylor_order=1, attn_axes="TL", n_head=4, embed_dim=64, patch_scale=8, enc_dec_type="fno", modes1=6, modes2=6)
    losses = []
    for _ in range(2):
        losses.append(float(tante_amd.train_step(m, opt, {k: v.to(dev) for k, v in batch.items()}, fmt, 2)))
        ropt.zero_grad()
        yo, y_ref = O.rollout(w, cfg, batch, 2)
        lo = ((yo - y_ref) ** 2).mean()
        lo.backward()
        torch.nn.utils.clip_grad_norm_(list(w.values()), 1.0)
        ropt.step()
        assert abs(losses[-1] - float(lo.detach())) < 1e-4 * max(1.0, float(lo.detach()))
    for k, p in m.named_parameters():
        a, b = p.detach().cpu(), w[k].detach()
        if a.is_complex():
            a, b = torch.view_as_real(a), torch.view_as_real(b)
        assert float((a - b).abs().max()) <= 2.2 * 1e-3 * 2, k
        assert rel_err(a, b) < 1e-2, k
    assert losses[1] < losses[0]


def test_cfg3_training_overfits_a_fixed_batch(dev):
    """End-to-end health of the train path at full cfg3 size (bf16, dropout 0.1, 4-step BPTT, clip + AdamW through the flat buckets, every
    fused epilogue / deferred weight-gradient launch of DESIGN 4.3): 25 steps on one smooth batch must cut the loss at least 4x."""
    import os
    import tante_amd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = tante_amd.load_config(os.path.join(root, "configs", "tante_trl.yaml")); wl = cfg["workload"]
    md = tante_amd.TanteMetadata(n_fields=wl["n_fields"], spatial_resolution=tuple(wl["spatial_resolution"]))
    torch.manual_seed(211)
    m = tante_amd.build_model(cfg, md, dropout=0.1).to(dev).train().set_compute("bf16")
    opt = tante_amd.FlatAdamW(m.parameters(), lr=1e-3, weight_decay=1e-5, max_norm=1.0)
    B, n, T = 4, wl["n_steps_output"], wl["n_steps_input"]
    g = torch.Generator().manual_seed(1)
    base = torch.randn(B, 1, *wl["spatial_resolution"], wl["n_fields"], generator=g)
    drift = 0.05 * torch.randn(B, 1, *wl["spatial_resolution"], wl["n_fields"], generator=g)
    frames = torch.cat([base + k * drift for k in range(T + n)], dim=1)
    batch = {"input": frames[:, :T].to(dev), "output": frames[:, T:].to(dev)}
    fmt = tante_amd.DefaultChannelsFirstFormatter(md)
    losses = [float(tante_amd.train_step(m, opt, batch, fmt, n, 1)) for _ in range(25)]
    assert all(l == l for l in losses), losses
    assert losses[-1] < 0.25 * losses[0], (losses[0], losses[-1])


def test_cfg2_rollout_bf16_tracks_fp32_over_eight_steps(dev):
    """The re-fed bf16 rollout (fused kernels, frame-encoding cache) against the fp32 rollout (unfused parity kernels) of the same
    weights: the relative difference of frame k grows about linearly (2e-4 per step measured) and must stay far inside the 1e-2 bar."""
    import tante_amd
    m = _cfg2_model(dev)
    md = tante_amd.TanteMetadata(n_fields=11, spatial_resolution=(256, 256))
    fmt = tante_amd.DefaultChannelsFirstFormatter(md)
    g = torch.Generator().manual_seed(5)
    batch = {"input": torch.randn(2, 4, 256, 256, 11, generator=g).to(dev), "output": torch.randn(2, 8, 256, 256, 11, generator=g).to(dev)}
    with torch.no_grad():
        y32, _ = tante_amd.rollout_model(m.set_compute("fp32"), batch, fmt, 8)
        y16, _ = tante_amd.rollout_model(m.set_compute("bf16"), batch, fmt, 8)
    errs = [float((y16[:, t] - y32[:, t]).norm() / y32[:, t].norm()) for t in range(8)]
    assert all(e == e for e in errs) and errs[-1] < 5e-3 and errs[0] < 1e-3, errs

