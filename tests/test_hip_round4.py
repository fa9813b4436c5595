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
        operands and fp32 accumulation, but stage 3's K = 512 is summed as four 128-deep partials instead of one chain and stage 1's
        k order differs -- fp32 rounding of the sum order, and a bf16 re-rounding of an intermediate where that flips it: 2e-4;
    (3) z against the ORACLE's encoder (fp32 CPU, enc_dec_cnn.py:217-229) on the same frame at the bf16 bar;
    (4) two launches give the same bits (the partials are added in a fixed order whatever the arrival order of the workgroups)."""
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
        monkeypatch.setenv("TANTE_NO_TAIL_ENC", "1")
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
