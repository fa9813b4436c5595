"""CPU ORACLE for the TANTE Taylor-rollout hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-the-maths restatement, in plain torch-CPU fp32 (or fp64) tensor
arithmetic over a flat ``name -> tensor`` weight dictionary, of what the reference
(zwu88/TANTE, mounted read-only at /root/reference in the build container) computes
on its rollout path.  It exists to CHECK the hand-written HIP path; it is never the
thing measured or shipped.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product package
(``tante_amd``) never imports it and has no CPU fallback.

Pinning: every function below is checked in ``tests/test_oracle_golden.py`` against
golden vectors produced by running the reference itself on CPU in the build
container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).  Parity is
therefore PINNED (not "parity unpinned").

Each function cites the reference file:line it follows.  Weight names are the
reference ``state_dict()`` keys.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
W = Dict[str, Tensor]

# models/enc_dec_cnn.py:39-46 -- patch_scale -> per-stage kernel sizes
PATCH_MAP = {64: (4, 4, 4), 32: (4, 4, 2), 16: (4, 2, 2), 8: (2, 2, 2), 4: (2, 2, 1), 2: (2, 1, 1)}


def sub(w: W, prefix: str) -> W:
    """View of the weights below ``prefix`` (prefix stripped)."""
    n = len(prefix)
    return {k[n:]: v for k, v in w.items() if k.startswith(prefix)}


# ----------------------------------------------------------------------------------------
# elementwise helpers
# ----------------------------------------------------------------------------------------
# Two spellings of the same arithmetic.  The default writes every operation out (this is what the parity tests check the HIP path
# against, in fp32 or fp64, with autograd).  FAST swaps the five hottest helpers for torch's fused CPU ops (F.layer_norm, F.gelu,
# F.linear, scaled_dot_product_attention): same maths, one pass over the data instead of 6-8 -- the written-out form runs ~2x slower
# than the reference's nn.Module forward on the same host, which would halve the CPU baseline and flatter the GPU/CPU ratio.
# bench.py's cpu_baseline leg times the FAST form (tools/cpu_reference_time.py holds it to +-10 % of the real reference's time in
# the build container); tests/test_oracle_golden.py pins BOTH forms to the reference's golden vectors.
_FAST = False


def set_fast(on: bool) -> bool:
    global _FAST
    old, _FAST = _FAST, bool(on)
    return old


def gelu_erf(x: Tensor) -> Tensor:
    """nn.GELU() default (exact/erf form): enc_dec_cnn.py:215,260 ; attn_backbone.py:113,116,119."""
    if _FAST:
        return F.gelu(x)
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def gelu_tanh(x: Tensor) -> Tensor:
    """nn.GELU(approximate='tanh'): attn_backbone.py:54."""
    if _FAST:
        return F.gelu(x, approximate="tanh")
    c = math.sqrt(2.0 / math.pi)
    return 0.5 * x * (1.0 + torch.tanh(c * (x + 0.044715 * x * x * x)))


def layer_norm(x: Tensor, g: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    """nn.LayerNorm over the last dim, biased variance (attn_backbone.py:47,50)."""
    if _FAST:
        return F.layer_norm(x, (x.shape[-1],), g, b, eps)
    mu = x.mean(-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdim=True)
    return xc * torch.rsqrt(var + eps) * g + b


def linear(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    if _FAST:
        return F.linear(x, w, b)
    y = x @ w.t()
    return y if b is None else y + b


# ----------------------------------------------------------------------------------------
# patch encoder / derivative-head decoder  (models/enc_dec_cnn.py)
# ----------------------------------------------------------------------------------------
def _stride_pad(P: int, overlap: float) -> Tuple[int, int]:
    """enc_dec_cnn.py:66-81 / 130-146: stride = max(1, round(P (1-overlap))), pad = (P-1)//2."""
    return max(1, int(round(P * (1.0 - overlap)))), (P - 1) // 2


def real_conv2d(x: Tensor, w: Tensor, b: Tensor, P: int, overlap: float) -> Tensor:
    """RealConv2d.forward, enc_dec_cnn.py:97-110: strided conv, then adaptive average pooling to
    exactly (H//P, W//P).  (B, Cin, H, W) -> (B, Cout, H//P, W//P)."""
    s, p = _stride_pad(P, overlap)
    H, Wd = x.shape[-2:]
    assert H % P == 0 and Wd % P == 0
    y = F.conv2d(x, w, b, stride=s, padding=p)
    return _adaptive_avg_pool(y, H // P, Wd // P)


def _adaptive_avg_pool(y: Tensor, th: int, tw: int) -> Tensor:
    """adaptive_avg_pool2d: output cell i averages input rows floor(i*H/th) .. ceil((i+1)*H/th)-1."""
    H, Wd = y.shape[-2:]
    if _FAST:
        return F.adaptive_avg_pool2d(y, (th, tw))     # the reference runs the pool even when it is an identity (enc_dec_cnn.py:99-103): so does the timed form
    if (H, Wd) == (th, tw):
        return y
    rows = []
    for i in range(th):
        h0, h1 = (i * H) // th, -((-(i + 1) * H) // th)
        cols = []
        for j in range(tw):
            w0, w1 = (j * Wd) // tw, -((-(j + 1) * Wd) // tw)
            cols.append(y[..., h0:h1, w0:w1].mean(dim=(-2, -1)))
        rows.append(torch.stack(cols, -1))
    return torch.stack(rows, -2)


def real_transconv2d(x: Tensor, w: Tensor, b: Tensor, P: int, overlap: float) -> Tensor:
    """RealTransConv2d.forward, enc_dec_cnn.py:164-184: transposed conv, then bilinear resize
    (align_corners=False) to exactly (H*P, W*P) when the deconv output has another size."""
    s, p = _stride_pad(P, overlap)
    H, Wd = x.shape[-2:]
    y = F.conv_transpose2d(x, w, b, stride=s, padding=p)
    if y.shape[-2:] != (H * P, Wd * P):
        y = _bilinear_resize(y, H * P, Wd * P)
    return y


def _bilinear_resize(y: Tensor, th: int, tw: int) -> Tensor:
    """F.interpolate(mode='bilinear', align_corners=False): src = (dst + .5) * in/out - .5, clamped at 0."""
    def axis(n_in, n_out):
        d = torch.arange(n_out, dtype=torch.float64)
        srcf = ((d + 0.5) * (n_in / n_out) - 0.5).clamp_(min=0.0)
        i0 = srcf.floor().long().clamp_(max=n_in - 1)
        i1 = (i0 + 1).clamp_(max=n_in - 1)
        lam = (srcf - i0.double()).to(y.dtype)
        return i0, i1, lam
    h0, h1, lh = axis(y.shape[-2], th)
    w0, w1, lw = axis(y.shape[-1], tw)
    top = y[..., h0, :] * (1 - lh)[:, None] + y[..., h1, :] * lh[:, None]
    return top[..., w0] * (1 - lw) + top[..., w1] * lw


def enc_cnn(w: W, x: Tensor, patch_scale: int, overlap: float = 0.0) -> Tensor:
    """enc_CNN.forward, enc_dec_cnn.py:217-229.  (B,T,D,H,W) -> (B,T,Hp,Wp,C); GELU(erf) after
    stages 1 and 2 only."""
    B, T = x.shape[:2]
    P = PATCH_MAP[patch_scale]
    z = x.reshape(B * T, *x.shape[2:])
    for i in (1, 2, 3):
        z = real_conv2d(z, w[f"enc_conv_{i}.conv.weight"], w[f"enc_conv_{i}.conv.bias"], P[i - 1], overlap)
        if i < 3:
            z = gelu_erf(z)
    return z.reshape(B, T, *z.shape[1:]).permute(0, 1, 3, 4, 2)


def dec_cnn(w: W, x: Tensor, patch_scale: int, overlap: float = 0.0) -> Tensor:
    """dec_CNN.forward, enc_dec_cnn.py:263-277.  (B,T,Hp,Wp,C) -> (B,T,D,H,W); stage kernel sizes
    are Patch_map reversed (P[2], P[1], P[0]); GELU(erf) after stages 1 and 2 only."""
    B, T = x.shape[:2]
    P = PATCH_MAP[patch_scale]
    z = x.permute(0, 1, 4, 2, 3).reshape(B * T, x.shape[4], x.shape[2], x.shape[3])
    for i in (1, 2, 3):
        z = real_transconv2d(z, w[f"dec_conv_{i}.deconv.weight"], w[f"dec_conv_{i}.deconv.bias"], P[3 - i], overlap)
        if i < 3:
            z = gelu_erf(z)
    return z.reshape(B, T, *z.shape[1:])


# ----------------------------------------------------------------------------------------
# FiLM time encoding, step interpreter, positional tables  (models/tante.py)
# ----------------------------------------------------------------------------------------
def t_series(in_T: int, frame_interval: float) -> Tensor:
    """tante.py:279-285.  Note the quirk: [0, -0*dt, -1*dt, ...] reversed -> 0 appears twice,
    e.g. t_series(4, 1.0) = [-2, -1, 0, 0]."""
    seq = [0.0] + [-i * frame_interval for i in range(in_T - 1)]
    seq.reverse()
    return torch.tensor(seq)


def _film_mlp(w: W, name: str, t: Tensor) -> Tensor:
    h = torch.relu(linear(t[..., None], w[f"{name}.0.weight"], w[f"{name}.0.bias"]))
    return linear(h, w[f"{name}.2.weight"], w[f"{name}.2.bias"])


def film(w: W, x: Tensor, t: Tensor) -> Tensor:
    """film.forward, tante.py:218-230: returns x + (x*scale(t) + shift(t)).  5-D x: t is (T,),
    broadcast as (1,T,1,1,C); 3-D x: t is (B,), broadcast as (B,1,C)."""
    scale = _film_mlp(w, "condition_to_scale", t)
    shift = _film_mlp(w, "condition_to_shift", t)
    if x.dim() == 5:
        scale, shift = scale[None, :, None, None, :], shift[None, :, None, None, :]
    elif x.dim() == 3:
        scale, shift = scale[:, None, :], shift[:, None, :]
    return x + (x * scale + shift)


def interprator(w: W, x: Tensor, out_T: float, ep: float = 1.001) -> Tensor:
    """interprator.forward, tante.py:191-201.  x (B,L,C) -> per-token scalar via MLP C->C/2->C/4->1
    (ReLU); values clamped into [0, out_T-1] (straight-through in the reference; forward value is the
    clamp); mean over tokens; + ep."""
    h = torch.relu(linear(x, w["interprete.0.weight"], w["interprete.0.bias"]))
    h = torch.relu(linear(h, w["interprete.2.weight"], w["interprete.2.bias"]))
    t = linear(h, w["interprete.4.weight"], w["interprete.4.bias"])[..., 0]
    td = t.detach()
    t = t + torch.relu(-td) - torch.relu(td - (out_T - 1))
    return t.mean(dim=1) + ep


# ----------------------------------------------------------------------------------------
# transformer block and axis-factorised backbone  (models/attn_backbone.py)
# ----------------------------------------------------------------------------------------
def mha_self(w: W, h: Tensor, n_head: int, causal: bool) -> Tensor:
    """nn.MultiheadAttention(batch_first, bias) self-attention as used at attn_backbone.py:74-80:
    packed in_proj (3C,C), heads split along C, scores scaled by 1/sqrt(d_h), bool causal mask
    (True above the diagonal = blocked, attn_backbone.py:35-36), softmax, out_proj."""
    Bp, L, C = h.shape
    d = C // n_head
    qkv = linear(h, w["attn.in_proj_weight"], w["attn.in_proj_bias"])
    q, k, v = (t.reshape(Bp, L, n_head, d).transpose(1, 2) for t in qkv.split(C, dim=-1))
    if _FAST:
        o = F.scaled_dot_product_attention(q, k, v, is_causal=causal).transpose(1, 2).reshape(Bp, L, C)
        return linear(o, w["attn.out_proj.weight"], w["attn.out_proj.bias"])
    s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(d))
    if causal:
        blocked = torch.triu(torch.ones(L, L, dtype=torch.bool), diagonal=1)
        s = s.masked_fill(blocked, float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(Bp, L, C)
    return linear(o, w["attn.out_proj.weight"], w["attn.out_proj.bias"])


def transformer_block(w: W, x: Tensor, n_head: int, causal: bool = False) -> Tensor:
    """TransformerBlock.forward (dropout 0 / eval), attn_backbone.py:59-83:
    x += MHA(LN1(x)); x += W2 gelu_tanh(W1 LN2(x))."""
    x = x + mha_self(w, layer_norm(x, w["ln1.weight"], w["ln1.bias"]), n_head, causal)
    h = layer_norm(x, w["ln2.weight"], w["ln2.bias"])
    h = gelu_tanh(linear(h, w["mlp.0.weight"], w["mlp.0.bias"]))
    return x + linear(h, w["mlp.2.weight"], w["mlp.2.bias"])


def _axis_mlp(w: W, name: str, x: Tensor) -> Tensor:
    """Linear(n,n) -> GELU(erf) -> Linear(n,n) over the LAST dim (attn_backbone.py:111-119)."""
    h = gelu_erf(linear(x, w[f"{name}.0.weight"], w[f"{name}.0.bias"]))
    return linear(h, w[f"{name}.2.weight"], w[f"{name}.2.bias"])


def attn_backbone(w: W, x: Tensor, attn_axes: str, n_head: int) -> Tensor:
    """Attn_Backbone.forward, attn_backbone.py:134-191.  x (B,T,H,W,C).  Three residual axis
    propagators (over H, then W, then T), then one TransformerBlock per letter on a regrouping
    (batch', L, C) of the tokens; T is causal, all others are not."""
    B, T, H, Wd, C = x.shape
    # l.140-146: the MLPs act on the H / W / T axis with every other index (incl. C) as batch
    x = x + _axis_mlp(w, "vertical_propagator", x.permute(0, 1, 3, 4, 2)).permute(0, 1, 4, 2, 3)
    x = x + _axis_mlp(w, "horizontal_propagator", x.permute(0, 1, 2, 4, 3)).permute(0, 1, 2, 4, 3)
    x = x + _axis_mlp(w, "temporal_propagator", x.permute(0, 2, 3, 4, 1)).permute(0, 4, 1, 2, 3)
    ci = 0
    for i, a in enumerate(attn_axes):
        bw = sub(w, f"blocks.{i}.")
        if a == "T":    # (b h w) t c, causal                                         l.149-152
            y = transformer_block(bw, x.permute(0, 2, 3, 1, 4).reshape(B * H * Wd, T, C), n_head, True)
            x = y.reshape(B, H, Wd, T, C).permute(0, 3, 1, 2, 4)
        elif a == "H":  # (b t w) h c                                                 l.154-157
            y = transformer_block(bw, x.permute(0, 1, 3, 2, 4).reshape(B * T * Wd, H, C), n_head)
            x = y.reshape(B, T, Wd, H, C).permute(0, 1, 3, 2, 4)
        elif a == "W":  # (b t h) w c                                                 l.159-162
            x = transformer_block(bw, x.reshape(B * T * H, Wd, C), n_head).reshape(B, T, H, Wd, C)
        elif a == "L":  # (b t) (h w) c                                               l.164-167
            x = transformer_block(bw, x.reshape(B * T, H * Wd, C), n_head).reshape(B, T, H, Wd, C)
        elif a == "Y":  # (b w) (t h) c                                               l.169-172
            y = transformer_block(bw, x.permute(0, 3, 1, 2, 4).reshape(B * Wd, T * H, C), n_head)
            x = y.reshape(B, Wd, T, H, C).permute(0, 2, 3, 1, 4)
        elif a == "X":  # (b h) (t w) c                                               l.174-177
            y = transformer_block(bw, x.permute(0, 2, 1, 3, 4).reshape(B * H, T * Wd, C), n_head)
            x = y.reshape(B, H, T, Wd, C).permute(0, 2, 1, 3, 4)
        elif a == "A":  # b (t h w) c                                                 l.179-182
            x = transformer_block(bw, x.reshape(B, T * H * Wd, C), n_head).reshape(B, T, H, Wd, C)
        elif a == "C":  # every scalar lifted 1 -> E; attention over the C axis; keep E-channel -1   l.184-189
            cw = sub(w, f"channel_blocks.{ci}.")
            ci += 1
            z = x.reshape(B * T * H * Wd, C, 1)
            z = linear(gelu_erf(linear(z, cw["0.weight"], cw["0.bias"])), cw["2.weight"], cw["2.bias"])
            x = transformer_block(bw, z, n_head)[..., -1].reshape(B, T, H, Wd, C)
        else:
            raise ValueError(f"invalid axis letter {a!r}")
    return x


# ----------------------------------------------------------------------------------------
# TANTE forward  (models/tante.py:125-176)
# ----------------------------------------------------------------------------------------
class TanteCfg:
    """The ctor arguments of models.TANTE that shape the arithmetic (tante.py:38-60)."""

    def __init__(self, in_T, n_fields, resolution, taylor_order=1, frame_interval=1.0, output_length=1,
                 attn_axes="THWTHWTHW", expanded_channel=128, n_head=8, mlp_ratio=1.0, embed_dim=256,
                 patch_scale=32, overlap_ratio=0.0, deg=True, enc_dec_type="cnn", modes1=32, modes2=32):
        self.in_T, self.n_fields, self.resolution = in_T, n_fields, tuple(resolution)
        self.taylor_order, self.frame_interval, self.output_length = taylor_order, frame_interval, output_length
        self.attn_axes = attn_axes.replace(" ", "")
        self.blocks_axes = [p.strip() for p in self.attn_axes.split("-")]   # tante.py:79
        if len(self.blocks_axes) != taylor_order:                            # tante.py:80-83
            raise ValueError("Block allocation doesn't match expansion order")
        self.expanded_channel, self.n_head, self.mlp_ratio = expanded_channel, n_head, mlp_ratio
        self.embed_dim, self.patch_scale, self.overlap_ratio, self.deg = embed_dim, patch_scale, overlap_ratio, deg
        self.enc_dec_type, self.modes = enc_dec_type, (modes1, modes2)   # tante.py:95-102: 'fno' swaps in enc_FNO / dec_FNO


def taylor_coeff(i: int, frame_interval: float, order: int) -> float:
    """(i * dt)^k / k!  -- tante.py:168."""
    return (i * frame_interval) ** order / math.factorial(order)


def tante_embed(w: W, cfg: TanteCfg, inp: Tensor) -> Tensor:
    """tante.py:132-141: encoder -> FiLM(t_seq) -> + s_emb (1,Hp,Wp,C) -> + t_emb (1,T,C)."""
    if cfg.enc_dec_type == "fno":
        from .spectral_oracle import enc_fno
        x = enc_fno(sub(w, "encoder."), inp, cfg.patch_scale, cfg.overlap_ratio, cfg.modes)
    else:
        x = enc_cnn(sub(w, "encoder."), inp, cfg.patch_scale, cfg.overlap_ratio)
    x = film(sub(w, "t_encode."), x, t_series(cfg.in_T, cfg.frame_interval).to(x.dtype))
    return x + w["s_emb"][:, None] + w["t_emb"][:, :, None, None, :]


def tante_forward(w: W, cfg: TanteCfg, inp: Tensor, out_T: float = 1):
    """TANTE.forward, tante.py:125-176.  inp (B, T>=in_T, D, H, W).  Returns (B, n_out, D, H, W), and
    additionally R_t (B,) when deg=False.

    deg=False follows the evidently intended semantics (fixture g13): the reference
    glue at tante.py:149-152 raises as shipped; intended is d3 = 'b 1 h w c -> b (h w) c',
    rt = interprator(d3, out_T), d3 = film3d(d3, rt), back to 5-D, decode."""
    if inp.shape[1] != cfg.in_T:
        inp = inp[:, -cfg.in_T:]
    x = tante_embed(w, cfg, inp)
    B, T, Hp, Wp, C = x.shape
    ders: List[Tensor] = []
    rts: List[Tensor] = []
    for i, axes in enumerate(cfg.blocks_axes):
        x = attn_backbone(sub(w, f"blocks.{i}."), x, axes, cfg.n_head)   # chained: order i+1 sees order i
        d = x[:, -1:]
        if not cfg.deg:
            d3 = d.reshape(B, Hp * Wp, C)
            rt = interprator(sub(w, f"interprators.{i}."), d3, out_T)
            rts.append(rt)
            d = film(sub(w, f"modifiers.{i}."), d3, rt).reshape(B, 1, Hp, Wp, C)
        if cfg.enc_dec_type == "fno":
            from .spectral_oracle import dec_fno
            ders.append(dec_fno(sub(w, f"decoders.{i}."), d, cfg.patch_scale, cfg.overlap_ratio, cfg.modes))
        else:
            ders.append(dec_cnn(sub(w, f"decoders.{i}."), d, cfg.patch_scale, cfg.overlap_ratio))
    if cfg.deg:
        n_out, R_t = cfg.output_length, None
    else:
        R_t = torch.stack(rts, dim=1).mean(dim=1)
        n_out = math.floor(float(R_t[0]))        # tante.py:163 -- sample 0 decides for the batch
    last = inp[:, -1:]
    outs = []
    for i in range(1, n_out + 1):
        o = last
        for k in range(1, cfg.taylor_order + 1):
            o = o + ders[k - 1] * taylor_coeff(i, cfg.frame_interval, k)
        outs.append(o)
    y = torch.cat(outs, dim=1)
    return y if cfg.deg else (y, R_t)


# ----------------------------------------------------------------------------------------
# rollout harness semantics  (trainer/*.py, data/datamodule.py)
# ----------------------------------------------------------------------------------------
def format_input(batch: Dict[str, Tensor]) -> Tuple[Tensor, Tensor]:
    """DefaultChannelsFirstFormatter.process_input, datamodule.py:185-189: 'b t h w c -> b t c h w'
    + nan_to_num on input and reference."""
    x = torch.nan_to_num(batch["input"].permute(0, 1, 4, 2, 3))
    return x, torch.nan_to_num(batch["output"])


def rollout(w: W, cfg: TanteCfg, batch: Dict[str, Tensor], n_steps: int) -> Tuple[Tensor, Tensor]:
    """Trainer.rollout_model / Evaler.rollout_model, trainer.py:144-159 ; evaler.py:121-138.
    Sliding-window re-feed; prediction returned channels-last and truncated to n_steps."""
    x, y_ref = format_input(batch)
    preds, produced = [], 0
    while produced < n_steps:
        y = tante_forward(w, cfg, x)
        produced += y.shape[1]
        if produced < n_steps:
            x = torch.cat([x[:, y.shape[1]:], y], dim=1)
        preds.append(y.permute(0, 1, 3, 4, 2))
    return torch.cat(preds, dim=1)[:, :n_steps], y_ref


def rollout_adaptive(w: W, cfg: TanteCfg, batch: Dict[str, Tensor], n_steps: int, out_T: float,
                     per_sample: bool) -> Tuple[Tensor, Tensor, Tensor]:
    """R_Trainer.rollout_model (per_sample=True, out_T=1.5: r_trainer.py:112-133) and
    R_Evaler.rollout_model (per_sample=False, out_T=n_steps_rollout: r_evaler.py:87-105)."""
    x_all, y_ref = format_input(batch)
    chunks = [x_all[i:i + 1] for i in range(x_all.shape[0])] if per_sample else [x_all]
    rts, ys = [], []
    for x in chunks:
        preds, produced = [], 0
        while produced < n_steps:
            y, rt = tante_forward(w, cfg, x, out_T)
            produced += y.shape[1]
            if produced < n_steps:
                x = torch.cat([x[:, y.shape[1]:], y], dim=1)
            preds.append(y.permute(0, 1, 3, 4, 2))
            rts.append(rt)
        ys.append(torch.cat(preds, dim=1)[:, :n_steps])
    return torch.cat(ys, dim=0), y_ref, torch.cat(rts, dim=0)


# ----------------------------------------------------------------------------------------
# losses / metrics  (trainer/metrics.py) -- channels-last (B,T,H,W,C)
# ----------------------------------------------------------------------------------------
def mse(x: Tensor, y: Tensor) -> Tensor:
    """MSE.eval, metrics.py:53-60: mean over (H,W) -> (B,T,C)."""
    return ((x - y) ** 2).mean(dim=(-3, -2))


def eval_rt(rt: Tensor, eps: float = 0.5, n: float = 2.0):
    """MSE.eval_rt, metrics.py:62-80: step-size band regulariser."""
    r = rt.mean()
    up, down = min(1 + eps, 4), max(1 + eps, 4)
    loss = 0.0
    if r < up:
        loss = loss + 5e-3 * (up - r) ** n
    if r > down:
        loss = loss + 1e-1 * (r - down) ** n
    return loss


def mse_with_rt(x: Tensor, y: Tensor, rt: Optional[Tensor], eps: float = 0.5, n: float = 2.0):
    """Metric.forward, metrics.py:20-43."""
    if rt is None:
        return mse(x, y)
    return mse(x, y).mean() + eval_rt(rt, eps, n)


def l2re(x: Tensor, y: Tensor, eps: float = 1e-7) -> Tensor:
    """L2RE.eval, metrics.py:100-111: ||x-y||_2 / (||y||_2 + eps) over (T,H,W) -> (B,C)."""
    B, C = x.shape[0], x.shape[-1]
    d = (x - y).reshape(B, -1, C)
    return torch.sqrt((d * d).sum(1)) / (torch.sqrt((y.reshape(B, -1, C) ** 2).sum(1)) + eps)


def _norm(y: Tensor, dims, mode: str) -> Tensor:
    if mode == "norm":
        return (y * y).mean(dim=dims)
    return y.var(dim=dims, unbiased=True)        # torch.std(...)**2, metrics.py:94,126


def nmse(x, y, eps=1e-7, mode="norm"):
    """NMSE.eval, metrics.py:82-98."""
    return mse(x, y) / (_norm(y, (-3, -2), mode) + eps)


def nnmse(x, y, eps=1e-7, mode="norm"):
    """NNMSE.eval, metrics.py:114-130: normaliser over (H,W,C); numerator mean_C(MSE)."""
    return mse(x, y).mean(dim=-1) / (_norm(y, (-3, -2, -1), mode) + eps)


def vrmse(x, y, eps=1e-7):
    """VRMSE.eval = sqrt(NMSE(norm_mode='std')), metrics.py:140-164."""
    return torch.sqrt(nmse(x, y, eps, "std"))


# ----------------------------------------------------------------------------------------
# optimiser step / LR schedule  (trainer/trainer.py:191-196 ; optim/schedulers.py)
# ----------------------------------------------------------------------------------------
def clip_grad_norm(grads: Sequence[Tensor], max_norm: float = 1.0) -> Tuple[List[Tensor], Tensor]:
    """torch.nn.utils.clip_grad_norm_: total 2-norm; scale by min(1, max_norm/(norm+1e-6))."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return [g * coef for g in grads], total


def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, wd: float,
               b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8) -> Tuple[Tensor, Tensor, Tensor]:
    """torch.optim.AdamW (decoupled weight decay), one parameter, step counted from 1."""
    p = p * (1.0 - lr * wd)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    denom = torch.sqrt(v) / math.sqrt(1 - b2 ** step) + eps
    return p - (lr / (1 - b1 ** step)) * m / denom, m, v


def warmup_cosine_lr(epoch: int, base_lr: float, warmup_epochs: int, max_epochs: int,
                     warmup_start_lr: float, eta_min: float) -> float:
    """LinearWarmupCosineAnnealingLR in closed form (schedulers.py:97-123); the reference steps the
    chainable form (l.50-95) once per epoch, which evaluates to the same value up to rounding."""
    if epoch < warmup_epochs:
        return warmup_start_lr + epoch * (base_lr - warmup_start_lr) / max(1, warmup_epochs - 1)
    return eta_min + 0.5 * (base_lr - eta_min) * (
        1 + math.cos(math.pi * (epoch - warmup_epochs) / (max_epochs - warmup_epochs)))
