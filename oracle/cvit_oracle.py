"""CPU ORACLE for the CViT path (models/cvit.py).  TEST INFRASTRUCTURE ONLY -- same rules as tante_oracle.py:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the product never does.

Functional torch-CPU restatement over the reference state_dict keys, pinned in tests/test_oracle_golden.py against
the g11_cvit_* fixtures (outputs of the reference itself, tests/golden/make_golden.py)."""
from __future__ import annotations

import math
from typing import Optional

import torch

from .tante_oracle import W, Tensor, gelu_erf, layer_norm, linear, sub


class CvitCfg:
    """The ctor arguments of models.CViT that shape the arithmetic (cvit.py:334-353)."""

    def __init__(self, in_T, n_fields, resolution, out_steps=4, patch_size=(1, 16, 16), grid_size=(128, 128), latent_dim=256,
                 emb_dim=256, depth=3, num_heads=8, dec_emb_dim=256, dec_num_heads=8, dec_depth=1, num_mlp_layers=1, mlp_ratio=1,
                 eps=1e5, layer_norm_eps=1e-5, embedding_type="grid"):
        self.in_T, self.n_fields, self.resolution, self.out_steps = in_T, n_fields, tuple(resolution), out_steps
        self.patch_size, self.grid_size, self.latent_dim, self.emb_dim, self.depth = tuple(patch_size), tuple(grid_size), latent_dim, emb_dim, depth
        self.num_heads, self.dec_emb_dim, self.dec_num_heads, self.dec_depth = num_heads, dec_emb_dim, dec_num_heads, dec_depth
        self.num_mlp_layers, self.mlp_ratio, self.eps, self.layer_norm_eps = num_mlp_layers, mlp_ratio, eps, layer_norm_eps
        self.embedding_type = embedding_type


def mha(w: W, q_in: Tensor, kv_in: Tensor, n_head: int) -> Tensor:
    """nn.MultiheadAttention(batch_first)(q, kv, kv): packed in_proj rows [0,C) -> q, [C,2C) -> k, [2C,3C) -> v (cvit.py:125,162)."""
    B, Lq, C = q_in.shape
    Lk = kv_in.shape[1]
    d = C // n_head
    Wi, bi = w["attn.in_proj_weight"], w["attn.in_proj_bias"]
    q = linear(q_in, Wi[:C], bi[:C]).reshape(B, Lq, n_head, d).transpose(1, 2)
    k = linear(kv_in, Wi[C:2 * C], bi[C:2 * C]).reshape(B, Lk, n_head, d).transpose(1, 2)
    v = linear(kv_in, Wi[2 * C:], bi[2 * C:]).reshape(B, Lk, n_head, d).transpose(1, 2)
    p = torch.softmax((q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(d)), dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B, Lq, C)
    return linear(o, w["attn.out_proj.weight"], w["attn.out_proj.bias"])


def mlp_block(w: W, x: Tensor) -> Tensor:
    """MlpBlock: fc2(gelu_erf(fc1(x)))  (cvit.py:95-110)."""
    return linear(gelu_erf(linear(x, w["fc1.weight"], w["fc1.bias"])), w["fc2.weight"], w["fc2.bias"])


def self_attn_block(w: W, x: Tensor, n_head: int, eps: float) -> Tensor:
    """SelfAttnBlock.forward (cvit.py:129-139)."""
    h = layer_norm(x, w["layer_norm1.weight"], w["layer_norm1.bias"], eps)
    x = mha(w, h, h, n_head) + x
    return x + mlp_block(sub(w, "mlp."), layer_norm(x, w["layer_norm2.weight"], w["layer_norm2.bias"], eps))


def cross_attn_block(w: W, q_in: Tensor, kv_in: Tensor, n_head: int, eps: float) -> Tensor:
    """CrossAttnBlock.forward (cvit.py:158-169).  layer_norm2 is used TWICE: on the keys/values (l.160) and again on the
    post-attention stream (l.165)."""
    q = layer_norm(q_in, w["layer_norm1.weight"], w["layer_norm1.bias"], eps)
    kv = layer_norm(kv_in, w["layer_norm2.weight"], w["layer_norm2.bias"], eps)
    x = mha(w, q, kv, n_head) + q_in
    return x + mlp_block(sub(w, "mlp."), layer_norm(x, w["layer_norm2.weight"], w["layer_norm2.bias"], eps))


def patch_embed(w: W, x: Tensor, patch) -> Tensor:
    """PatchEmbed: Conv3d kernel = stride = (pt, ph, pw) over (b c t h w) -> (b, t', h'w', emb)  (cvit.py:58-93)."""
    b, t, c, h, wd = x.shape
    pt, ph, pw = patch
    Wc = w["conv.weight"]                                    # (emb, c, pt, ph, pw)
    xp = x.reshape(b, t // pt, pt, c, h // ph, ph, wd // pw, pw).permute(0, 1, 4, 6, 3, 2, 5, 7)   # b t' h' w' c pt ph pw
    y = xp.reshape(b, t // pt, (h // ph) * (wd // pw), -1) @ Wc.reshape(Wc.shape[0], -1).T + w["conv.bias"]
    return y


def encoder(w: W, cfg: CvitCfg, x: Tensor) -> Tensor:
    """Encoder.forward (cvit.py:289-306): patch embed + t_emb + s_emb -> TimeAggregation (1 latent, depth 2, l.186-211) -> LN
    -> depth x SelfAttnBlock over the (t' s) tokens."""
    eps = cfg.layer_norm_eps
    y = patch_embed(sub(w, "patch_embed."), x, cfg.patch_size)              # (b, t, s, d)
    y = y + w["t_emb"][:, :, None, :] + w["s_emb"][:, None, :, :]
    b, t, s, d = y.shape
    kv = y.permute(0, 2, 1, 3).reshape(b * s, t, d)
    lat = w["time_agg.latents"][None].expand(b * s, -1, -1)
    for i in range(2):                                                      # time_agg depth is fixed to 2 (l.265)
        lat = cross_attn_block(sub(w, f"time_agg.CrossAttnBlocks.{i}."), lat, kv, cfg.num_heads, eps)
    tl = lat.shape[1]
    y = lat.reshape(b, s, tl, d).permute(0, 2, 1, 3)                        # (b, t', s, d)
    y = layer_norm(y, w["layer_norm.weight"], w["layer_norm.bias"], eps).reshape(b, tl * s, d)
    for i in range(cfg.depth):
        y = self_attn_block(sub(w, f"SelfAttnBlocks.{i}."), y, cfg.num_heads, eps)
    return y


def generate_coords(h: int, wd: int) -> Tensor:
    """cvit.py:469-479: the (h * w, 2) query grid on [0,1]^2, 'ij' order."""
    xs, ys = torch.meshgrid(torch.linspace(0, 1, h), torch.linspace(0, 1, wd), indexing="ij")
    return torch.stack([xs.flatten(), ys.flatten()], dim=-1)


def coord_embedding(w: W, cfg: CvitCfg, coords: Tensor) -> Tensor:
    """cvit.py:434-446: grid = normalised exp(-eps |x - g|^2) weights over the latent grid -> Linear -> LN; fourier = [cos, sin] of
    coords @ kernel (l.308-331); mlp = MlpBlock(2 -> d -> d) -> LN."""
    if cfg.embedding_type == "grid":
        d2 = ((coords[:, None, :] - w["grid"][None, :, :]) ** 2).sum(dim=2)
        e = torch.exp(-cfg.eps * d2)
        wts = e / e.sum(dim=1, keepdim=True)
        c = wts @ w["latents"]
        c = linear(c, w["embedding.0.weight"], w["embedding.0.bias"])
        return layer_norm(c, w["embedding.1.weight"], w["embedding.1.bias"], cfg.layer_norm_eps)
    if cfg.embedding_type == "fourier":
        dp = coords @ w["embedding.0.kernel"]
        return torch.cat([torch.cos(dp), torch.sin(dp)], dim=-1)
    c = mlp_block(sub(w, "embedding.0."), coords)
    return layer_norm(c, w["embedding.1.weight"], w["embedding.1.bias"], cfg.layer_norm_eps)


def mlp_head(w: W, cfg: CvitCfg, x: Tensor) -> Tensor:
    """Mlp.forward (cvit.py:234-242): num_layers x [x = LN(x + gelu(dense(x)))] -> output layer."""
    for i in range(cfg.num_mlp_layers):
        x = x + gelu_erf(linear(x, w[f"dense_layers.{i}.weight"], w[f"dense_layers.{i}.bias"]))
        x = layer_norm(x, w[f"layer_norms.{i}.weight"], w[f"layer_norms.{i}.bias"], cfg.layer_norm_eps)
    return linear(x, w["output_layer.weight"], w["output_layer.bias"])


def cvit_forward(w: W, cfg: CvitCfg, x: Tensor, input_coords: Optional[Tensor] = None) -> Tensor:
    """CViT.forward (cvit.py:427-466).  x (b, t, c, h, w) -> (b, out_steps, c, h, w), or (b, out_steps, n, c) for query points."""
    b, t, c, h, wd = x.shape
    coords = generate_coords(h, wd) if input_coords is None else input_coords
    q = coord_embedding(w, cfg, coords)[None].expand(b, -1, -1)
    y = encoder(sub(w, "Encoder."), cfg, x)
    y = linear(layer_norm(y, w["norm1.weight"], w["norm1.bias"], cfg.layer_norm_eps), w["E2D.weight"], w["E2D.bias"])
    for i in range(cfg.dec_depth):   # l.455-456: the QUERIES stay the coordinate embedding; each block's output becomes the next keys/values
        y = cross_attn_block(sub(w, f"CrossAttnBlocks.{i}."), q, y, cfg.dec_num_heads, cfg.layer_norm_eps)
    q = layer_norm(y, w["norm2.weight"], w["norm2.bias"], cfg.layer_norm_eps)
    o = mlp_head(sub(w, "mlp."), cfg, q)                                    # (b, n, out_steps * c)
    n = o.shape[1]
    o = o.reshape(b, n, cfg.out_steps, c)
    if input_coords is None:
        return o.reshape(b, h, wd, cfg.out_steps, c).permute(0, 3, 4, 1, 2)
    return o.permute(0, 2, 1, 3)
