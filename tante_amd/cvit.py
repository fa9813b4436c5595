"""CViT on the HIP kernels  (reference models/cvit.py).

Same constructor arguments, forward(x, input_coords=None) contract and state_dict keys as the reference's models.CViT.
Every arithmetic step is a C-ABI call of libtante_hip: im2col + tante_gemm (Conv3d patch embed), tante_gemm with folded
LayerNorm (all projections / MLPs), tante_cross_attention (self / cross / time-aggregation attention), tante_grid_embed (the
eps-Gaussian latent-grid interpolation, evaluated exactly but sparsely), tante_layernorm_affine.  torch allocates and reshapes.

Reference quirks kept on purpose (pinned by the g11 fixtures):
  * CrossAttnBlock applies layer_norm2 twice -- to the keys/values and again to the post-attention stream (cvit.py:160,165);
  * the decoder loop feeds each block's OUTPUT back as the next block's keys/values while the queries stay the coordinate
    embedding (cvit.py:455-456);
  * TimeAggregation depth is fixed to 2 with a single latent (cvit.py:262-269); the default eps is 1e5 (l.350).

Inference path only (no autograd graph); no CPU fallback."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import kernels as K
from . import stages as S
from .attn_backbone import _PackCache, _no_autograd, resolve_compute
from .tante import TanteMetadata, s_emb_init, t_emb_init
from . import options as _O

# host switches of this module (options.set_option, or TANTE_<NAME>=<int> when the package is imported)
CVIT_FUSED = _O.register("TANTE_CVIT_FUSED", True, __name__, "CVIT_FUSED")                # the block tail as one chain512 launch
CVIT_CHAIN_QKV = _O.register("TANTE_CVIT_CHAIN_QKV", False, __name__, "CVIT_CHAIN_QKV")   # measured SLOWER (B = 1: 1.02 -> 1.11 ms): off
# Block tails of at most this many token rows run as three wave-per-tile GEMM launches (gemm.hip: gemm_small_kernel, every CU takes part)
# instead of the one-launch chain, whose 16-token workgroups pull all 1.5 MB of weights through 16 CUs at B = 1.  Measured at cfg4 B = 1
# (profiles/r06_cvit_small_probe.log): 7.1 + 5.9 + 7.1 us of kernels + two boundaries against the chain's 22.8 us -- 0.717-0.722 against
# 0.717-0.741 ms per forward: a wash, so the chain (24 launches fewer) stays the default.  0: always the chain.
CVIT_SMALL_ROWS = _O.register("TANTE_CVIT_SMALL_ROWS", 0, __name__, "CVIT_SMALL_ROWS")


class MlpBlock(nn.Module):
    """fc2(gelu_erf(fc1(x)))  (cvit.py:95-110)."""

    def __init__(self, in_dim=256, dim=256, out_dim=256, kernel_init=True):
        super().__init__()
        self.fc1 = nn.Linear(in_dim, dim)
        self.fc2 = nn.Linear(dim, out_dim)
        if kernel_init:
            nn.init.xavier_uniform_(self.fc1.weight)
            nn.init.xavier_uniform_(self.fc2.weight)


class _AttnBlock(nn.Module):
    """Parameter layout shared by SelfAttnBlock / CrossAttnBlock (cvit.py:112-169)."""

    def __init__(self, num_heads, emb_dim, mlp_ratio, layer_norm_eps=1e-5):
        super().__init__()
        self.num_heads, self.emb_dim, self.eps = num_heads, emb_dim, layer_norm_eps
        self.attn = nn.MultiheadAttention(embed_dim=emb_dim, num_heads=num_heads, batch_first=True)
        self.layer_norm1 = nn.LayerNorm(emb_dim, eps=layer_norm_eps)
        self.layer_norm2 = nn.LayerNorm(emb_dim, eps=layer_norm_eps)
        self.mlp = MlpBlock(emb_dim, emb_dim * mlp_ratio, emb_dim)
        self._cache = _PackCache()

    def _packed(self, compute: int):
        a, m = self.attn, self.mlp
        n1, n2 = self.layer_norm1, self.layer_norm2
        params = [n1.weight, n1.bias, n2.weight, n2.bias, a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias,
                  m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias]
        C_ = self.emb_dim

        def build():
            Wi, bi = a.in_proj_weight.detach(), a.in_proj_bias.detach()
            pw = K.pack_weight
            return dict(
                # LayerNorm affine folded into the consuming projection (gamma scales the columns, beta joins the bias)
                q_ln1=pw(Wi[:C_].contiguous(), bi[:C_].contiguous(), compute, gamma=n1.weight, beta=n1.bias),
                kv_ln2=pw(Wi[C_:].contiguous(), bi[C_:].contiguous(), compute, gamma=n2.weight, beta=n2.bias),
                qkv_ln1=pw(Wi, bi, compute, gamma=n1.weight, beta=n1.bias),
                out=pw(a.out_proj.weight, a.out_proj.bias, compute),
                fc1_ln2=pw(m.fc1.weight, m.fc1.bias, compute, gamma=n2.weight, beta=n2.bias),
                fc2=S.pack_linear_chunks(m.fc2.weight.detach(), m.fc2.bias, compute))
        return self._cache.get(compute, params, build)

    def _chain_ok(self, compute: int, M: int, resid_rows: int) -> bool:
        """The one-launch tail (cvit_fused.hip): bf16, width 512, mlp_ratio 1, whole 16-token tiles."""
        return (compute == L.BF16 and self.emb_dim == 512 and self.mlp.fc1.out_features == 512 and M % 16 == 0 and resid_rows % 16 == 0
                and CVIT_FUSED)

    def _chain_packed(self, extra=None):
        """out_proj | fc1 (LN2 folded) | fc2 [| the Mlp's dense layer] as one fragment stream + biases; extra = (norm2, Mlp) for mode 1."""
        a, m, n2 = self.attn, self.mlp, self.layer_norm2
        params = [n2.weight, n2.bias, a.out_proj.weight, a.out_proj.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias]
        if extra is not None:
            norm2, mlp = extra
            ln = mlp.layer_norms[0]
            params += [norm2.weight, norm2.bias, mlp.dense_layers[0].weight, mlp.dense_layers[0].bias, ln.weight, ln.bias, mlp.output_layer.weight,
                       mlp.output_layer.bias]

        def build():
            mats = [K.pack_chain_matrix(a.out_proj.weight, a.out_proj.bias), K.pack_chain_matrix(m.fc1.weight, m.fc1.bias, n2.weight, n2.bias),
                    K.pack_chain_matrix(m.fc2.weight, m.fc2.bias)]
            tail = None
            if extra is not None:
                mats.append(K.pack_chain_matrix(mlp.dense_layers[0].weight, mlp.dense_layers[0].bias))
                wout, bout = K.pack_chain_output(mlp.output_layer.weight, mlp.output_layer.bias, ln.weight, ln.bias)
                tail = (norm2.weight.detach().float().contiguous(), norm2.bias.detach().float().contiguous(), float(norm2.eps), float(ln.eps), wout, bout,
                        mlp.output_layer.out_features)
            return torch.cat([f for f, _ in mats]), torch.cat([b for _, b in mats]), tail
        return self._cache.get(("chain", extra is not None), params, build)

    def _tail(self, attn_o: torch.Tensor, resid: torch.Tensor, pk, compute: int, model_tail=None) -> torch.Tensor:
        """x = out_proj(attn) + resid;  return x + fc2(gelu(fc1(LN2(x)))).  resid may hold fewer rows than attn_o (row t % rows: the decoder's
        shared queries).  model_tail = (norm2, Mlp): continue through norm2 -> Mlp -> output layer in the same launch, -> (M, out_dim)."""
        adt = K.act_torch_dtype(compute)
        M = attn_o.shape[0]
        small = model_tail is None and M <= CVIT_SMALL_ROWS and resid.shape[0] == M
        if not small and self._chain_ok(compute, M, resid.shape[0]) and (model_tail is None or _model_tail_ok(model_tail)):
            w, bias, tail = self._chain_packed(model_tail)
            out = torch.empty(M, self.emb_dim if tail is None else tail[6], dtype=torch.float32, device=attn_o.device)
            return K.cvit_chain512(attn_o, resid, w, bias, float(self.eps), M, out, tail)
        if resid.shape[0] != M:
            resid = resid.unsqueeze(0).expand(M // resid.shape[0], -1, -1).reshape(M, self.emb_dim).contiguous()
        if model_tail is not None:
            norm2, mlp = model_tail
            x = self._tail(attn_o, resid, pk, compute)
            return mlp.run(K.layernorm_affine(x, norm2.weight, norm2.bias, norm2.eps), compute)
        x = torch.empty(M, self.emb_dim, dtype=torch.float32, device=attn_o.device)
        K.linear(attn_o, pk["out"], x, M=M, residual=resid)
        h = torch.empty(M, pk["fc1_ln2"].N, dtype=adt, device=x.device)
        K.linear(x, pk["fc1_ln2"], h, M=M, act=L.ACT_GELU_ERF, ln=True, ln_eps=self.eps)
        if len(pk["fc2"]) == 1:
            y = torch.empty_like(x)
            return K.linear(h, pk["fc2"][0], y, M=M, residual=x)
        return S.linear_chunks(h, pk["fc2"], torch.float32) + x    # hidden > 512: chunked accumulate, then the residual add


def _model_tail_ok(model_tail) -> bool:
    norm2, mlp = model_tail
    return mlp.num_layers == 1 and mlp.output_layer.out_features <= 16 and mlp.dense_layers[0].in_features == 512 and mlp.dense_layers[0].out_features == 512


class SelfAttnBlock(_AttnBlock):
    def _chain_qkv_packed(self, nxt: "SelfAttnBlock"):
        """This block's tail matrices + the NEXT block's LN1-folded input projection as one six-matrix fragment stream."""
        a, m, n2 = self.attn, self.mlp, self.layer_norm2
        na, n1 = nxt.attn, nxt.layer_norm1
        params = [n2.weight, n2.bias, a.out_proj.weight, a.out_proj.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias,
                  n1.weight, n1.bias, na.in_proj_weight, na.in_proj_bias]

        def build():
            C_ = self.emb_dim
            Wi, bi = na.in_proj_weight.detach(), na.in_proj_bias.detach()
            mats = [K.pack_chain_matrix(a.out_proj.weight, a.out_proj.bias), K.pack_chain_matrix(m.fc1.weight, m.fc1.bias, n2.weight, n2.bias),
                    K.pack_chain_matrix(m.fc2.weight, m.fc2.bias)]
            mats += [K.pack_chain_matrix(Wi[p * C_:(p + 1) * C_], bi[p * C_:(p + 1) * C_], n1.weight, n1.bias) for p in range(3)]
            return torch.cat([f for f, _ in mats]), torch.cat([b for _, b in mats])
        return self._cache.get(("chain_qkv", id(nxt)), params, build)

    def run(self, x: torch.Tensor, nb: int, Lq: int, compute: int, qkv: Optional[torch.Tensor] = None, next_blk: Optional["SelfAttnBlock"] = None):
        """x (nb * Lq, C) fp32 -> block(x)   (cvit.py:129-139).  qkv: this block's q | k | v rows if the previous block's launch already
        projected them.  next_blk: -> (block(x), the next block's q | k | v) when the one-launch tail applies, (block(x), None) otherwise."""
        pk = self._packed(compute)
        adt = K.act_torch_dtype(compute)
        C_, nh = self.emb_dim, self.num_heads
        M = nb * Lq
        if qkv is None:
            qkv = torch.empty(M, 3 * C_, dtype=adt, device=x.device)
            K.linear(x, pk["qkv_ln1"], qkv, M=M, ln=True, ln_eps=self.eps)
        o = torch.empty(M, C_, dtype=adt, device=x.device)
        K.cross_attention(qkv, qkv[:, C_:], qkv[:, 2 * C_:], o, nb, nh, C_ // nh, Lq, Lq, 3 * C_, 3 * C_, C_)
        if next_blk is None:
            return self._tail(o, x, pk, compute)
        if (self._chain_ok(compute, M, M) and next_blk.emb_dim == C_ and next_blk.num_heads * 64 == C_
                and CVIT_CHAIN_QKV):      # measured SLOWER (B = 1: 1.02 -> 1.11 ms, B = 4: 1.80 -> 1.86): off
            w, bias = self._chain_qkv_packed(next_blk)
            out = torch.empty(M, C_, dtype=torch.float32, device=x.device)
            nqkv = torch.empty(M, 3 * C_, dtype=torch.bfloat16, device=x.device)
            L.check(L.lib().tante_cvit_chain512_qkv(o.data_ptr(), x.data_ptr(), M, w.data_ptr(), bias.data_ptr(), float(self.eps), float(next_blk.eps), M,
                                                    out.data_ptr(), nqkv.data_ptr(), K._stream()), "tante_cvit_chain512_qkv")
            return out, nqkv
        return self._tail(o, x, pk, compute), None


class CrossAttnBlock(_AttnBlock):
    def project_queries(self, q_in: torch.Tensor, compute: int) -> torch.Tensor:
        """LN1 + the query projection of q_in (rows, C) fp32 -> (rows, C) in the activation dtype."""
        pk = self._packed(compute)
        q = torch.empty(q_in.shape[0], self.emb_dim, dtype=K.act_torch_dtype(compute), device=q_in.device)
        return K.linear(q_in, pk["q_ln1"], q, M=q_in.shape[0], ln=True, ln_eps=self.eps)

    def run(self, q_in: torch.Tensor, kv_in: torch.Tensor, nb: int, Lq: int, Lk: int, compute: int, model_tail=None, q_proj=None) -> torch.Tensor:
        """q_in (nb * Lq, C) -- or (Lq, C): the SAME queries for every sample, projected once -- and kv_in (nb * Lk, C) fp32
        -> block(q_in, kv_in)   (cvit.py:158-169).  q_proj: project_queries(q_in) computed earlier (input-independent queries)."""
        pk = self._packed(compute)
        adt = K.act_torch_dtype(compute)
        C_, nh = self.emb_dim, self.num_heads
        shared = q_in.shape[0] == Lq and nb > 1
        q = q_proj if q_proj is not None else self.project_queries(q_in, compute)
        kv = torch.empty(nb * Lk, 2 * C_, dtype=adt, device=q_in.device)
        K.linear(kv_in, pk["kv_ln2"], kv, M=nb * Lk, ln=True, ln_eps=self.eps)
        o = torch.empty(nb * Lq, C_, dtype=adt, device=q_in.device)
        K.cross_attention(q, kv, kv[:, C_:], o, nb, nh, C_ // nh, Lq, Lk, C_, 2 * C_, C_, shared_q=shared)
        return self._tail(o, q_in, pk, compute, model_tail=model_tail)


class TimeAggregation(nn.Module):
    """Learned latents attend over the T frames of every spatial token  (cvit.py:171-211)."""

    def __init__(self, emb_dim, depth, num_heads=8, num_latents=64, mlp_ratio=1, layer_norm_eps=1e-5):
        super().__init__()
        self.emb_dim, self.depth, self.num_latents = emb_dim, depth, num_latents
        self.latents = nn.Parameter(torch.randn(num_latents, emb_dim))
        self.CrossAttnBlocks = nn.ModuleList([CrossAttnBlock(num_heads, emb_dim, mlp_ratio, layer_norm_eps) for _ in range(depth)])
        self._rep = _PackCache()

    def run(self, x_bs_t: torch.Tensor, nbs: int, T: int, compute: int) -> torch.Tensor:
        """x ((b s) t, d) fp32 -> latents ((b s) t', d)."""
        # the latents repeated per (b, s) token: input independent, so built once per weight version (the fused block tail reads its
        # residual rows one per query row)
        lat = self._rep.get(nbs, [self.latents], lambda: self.latents.detach().unsqueeze(0).expand(nbs, -1, -1)
                            .reshape(nbs * self.num_latents, self.emb_dim).contiguous())
        for i, blk in enumerate(self.CrossAttnBlocks):
            # the first block's queries ARE the (repeated) latents: their LayerNorm1 + projection is input-independent too
            qp = None
            if i == 0:
                a = blk.attn
                qp = self._rep.get(("qproj", nbs, compute), [self.latents, blk.layer_norm1.weight, blk.layer_norm1.bias, a.in_proj_weight, a.in_proj_bias],
                                   lambda: blk.project_queries(lat, compute))
            lat = blk.run(lat, x_bs_t, nbs, self.num_latents, T, compute, q_proj=qp)
        return lat


class PatchEmbed(nn.Module):
    """Conv3d with kernel = stride = (pt, ph, pw) as im2col + GEMM  (cvit.py:58-93)."""

    def __init__(self, n_channel, patch_size=(1, 16, 16), emb_dim=768, use_norm=False, kernel_init=False, layer_norm_eps=1e-5):
        super().__init__()
        self.patch_size, self.use_norm = tuple(patch_size), use_norm
        if self.patch_size[0] != 1:
            # (the reference itself cannot run them: Encoder.forward repeats s_emb over the INPUT's t while the patch-embedded sequence has
            # t / patch_size[0] steps, cvit.py:293-296 -> a shape error in `x + t_emb + s_emb`; verified with patch_size (2, 8, 8))
            raise NotImplementedError("temporal patches (patch_size[0] > 1): the reference's Encoder.forward raises for them (cvit.py:293-296); "
                                      "every shipped config uses 1")
        self.conv = nn.Conv3d(n_channel, emb_dim, kernel_size=self.patch_size, stride=self.patch_size)
        if use_norm:
            self.layer_norm = nn.LayerNorm(emb_dim, eps=layer_norm_eps)
        self._cache = _PackCache()

    def run(self, x: torch.Tensor, compute: int, nhwc: bool = False) -> torch.Tensor:
        """x (b, t, c, h, w) fp32 contiguous -- or, with nhwc, the channels-first VIEW of a contiguous (b, t, h, w, c) tensor, which is what
        the formatter hands over -> (b * t * s, emb) fp32, s = (h / ph)(w / pw) row-major."""
        b, t, c, h, w = x.shape
        _, ph, pw = self.patch_size
        chunks = self._cache.get(compute, [self.conv.weight, self.conv.bias],
                                 lambda: S.pack_linear_chunks(self.conv.weight.detach().reshape(self.conv.weight.shape[0], -1),
                                                              self.conv.bias, compute))
        cols = K.im2col(x.permute(0, 1, 3, 4, 2) if nhwc else x, not nhwc, b * t, c, h, w, ph, pw, ph, pw, 0, 0, 0, K.act_torch_dtype(compute))
        y = S.linear_chunks(cols, chunks, torch.float32)
        if self.use_norm:
            y = K.layernorm_affine(y, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps)
        return y


class Encoder(nn.Module):
    """cvit.py:248-306."""

    def __init__(self, n_channel, patch_size=(1, 16, 16), emb_dim=256, depth=3, num_heads=8, mlp_ratio=1, out_dim=1,
                 layer_norm_eps=1e-5, THW_shape=(4, 128, 384)):
        super().__init__()
        self.depth, self.emb_dim = depth, emb_dim
        self.patch_embed = PatchEmbed(n_channel, patch_size, emb_dim)
        self.time_agg = TimeAggregation(num_latents=1, emb_dim=emb_dim, depth=2, num_heads=num_heads, mlp_ratio=mlp_ratio,
                                        layer_norm_eps=layer_norm_eps)
        self.layer_norm = nn.LayerNorm(emb_dim, eps=layer_norm_eps)
        t, h, w = THW_shape
        self.t_emb = nn.Parameter(t_emb_init(emb_dim, t // patch_size[0]))
        self.s_emb = nn.Parameter(s_emb_init(emb_dim, (h // patch_size[1], w // patch_size[2]), flatten=True))   # (1, S, D)
        self.SelfAttnBlocks = nn.ModuleList([SelfAttnBlock(num_heads, emb_dim, mlp_ratio, layer_norm_eps) for _ in range(depth)])

    def run(self, x: torch.Tensor, compute: int, nhwc: bool = False):
        """x (b, t, c, h, w) -> tokens (b * t' * s, d) fp32, (t' * s).  nhwc: see PatchEmbed.run."""
        b, t = x.shape[:2]
        d = self.emb_dim
        y = self.patch_embed.run(x, compute, nhwc)                 # rows (b, t, s)
        s = y.shape[0] // (b * t)
        # + t_emb[t] + s_emb[s] and 'b t s d -> (b s) t d' (the layout the time aggregation attends over) in one pass
        kv = torch.empty_like(y)
        L.check(L.lib().tante_pos_embed_tmajor(y.data_ptr(), self.t_emb.detach().view(t, d).contiguous().data_ptr(),
                                               self.s_emb.detach().view(s, d).contiguous().data_ptr(), b, t, s, d, kv.data_ptr(), K._stream()),
                "tante_pos_embed_tmajor")
        lat = self.time_agg.run(kv, b * s, t, compute)                                   # ((b s) t', d), t' = 1
        tl = self.time_agg.num_latents
        if tl != 1:
            lat = lat.view(b, s, tl, d).permute(0, 2, 1, 3).contiguous().view(b * tl * s, d)
        y = K.layernorm_affine(lat, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps)
        blocks = list(self.SelfAttnBlocks)
        qkv = None
        for i, blk in enumerate(blocks):      # a block's launch also projects the next block's q | k | v where the one-launch tail applies
            if i + 1 < len(blocks):
                y, qkv = blk.run(y, b, tl * s, compute, qkv=qkv, next_blk=blocks[i + 1])
            else:
                y = blk.run(y, b, tl * s, compute, qkv=qkv)
        return y, tl * s


class FourierEmbs(nn.Module):
    """cvit.py:308-331."""

    def __init__(self, embed_scale: float, embed_dim: int, D: int = 2):
        super().__init__()
        self.embed_scale, self.embed_dim = embed_scale, embed_dim
        self.kernel = nn.Parameter(torch.randn(D, embed_dim // 2) * embed_scale)


class Mlp(nn.Module):
    """num_layers x [x = LN(x + gelu(dense(x)))] -> output layer  (cvit.py:213-242)."""

    def __init__(self, in_dim, num_layers, hidden_dim, out_dim, kernel_init=True, layer_norm_eps=1e-5):
        super().__init__()
        self.num_layers = num_layers
        self.dense_layers = nn.ModuleList([nn.Linear(hidden_dim if i > 0 else in_dim, hidden_dim) for i in range(num_layers)])
        self.output_layer = nn.Linear(hidden_dim, out_dim)
        self.layer_norms = nn.ModuleList([nn.LayerNorm(hidden_dim, eps=layer_norm_eps) for _ in range(num_layers)])
        self._cache = _PackCache()

    def run(self, x: torch.Tensor, compute: int) -> torch.Tensor:
        params = [p for m in list(self.dense_layers) + [self.output_layer] for p in (m.weight, m.bias)]
        pk = self._cache.get(compute, params, lambda: [K.pack_weight(m.weight, m.bias, compute)
                                                       for m in list(self.dense_layers) + [self.output_layer]])
        M = x.shape[0]
        for i in range(self.num_layers):
            y = torch.empty(M, pk[i].N, dtype=torch.float32, device=x.device)
            K.linear(x, pk[i], y, M=M, act=L.ACT_GELU_ERF, residual=x)              # x + gelu(dense(x))
            ln = self.layer_norms[i]
            x = K.layernorm_affine(y, ln.weight, ln.bias, ln.eps)
        out = torch.empty(M, pk[-1].N, dtype=torch.float32, device=x.device)
        return K.linear(x, pk[-1], out, M=M)


def generate_coords(h: int, w: int, device) -> torch.Tensor:
    """The (h * w, 2) query grid on [0, 1]^2 in 'ij' order  (cvit.py:469-479)."""
    xs, ys = torch.meshgrid(torch.linspace(0, 1, h, device=device), torch.linspace(0, 1, w, device=device), indexing="ij")
    return torch.stack([xs.flatten(), ys.flatten()], dim=-1).contiguous()


class CViT(nn.Module):
    def __init__(self, in_T, dset_metadata: TanteMetadata = None, out_steps=4, patch_size: tuple = (1, 16, 16),
                 grid_size: tuple = (128, 128), latent_dim: int = 256, emb_dim: int = 256, depth: int = 3, num_heads: int = 8,
                 dec_emb_dim: int = 256, dec_num_heads: int = 8, dec_depth: int = 1, num_mlp_layers: int = 1, mlp_ratio: int = 1,
                 eps: float = 1e5, layer_norm_eps: float = 1e-5, embedding_type: str = "grid"):
        super().__init__()
        n_channel = dset_metadata.n_fields if dset_metadata else 4
        self.T = in_T
        self.H, self.W = dset_metadata.spatial_resolution if dset_metadata else (128, 384)
        self.embedding_type, self.eps, self.dec_depth, self.out_steps = embedding_type, eps, dec_depth, out_steps
        self.dec_emb_dim, self.ln_eps = dec_emb_dim, layer_norm_eps
        out_dim = n_channel * out_steps
        if embedding_type == "grid":
            n_x, n_y = grid_size
            self.latents = nn.Parameter(torch.randn(n_x * n_y, latent_dim))
            xx, yy = np.meshgrid(np.linspace(0, 1, n_x), np.linspace(0, 1, n_y), indexing="ij")
            grid = torch.tensor(np.hstack([xx.flatten()[:, None], yy.flatten()[:, None]])).to(dtype=self.latents.dtype)
            self.grid = nn.Parameter(grid)
            self.embedding = nn.Sequential(nn.Linear(latent_dim, dec_emb_dim), nn.LayerNorm(dec_emb_dim, eps=layer_norm_eps))
        elif embedding_type == "fourier":
            self.embedding = nn.Sequential(FourierEmbs(embed_scale=2 * np.pi, embed_dim=dec_emb_dim))
        elif embedding_type == "mlp":
            self.embedding = nn.Sequential(MlpBlock(2, dec_emb_dim, dec_emb_dim), nn.LayerNorm(dec_emb_dim, eps=layer_norm_eps))
        else:
            raise ValueError(f"embedding_type must be 'grid', 'fourier' or 'mlp', got {embedding_type!r}")
        self.Encoder = Encoder(n_channel=n_channel, patch_size=patch_size, emb_dim=emb_dim, depth=depth, num_heads=num_heads,
                               mlp_ratio=mlp_ratio, layer_norm_eps=layer_norm_eps, THW_shape=(self.T, self.H, self.W))
        self.E2D = nn.Linear(emb_dim, dec_emb_dim)
        self.CrossAttnBlocks = nn.ModuleList([CrossAttnBlock(dec_num_heads, dec_emb_dim, mlp_ratio, layer_norm_eps) for _ in range(dec_depth)])
        self.mlp = Mlp(in_dim=dec_emb_dim, num_layers=num_mlp_layers, hidden_dim=dec_emb_dim, out_dim=out_dim, layer_norm_eps=layer_norm_eps)
        self.norm1 = nn.LayerNorm(emb_dim, eps=layer_norm_eps)
        self.norm2 = nn.LayerNorm(dec_emb_dim, eps=layer_norm_eps)
        self.compute: Optional[str] = None
        self._cache = _PackCache()
        self._coord_cache = _PackCache()

    def set_compute(self, mode: Optional[str]):
        if mode is not None and mode not in K.COMPUTE:
            raise ValueError("compute must be None, 'fp32' or 'bf16'")
        self.compute = mode
        return self

    # ---- coordinate embedding (cvit.py:434-446): input independent, so the default full-grid queries are cached per weight version
    def _embed_coords(self, coords: torch.Tensor, compute: int) -> torch.Tensor:
        n = coords.shape[0]
        if self.embedding_type == "grid":
            lin, ln = self.embedding[0], self.embedding[1]
            c = K.grid_embed(coords, self.grid.detach().contiguous(), self.latents.detach().contiguous(), float(self.eps))
            chunks = self._cache.get((compute, "emb"), [lin.weight, lin.bias], lambda: S.pack_linear_chunks(lin.weight.detach(), lin.bias, compute))
            c = S.linear_chunks(c, chunks, torch.float32)
            return K.layernorm_affine(c, ln.weight, ln.bias, ln.eps)
        if self.embedding_type == "fourier":
            return K.fourier_embed(coords, self.embedding[0].kernel.detach().contiguous())
        blk, ln = self.embedding[0], self.embedding[1]
        pk = self._cache.get((compute, "emb"), [blk.fc1.weight, blk.fc1.bias, blk.fc2.weight, blk.fc2.bias],
                             lambda: (K.pack_weight(blk.fc1.weight, blk.fc1.bias, compute), S.pack_linear_chunks(blk.fc2.weight.detach(), blk.fc2.bias, compute)))
        h = torch.empty(n, pk[0].N, dtype=K.act_torch_dtype(compute), device=coords.device)
        K.linear(coords, pk[0], h, M=n, act=L.ACT_GELU_ERF)
        c = S.linear_chunks(h, pk[1], torch.float32)
        return K.layernorm_affine(c, ln.weight, ln.bias, ln.eps)

    def encode(self, x: torch.Tensor):
        """The input-dependent half of forward that does NOT depend on the query points: Encoder -> norm1 -> E2D (cvit.py:437-448) ->
        the decoder's first keys / values.  Evaler.rollout_cvit (trainer/evaler.py:140-165) evaluates the full field in query chunks and
        the reference runs the whole model per chunk; `forward(x, coords, encoded=model.encode(x))` runs this half once per window
        (harness.rollout_cvit_eval) -- the same launches on the same data, so the same bits."""
        return self.forward(x, None, _encode_only=True)

    def forward(self, x: torch.Tensor, input_coords: Optional[torch.Tensor] = None, encoded=None, _encode_only: bool = False) -> torch.Tensor:
        """x (b, t, c, h, w) -> (b, out_steps, c, h, w); with input_coords (n, 2): (b, out_steps, n, c)   (cvit.py:427-466).
        encoded: the result of encode(x) for this x (inference only)."""
        if not x.is_cuda:
            raise RuntimeError("tante_amd.CViT runs on the GPU only (no CPU fallback); move the input to cuda")
        compute = resolve_compute(self.compute)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            if encoded is not None or _encode_only:
                raise ValueError("encode() / encoded= are inference-path options")
            return cvit_train_forward(self, x.detach().to(torch.float32).contiguous(), input_coords, compute)
        x = x.detach()
        # the formatter's 'b t h w c -> b t c h w' view of a channels-last batch is read in place by the patch gather (no copy)
        nhwc = x.dtype == torch.float32 and x.dim() == 5 and not x.is_contiguous() and x.permute(0, 1, 3, 4, 2).is_contiguous()
        if not nhwc:
            x = x.to(torch.float32).contiguous()
        b, t, c, h, w = x.shape
        if _encode_only:
            y, s = self.Encoder.run(x, compute, nhwc)
            e2d = self._cache.get((compute, "e2d"), [self.norm1.weight, self.norm1.bias, self.E2D.weight, self.E2D.bias],
                                  lambda: K.pack_weight(self.E2D.weight, self.E2D.bias, compute, gamma=self.norm1.weight, beta=self.norm1.bias))
            kv = torch.empty(b * s, self.dec_emb_dim, dtype=torch.float32, device=x.device)
            K.linear(y, e2d, kv, M=b * s, ln=True, ln_eps=self.norm1.eps)
            return (kv, s, compute, (b, t, c, h, w))
        if input_coords is None:
            params = [p for p in self.parameters()]
            q1 = self._coord_cache.get((compute, h, w), params, lambda: self._embed_coords(generate_coords(h, w, x.device), compute))
        else:
            q1 = self._embed_coords(input_coords.detach().to(x.device, torch.float32).contiguous(), compute)
        n = q1.shape[0]
        d = self.dec_emb_dim
        q = q1                                                     # 'n d -> b n d' is never materialised: the blocks take the shared rows
        if encoded is not None:
            kv0, s, ecompute, eshape = encoded
            if ecompute != compute or tuple(eshape) != (b, t, c, h, w):
                raise ValueError("encoded= was computed for another input shape or compute mode")
            kv = kv0      # (the blocks write fresh tensors: the cached rows stay intact for the next chunk)
        else:
            y, s = self.Encoder.run(x, compute, nhwc)                                          # (b * s, emb)
            e2d = self._cache.get((compute, "e2d"), [self.norm1.weight, self.norm1.bias, self.E2D.weight, self.E2D.bias],
                                  lambda: K.pack_weight(self.E2D.weight, self.E2D.bias, compute, gamma=self.norm1.weight, beta=self.norm1.bias))
            kv = torch.empty(b * s, d, dtype=torch.float32, device=x.device)
            K.linear(y, e2d, kv, M=b * s, ln=True, ln_eps=self.norm1.eps)                      # E2D(norm1(x))
        Lk = s
        for i, blk in enumerate(self.CrossAttnBlocks):   # queries stay the coordinate embedding; the output becomes the next keys/values
            last = i == len(self.CrossAttnBlocks) - 1     # the last block carries norm2 -> Mlp -> output layer (one launch where it fuses)
            # the default full-grid queries are input-independent, and so is every block's LayerNorm1 + query projection of them: cached
            # with the embedding, per weight version (34 GFLOP and ~100 us per forward at 65 536 queries)
            qp = self._coord_cache.get((compute, h, w, "qproj", i), params, lambda: blk.project_queries(q, compute)) if input_coords is None else None
            kv = blk.run(q, kv, b, n, Lk, compute, model_tail=(self.norm2, self.mlp) if last else None, q_proj=qp)
            Lk = n
        o = kv.view(b, n, self.out_steps, c)
        if input_coords is None:
            return o.view(b, h, w, self.out_steps, c).permute(0, 3, 4, 1, 2)                   # 'b (h w) (t d) -> b t d h w'
        return o.permute(0, 2, 1, 3)                                                           # 'b n (t d) -> b t n d'


# ---- differentiable (training) forward: the same arithmetic from tante_amd.autograd ops ---------------------------------------------
def _fold(w, b, ln):
    """LayerNorm affine folded into the consuming projection (parameter-sized torch expressions; autograd distributes the gradients)."""
    return w * ln.weight[None, :], b + w @ ln.bias


def _attn_tail_train(blk, o, resid, compute):
    from .autograd import ActFn, LayerNormFn, LinearFn
    adt = K.act_torch_dtype(compute)
    a, m = blk.attn, blk.mlp
    x = LinearFn.apply(o, a.out_proj.weight, a.out_proj.bias, resid, compute, torch.float32)
    w1, b1 = _fold(m.fc1.weight, m.fc1.bias, blk.layer_norm2)
    h = ActFn.apply(LinearFn.apply(LayerNormFn.apply(x, blk.eps, adt), w1, b1, None, compute, adt), L.ACT_GELU_ERF, adt)
    return LinearFn.apply(h, m.fc2.weight, m.fc2.bias, x, compute, torch.float32)


def _self_block_train(blk, x, nb, Lq, compute):
    from .autograd import CrossAttentionFn, LayerNormFn, LinearFn
    adt = K.act_torch_dtype(compute)
    C_, nh = blk.emb_dim, blk.num_heads
    w, b = _fold(blk.attn.in_proj_weight, blk.attn.in_proj_bias, blk.layer_norm1)
    qkv = LinearFn.apply(LayerNormFn.apply(x, blk.eps, adt), w, b, None, compute, adt)
    o = CrossAttentionFn.apply(qkv, qkv, 0, C_, 2 * C_, nb, nh, C_ // nh, Lq, Lq)
    return _attn_tail_train(blk, o, x, compute)


def _cross_block_train(blk, q_in, kv_in, nb, Lq, Lk, compute):
    from .autograd import CrossAttentionFn, LayerNormFn, LinearFn
    adt = K.act_torch_dtype(compute)
    C_, nh = blk.emb_dim, blk.num_heads
    Wi, bi = blk.attn.in_proj_weight, blk.attn.in_proj_bias
    wq, bq = _fold(Wi[:C_], bi[:C_], blk.layer_norm1)
    wkv, bkv = _fold(Wi[C_:], bi[C_:], blk.layer_norm2)
    q = LinearFn.apply(LayerNormFn.apply(q_in, blk.eps, adt), wq, bq, None, compute, adt)
    kv = LinearFn.apply(LayerNormFn.apply(kv_in, blk.eps, adt), wkv, bkv, None, compute, adt)
    o = CrossAttentionFn.apply(q, kv, 0, 0, C_, nb, nh, C_ // nh, Lq, Lk)
    return _attn_tail_train(blk, o, q_in, compute)


def cvit_train_forward(m: CViT, x: torch.Tensor, input_coords, compute: int) -> torch.Tensor:
    """CViT.forward with an autograd graph of HIP ops (one launch per layer; the backward kernels are tante_cross_attention_bwd,
    tante_layernorm_affine_bwd, tante_grid_embed_bwd and the TANTE train path's dgrad / wgrad / LayerNorm / activation kernels)."""
    from .autograd import (ActFn, DropoutAddFn, FilmPosFn, FourierEmbedFn, GridEmbedFn, LayerNormAffineFn, LayerNormFn, LinearFn)
    adt = K.act_torch_dtype(compute)
    b, t, c, h, w = x.shape
    dev = x.device
    coords = generate_coords(h, w, dev) if input_coords is None else input_coords.detach().to(dev, torch.float32).contiguous()
    n, d = coords.shape[0], m.dec_emb_dim
    # ---- coordinate embedding
    if m.embedding_type == "grid":
        lin, ln = m.embedding[0], m.embedding[1]
        ce = GridEmbedFn.apply(coords, m.grid, m.latents, float(m.eps))
        q1 = LayerNormAffineFn.apply(LinearFn.apply(ce, lin.weight, lin.bias, None, compute, torch.float32), ln.weight, ln.bias, ln.eps)
    elif m.embedding_type == "fourier":
        q1 = FourierEmbedFn.apply(coords, m.embedding[0].kernel)
    else:
        blk, ln = m.embedding[0], m.embedding[1]
        hcoord = ActFn.apply(LinearFn.apply(coords, blk.fc1.weight, blk.fc1.bias, None, compute, adt), L.ACT_GELU_ERF, adt)
        q1 = LayerNormAffineFn.apply(LinearFn.apply(hcoord, blk.fc2.weight, blk.fc2.bias, None, compute, torch.float32), ln.weight, ln.bias, ln.eps)
    q = q1.unsqueeze(0).expand(b, n, d).reshape(b * n, d).contiguous()
    # ---- encoder
    enc = m.Encoder
    pe = enc.patch_embed
    _, ph, pw = pe.patch_size
    cols = K.im2col(x.view(b * t, c, h, w), True, b * t, c, h, w, ph, pw, ph, pw, 0, 0, 0, adt)
    y = LinearFn.apply(cols, pe.conv.weight.view(pe.conv.weight.shape[0], -1), pe.conv.bias, None, compute, torch.float32)
    if pe.use_norm:
        y = LayerNormAffineFn.apply(y, pe.layer_norm.weight, pe.layer_norm.bias, pe.layer_norm.eps)
    s = y.shape[0] // (b * t)
    e = enc.emb_dim
    ones = torch.ones(t, e, dtype=torch.float32, device=dev)
    y = FilmPosFn.apply(y, ones, enc.t_emb.view(t, e), enc.s_emb.view(s, e), t, s)
    kv = y.view(b, t, s, e).permute(0, 2, 1, 3).reshape(b * s * t, e).contiguous()
    ta = enc.time_agg
    lat = ta.latents.unsqueeze(0).expand(b * s, -1, -1).reshape(b * s * ta.num_latents, e).contiguous()
    for blk in ta.CrossAttnBlocks:
        lat = _cross_block_train(blk, lat, kv, b * s, ta.num_latents, t, compute)
    tl = ta.num_latents
    if tl != 1:
        lat = lat.view(b, s, tl, e).permute(0, 2, 1, 3).reshape(b * tl * s, e).contiguous()
    y = LayerNormAffineFn.apply(lat, enc.layer_norm.weight, enc.layer_norm.bias, enc.layer_norm.eps)
    for blk in enc.SelfAttnBlocks:
        y = _self_block_train(blk, y, b, tl * s, compute)
    # ---- decoder
    we, be = _fold(m.E2D.weight, m.E2D.bias, m.norm1)
    kvd = LinearFn.apply(LayerNormFn.apply(y, m.norm1.eps, adt), we, be, None, compute, torch.float32)
    Lk = tl * s
    for blk in m.CrossAttnBlocks:
        kvd = _cross_block_train(blk, q, kvd, b, n, Lk, compute)
        Lk = n
    z = LayerNormAffineFn.apply(kvd, m.norm2.weight, m.norm2.bias, m.norm2.eps)
    head = m.mlp
    for i in range(head.num_layers):
        hh = ActFn.apply(LinearFn.apply(z, head.dense_layers[i].weight, head.dense_layers[i].bias, None, compute, adt), L.ACT_GELU_ERF, adt)
        z = DropoutAddFn.apply(hh, z, 0.0)                                   # x + gelu(dense(x))
        ln = head.layer_norms[i]
        z = LayerNormAffineFn.apply(z, ln.weight, ln.bias, ln.eps)
    o = LinearFn.apply(z, head.output_layer.weight, head.output_layer.bias, None, compute, torch.float32).view(b, n, m.out_steps, c)
    if input_coords is None:
        return o.view(b, h, w, m.out_steps, c).permute(0, 3, 4, 1, 2)
    return o.permute(0, 2, 1, 3)
