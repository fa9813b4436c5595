"""TANTE operator on MI355X -- host side of the Taylor-expansion rollout path.

Mirrors ``models.TANTE`` (models/tante.py:37-176), ``enc_CNN`` / ``dec_CNN`` (models/enc_dec_cnn.py:187-277),
``film`` and ``interprator`` (models/tante.py:178-230): same constructor arguments, same parameter names,
shapes and default initialisation (``state_dict()`` is interchangeable with the reference's), same
``forward(input[B,T,D,H,W], out_T=1)`` contract.  The torch.nn modules are parameter containers; the
arithmetic runs in libtante_hip.so:

  patch embed   3 x (kernel = stride conv as a GEMM over non-overlapping patches, GELU in the epilogue);
                the third stage's epilogue applies FiLM(t_seq), + s_emb, + t_emb (tante.py:136-141) and
                writes the fp32 token stream directly -- the encoder never materialises NCHW intermediates.
  backbone      attn_backbone.Attn_Backbone.forward_tokens, in place on the token stream.
  heads         per order: 3 x (ConvTranspose kernel = stride as GEMM + pixel-shuffle scatter epilogue),
                reading the last time slot of the token stream by stride (no slice copy).
  Taylor sum    one pass over (last frame, K derivative fields) -> n_out frames (tante.py:165-171).
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib as L
from . import options as _O
from . import kernels as K
from . import stages as S
from .attn_backbone import Attn_Backbone, _PackCache, _gpu_only, _no_autograd, _wants_grad, resolve_compute

# models/enc_dec_cnn.py:39-46
Patch_map = {64: (4, 4, 4), 32: (4, 4, 2), 16: (4, 2, 2), 8: (2, 2, 2), 4: (2, 2, 1), 2: (2, 1, 1)}


@dataclass
class TanteMetadata:
    """The constructor-argument record of the reference (data/dataset.py:43-63); the operator reads
    only n_fields and spatial_resolution (tante.py:64-66)."""
    dataset_name: str = "synthetic"
    n_spatial_dims: int = 2
    spatial_resolution: Tuple[int, ...] = (128, 384)
    field_names: Optional[Dict[int, List[str]]] = None
    boundary_condition_types: Optional[List[str]] = None
    n_files: int = 1
    n_trajectories_per_file: Optional[List[int]] = None
    n_steps_per_trajectory: Optional[List[int]] = None
    n_fields: int = 4

    @property
    def sample_shapes(self):
        return {"input_fields": [*self.spatial_resolution, self.n_fields],
                "output_fields": [*self.spatial_resolution, self.n_fields],
                "space_grid": [*self.spatial_resolution, self.n_spatial_dims]}


HEAD_MULTI = _O.register("TANTE_HEAD_MULTI", True, __name__, "HEAD_MULTI")      # every Taylor order's derivative head in one launch
HEAD_STREAMS = _O.register("TANTE_HEAD_STREAMS", True, __name__, "HEAD_STREAMS")  # ... reading each order's own stream buffer (no row copies)
HEAD_ENC = _O.register("TANTE_HEAD_ENC", True, __name__, "HEAD_ENC")          # ... and re-encoding the predicted frame in the same launch (head_enc.hip)


def _check_patch_cfg(patch_scale, overlap_ratio):
    """-> per-stage kernel sizes.  Stages with 'same' padding (4x4 kernels, enc_dec_cnn.py:78-81) or overlap (stride < kernel) take
    the general im2col / crop+resize route of stages.py; 1x1 / 2x2 non-overlapping stages keep the dedicated patch GEMM."""
    if not 0.0 <= overlap_ratio < 1.0:
        raise AssertionError("overlap_ratio must be in [0, 1).")
    return Patch_map[patch_scale]


def _film_pos(v: torch.Tensor, film: tuple) -> torch.Tensor:
    fa, fb, se, T, HW = film
    y = torch.empty_like(v)
    L.check(L.lib().tante_film_pos_fwd(v.data_ptr(), fa.data_ptr(), fb.data_ptr(), se.data_ptr(), v.shape[0], v.shape[1], T, HW, y.data_ptr(),
                                       K._stream()), "tante_film_pos_fwd")
    return y


class _ConvHolder(nn.Module):
    """Parameter container with the reference's attribute name (`conv` / `deconv`)."""

    def __init__(self, name: str, mod: nn.Module):
        super().__init__()
        setattr(self, name, mod)


class enc_CNN(nn.Module):
    """3-stage patch embed, (B,T,D,H,W) -> (B,T,Hp,Wp,C)   (enc_dec_cnn.py:187-229)."""

    def __init__(self, dset_metadata=None, embed_dim: int = 256, patch_scale=64, overlap_ratio=0.5):
        super().__init__()
        self.embed_dim = embed_dim
        self.P = _check_patch_cfg(patch_scale, overlap_ratio)
        self.overlap = overlap_ratio
        self.fused = True          # bf16: stages 2 + 3 in one launch when the shape allows
        cin = dset_metadata.n_fields if dset_metadata else 4
        shape = dset_metadata.spatial_resolution if dset_metadata else (128, 384)
        self.H, self.W = shape[0], shape[1]
        self.chans = [cin, embed_dim // 4, embed_dim // 2, embed_dim]
        for i in range(3):
            p = self.P[i]
            st, pd = S.stride_pad(p, overlap_ratio)
            setattr(self, f"enc_conv_{i + 1}", _ConvHolder("conv", nn.Conv2d(self.chans[i], self.chans[i + 1], (p, p), stride=(st, st),
                                                                             padding=(pd, pd))))
        tot = self.P[0] * self.P[1] * self.P[2]
        self.patch_shape = (self.H // tot, self.W // tot)
        self._cache = _PackCache()

    def _packed(self, compute: int):
        convs = [getattr(self, f"enc_conv_{i + 1}").conv for i in range(3)]
        params = [p for c in convs for p in (c.weight, c.bias)]

        def build():
            out = []
            for i, c in enumerate(convs):
                p, ci, co = self.P[i], self.chans[i], self.chans[i + 1]
                if self._general(i):   # im2col route: K-chunked dense weight in the gather's column order
                    out.append(S.pack_linear_chunks(S.conv_weight_2d(c.weight, 0 if i == 0 else 1), c.bias, compute))
                elif i == 0:   # first stage reads the channels-first input: k = (ci, kh, kw) is the native weight order
                    out.append(K.pack_weight(c.weight, c.bias, compute, L.W_LINEAR, N=co, K=ci * p * p))
                else:        # later stages read channels-last intermediates: k = (kh, kw, ci)
                    out.append(K.pack_weight(c.weight, c.bias, compute, L.W_CONV_NHWC, N=co, K=ci * p * p, P=p, C_other=ci))
            return out
        return self._cache.get(compute, params, build)

    def _general(self, i: int) -> bool:
        s_, p_ = S.stride_pad(self.P[i], self.overlap)
        return s_ != self.P[i] or p_ != 0 or self.chans[i] * self.P[i] ** 2 > S.KMAX

    def fuses_23(self, compute: int) -> bool:
        return (compute == L.BF16 and self.fused and self.P == (2, 2, 2) and self.overlap == 0.0 and K.enc23_supported(self.embed_dim))

    def packed_head_enc(self):
        """Encoder stream of the one-launch rollout tail (kernels.head_enc_fused), rebuilt when a conv parameter changes."""
        convs = [getattr(self, f"enc_conv_{i + 1}").conv for i in range(3)]
        params = [q for c in convs for q in (c.weight, c.bias)]
        return self._cache.get(-4, params, lambda: K.pack_head_enc(params, self.embed_dim, self.chans[0]))

    def forward_frames(self, inp: torch.Tensor, compute: int, item_stride: int, z: torch.Tensor) -> torch.Tensor:
        """inp (B, F, D, H, W) fp32 frames -> z (F, B, Hp*Wp, C) fp32: stages 1-3 WITHOUT FiLM, frame-major (the rollout's frame cache).
        Fused bf16 path only (fuses_23)."""
        B, F, D, H, W = inp.shape
        if not self.fuses_23(compute):
            raise RuntimeError("forward_frames needs the fused stage 2 + 3 encoder path")
        pk = self._packed(compute)
        adt = K.act_torch_dtype(compute)
        n_img = B * F
        h1 = torch.empty(n_img * (H // 2) * (W // 2), self.chans[1], dtype=adt, device=inp.device)
        K.patch_embed(inp, pk[0], h1, n_img=n_img, Hin=H, Win=W, Cin=self.chans[0], P=2, nchw=True, act=L.ACT_GELU_ERF, film=None,
                      imgs_per_item=F, item_stride=item_stride)
        convs = [self.enc_conv_2.conv, self.enc_conv_3.conv]
        params = [q for c in convs for q in (c.weight, c.bias)]
        st = self._cache.get(-3, params, lambda: K.pack_enc23(params, self.embed_dim))
        return K.enc23_frames(h1, n_img, F, H // 8, W // 8, self.embed_dim, st, z)

    def forward_tokens(self, inp: torch.Tensor, compute: int, film: Optional[tuple], item_stride: Optional[int] = None,
                       out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """inp (B,T,D,H,W) fp32 -> tokens (B*T*Hp*Wp, C) fp32; `film` = (a, b, s_emb, T, HW) is applied in the
        last stage's epilogue (None: plain encoder output).  inp may be a window view of a longer rollout buffer:
        frames contiguous, `item_stride` elements between batch items.  `out` (fused stages 2 + 3 only): destination rows."""
        B, T, D, H, W = inp.shape
        if (H, W) != (self.H, self.W) or D != self.chans[0]:
            raise ValueError(f"encoder built for {self.chans[0]} fields at {(self.H, self.W)}, got {tuple(inp.shape)}")
        pk = self._packed(compute)
        adt = K.act_torch_dtype(compute)
        n_img, h, w = B * T, H, W
        x = inp
        fuse23 = film is not None and self.fuses_23(compute)
        dst = out
        if dst is not None and not fuse23:
            raise ValueError("out= needs the fused stage 2 + 3 path")
        for i in range(3):
            p, ci, co = self.P[i], self.chans[i], self.chans[i + 1]
            last = i == 2
            if fuse23 and i == 1:   # stages 2 + 3 + FiLM / positional epilogue in one launch; the C/2 intermediate never exists
                convs = [self.enc_conv_2.conv, self.enc_conv_3.conv]
                params = [q for c in convs for q in (c.weight, c.bias)]
                st = self._cache.get(-3, params, lambda: K.pack_enc23(params, self.embed_dim))
                if dst is None:
                    dst = torch.empty(n_img * (h // 4) * (w // 4), self.embed_dim, dtype=torch.float32, device=inp.device)
                return K.enc23_fused(x, n_img, h // 4, w // 4, self.embed_dim, st, film, dst)
            if self._general(i):
                if i == 0:
                    x = x.contiguous().view(n_img, D, H, W)
                out, _, _ = S.conv_stage(x, i == 0, n_img, ci, h, w, p, self.overlap, pk[i], compute,
                                         L.ACT_NONE if last else L.ACT_GELU_ERF, torch.float32 if last else adt)
                if last and film is not None:
                    out = _film_pos(out, film)
                x, h, w = out, h // p, w // p
                continue
            out = torch.empty(n_img * (h // p) * (w // p), co, dtype=torch.float32 if last else adt, device=inp.device)
            K.patch_embed(x, pk[i], out, n_img=n_img, Hin=h, Win=w, Cin=ci, P=p, nchw=(i == 0),
                          act=L.ACT_NONE if last else L.ACT_GELU_ERF, film=film if last else None,
                          imgs_per_item=T if i == 0 else None, item_stride=(item_stride or T * D * H * W) if i == 0 else 0)
            x, h, w = out, h // p, w // p
        return x

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _gpu_only(x)
        B, T = x.shape[:2]
        if _wants_grad(self, x):      # enc_dec_cnn.py:217-229 is differentiable: HIP forward + HIP backward (train_forward.encoder_train)
            from .train_forward import encoder_train
            tok = encoder_train(self, x.to(torch.float32).contiguous(), resolve_compute(None))
            return tok.view(B, T, self.patch_shape[0], self.patch_shape[1], self.embed_dim)
        tok = self.forward_tokens(x.detach().float().contiguous(), resolve_compute(None), None)
        return tok.view(B, T, self.patch_shape[0], self.patch_shape[1], self.embed_dim)


class dec_CNN(nn.Module):
    """3-stage derivative head, (B,T,Hp,Wp,C) -> (B,T,D,H,W)   (enc_dec_cnn.py:232-277)."""

    def __init__(self, dset_metadata=None, embed_dim: int = 256, patch_scale=64, overlap_ratio=0.5):
        super().__init__()
        self.embed_dim = embed_dim
        Pm = _check_patch_cfg(patch_scale, overlap_ratio)
        self.P = (Pm[2], Pm[1], Pm[0])           # enc_dec_cnn.py:251-253: kernel sizes in reverse order
        self.overlap = overlap_ratio
        cout = dset_metadata.n_fields if dset_metadata else 4
        shape = dset_metadata.spatial_resolution if dset_metadata else (128, 384)
        self.H, self.W = shape[0], shape[1]
        self.chans = [embed_dim, embed_dim // 2, embed_dim // 4, cout]
        for i in range(3):
            p = self.P[i]
            st, pd = S.stride_pad(p, overlap_ratio)
            setattr(self, f"dec_conv_{i + 1}", _ConvHolder("deconv", nn.ConvTranspose2d(self.chans[i], self.chans[i + 1], (p, p),
                                                                                         stride=(st, st), padding=(pd, pd))))
        tot = Pm[0] * Pm[1] * Pm[2]
        self.patch_shape = (self.H // tot, self.W // tot)
        self._cache = _PackCache()

    def _packed(self, compute: int):
        convs = [getattr(self, f"dec_conv_{i + 1}").deconv for i in range(3)]
        params = [p for c in convs for p in (c.weight, c.bias)]

        def build():
            out = []
            for i, c in enumerate(convs):
                p, ci, co = self.P[i], self.chans[i], self.chans[i + 1]
                st_, pd_ = S.stride_pad(p, self.overlap)
                if st_ != p:       # overlapping taps: tap GEMM + gather-sum (stages.deconv_stage)
                    out.append(S.deconv_taps_pack(c.weight, c.bias, compute))
                    continue
                padded = pd_ != 0    # padded stages scatter channels-last, then crop + resize
                lay = L.W_DECONV_NCHW if (i == 2 and not padded) else L.W_DECONV_NHWC
                out.append(K.pack_weight(c.weight, c.bias, compute, lay, N=co * p * p, K=ci, P=p, C_other=co))
            return out
        return self._cache.get(compute, params, build)

    def packed_head(self):
        """Weight stream of the fused head kernel (kernels.head_fused), rebuilt when a deconv parameter changes."""
        convs = [getattr(self, f"dec_conv_{i + 1}").deconv for i in range(3)]
        params = [p for c in convs for p in (c.weight, c.bias)]
        return self._cache.get(-2, params, lambda: K.pack_head(params, self.embed_dim, self.chans[3]))

    def forward_tokens(self, src: torch.Tensor, n_img: int, compute: int, a_n0: int, a_s1: int, a_s0: int, a_off: int) -> torch.Tensor:
        """Rows (img, hp, wp) of `src` (gathered by the given strides, in elements) -> (n_img, D, H, W) fp32."""
        pk = self._packed(compute)
        adt = K.act_torch_dtype(compute)
        h, w = self.patch_shape
        x = src
        for i in range(3):
            p, co = self.P[i], self.chans[i + 1]
            last = i == 2
            if S.stride_pad(p, self.overlap) != (p, 0):   # padded stage: unpadded deconv -> crop + bilinear resize -> act
                rows = dict(a_n0=a_n0, a_s1=a_s1, a_s0=a_s0, a_off=a_off) if i == 0 else {}
                x = S.deconv_stage(x, n_img, h, w, p, self.overlap, pk[i], co, compute, L.ACT_NONE if last else L.ACT_GELU_ERF, last,
                                   torch.float32 if last else adt, **rows)
                h, w = h * p, w * p
                continue
            if last:
                out = torch.empty(n_img, co, h * p, w * p, dtype=torch.float32, device=src.device)
            else:
                out = torch.empty(n_img, h * p, w * p, co, dtype=adt, device=src.device)
            if i == 0:
                K.deconv(x, pk[i], out, n_img=n_img, Hi=h, Wi=w, P=p, Cout=co, nchw_out=last,
                         act=L.ACT_NONE if last else L.ACT_GELU_ERF, a_n0=a_n0, a_s1=a_s1, a_s0=a_s0, a_off=a_off)
            else:
                K.deconv(x, pk[i], out, n_img=n_img, Hi=h, Wi=w, P=p, Cout=co, nchw_out=last,
                         act=L.ACT_NONE if last else L.ACT_GELU_ERF)
            x, h, w = out, h * p, w * p
        return x

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _gpu_only(x)
        B, T, Hp, Wp, C_ = x.shape
        if _wants_grad(self, x):      # enc_dec_cnn.py:263-277
            from .train_forward import decoder_train
            y = decoder_train(self, x.to(torch.float32).reshape(B * T * Hp * Wp, C_).contiguous(), B * T, resolve_compute(None))
            return y.view(B, T, *y.shape[1:])
        src = x.detach().float().contiguous()
        y = self.forward_tokens(src, B * T, resolve_compute(None), B * T * Hp * Wp, 0, C_, 0)
        return y.view(B, T, *y.shape[1:])


class film(nn.Module):
    """x + (x * scale(t) + shift(t))   (tante.py:203-230).  Parameter container + table builder."""

    def __init__(self, h_dim=768, in_dim=1):
        super().__init__()
        self.h_dim = h_dim
        self.condition_to_scale = nn.Sequential(nn.Linear(in_dim, h_dim // 2), nn.ReLU(), nn.Linear(h_dim // 2, h_dim))
        self.condition_to_shift = nn.Sequential(nn.Linear(in_dim, h_dim // 2), nn.ReLU(), nn.Linear(h_dim // 2, h_dim))

    def tables(self, t: torch.Tensor, add: Optional[torch.Tensor] = None):
        """-> (a, b) with a = 1 + scale(t), b = shift(t) (+ add); film(x, t) = x * a + b."""
        s, h = self.condition_to_scale, self.condition_to_shift
        return K.film_table(t, (s[0].weight, s[0].bias, s[2].weight, s[2].bias, h[0].weight, h[0].bias, h[2].weight, h[2].bias),
                            self.h_dim, add)

    def forward(self, x: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        _gpu_only(x)
        if _wants_grad(self, x, t):   # tante.py:218-230: x + (x * scale(t) + shift(t)); the two small MLPs are torch (parameter-sized), the
            from .autograd import FilmPosFn      # broadcast multiply-add over the tokens and its backward are HIP (FilmPosFn)
            tt = t.to(x.device, torch.float32).reshape(-1, 1)
            fa = (1.0 + self.condition_to_scale(tt)).float().contiguous()
            fb = self.condition_to_shift(tt).float().contiguous()
            C_ = x.shape[-1]
            if x.dim() == 5:          # (B,T,H,W,C), t (T,): table row = time slot
                B, T, H, W = x.shape[:4]
                zero = torch.zeros(H * W, C_, dtype=torch.float32, device=x.device)
                y = FilmPosFn.apply(x.to(torch.float32).reshape(B * T * H * W, C_).contiguous(), fa, fb, zero, T, H * W)
            elif x.dim() == 3:        # (B,L,C), t (B,): table row = batch item
                B, Lq = x.shape[:2]
                zero = torch.zeros(Lq, C_, dtype=torch.float32, device=x.device)
                y = FilmPosFn.apply(x.to(torch.float32).reshape(B * Lq, C_).contiguous(), fa, fb, zero, B, Lq)
            else:
                raise ValueError("film expects a 3-D or 5-D tensor")
            return y.view(x.shape)
        x = x.detach().float().contiguous()
        a, b = self.tables(t.detach().float().contiguous().to(x.device))
        C_ = x.shape[-1]
        y = torch.empty_like(x)
        if x.dim() == 5:      # (B,T,H,W,C), t (T,): one table row per time slot
            B, T = x.shape[:2]
            hw = x.shape[2] * x.shape[3]
            for bi in range(B):
                K.film_apply(x, bi * T * hw * C_, hw * C_, y[bi], T * hw, C_, hw, a, b)
        elif x.dim() == 3:    # (B,L,C), t (B,)
            B, Lq = x.shape[:2]
            K.film_apply(x, 0, Lq * C_, y, B * Lq, C_, Lq, a, b)
        else:
            raise ValueError("film expects a 3-D or 5-D tensor")
        return y


class interprator(nn.Module):
    """Adaptive step-size head (tante.py:178-201): per-token MLP C -> C/2 -> C/4 -> 1, clamp, mean, + ep."""

    def __init__(self, h_dim=768, sp_dim=16, ep=1.001):
        super().__init__()
        self.sp_dim, self.ep, self.h_dim = sp_dim, ep, h_dim
        self.interprete = nn.Sequential(nn.Linear(h_dim, h_dim // 2), nn.ReLU(), nn.Linear(h_dim // 2, h_dim // 4), nn.ReLU(),
                                        nn.Linear(h_dim // 4, 1))
        self._cache = _PackCache()

    def _packed(self, compute):
        lin = [self.interprete[0], self.interprete[2], self.interprete[4]]
        return self._cache.get(compute, [p for l in lin for p in (l.weight, l.bias)],
                               lambda: [K.pack_weight(l.weight, l.bias, compute) for l in lin])

    def forward_tokens(self, src: torch.Tensor, B: int, out_T: float, compute: int, a_n0: int, a_s1: int, a_s0: int, a_off: int):
        pk = self._packed(compute)
        n = B * self.sp_dim
        adt = K.act_torch_dtype(compute)
        h1 = torch.empty(n, self.h_dim // 2, dtype=adt, device=src.device)
        K.linear(src, pk[0], h1, M=n, act=L.ACT_RELU, a_n0=a_n0, a_s1=a_s1, a_s0=a_s0, a_off=a_off)
        h2 = torch.empty(n, self.h_dim // 4, dtype=adt, device=src.device)
        K.linear(h1, pk[1], h2, M=n, act=L.ACT_RELU)
        t = torch.empty(n, 1, dtype=torch.float32, device=src.device)
        K.linear(h2, pk[2], t, M=n)
        return K.rt_reduce(t, B, self.sp_dim, out_T, self.ep)

    def forward(self, x: torch.Tensor, out_T) -> torch.Tensor:
        _gpu_only(x)
        B, Lq, C_ = x.shape
        if _wants_grad(self, x):      # tante.py:191-201 incl. the straight-through clamp's gradient (RtReduceFn)
            from .train_forward import interprator_train
            return interprator_train(self, x.to(torch.float32).reshape(B * Lq, C_).contiguous(), B, out_T, resolve_compute(None))
        src = x.detach().float().contiguous()
        return self.forward_tokens(src, B, out_T, resolve_compute(None), B * Lq, 0, C_, 0)


def get_1d_sincos_pos_embed_from_grid(embed_dim, pos):
    """tante.py:232-242 (initial value of a learned parameter; checkpoints override it)."""
    omega = torch.arange(embed_dim // 2, dtype=torch.float32) / (embed_dim / 2.0)
    omega = 1.0 / 10000 ** omega
    out = pos.reshape(-1)[:, None] * omega[None, :]
    return torch.cat([torch.sin(out), torch.cos(out)], dim=1)


def t_emb_init(embed_dim, length):
    return get_1d_sincos_pos_embed_from_grid(embed_dim, torch.arange(length, dtype=torch.float32)).unsqueeze(0)


def s_emb_init(embed_dim, grid_size, *, flatten: bool = False):
    """tante.py:251-276, including its meshgrid('ij') of (w, h) reshaped to (2,1,H,W)."""
    H, W = grid_size
    gw, gh = torch.meshgrid(torch.arange(W, dtype=torch.float32), torch.arange(H, dtype=torch.float32), indexing="ij")
    grid = torch.stack([gh, gw], dim=0).reshape(2, 1, H, W)
    emb = torch.cat([get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[0]),
                     get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[1])], dim=1)
    return emb.unsqueeze(0) if flatten else emb.view(H, W, embed_dim).unsqueeze(0)


def t_series(IP, frame_interval):
    """tante.py:279-285: [.., -2dt, -dt, 0, 0] -- the duplicated zero is the reference's behaviour."""
    seq = [0.0] + [-i * frame_interval for i in range(IP - 1)]
    seq.reverse()
    return torch.tensor(seq)


class TANTE(nn.Module):
    def __init__(self, in_T, dset_metadata: TanteMetadata = None, taylor_order: int = 1, frame_interval: float = 1.0,
                 output_length=1, attn_axes: str = "THWTHWTHW", expanded_channel: int = 128, n_head: int = 8,
                 mlp_ratio: float = 1.0, dropout: float = 0.0, enc_dec_type: str = "cnn", embed_dim: int = 256,
                 modes1: int = 32, modes2: int = 32, patch_scale: int = 32, overlap_ratio: float = 0.0, deg: bool = True):
        super().__init__()
        n_channel = dset_metadata.n_fields if dset_metadata else 4
        shape = dset_metadata.spatial_resolution if dset_metadata else (128, 384)
        self.T = in_T
        self.D, self.H, self.W = n_channel, shape[0], shape[1]
        self.H_p, self.W_p = shape[0] // patch_scale, shape[1] // patch_scale
        self.C = embed_dim
        self.taylor_order, self.frame_interval, self.output_length, self.deg = taylor_order, frame_interval, output_length, deg
        self.attn_axes = attn_axes.replace(" ", "")
        if set(self.attn_axes) - {"T", "H", "W", "L", "A", "C", "X", "Y", "-"}:   # the reference's set has the typo 'X,'
            raise ValueError("There are invalid letters")
        self.blocks_axes = [p.strip() for p in self.attn_axes.split("-")]
        if len(self.blocks_axes) != taylor_order:
            raise ValueError(f"Block allocation doesn't match expansion order: expected {taylor_order} parts, "
                             f"got {len(self.blocks_axes)} (input='{self.attn_axes}').")
        self.decoders = nn.ModuleList()
        if enc_dec_type == "cnn":                                                                    # tante.py:86-93
            self.encoder = enc_CNN(dset_metadata=dset_metadata, embed_dim=embed_dim, patch_scale=patch_scale, overlap_ratio=overlap_ratio)
            for _ in range(taylor_order):
                self.decoders.append(dec_CNN(dset_metadata=dset_metadata, embed_dim=embed_dim, patch_scale=patch_scale,
                                             overlap_ratio=overlap_ratio))
        elif enc_dec_type == "fno":                                                                  # tante.py:94-101
            from .spectral import enc_FNO, dec_FNO
            self.encoder = enc_FNO(dset_metadata=dset_metadata, embed_dim=embed_dim, modes=(modes1, modes2), patch_scale=patch_scale,
                                   overlap_ratio=overlap_ratio)
            for _ in range(taylor_order):
                self.decoders.append(dec_FNO(dset_metadata=dset_metadata, embed_dim=embed_dim, modes=(modes1, modes2),
                                             patch_scale=patch_scale, overlap_ratio=overlap_ratio))
        else:
            raise ValueError(f"enc_dec_type must be 'cnn' or 'fno', got {enc_dec_type!r}")
        self.enc_dec_type = enc_dec_type
        self.blocks = nn.ModuleList()
        for block_axes in self.blocks_axes:
            self.blocks.append(Attn_Backbone(tensor_shape=(self.T, self.H_p, self.W_p, self.C), attn_axes=block_axes,
                                             expanded_channel=expanded_channel, n_head=n_head, mlp_ratio=mlp_ratio, dropout=dropout))
        self.t_emb = nn.Parameter(t_emb_init(self.C, self.T))
        self.s_emb = nn.Parameter(s_emb_init(self.C, (self.H_p, self.W_p), flatten=False))
        self.t_seq = t_series(self.T, frame_interval)      # plain attribute, not in the state_dict (tante.py:118)
        self.t_encode = film(self.C, in_dim=1)
        if not self.deg:
            self.interprators = nn.ModuleList([interprator(self.C, self.H_p * self.W_p) for _ in range(taylor_order)])
            self.modifiers = nn.ModuleList([film(self.C, in_dim=1) for _ in range(taylor_order)])
        self.compute: Optional[str] = None       # None: follow torch.autocast; "fp32" / "bf16": pinned
        self.fused_head = True                   # bf16: fused derivative head + Taylor accumulation when the shape allows
        self._film_cache = _PackCache()

    def set_compute(self, mode: Optional[str]):
        if mode is not None and mode not in K.COMPUTE:
            raise ValueError("compute must be None, 'fp32' or 'bf16'")
        self.compute = mode
        return self

    def _apply(self, fn, *a, **k):               # keep t_seq with the parameters on .to()/.cuda()
        super()._apply(fn, *a, **k)
        self.t_seq = fn(self.t_seq)
        return self

    def _time_tables(self):
        """FiLM(t_seq) scale/shift with t_emb folded into the shift: input independent, rebuilt only when
        the weights change."""
        te = self.t_encode
        params = list(te.parameters()) + [self.t_emb]
        return self._film_cache.get(0, params, lambda: te.tables(self.t_seq.to(self.t_emb.device, torch.float32).contiguous(),
                                                                 self.t_emb.view(self.T, self.C)))

    # ---- frame-encoding cache (rollout loops) -------------------------------------------------------------------------------------
    # The encoder is per frame and FiLM(t) / the positional embeddings are applied after it, so a frame's encoding before FiLM does not
    # depend on the window it is read in: a sliding-window rollout needs each frame encoded ONCE, not once per window containing it
    # (T times).  encode_frame() writes that pre-FiLM encoding; forward(enc_cache=...) skips the encoder and lets the first propagator
    # kernel apply FiLM while it loads the planes.  Same arithmetic per token, term for term.
    def enc_cache_supported(self) -> bool:
        return self._enc_cache_fused() or self._enc_cache_frames()

    def _enc_cache_fused(self) -> bool:
        """enc_CNN with the fused stages: the first propagator launch applies FiLM while it loads the cached planes."""
        compute = resolve_compute(self.compute)
        return bool(self.deg and type(self.encoder).__name__ == "enc_CNN" and self.encoder.fuses_23(compute)
                    and K.axis_hw_supported(self.H_p, self.W_p, self.C))

    def _enc_cache_frames(self) -> bool:
        """The spectral encoder (round 5): enc_FNO is per frame too (enc_dec_fno.py:224-273: every layer acts on (B T) images), so the
        rollout encodes each frame once and a window is T cached encodings + one FiLM / positional pass over them
        (tante_film_pos_fwd_frames: the same expression per token as the dense pass).  At cfg5 the encoder was 42 % of a model call and
        three of its four frames had been encoded by the previous calls."""
        return bool(self.deg and type(self.encoder).__name__ == "enc_FNO" and self.C == 256 and 1 <= self.T <= 8)

    def encode_frames(self, frames: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
        """frames: (B, F, D, H, W) fp32 view (contiguous frames, any batch stride) -> z (F, B, Hp*Wp, C) fp32, the encoder output before
        FiLM in the frame-major layout forward(enc_cache=...) reads (one launch pair for the F frames)."""
        compute = resolve_compute(self.compute)
        if self._enc_cache_frames() and not self._enc_cache_fused():
            B, F = frames.shape[:2]
            # one frame: the rows (b, hw) ARE the cache entry -- the encoder's last GEMM writes them there
            want = z.view(B * self.H_p * self.W_p, self.C) if (F == 1 and z.is_contiguous()) else None
            y = self.encoder.forward_tokens(frames.detach().to(torch.float32), compute, None, out=want)        # rows (b, f, hw)
            if want is None or y.data_ptr() != want.data_ptr():
                z.copy_(y.view(B, F, self.H_p * self.W_p, self.C).transpose(0, 1))
            return z
        self.encoder.forward_frames(frames, compute, frames.stride(0), z)
        return z

    def encode_frame(self, frame: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
        """One frame: (B, 1, D, H, W) -> z (B, Hp*Wp, C)."""
        return self.encode_frames(frame, z.view(1, *z.shape))

    def tail_fused_supported(self) -> bool:
        """The token-local tail of a call -- every order's derivative head, the Taylor sum and (in a rollout) the re-encoding of the
        predicted frame -- as ONE launch (csrc/head_enc.hip): bf16, C = 256, the three 2 x 2 stages of patch_scale 8, one output frame."""
        compute = resolve_compute(self.compute)
        return bool(HEAD_ENC and self.deg and self.output_length == 1 and 1 <= self.taylor_order <= 4 and compute == L.BF16 and self.fused_head
                    and type(self.encoder).__name__ == "enc_CNN" and getattr(self.decoders[0], 'P', None) == (2, 2, 2)
                    and self.decoders[0].overlap == 0.0 and K.head_enc_supported(self.C, self.D, self.H_p, self.W_p)
                    and all(b.takes_x_in(compute) for b in self.blocks[1:self.taylor_order]))

    def forward(self, input: torch.Tensor, out_T=1, out: Optional[torch.Tensor] = None, enc_cache: Optional[tuple] = None,
                enc_next: Optional[torch.Tensor] = None, per_sample_counts: bool = False):
        """`out` (optional, deg=True only): a (B, output_length, D, H, W) fp32 view with contiguous frames (e.g. the next
        slots of a rollout buffer) that receives the prediction instead of a fresh tensor.
        `enc_cache` = (z, t_stride, b_stride): the window's frames already encoded by encode_frame (frame t of item b at
        z + t * t_stride + b * b_stride); see enc_cache_supported().
        `enc_next` (tail_fused_supported() only): a contiguous (B, Hp*Wp, C) fp32 tensor that receives the pre-FiLM encoding of the
        PREDICTED frame -- what encode_frame would compute from it -- out of the same launch that writes the frame.
        `per_sample_counts` (deg=False): the reference lets sample 0's floor(R_t[0]) decide the frame count of the whole batch
        (tante.py:163), which is why R_Trainer loops over the samples one at a time (r_trainer.py:118-119).  With this flag the call
        produces max_i floor(R_t[i]) frames; frame j of sample i is what a single-sample call would give for j < floor(R_t[i]) (every op
        of the path is per sample), and the caller keeps floor(R_t[i]) frames of sample i (rollout.rollout_adaptive).
        With autograd enabled the differentiable path (train_forward.py: HIP forward + HIP backward kernels) runs."""
        if not input.is_cuda:
            raise RuntimeError("tante_amd.TANTE runs on the GPU only (no CPU fallback); move the input to cuda")
        if input.shape[1] != self.T:
            input = input[:, -self.T:]
        if torch.is_grad_enabled() and (input.requires_grad or any(p.requires_grad for p in self.parameters())):
            from .train_forward import tante_train_forward
            if out is not None:
                raise ValueError("out= is an inference-path option")
            return tante_train_forward(self, input.to(torch.float32).contiguous(), resolve_compute(self.compute), out_T,
                                       per_sample_counts=per_sample_counts)
        inp = input.detach().to(torch.float32)
        B, T, D, H, W = inp.shape
        frame = D * H * W
        if inp.stride()[1:] != (frame, H * W, W, 1) or inp.stride(0) % 4 or inp.data_ptr() % 16:
            inp = inp.contiguous()      # only a (B, T_total, D, H, W) time window is read in place
        bstride = inp.stride(0)
        compute = resolve_compute(self.compute)
        Hp, Wp, C_ = self.H_p, self.W_p, self.C
        HW = Hp * Wp
        fa, fb = self._time_tables()
        film = (fa, fb, self.s_emb.view(HW, C_), T, HW)
        film_frames = None
        if enc_cache is not None:
            if not self.enc_cache_supported():
                raise RuntimeError("enc_cache: this model / compute mode has no frame-encoding cache path")
            x = torch.empty(B * T * HW, C_, dtype=torch.float32, device=inp.device)
            if not self._enc_cache_fused():      # cached frames + FiLM / positional terms in one pass (tante.py:136-141 over the window)
                zc = enc_cache[0]
                fr = L.Frames()
                for t in range(T):
                    f = zc[t]
                    if tuple(f.shape) != (B, HW, C_) or f.dtype != torch.float32 or f.stride(2) != 1 or f.stride(1) != C_ or f.data_ptr() % 16:
                        raise ValueError("enc_cache: frames must be (B, Hp*Wp, C) fp32 with contiguous rows")
                    fr.f[t], fr.bstride[t] = f.data_ptr(), f.stride(0)
                vp0 = self.blocks[0].vertical_propagator
                if (compute == L.BF16 and not K.axis_hw_supported(Hp, Wp, C_, compute)
                        and L.lib().tante_axis_mlp_film_supported(B, T, Hp, Wp * C_, C_)
                        and all(q.data_ptr() % 16 == 0 for q in (vp0[0].weight, vp0[0].bias, vp0[2].weight, vp0[2].bias, fa, fb, self.s_emb))):
                    # the first backbone's vertical propagator applies FiLM + the positional terms while it loads its tiles
                    # (tante_axis_mlp_film): x is produced by that launch, the pass below is not run
                    film_frames = (fr, zc, fa, fb, self.s_emb.view(HW, C_))
                else:
                    L.check(L.lib().tante_film_pos_fwd_frames(C.byref(fr), fa.data_ptr(), fb.data_ptr(), self.s_emb.view(HW, C_).data_ptr(), B, T, HW, C_,
                                                              x.data_ptr(), K._stream()), "tante_film_pos_fwd_frames")
                enc_cache = None
        else:
            x = self.encoder.forward_tokens(inp, compute, film, bstride)                               # tante.py:132-141
        last_slot = dict(a_n0=HW, a_s1=T * HW * C_, a_s0=C_, a_off=(T - 1) * HW * C_)               # x[:, -1:] by stride
        fused_head = (compute == L.BF16 and self.fused_head and getattr(self.decoders[0], 'P', None) == (2, 2, 2) and self.decoders[0].overlap == 0.0
                      and K.head_fused_supported(C_, D))
        if out is not None:
            if not self.deg:
                raise ValueError("out= is only meaningful with a fixed output length (deg=True)")
            if tuple(out.shape) != (B, self.output_length, D, H, W) or out.dtype != torch.float32 or not out.is_cuda \
                    or out.stride()[1:] != (frame, H * W, W, 1) or out.stride(0) % 4 or out.data_ptr() % 16:
                raise ValueError("out must be a (B, output_length, D, H, W) fp32 CUDA view with contiguous frames")
        derivs, r_t, srcs = [], [], []
        # one prediction frame, several Taylor orders: ONE head launch after the last backbone (tante_head_fused_multi) instead of one per
        # order -- the frame is read and written once instead of taylor_order times.  The earlier orders' last-slot rows are copied aside
        # (8 MB each at cfg2) because the later backbones update the stream in place.  TANTE_HEAD_MULTI=0: one launch per order (A/B).
        tail_fused = self.tail_fused_supported()
        if enc_next is not None:
            if not tail_fused or not self.encoder.fuses_23(compute):
                raise RuntimeError("enc_next: this model / compute mode has no fused head + re-encoding path (tail_fused_supported())")
            if tuple(enc_next.shape) != (B, HW, C_) or enc_next.dtype != torch.float32 or not enc_next.is_cuda or not enc_next.is_contiguous():
                raise ValueError("enc_next must be a contiguous (B, Hp*Wp, C) fp32 CUDA tensor")
        multi_head = (self.deg and fused_head and self.output_length == 1 and 2 <= self.taylor_order <= 4 and HEAD_MULTI) or tail_fused
        saved_rows = []
        # ... or not copied at all: every later backbone writes a stream buffer of its own (its first launch, the H + W propagator pass,
        # runs out of place: tante_axis_hw_oop), so the earlier orders' streams stay intact for the head.  TANTE_HEAD_STREAMS=0: copies.
        streams = tail_fused or (multi_head and HEAD_STREAMS and all(b.takes_x_in(compute) for b in self.blocks[1:self.taylor_order]))
        for i in range(self.taylor_order):
            if streams and i > 0:
                x_prev, x = x, torch.empty_like(x)
                self.blocks[i].forward_tokens(x, B, compute, x_in=x_prev)
            else:
                self.blocks[i].forward_tokens(x, B, compute, film_src=(enc_cache + (film,)) if (enc_cache is not None and i == 0) else None,
                                              film_frames=film_frames if i == 0 else None)  # l.146 (chained)
            if self.deg:
                if multi_head:
                    if i + 1 < self.taylor_order:
                        saved_rows.append(x if streams else
                                          x.view(B, T, HW * C_)[:, T - 1].clone(memory_format=torch.contiguous_format).view(B * HW, C_))      # (a COPY also at B = 1)
                        continue
                    if out is None:
                        out = torch.empty(B, 1, D, H, W, dtype=torch.float32, device=x.device)
                    coefs = [self.frame_interval ** (k + 1) / math.factorial(k + 1) for k in range(self.taylor_order)]
                    if tail_fused:      # heads + Taylor sum (+ the predicted frame's encoding for the next call) in one launch
                        K.head_enc_fused(saved_rows + [x], HW, T * HW * C_, C_, (T - 1) * HW * C_, B, Hp, Wp, C_, D,
                                         [self.decoders[k].packed_head() for k in range(self.taylor_order)], coefs, out, out.stride(0),
                                         inp, (T - 1) * frame, bstride,
                                         enc_stream=self.encoder.packed_head_enc() if enc_next is not None else None, z=enc_next)
                        continue
                    K.head_fused_multi(saved_rows + [x], HW, T * HW * C_, C_, (T - 1) * HW * C_, B, Hp, Wp, C_, D,
                                       [self.decoders[k].packed_head() for k in range(self.taylor_order)], coefs, out, out.stride(0),
                                       inp, (T - 1) * frame, bstride, streams=streams)                          # l.147,153,165-171
                elif fused_head:
                    if out is None:
                        out = torch.empty(B, self.output_length, D, H, W, dtype=torch.float32, device=x.device)
                    coefs = [(j * self.frame_interval) ** (i + 1) / math.factorial(i + 1) for j in range(1, self.output_length + 1)]
                    K.head_fused(x, HW, T * HW * C_, C_, (T - 1) * HW * C_, B, Hp, Wp, C_, D, self.decoders[i].packed_head(), out,
                                 out.stride(0), coefs, inp if i == 0 else None, (T - 1) * frame, bstride)     # l.147,153,165-171
                else:
                    derivs.append(self.decoders[i].forward_tokens(x, B, compute, **last_slot))      # l.147,153
            else:
                # intended semantics of l.148-152 (the shipped glue raises): d3 = last slot as (B, L, C);
                # rt = interprator(d3, out_T); d3 = film3d(d3, rt); decode
                rt = self.interprators[i].forward_tokens(x, B, out_T, compute, **last_slot)
                r_t.append(rt)
                ma, mb = self.modifiers[i].tables(rt)
                d3 = torch.empty(B * HW, C_, dtype=torch.float32, device=x.device)
                K.film_apply(x, (T - 1) * HW * C_, T * HW * C_, d3, B * HW, C_, HW, ma, mb)
                srcs.append(d3)
        if self.deg:
            n_out, R_t = self.output_length, None
            if fused_head:
                return out
        else:
            R_t = torch.stack(r_t, dim=1).mean(dim=1)
            # l.163: sample 0 decides for the batch (host sync, as in the reference); per_sample_counts: enough frames for every sample
            n_out = int(torch.floor(R_t).max()) if per_sample_counts else math.floor(float(R_t[0]))
            if n_out >= 1 and fused_head and n_out <= 8:
                out = torch.empty(B, n_out, D, H, W, dtype=torch.float32, device=x.device)
                for i, d3 in enumerate(srcs):
                    coefs = [(j * self.frame_interval) ** (i + 1) / math.factorial(i + 1) for j in range(1, n_out + 1)]
                    K.head_fused(d3, B * HW, 0, C_, 0, B, Hp, Wp, C_, D, self.decoders[i].packed_head(), out, out.stride(0), coefs,
                                 inp if i == 0 else None, (T - 1) * frame, bstride)
                return out, R_t
            derivs = [self.decoders[i].forward_tokens(d3, B, compute, B * HW, 0, C_, 0) for i, d3 in enumerate(srcs)]
        if n_out < 1:
            out = torch.empty(B, 0, D, H, W, dtype=torch.float32, device=x.device)
        else:
            if out is None:
                out = torch.empty(B, n_out, D, H, W, dtype=torch.float32, device=x.device)
            K.taylor(inp, (T - 1) * frame, bstride, derivs, self.frame_interval, n_out, out, B, frame,
                     out_bstride=out.stride(0))                                                     # l.165-171
        return out if self.deg else (out, R_t)
