"""Spectral operator path on the HIP kernels: SpectralLayer, enc_FNO, dec_FNO  (reference models/enc_dec_fno.py:184-323).

Same constructor arguments and state_dict keys as the reference (`weight` is a complex64 parameter, `w0` a 1x1 Conv2d,
`enc_conv_*.conv` / `dec_conv_*.deconv` the strided stages).  The layer itself is one C-ABI call (tante_spectral_layer): hipFFT
R2C -> low-mode complex contraction (both 'ortho' factors folded in) -> hipFFT C2R -> 1x1 conv + sum + activation.
Inference path only (no autograd graph); no CPU fallback."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import _lib as L
from . import kernels as K
from . import stages as S
from .attn_backbone import _PackCache, _no_autograd, resolve_compute

# models/enc_dec_fno.py:39-46 -- the spectral encoder's OWN two-stage patch table
Patch_map_fno = {64: (8, 8), 32: (8, 4), 16: (4, 4), 8: (4, 2), 4: (2, 2), 2: (2, 1)}


class _Holder(nn.Module):
    def __init__(self, name: str, mod: nn.Module):
        super().__init__()
        setattr(self, name, mod)


class SpectralLayer(nn.Module):
    """enc_dec_fno.py:184-222."""

    def __init__(self, in_channels, out_channels, modes1, modes2):
        super().__init__()
        self.in_channels, self.out_channels, self.modes1, self.modes2 = in_channels, out_channels, modes1, modes2
        self.weight = nn.Parameter(torch.randn(in_channels, out_channels, modes1, modes2, dtype=torch.cfloat)
                                   * (1.0 / (in_channels * out_channels) ** 0.5))
        self.w0 = nn.Conv2d(in_channels, out_channels, kernel_size=1, bias=True)
        self._cache = _PackCache()

    def _planes(self):
        return self._cache.get(0, [self.weight], lambda: (self.weight.detach().real.contiguous(), self.weight.detach().imag.contiguous()))

    def run(self, x: torch.Tensor, act: int = L.ACT_NONE, compute: int = L.F32, bf16_out: bool = False, nhwc_out: bool = False) -> torch.Tensor:
        """x (n, Cin, H, W) fp32 (contiguous, or contiguous images with a batch stride) -> act(layer(x)); compute: the model's mode;
        bf16_out: the consumer rounds to bf16 anyway; nhwc_out: channels-last rows (n H W, Cout) instead of the image (kernels.spectral_layer)."""
        if x.dim() != 4 or x.size(1) != self.in_channels:
            raise AssertionError("SpectralLayer expects (B, Cin, H, W)")
        re, im = self._planes()
        w0 = self.w0.weight.detach().view(self.out_channels, self.in_channels)
        return K.spectral_layer(x, re, im, self.modes1, self.modes2, w0, self.w0.bias.detach(), act, compute, bf16_out, nhwc_out)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _no_autograd(self)
        return self.run(x.detach().float().contiguous())


class enc_FNO(nn.Module):
    """(B,T,D,H,W) -> (B,T,Hp,Wp,C): spectral -> GELU -> conv(P0) -> GELU -> spectral -> GELU -> conv(P1)  (enc_dec_fno.py:224-273)."""

    def __init__(self, dset_metadata=None, embed_dim: int = 256, modes: Tuple[int, int] = (32, 32), patch_scale=64, overlap_ratio=0.5):
        super().__init__()
        self.embed_dim = embed_dim
        m1, m2 = modes
        self.P = Patch_map_fno[patch_scale]
        self.overlap = overlap_ratio
        cin = dset_metadata.n_fields if dset_metadata else 4
        shape = dset_metadata.spatial_resolution if dset_metadata else (128, 384)
        self.H, self.W = shape[0], shape[1]
        C_ = embed_dim
        self.chans = [cin, C_ // 8, C_ // 4, C_ // 2, C_]
        self.enc_spectral_1 = SpectralLayer(cin, C_ // 8, m1, m2)
        self.enc_conv_1 = self._conv(C_ // 8, C_ // 4, self.P[0])
        self.enc_spectral_2 = SpectralLayer(C_ // 4, C_ // 2, m1 // self.P[0], m2 // self.P[0])
        self.enc_conv_2 = self._conv(C_ // 2, C_, self.P[1])
        self.patch_shape = (self.H // (self.P[0] * self.P[1]), self.W // (self.P[0] * self.P[1]))
        self._cache = _PackCache()

    def _conv(self, ci, co, p):
        st, pd = S.stride_pad(p, self.overlap)
        return _Holder("conv", nn.Conv2d(ci, co, (p, p), stride=(st, st), padding=(pd, pd)))

    def _packed(self, compute: int):
        convs = [self.enc_conv_1.conv, self.enc_conv_2.conv]
        params = [p for c in convs for p in (c.weight, c.bias)]
        return self._cache.get(compute, params,
                               lambda: [S.pack_linear_chunks(S.conv_weight_2d(c.weight, 0), c.bias, compute) for c in convs])

    def forward_tokens(self, inp: torch.Tensor, compute: int, film: Optional[tuple], item_stride: Optional[int] = None,
                       out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """out (no FiLM): a contiguous (B T Hp Wp, C) fp32 tensor that receives the token rows (the rollout's frame cache)."""
        B, T, D, H, W = inp.shape
        if (H, W) != (self.H, self.W) or D != self.chans[0]:
            raise ValueError(f"encoder built for {self.chans[0]} fields at {(self.H, self.W)}, got {tuple(inp.shape)}")
        pk = self._packed(compute)
        n = B * T
        # one frame of every batch item out of a longer buffer (the rollout's re-encoding of a predicted frame): read in place
        z = inp[:, 0] if (T == 1 and not inp.is_contiguous() and inp[0].is_contiguous()) else inp.contiguous().view(n, D, H, W)
        # (bf16 mode: the conv's patch gather rounds the image to bf16 anyway -- the spectral layer's last kernel does it while storing)
        z = self.enc_spectral_1.run(z, L.ACT_GELU_ERF, compute, bf16_out=True)
        # (channels-first for the spectral layer: written by the patch GEMM's epilogue, stages.conv_stage)
        y, h, w = S.conv_stage(z, True, n, self.chans[1], H, W, self.P[0], self.overlap, pk[0], compute, L.ACT_GELU_ERF, torch.float32, nchw_out=True)
        z = self.enc_spectral_2.run(y, L.ACT_GELU_ERF, compute)
        y, h, w = S.conv_stage(z, True, n, self.chans[3], h, w, self.P[1], self.overlap, pk[1], compute, L.ACT_NONE, torch.float32,
                               out=out if film is None else None)
        if film is not None:
            fa, fb, se, Tt, HW = film
            out = torch.empty_like(y)
            L.check(L.lib().tante_film_pos_fwd(y.data_ptr(), fa.data_ptr(), fb.data_ptr(), se.data_ptr(), y.shape[0], y.shape[1], Tt, HW,
                                               out.data_ptr(), K._stream()), "tante_film_pos_fwd")
            y = out
        return y

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _no_autograd(self)
        B, T = x.shape[:2]
        tok = self.forward_tokens(x.detach().float().contiguous(), resolve_compute(None), None)
        return tok.view(B, T, self.patch_shape[0], self.patch_shape[1], self.embed_dim)


class dec_FNO(nn.Module):
    """(B,T,Hp,Wp,C) -> (B,T,D,H,W): deconv(P1) -> GELU -> spectral -> GELU -> deconv(P0) -> GELU -> spectral  (enc_dec_fno.py:276-323)."""

    def __init__(self, dset_metadata=None, embed_dim: int = 256, modes: Tuple[int, int] = (32, 32), patch_scale=64, overlap_ratio=0.5):
        super().__init__()
        self.embed_dim = embed_dim
        m1, m2 = modes
        self.Pf = Patch_map_fno[patch_scale]
        self.overlap = overlap_ratio
        cout = dset_metadata.n_fields if dset_metadata else 4
        shape = dset_metadata.spatial_resolution if dset_metadata else (128, 384)
        self.H, self.W = shape[0], shape[1]
        C_ = embed_dim
        self.chans = [C_, C_ // 2, C_ // 4, C_ // 8, cout]
        self.dec_conv_1 = self._deconv(C_, C_ // 2, self.Pf[1])
        self.dec_spectral_1 = SpectralLayer(C_ // 2, C_ // 4, m1 // self.Pf[0], m2 // self.Pf[0])
        self.dec_conv_2 = self._deconv(C_ // 4, C_ // 8, self.Pf[0])
        self.dec_spectral_2 = SpectralLayer(C_ // 8, cout, m1, m2)
        self.patch_shape = (self.H // (self.Pf[0] * self.Pf[1]), self.W // (self.Pf[0] * self.Pf[1]))
        self._cache = _PackCache()

    def _deconv(self, ci, co, p):
        st, pd = S.stride_pad(p, self.overlap)
        return _Holder("deconv", nn.ConvTranspose2d(ci, co, (p, p), stride=(st, st), padding=(pd, pd)))

    def _packed(self, compute: int):
        dcs = [(self.dec_conv_1.deconv, self.Pf[1]), (self.dec_conv_2.deconv, self.Pf[0])]
        params = [p for c, _ in dcs for p in (c.weight, c.bias)]

        def build():   # channels-first outputs: both stages feed a spectral layer
            out = []
            for c, p in dcs:
                st_, pd_ = S.stride_pad(p, self.overlap)
                if st_ != p:
                    out.append(S.deconv_taps_pack(c.weight, c.bias, compute))
                else:
                    out.append(K.pack_weight(c.weight, c.bias, compute, L.W_DECONV_NCHW if pd_ == 0 else L.W_DECONV_NHWC,
                                             N=c.weight.shape[1] * p * p, K=c.weight.shape[0], P=p, C_other=c.weight.shape[1]))
            return out
        return self._cache.get(compute, params, build)

    def forward_tokens(self, src: torch.Tensor, n_img: int, compute: int, a_n0: int, a_s1: int, a_s0: int, a_off: int) -> torch.Tensor:
        pk = self._packed(compute)
        h, w = self.patch_shape
        p1, p0 = self.Pf[1], self.Pf[0]
        z = S.deconv_stage(src, n_img, h, w, p1, self.overlap, pk[0], self.chans[1], compute, L.ACT_GELU_ERF, True, torch.float32,
                           a_n0=a_n0, a_s1=a_s1, a_s0=a_s0, a_off=a_off)
        h, w = h * p1, w * p1
        # channels-last rows for the row GEMM behind it: written that way by the layer's last kernel (bf16 mode), by a layout copy otherwise
        rows = self.dec_spectral_1.run(z, L.ACT_GELU_ERF, compute, nhwc_out=True)
        z = S.deconv_stage(rows, n_img, h, w, p0, self.overlap, pk[1], self.chans[3], compute, L.ACT_GELU_ERF, True, torch.float32)
        return self.dec_spectral_2.run(z, L.ACT_NONE, compute)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _no_autograd(self)
        B, T, Hp, Wp, C_ = x.shape
        src = x.detach().float().contiguous()
        y = self.forward_tokens(src, B * T, resolve_compute(None), B * T * Hp * Wp, 0, C_, 0)
        return y.view(B, T, *y.shape[1:])


# ---- differentiable (training) forward of the spectral encoder / decoder ---------------------------------------------------------------
def _conv_stage_train(z_nchw: torch.Tensor, conv: nn.Conv2d, P: int, overlap: float, compute: int, out_dtype: torch.dtype, nhwc=None):
    """RealConv2d on the train path: channels-last im2col (its backward is the gather-sum col2im) + LinearFn; -> (rows, Cout), (h, w).
    nhwc = (n, C, H, W): z_nchw is already the channels-last image (n, H, W, C) (the CNN encoder's stage outputs are token rows)."""
    from .autograd import AvgPoolFn, Im2colFn, LinearFn
    n, C_, H, W = nhwc if nhwc is not None else z_nchw.shape
    st, pd = S.stride_pad(P, overlap)
    Ho, Wo = (H + 2 * pd - P) // st + 1, (W + 2 * pd - P) // st + 1
    if H % P or W % P:
        raise ValueError("To enforce (H//P, W//P), input H and W must be divisible by patch_size.")
    img = z_nchw.reshape(n, H, W, C_) if nhwc is not None else z_nchw.permute(0, 2, 3, 1)
    cols = Im2colFn.apply(img.contiguous(), n, C_, H, W, P, st, pd, K.act_torch_dtype(compute))
    w2d = conv.weight.permute(0, 2, 3, 1).reshape(conv.weight.shape[0], -1).contiguous()   # columns (kh, kw, c), a parameter-sized re-layout
    if (Ho, Wo) == (H // P, W // P):
        return LinearFn.apply(cols, w2d, conv.bias, None, compute, out_dtype), Ho, Wo
    # overlapping stage (stride < kernel): the conv's (Ho, Wo) grid is adaptive-average-pooled to (H // P, W // P) before the activation
    y = LinearFn.apply(cols, w2d, conv.bias, None, compute, K.act_torch_dtype(compute))
    return AvgPoolFn.apply(y, n, Ho, Wo, conv.weight.shape[0], H // P, W // P, out_dtype), H // P, W // P


def _deconv_stage_train(rows: torch.Tensor, dc: nn.ConvTranspose2d, n_img: int, h: int, w: int, P: int, overlap: float, compute: int,
                        nchw_out: bool = True, out_dtype: torch.dtype = torch.float32):
    """RealTransConv2d on the train path -> (n_img, Cout, h P, w P) (or channels-last (n_img, h P, w P, Cout)) pre-activation."""
    from .autograd import Col2imFn, CropResizeFn, DeconvFn, LinearFn
    st, pd = S.stride_pad(P, overlap)
    Cout = dc.weight.shape[1]
    adt = K.act_torch_dtype(compute)
    if st != P:
        # overlapping taps (enc_dec_cnn.py:128-184): tap matrix by one GEMM -> gather-sum + bias -> bilinear resize to (h P, w P)
        if dc.weight.shape[0] > S.KMAX:
            raise NotImplementedError("overlapping transposed conv with more than 512 input channels")
        w2 = dc.weight.permute(2, 3, 1, 0).reshape(-1, dc.weight.shape[0]).contiguous()   # rows (kh, kw, co), a parameter-sized re-layout
        taps = LinearFn.apply(rows, w2, None, None, compute, adt)
        full = Col2imFn.apply(taps, dc.bias, n_img, h, w, P, st, pd, Cout, adt)      # (n, Hf, Wf, Cout)
        return CropResizeFn.apply(full, n_img, Cout, full.shape[1], full.shape[2], (0, 0), h * P, w * P, nchw_out, out_dtype)
    if pd == 0:
        return DeconvFn.apply(rows, dc.weight, dc.bias, n_img, h, w, P, nchw_out, compute, out_dtype)
    full = DeconvFn.apply(rows, dc.weight, dc.bias, n_img, h, w, P, False, compute, adt)   # (n, hP, wP, Cout)
    return CropResizeFn.apply(full, n_img, Cout, h * P - 2 * pd, w * P - 2 * pd, (pd, pd), h * P, w * P, nchw_out, out_dtype)


def enc_fno_train(enc: enc_FNO, inp: torch.Tensor, compute: int) -> torch.Tensor:
    """enc_FNO.forward with an autograd graph: (B, T, D, H, W) -> tokens (B*T*Hp*Wp, C) fp32 (before FiLM / positional terms)."""
    from .autograd import ActFn, SpectralLayerFn
    B, T, D, H, W = inp.shape
    n = B * T
    l1, l2 = enc.enc_spectral_1, enc.enc_spectral_2
    z = ActFn.apply(SpectralLayerFn.apply(inp.reshape(n, D, H, W), l1.weight, l1.w0.weight, l1.w0.bias, l1.modes1, l1.modes2),
                    L.ACT_GELU_ERF, torch.float32)
    y, h, w = _conv_stage_train(z, enc.enc_conv_1.conv, enc.P[0], enc.overlap, compute, torch.float32)
    y = ActFn.apply(y, L.ACT_GELU_ERF, torch.float32)
    z = y.view(n, h, w, -1).permute(0, 3, 1, 2).contiguous()
    z = ActFn.apply(SpectralLayerFn.apply(z, l2.weight, l2.w0.weight, l2.w0.bias, l2.modes1, l2.modes2), L.ACT_GELU_ERF, torch.float32)
    y, h, w = _conv_stage_train(z, enc.enc_conv_2.conv, enc.P[1], enc.overlap, compute, torch.float32)
    return y


def dec_fno_train(dec: dec_FNO, rows: torch.Tensor, n_img: int, compute: int) -> torch.Tensor:
    """dec_FNO.forward with an autograd graph: last-slot tokens (n_img*Hp*Wp, C) -> (n_img, D, H, W) fp32."""
    from .autograd import ActFn, SpectralLayerFn
    h, w = dec.patch_shape
    p1, p0 = dec.Pf[1], dec.Pf[0]
    l1, l2 = dec.dec_spectral_1, dec.dec_spectral_2
    z = ActFn.apply(_deconv_stage_train(rows, dec.dec_conv_1.deconv, n_img, h, w, p1, dec.overlap, compute), L.ACT_GELU_ERF, torch.float32)
    h, w = h * p1, w * p1
    z = ActFn.apply(SpectralLayerFn.apply(z, l1.weight, l1.w0.weight, l1.w0.bias, l1.modes1, l1.modes2), L.ACT_GELU_ERF, torch.float32)
    r2 = z.permute(0, 2, 3, 1).contiguous().view(n_img * h * w, -1)
    z = ActFn.apply(_deconv_stage_train(r2, dec.dec_conv_2.deconv, n_img, h, w, p0, dec.overlap, compute), L.ACT_GELU_ERF, torch.float32)
    return SpectralLayerFn.apply(z, l2.weight, l2.w0.weight, l2.w0.bias, l2.modes1, l2.modes2)
