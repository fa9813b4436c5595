"""General encoder / decoder stages on the HIP kernels: RealConv2d / RealTransConv2d for every patch size, 'same' padding and
(for the conv) overlap ratio (reference models/enc_dec_cnn.py:49-184), and dense layers with K > 512.

The shipped configs (patch_scale 8: three 2x2 stages, no padding) keep their dedicated patch-GEMM / fused-head kernels; these
helpers are the general route: im2col gather -> tante_gemm -> (adaptive average pool | crop + bilinear resize) -> activation.
No torch arithmetic: torch only allocates and reshapes (weight re-layouts are cached until a parameter changes)."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from . import _lib as L
from . import kernels as K
from . import options as _O

KMAX = 512   # register-stationary K limit of tante_gemm


def stride_pad(P: int, overlap: float) -> Tuple[int, int]:
    """enc_dec_cnn.py:66-81 / 128-143: stride = max(1, round(P (1 - overlap))), 'same' padding = (P - 1) // 2."""
    return max(1, int(round(P * (1.0 - overlap)))), (P - 1) // 2


def pack_linear_chunks(weight2d: torch.Tensor, bias: Optional[torch.Tensor], compute: int) -> List[K.PackedWeight]:
    """(N, K) weight -> packed K-chunks of <= 512 columns (bias rides the first chunk)."""
    N, Kt = weight2d.shape
    out = []
    for k0 in range(0, Kt, KMAX):
        wk = weight2d[:, k0:k0 + KMAX].contiguous()
        out.append(K.pack_weight(wk, bias if k0 == 0 else None, compute, L.W_LINEAR, N=N, K=wk.shape[1]))
    return out


def linear_chunks(a: torch.Tensor, chunks: List[K.PackedWeight], out_dtype: torch.dtype, act: int = L.ACT_NONE) -> torch.Tensor:
    """out = act(a @ W^T + b) for a dense (M, K) activation matrix of any K: the K-chunks accumulate through the GEMM's fp32
    residual operand (chunk i reads `out` as residual and writes it back in place)."""
    M, Kt = a.shape
    if len(chunks) == 1:
        out = torch.empty(M, chunks[0].N, dtype=out_dtype, device=a.device)
        return K.linear(a, chunks[0], out, M=M, act=act)
    acc = torch.empty(M, chunks[0].N, dtype=torch.float32, device=a.device)
    k0 = 0
    for i, pw in enumerate(chunks):
        K.linear(a, pw, acc, M=M, a_n0=M, a_s0=Kt, a_off=k0, residual=acc if i else None)
        k0 += pw.K
    if act != L.ACT_NONE or out_dtype != torch.float32:
        out = torch.empty(M, chunks[0].N, dtype=out_dtype, device=a.device)
        L.check(L.lib().tante_act_fwd(acc.data_ptr(), L.F32, out.data_ptr(), K._DT[out_dtype], acc.numel(), act, K._stream()), "tante_act_fwd")
        return out
    return acc


# False: non-overlapping channels-first conv stages always run as tante_im2col + the dense GEMM (the A/B and parity reference of the fused route)
PATCH_GEMM = _O.register("TANTE_CONV_PATCH_GEMM", True, __name__, "PATCH_GEMM")


def conv_weight_2d(w: torch.Tensor, korder: int) -> torch.Tensor:
    """Conv weight (Cout, Cin, kh, kw[, ...]) -> (Cout, K) in the im2col column order."""
    w = w.detach()
    if korder == 0:
        return w.reshape(w.shape[0], -1)
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)


def conv_stage(x: torch.Tensor, nchw: bool, n_img: int, Cin: int, H: int, W: int, P: int, overlap: float, chunks: List[K.PackedWeight],
               compute: int, act: int, out_dtype: torch.dtype, nchw_out: bool = False,
               out: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, int, int]:
    """RealConv2d.forward (enc_dec_cnn.py:97-110): conv(kernel P, stride / padding from overlap) -> adaptive_avg_pool2d to
    (H // P, W // P) -> act.  Returns the channels-last (n_img * Ht * Wt, Cout) matrix and (Ht, Wt).  x may be a bf16 image (a producer
    that rounded for the bf16 patch gather already).  nchw_out: the (n_img, Cout, Ht, Wt) fp32 image instead (a spectral layer follows):
    written by the patch GEMM's own epilogue where that route applies (round 5; the generic kernel's pixel-shuffle epilogue had been
    measured slower than a layout copy -- 96 against 40 + 20 us at cfg5), by a layout copy of the rows otherwise.
    out: where the rows may be written (used when the patch-gather GEMM runs and the shape / dtype match; the caller checks identity)."""
    if H % P or W % P:
        raise ValueError("To enforce (H//P, W//P), input H and W must be divisible by patch_size.")
    s, p = stride_pad(P, overlap)
    Hc, Wc = (H + 2 * p - P) // s + 1, (W + 2 * p - P) // s + 1
    Ht, Wt = H // P, W // P
    adt = K.act_torch_dtype(compute)
    if (PATCH_GEMM and nchw and s == P and (Hc, Wc) == (Ht, Wt) and ((P == 2 and p == 0) or (P == 4 and p == 1)) and len(chunks) == 1
            and compute == L.BF16 and chunks[0].K == Cin * P * P and chunks[0].K in (256, 512) and chunks[0].N % 4 == 0 and n_img * Ht * Wt >= 4096
            and x.is_contiguous() and x.dtype in (torch.bfloat16, torch.float32) and x.data_ptr() % 16 == 0
            and act in (L.ACT_NONE, L.ACT_GELU_ERF)):
        # the gather IS the GEMM's fragment load (gemm.hip patch_frag): no (M, K) matrix in HBM; bit-identical to the two launches below
        if nchw_out and out_dtype == torch.float32:
            y = torch.empty(n_img, chunks[0].N, Ht, Wt, dtype=torch.float32, device=x.device)
            K.patch_embed(x, chunks[0], y, n_img=n_img, Hin=H, Win=W, Cin=Cin, P=P, nchw=True, act=act, pad=p, nchw_out=True)
            return y, Ht, Wt
        take = (out is not None and not nchw_out and out.dtype == out_dtype and out.is_contiguous() and tuple(out.shape) == (n_img * Ht * Wt, chunks[0].N))
        y = out if take else torch.empty(n_img * Ht * Wt, chunks[0].N, dtype=out_dtype, device=x.device)
        K.patch_embed(x, chunks[0], y, n_img=n_img, Hin=H, Win=W, Cin=Cin, P=P, nchw=True, act=act, pad=p)
        return (_rows_to_nchw(y, n_img, Ht, Wt) if nchw_out else y), Ht, Wt
    cols = K.im2col(x, nchw, n_img, Cin, H, W, P, P, s, s, p, p, 0 if nchw else 1, adt)
    same = (Hc, Wc) == (Ht, Wt)
    y = linear_chunks(cols, chunks, out_dtype if same else adt, act if same else L.ACT_NONE)
    if not same:
        y = K.avgpool_nhwc(y, n_img, Hc, Wc, chunks[0].N, Ht, Wt, act, out_dtype)
    return (_rows_to_nchw(y, n_img, Ht, Wt) if nchw_out else y), Ht, Wt


def _rows_to_nchw(y2d: torch.Tensor, n: int, h: int, w: int) -> torch.Tensor:
    """channels-last rows (n*h*w, C) -> (n, C, h, w) fp32 (layout change only)."""
    return y2d.view(n, h, w, -1).permute(0, 3, 1, 2).float().contiguous()


def deconv_taps_pack(weight: torch.Tensor, bias: Optional[torch.Tensor], compute: int):
    """ConvTranspose2d weight (Cin, Cout, P, P) -> (packed (P*P*Cout, Cin) tap GEMM weight without bias, bias) for overlapping stages."""
    w = weight.detach()
    w2 = w.permute(2, 3, 1, 0).reshape(-1, w.shape[0]).contiguous()
    if w2.shape[1] > KMAX:
        raise NotImplementedError("overlapping transposed conv with more than 512 input channels")
    return K.pack_weight(w2, None, compute, L.W_LINEAR, N=w2.shape[0], K=w2.shape[1]), (None if bias is None else bias.detach())


def deconv_stage(x: torch.Tensor, n_img: int, Hi: int, Wi: int, P: int, overlap: float, pw: K.PackedWeight, Cout: int, compute: int,
                 act: int, nchw_out: bool, out_dtype: torch.dtype, **rows) -> torch.Tensor:
    """RealTransConv2d.forward (enc_dec_cnn.py:164-184): ConvTranspose2d(kernel P, stride, padding) -> bilinear resize to
    (Hi * P, Wi * P) when the size differs -> act.  x: channels-last rows (n_img * Hi * Wi, Cin) (or gathered by **rows)."""
    s, p = stride_pad(P, overlap)
    Ho, Wo = Hi * P, Wi * P
    if s != P:   # overlapping taps: tap matrix by one GEMM (pw = taps packing, see deconv_taps_chunks) -> gather-sum -> resize
        M = n_img * Hi * Wi
        if rows:
            a_n0, a_s1, a_s0, a_off = rows["a_n0"], rows["a_s1"], rows["a_s0"], rows["a_off"]
        else:
            a_n0, a_s1, a_s0, a_off = M, 0, x.shape[-1], 0
        taps, bias = pw
        cols = torch.empty(M, taps.N, dtype=K.act_torch_dtype(compute), device=x.device)
        K.linear(x, taps, cols, M=M, a_n0=a_n0, a_s1=a_s1, a_s0=a_s0, a_off=a_off)
        full = K.col2im_nhwc(cols, n_img, Hi, Wi, P, s, p, Cout, bias, K.act_torch_dtype(compute))
        Hf, Wf = full.shape[1], full.shape[2]
        shape = (n_img, Cout, Ho, Wo) if nchw_out else (n_img, Ho, Wo, Cout)
        out = torch.empty(shape, dtype=out_dtype, device=x.device)
        ostr = (Cout * Ho * Wo, Ho * Wo, Wo, 1) if nchw_out else (Ho * Wo * Cout, 1, Wo * Cout, Cout)
        return K.resize_bilinear(full, n_img, Cout, Hf, Wf, (0, 0), (Hf * Wf * Cout, 1, Wf * Cout, Cout), Ho, Wo, out, ostr, act)
    shape = (n_img, Cout, Ho, Wo) if nchw_out else (n_img, Ho, Wo, Cout)
    if p == 0:
        out = torch.empty(shape, dtype=out_dtype, device=x.device)
        return K.deconv(x, pw, out, n_img=n_img, Hi=Hi, Wi=Wi, P=P, Cout=Cout, nchw_out=nchw_out, act=act, **rows)
    # padding p crops p pixels off every side of the unpadded result; the reference then resizes (Ho - 2p, Wo - 2p) back to (Ho, Wo)
    full = torch.empty(n_img, Ho, Wo, Cout, dtype=K.act_torch_dtype(compute), device=x.device)
    K.deconv(x, pw, full, n_img=n_img, Hi=Hi, Wi=Wi, P=P, Cout=Cout, nchw_out=False, act=L.ACT_NONE, **rows)
    out = torch.empty(shape, dtype=out_dtype, device=x.device)
    ostr = (Cout * Ho * Wo, Ho * Wo, Wo, 1) if nchw_out else (Ho * Wo * Cout, 1, Wo * Cout, Cout)
    return K.resize_bilinear(full, n_img, Cout, Ho - 2 * p, Wo - 2 * p, (p, p), (Ho * Wo * Cout, 1, Wo * Cout, Cout), Ho, Wo, out, ostr, act)
