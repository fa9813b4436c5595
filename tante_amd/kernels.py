"""Thin torch-tensor front end of the C ABI (include/tante_hip.h).

PyTorch supplies device memory and the current stream only; every computation below is a call into
libtante_hip.so.  Nothing here falls back to ATen arithmetic: CPU tensors or a missing library raise.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import torch

from . import _lib as L

_DT = {torch.float32: L.F32, torch.bfloat16: L.BF16}
COMPUTE = {"fp32": L.F32, "float32": L.F32, "bf16": L.BF16, "bfloat16": L.BF16}


def _stream() -> int:
    """Raw handle of torch's current stream on the current device.  torch.cuda.current_stream() builds a Stream object per call
    (~8 us; 400 launches per train step each way made that 6 ms of a 27 ms host-side step)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _dev(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("tante_amd kernels need CUDA/HIP tensors (no CPU fallback)")
        if not t.is_contiguous():
            raise RuntimeError("tante_amd kernels need contiguous tensors")


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def act_torch_dtype(compute: int) -> torch.dtype:
    return torch.bfloat16 if compute == L.BF16 else torch.float32


class PackedWeight:
    """A weight in the GEMM's streaming layout (+ its fp32 bias, LayerNorm shift folded in)."""
    __slots__ = ("w", "bias", "N", "K", "compute", "geom")

    def __init__(self, w, bias, N, K, compute, geom):
        self.w, self.bias, self.N, self.K, self.compute, self.geom = w, bias, N, K, compute, geom


def pack_geom(N: int, K: int, compute: int) -> L.PackGeom:
    g = L.PackGeom()
    L.check(L.lib().tante_pack_geom(N, K, compute, C.byref(g)), "tante_pack_geom")
    return g


def pack_weight(w: torch.Tensor, bias: Optional[torch.Tensor], compute: int, layout: int = L.W_LINEAR,
                N: Optional[int] = None, K: Optional[int] = None, P: int = 0, C_other: int = 0,
                gamma: Optional[torch.Tensor] = None, beta: Optional[torch.Tensor] = None) -> PackedWeight:
    w = w.detach()
    _dev(w, bias, gamma, beta)
    if w.dtype != torch.float32:
        raise RuntimeError("master weights are fp32")
    if layout == L.W_LINEAR:
        N = w.shape[0] if N is None else N
        K = w[0].numel() if K is None else K
    assert N is not None and K is not None
    g = pack_geom(N, K, compute)
    wp = torch.empty(g.bytes, dtype=torch.uint8, device=w.device)
    bp = torch.empty(g.n_pad, dtype=torch.float32, device=w.device)
    L.check(L.lib().tante_pack_weight(_p(w), _p(None if bias is None else bias.detach()),
                                      _p(None if gamma is None else gamma.detach()),
                                      _p(None if beta is None else beta.detach()), layout, N, K, P, C_other, compute,
                                      _p(wp), _p(bp), _stream()), "tante_pack_weight")
    return PackedWeight(wp, bp, N, K, compute, g)


def _base(pw: PackedWeight, a: torch.Tensor, M: int, out: torch.Tensor, act: int) -> L.Gemm:
    _dev(a, out)
    g = L.Gemm()
    g.a, g.a_dtype, g.M, g.K = _p(a), _DT[a.dtype], M, pw.K
    g.w, g.bias, g.N, g.compute = _p(pw.w), _p(pw.bias), pw.N, pw.compute
    g.act, g.out, g.out_dtype = act, _p(out), _DT[out.dtype]
    g.a_n0 = 1
    return g


def _run(g: L.Gemm):
    L.check(L.lib().tante_gemm(C.byref(g), _stream()), "tante_gemm")


def linear(a: torch.Tensor, pw: PackedWeight = None, out: torch.Tensor = None, *, M: Optional[int] = None, act: int = L.ACT_NONE,
           ln: bool = False, ln_eps: float = 1e-5, residual: Optional[torch.Tensor] = None,
           a_n0: Optional[int] = None, a_s1: int = 0, a_s0: Optional[int] = None, a_off: int = 0,
           out_ld: Optional[int] = None, drop_p: float = 0.0, drop_seed: int = 0, dact: Optional[torch.Tensor] = None,
           dact_kind: int = L.ACT_NONE):
    """out[M, N] = act(norm?(rows(a)) @ W^T + b) (+ residual).  Row r of `a` starts at element
    (r // a_n0) * a_s1 + (r % a_n0) * a_s0 + a_off  (default: dense rows of length K).
    Training epilogues (see TanteGemm in include/tante_hip.h): drop_p > 0 drops the product before the residual is added;
    dact multiplies it by act'(dact) (an (M, N) pre-activation tensor)."""
    K = pw.K
    M = (a.numel() // K) if M is None else M
    g = _base(pw, a, M, out, act)
    g.a_mode = L.A_LINEAR
    g.a_n0 = M if a_n0 is None else a_n0
    g.a_s1, g.a_s0, g.a_off = a_s1, (K if a_s0 is None else a_s0), a_off
    g.ln, g.ln_eps = int(ln), ln_eps
    g.e_mode = L.E_LINEAR
    g.out_ld = pw.N if out_ld is None else out_ld
    if residual is not None:
        _dev(residual)
        if residual.dtype != torch.float32:
            raise RuntimeError("residual stream is fp32")
        g.residual, g.res_ld = _p(residual), pw.N
    if drop_p > 0.0:
        g.drop_p, g.drop_seed = float(drop_p), int(drop_seed)
    if dact is not None:
        _dev(dact)
        g.dact, g.dact_dtype, g.dact_kind = _p(dact), _DT[dact.dtype], dact_kind
    _run(g)
    return out


def linear_train_epilogue_ok(a: torch.Tensor, M: int, N: int, Kk: int, compute: int) -> bool:
    """Shapes for which tante_gemm offers the dropout / activation-gradient epilogues."""
    return compute == L.BF16 and a.dtype == torch.bfloat16 and M >= 4096 and Kk in (128, 256, 512) and N % 4 == 0


def patch_embed(a: torch.Tensor, pw: PackedWeight, out: torch.Tensor, *, n_img: int, Hin: int, Win: int, Cin: int, P: int,
                nchw: bool, act: int, film: Optional[tuple] = None, imgs_per_item: Optional[int] = None, item_stride: int = 0,
                elem_off: int = 0, pad: int = 0, nchw_out: bool = False):
    """Patch conv with kernel = stride = P as a GEMM over non-overlapping patches.  `a` is
    (n_img, Cin, Hin, Win) when nchw else (n_img, Hin, Win, Cin); out is channels-last
    (n_img, Hin/P, Win/P, N).  film = (film_a, film_b, s_emb, T, HW) selects the FiLM + positional epilogue.  pad: TanteGemm.a_pad (the
    'same' padding 1 of a P = 4 stage, channels-first bf16-compute only -- the library refuses what it cannot serve).  nchw_out: `out` is
    (n_img, N, Hin/P, Win/P) fp32, channels first (TANTE_E_DECONV_NCHW with Po = 1)."""
    M = n_img * (Hin // P) * (Win // P)
    if not a.is_cuda:
        raise RuntimeError("tante_amd kernels need CUDA/HIP tensors (no CPU fallback)")
    g = _base(pw, a if a.is_contiguous() else a.new_empty(0), M, out, act)
    g.a = a.data_ptr()        # may be a strided window view: addressing is explicit below
    g.a_mode = L.A_PATCH_NCHW if nchw else L.A_PATCH_NHWC
    g.Hin, g.Win, g.Cin, g.P, g.a_pad = Hin, Win, Cin, P, pad
    g.a_n0 = n_img if imgs_per_item is None else imgs_per_item
    g.a_s1, g.a_off = item_stride, elem_off
    g.out_ld = pw.N
    if nchw_out:
        if film is not None:
            raise ValueError("the FiLM epilogue writes channels-last rows")
        g.e_mode = L.E_DECONV_NCHW
        g.Hi, g.Wi, g.Po, g.Cout = Hin // P, Win // P, 1, pw.N
    elif film is None:
        g.e_mode = L.E_LINEAR
    else:
        fa, fb, se, T, HW = film
        _dev(fa, fb, se)
        g.e_mode = L.E_FILM
        g.film_a, g.film_b, g.s_emb, g.T, g.HW = _p(fa), _p(fb), _p(se), T, HW
    _run(g)
    return out


def deconv(a: torch.Tensor, pw: PackedWeight, out: torch.Tensor, *, n_img: int, Hi: int, Wi: int, P: int, Cout: int,
           nchw_out: bool, act: int, a_n0: Optional[int] = None, a_s1: int = 0, a_s0: Optional[int] = None, a_off: int = 0,
           dact: Optional[torch.Tensor] = None, dact_kind: int = L.ACT_NONE):
    """ConvTranspose2d with kernel = stride = P: one GEMM row per input pixel, N = Cout*P*P outputs
    scattered (pixel-shuffle) into (n_img, Hi*P, Wi*P, Cout) [channels-last] or (n_img, Cout, Hi*P, Wi*P)."""
    M = n_img * Hi * Wi
    g = _base(pw, a, M, out, act)
    g.a_mode = L.A_LINEAR
    g.a_n0 = M if a_n0 is None else a_n0
    g.a_s1, g.a_s0, g.a_off = a_s1, (pw.K if a_s0 is None else a_s0), a_off
    g.e_mode = L.E_DECONV_NCHW if nchw_out else L.E_DECONV_NHWC
    g.Hi, g.Wi, g.Po, g.Cout = Hi, Wi, P, Cout
    if dact is not None:      # channels-last output only: out = scatter(...) * act'(dact), dact laid out like out
        _dev(dact)
        g.dact, g.dact_dtype, g.dact_kind = _p(dact), _DT[dact.dtype], dact_kind
    _run(g)
    return out


def make_seq(letter: str, B: int, T: int, H: int, W: int) -> L.Seq:
    """Token regrouping of one axis letter over a (B,T,H,W) grid (attn_backbone.py:148-182)."""
    s = L.Seq()
    HW, THW = H * W, T * H * W
    if letter == "T":
        s.nseq, s.L, s.n_s0, s.S1, s.S0, s.n_l0, s.P1, s.P0 = B * HW, T, HW, THW, 1, T, 0, HW
    elif letter == "H":
        s.nseq, s.L, s.n_s0, s.S1, s.S0, s.n_l0, s.P1, s.P0 = B * T * W, H, W, HW, 1, H, 0, W
    elif letter == "W":
        s.nseq, s.L, s.n_s0, s.S1, s.S0, s.n_l0, s.P1, s.P0 = B * T * H, W, 1, W, 0, W, 0, 1
    elif letter == "L":
        s.nseq, s.L, s.n_s0, s.S1, s.S0, s.n_l0, s.P1, s.P0 = B * T, HW, 1, HW, 0, HW, 0, 1
    elif letter == "Y":   # (b w) (t h)
        s.nseq, s.L, s.n_s0, s.S1, s.S0, s.n_l0, s.P1, s.P0 = B * W, T * H, W, THW, 1, H, HW, W
    elif letter == "X":   # (b h) (t w)
        s.nseq, s.L, s.n_s0, s.S1, s.S0, s.n_l0, s.P1, s.P0 = B * H, T * W, H, THW, W, W, HW, 1
    elif letter == "A":
        s.nseq, s.L, s.n_s0, s.S1, s.S0, s.n_l0, s.P1, s.P0 = B, THW, 1, THW, 0, THW, 0, 1
    else:
        raise ValueError(f"invalid axis letter {letter!r}")
    return s


def dense_seq(nseq: int, Lq: int) -> L.Seq:
    s = L.Seq()
    s.nseq, s.L, s.n_s0, s.S1, s.S0, s.n_l0, s.P1, s.P0 = nseq, Lq, 1, Lq, 0, Lq, 0, 1
    return s


def attention(qkv: torch.Tensor, o: torch.Tensor, C_: int, n_head: int, seq: L.Seq, causal: bool):
    _dev(qkv, o)
    if qkv.dtype != o.dtype:
        raise RuntimeError("qkv and o must share a dtype")
    L.check(L.lib().tante_attention(_p(qkv), _p(o), _DT[qkv.dtype], C_, n_head, C.byref(seq), int(causal), _stream()),
            "tante_attention")
    return o


def axis_mlp(x: torch.Tensor, outer: int, n: int, inner: int, w1, b1, w2, b2, compute: int = L.F32):
    _dev(x, w1, b1, w2, b2)
    if x.dtype != torch.float32:
        raise RuntimeError("the residual stream is fp32")
    L.check(L.lib().tante_axis_mlp_c(_p(x), outer, n, inner, _p(w1.detach()), _p(b1.detach()), _p(w2.detach()), _p(b2.detach()), compute,
                                     _stream()), "tante_axis_mlp")
    return x


def axis_hw_supported(nH: int, nW: int, C_: int, compute: Optional[int] = None) -> bool:
    """Mirror of tante_axis_hw's shape rules.  Without `compute` the answer holds for both compute modes (the generic kernel's fp32
    staging is the larger one); with compute = BF16 whole-tile planes (nH, nW multiples of 16) may be larger (axis_hw_exact_kernel)."""
    def wbytes(m):           # axe_wbytes<m>() of pointwise.hip: both weights of an axis in fragment order + its biases
        return 2 * m * ((m + 1) // 2) * 1024 + 128 * m
    if (compute == L.BF16 and nH % 16 == 0 and nW % 16 == 0 and C_ % 16 == 0 and max(nH, nW) <= 64
            and nH * (nW * 16 + 32) * 4 + wbytes(nH // 16) + wbytes(nW // 16) <= 160 * 1024):
        return True
    n = max(nH, nW)
    rs = nW * 18
    while rs % 8 != 2:       # axis_row_stride() of pointwise.hip
        rs += 1
    return n <= 64 and C_ % 16 == 0 and nH * rs * 4 + 8 * n + 8 * n * n + 32 * n <= 160 * 1024


def axis_hw(x: torch.Tensor, BT: int, nH: int, nW: int, C_: int, vp, hp, compute: int):
    """Fused vertical + horizontal propagators; vp / hp = (w1, b1, w2, b2) of the H / W axis MLPs."""
    ws = [p.detach() for p in (*vp, *hp)]
    _dev(x, *ws)
    L.check(L.lib().tante_axis_hw(_p(x), BT, nH, nW, C_, *[_p(w) for w in ws], compute, _stream()), "tante_axis_hw")
    return x


def axis_hw_oop(xin: torch.Tensor, xout: torch.Tensor, BT: int, nH: int, nW: int, C_: int, vp, hp, compute: int):
    """axis_hw out of place (xin intact); only where axis_hw_train_supported says so."""
    ws = [p.detach() for p in (*vp, *hp)]
    _dev(xin, xout, *ws)
    L.check(L.lib().tante_axis_hw_oop(_p(xin), _p(xout), BT, nH, nW, C_, *[_p(w) for w in ws], compute, _stream()), "tante_axis_hw_oop")
    return xout


def axis_hw_train(x: torch.Tensor, BT: int, nH: int, nW: int, C_: int, vp, hp, compute: int):
    """Training forward of the H and W propagators in one launch, out of place: -> (y, x_mid) with x_mid the planes between the two
    (the W propagator's input); x is left intact.  Only where axis_hw_supported(..., BF16)'s whole-tile form applies."""
    ws = [p.detach() for p in (*vp, *hp)]
    _dev(x, *ws)
    y, xm = torch.empty_like(x), torch.empty_like(x)
    L.check(L.lib().tante_axis_hw_train(_p(x), _p(y), _p(xm), BT, nH, nW, C_, *[_p(w) for w in ws], compute, _stream()), "tante_axis_hw_train")
    return y, xm


def axis_hw_train_supported(nH: int, nW: int, C_: int, compute: int) -> bool:
    def wbytes(m):
        return 2 * m * ((m + 1) // 2) * 1024 + 128 * m
    return (compute == L.BF16 and nH % 16 == 0 and nW % 16 == 0 and C_ % 16 == 0 and max(nH, nW) <= 64
            and nH * (nW * 16 + 32) * 4 + wbytes(nH // 16) + wbytes(nW // 16) <= 160 * 1024)


def axis_hw_film(x: torch.Tensor, src: torch.Tensor, src_t_stride: int, src_b_stride: int, film: tuple, BT: int, nH: int, nW: int, C_: int,
                 vp, hp, compute: int):
    """axis_hw with the planes read from a frame-major pre-FiLM encoder cache; film = (a, b, s_emb, T, HW)."""
    fa, fb, se, T, HW = film
    ws = [p.detach() for p in (*vp, *hp)]
    _dev(x, src, fa, fb, se, *ws)
    if HW != nH * nW or src.dtype != torch.float32 or x.dtype != torch.float32:
        raise RuntimeError("axis_hw_film: fp32 planes of nH * nW tokens expected")
    L.check(L.lib().tante_axis_hw_film(_p(x), _p(src), src_t_stride, src_b_stride, _p(fa), _p(fb), _p(se), T, BT, nH, nW, C_,
                                       *[_p(w) for w in ws], compute, _stream()), "tante_axis_hw_film")
    return x


def film_table(t: torch.Tensor, film_params: Sequence[torch.Tensor], C_: int, add: Optional[torch.Tensor]):
    """film_params = (scale.0.weight, scale.0.bias, scale.2.weight, scale.2.bias, shift.0.weight, ...)."""
    _dev(t, add, *film_params)
    rows = t.numel()
    a = torch.empty(rows, C_, dtype=torch.float32, device=t.device)
    b = torch.empty(rows, C_, dtype=torch.float32, device=t.device)
    L.check(L.lib().tante_film_table(_p(t), rows, C_, *[_p(p.detach()) for p in film_params],
                                     _p(None if add is None else add.detach()), _p(a), _p(b), _stream()), "tante_film_table")
    return a, b


def film_apply(x: torch.Tensor, x_elem_off: int, x_bstride: int, y: torch.Tensor, rows: int, C_: int, rows_per: int, a, b):
    _dev(x, y, a, b)
    L.check(L.lib().tante_film_apply(x.data_ptr() + 4 * x_elem_off, x_bstride, _p(y), rows, C_, rows_per, _p(a), _p(b),
                                     _stream()), "tante_film_apply")
    return y


def taylor(last: torch.Tensor, last_elem_off: int, last_bstride: int, derivs: Sequence[torch.Tensor], dt: float, n_out: int,
           out: torch.Tensor, B: int, frame: int, out_elem_off: int = 0, out_bstride: Optional[int] = None):
    """`last` / `out` are base tensors addressed by (element offset, batch stride): they may be views into a rollout buffer."""
    if not (last.is_cuda and out.is_cuda):
        raise RuntimeError("tante_amd kernels need CUDA/HIP tensors (no CPU fallback)")
    _dev(*derivs)
    arr = (C.c_void_p * len(derivs))(*[d.data_ptr() for d in derivs])
    L.check(L.lib().tante_taylor(last.data_ptr() + 4 * last_elem_off, last_bstride, arr, len(derivs), float(dt), n_out,
                                 out.data_ptr() + 4 * out_elem_off, n_out * frame if out_bstride is None else out_bstride, B, frame,
                                 _stream()), "tante_taylor")
    return out


def rt_reduce(t: torch.Tensor, B: int, Lq: int, out_T: float, ep: float):
    _dev(t)
    rt = torch.empty(B, dtype=torch.float32, device=t.device)
    L.check(L.lib().tante_rt_reduce(_p(t), B, Lq, float(out_T), float(ep), _p(rt), _stream()), "tante_rt_reduce")
    return rt


def gather_last(z: torch.Tensor, n: int, E: int, out: torch.Tensor):
    _dev(z, out)
    L.check(L.lib().tante_gather_last(_p(z), n, E, _p(out), _stream()), "tante_gather_last")
    return out


# ---- fused TransformerBlock halves (bf16) ----------------------------------------------------------------
def attention_masked(qkv: torch.Tensor, o: torch.Tensor, C_: int, n_head: int, Bp: int, Lq: int, causal: bool, attn_mask: Optional[torch.Tensor],
                     key_padding_mask: Optional[torch.Tensor]):
    """Dense-sequence attention with additive fp32 masks: attn_mask (1, L, L) or (Bp * n_head, L, L), key_padding_mask (Bp, L)."""
    _dev(qkv, o, attn_mask, key_padding_mask)
    stride = 0 if attn_mask is None or attn_mask.shape[0] == 1 else Lq * Lq
    L.check(L.lib().tante_attention_masked(_p(qkv), _p(o), _DT[qkv.dtype], C_, n_head, Bp, Lq, int(causal), _p(attn_mask), stride,
                                           _p(key_padding_mask), _stream()), "tante_attention_masked")
    return o


def block_fused_supported(C_: int, n_head: int, hidden: int, Lq: int) -> bool:
    return bool(L.lib().tante_block_fused_supported(C_, n_head, hidden, Lq))


def pack_block(params: Sequence[torch.Tensor], C_: int, hidden: int) -> torch.Tensor:
    """params = (ln1.w, ln1.b, in_proj_w, in_proj_b, out_proj.w, out_proj.b, ln2.w, ln2.b, fc1.w, fc1.b, fc2.w, fc2.b)
    -> the block's weight stream (uint8 device buffer)."""
    ps = [p.detach() for p in params]
    _dev(*ps)
    st = torch.empty(L.lib().tante_block_stream_bytes(C_, hidden), dtype=torch.uint8, device=ps[0].device)
    L.check(L.lib().tante_pack_block(*[_p(p) for p in ps], C_, hidden, _p(st), _stream()), "tante_pack_block")
    return st


def block_fused(x: torch.Tensor, block_stream: torch.Tensor, C_: int, n_head: int, hidden: int, seq: L.Seq, causal: bool,
                eps: float, tprop: Optional[torch.Tensor] = None):
    """tprop (L = 4 only): the temporal propagator's 40 packed floats (w1, b1, w2, b2) -- applied to the rows inside the launch."""
    _dev(x, block_stream)
    if tprop is not None:
        _dev(tprop)
        L.check(L.lib().tante_block_fused_tprop(_p(x), _p(block_stream), C_, n_head, hidden, C.byref(seq), int(causal), eps, _p(tprop), _stream()),
                "tante_block_fused_tprop")
        return x
    L.check(L.lib().tante_block_fused(_p(x), _p(block_stream), C_, n_head, hidden, C.byref(seq), int(causal), eps, _stream()),
            "tante_block_fused")
    return x


def block_fused_train_supported(C_: int, n_head: int, hidden: int, Lq: int) -> bool:
    return C_ == 256 and n_head == 8 and hidden == 256 and 1 <= Lq <= 64


def pack_block_train(params: Sequence[torch.Tensor], C_: int, hidden: int) -> torch.Tensor:
    """params = (in_w folded, in_b folded, out_w, out_b, fc1_w folded, fc1_b folded, fc2_w, fc2_b) -> the training kernel's weight stream."""
    ps = [p.detach() for p in params]
    _dev(*ps)
    st = torch.empty(L.lib().tante_block_stream_bytes(C_, hidden), dtype=torch.uint8, device=ps[0].device)
    L.check(L.lib().tante_pack_block_train(*[_p(p) for p in ps], C_, hidden, _p(st), _stream()), "tante_pack_block_train")
    return st


def pack_block_tail_bwd(fc2_w: torch.Tensor, fc1_w_folded: torch.Tensor, out_w: torch.Tensor, C_: int, hidden: int) -> torch.Tensor:
    _dev(fc2_w, fc1_w_folded, out_w)
    st = torch.empty(L.lib().tante_block_tail_bwd_stream_bytes(C_, hidden), dtype=torch.uint8, device=fc2_w.device)
    L.check(L.lib().tante_pack_block_tail_bwd(_p(fc2_w.detach()), _p(fc1_w_folded.detach()), _p(out_w.detach()), C_, hidden, _p(st), _stream()),
            "tante_pack_block_tail_bwd")
    return st


def block_head_bwd(dqkv: torch.Tensor, xh1: torch.Tensor, st1: torch.Tensor, dx1: torch.Tensor, head_bwd_stream: torch.Tensor, C_: int) -> torch.Tensor:
    """-> dx (M, 256) fp32: the q | k | v data gradient + LayerNorm1 backward + the skip gradient in one launch (tante_block_head_bwd);
    head_bwd_stream = pack_block_tail_bwd on the three 256-row blocks of the folded in-projection weight."""
    _dev(dqkv, xh1, st1, dx1, head_bwd_stream)
    M = dx1.numel() // C_
    dx = torch.empty(M, C_, dtype=torch.float32, device=dx1.device)
    L.check(L.lib().tante_block_head_bwd(_p(dqkv), _p(xh1), _p(st1), _p(dx1), _p(head_bwd_stream), M, C_, _p(dx), _stream()), "tante_block_head_bwd")
    return dx


def block_tail_bwd(dout: torch.Tensor, hpre, xh2, st2, bwd_stream, C_: int, hidden: int, p_drop: float, seed_out: int, seed_mlp: int) -> dict:
    """-> {"dx1" fp32, "do", "dy2", "dhpre", "dy1" bf16}, each (M, 256) (tante_block_tail_bwd)."""
    _dev(dout, hpre, xh2, st2, bwd_stream)
    M = dout.numel() // C_
    dev = dout.device
    t = {"dx1": torch.empty(M, C_, dtype=torch.float32, device=dev)}
    for k in ("do", "dy2", "dhpre", "dy1"):
        t[k] = torch.empty(M, C_, dtype=torch.bfloat16, device=dev)
    L.check(L.lib().tante_block_tail_bwd(_p(dout), _p(hpre), _p(xh2), _p(st2), _p(bwd_stream), M, C_, hidden, float(p_drop), int(seed_out),
                                         int(seed_mlp), _p(t["dx1"]), _p(t["dy2"]), _p(t["dhpre"]), _p(t["dy1"]), _p(t["do"]), _stream()),
            "tante_block_tail_bwd")
    return t


def block_bwd_fused_supported(C_: int, n_head: int, hidden: int, Lq: int, causal: bool) -> bool:
    return bool(L.lib().tante_block_bwd_fused_supported(C_, n_head, hidden, Lq, int(causal)))


def block_bwd_fused(dout: torch.Tensor, xh1, st1, hpre, xh2, st2, tail_bwd_stream, block_stream, head_bwd_stream, C_: int, n_head: int,
                    hidden: int, seq: L.Seq, causal: bool, p_drop: float, seeds) -> dict:
    """The whole backward of a block in ONE launch (tante_block_bwd_fused) -> {"dx" fp32 (M, 256); "dy2", "dhpre", "dy1" (M, 256) and
    "dqkv" (M, 768) bf16: the row operands of the four weight gradients}.  seeds = (attention, out-proj, mlp) as the forward drew them."""
    _dev(dout, xh1, st1, hpre, xh2, st2, tail_bwd_stream, block_stream, head_bwd_stream)
    M = dout.numel() // C_
    dev = dout.device
    t = {"dx": torch.empty(M, C_, dtype=torch.float32, device=dev), "dqkv": torch.empty(M, 3 * C_, dtype=torch.bfloat16, device=dev)}
    for k in ("dy2", "dhpre", "dy1"):
        t[k] = torch.empty(M, C_, dtype=torch.bfloat16, device=dev)
    L.check(L.lib().tante_block_bwd_fused(_p(dout), _p(xh1), _p(st1), _p(hpre), _p(xh2), _p(st2), _p(tail_bwd_stream), _p(block_stream),
                                          _p(head_bwd_stream), C_, n_head, hidden, C.byref(seq), int(causal), float(p_drop), int(seeds[0]),
                                          int(seeds[1]), int(seeds[2]), _p(t["dx"]), _p(t["dy2"]), _p(t["dhpre"]), _p(t["dy1"]), _p(t["dqkv"]),
                                          _stream()), "tante_block_bwd_fused")
    return t


def block_fused_train(x: torch.Tensor, block_stream: torch.Tensor, C_: int, n_head: int, hidden: int, seq: L.Seq, causal: bool, eps: float,
                      p_drop: float, seeds, need_x1: bool = True, need_qkv: bool = True) -> dict:
    """Training forward of a whole block in one launch (tante_block_fused_train): -> the block output and every saved tensor of the
    unfused operators (see include/tante_hip.h).  x (tokens, 256) fp32 is left untouched.  need_x1=False skips the fp32 residual after
    the attention half (25 MB per launch at cfg3): the fused tail backward works from LayerNorm2's image and statistics.  need_qkv=False
    skips the packed projection (37.8 MB): the one-launch backward (block_bwd_fused) recomputes q | k | v from LayerNorm1's image."""
    _dev(x, block_stream)
    M = x.numel() // C_
    dev = x.device
    bf = lambda n: torch.empty(M, n, dtype=torch.bfloat16, device=dev)      # noqa: E731
    f32 = lambda n: torch.empty(M, n, dtype=torch.float32, device=dev)      # noqa: E731
    t = {"out": f32(C_), "xh1": bf(C_), "qkv": bf(3 * C_) if need_qkv else None, "o": bf(C_), "xh2": bf(C_), "hpre": bf(hidden), "act": bf(hidden),
         "st1": f32(2), "x1": f32(C_) if need_x1 else None, "st2": f32(2)}
    tr = L.BlockTrain(*[(t[k].data_ptr() if t[k] is not None else None) for k in ("out", "xh1", "qkv", "o", "xh2", "hpre", "act", "st1", "x1", "st2")],
                      float(p_drop), int(seeds[0]), int(seeds[1]), int(seeds[2]))
    L.check(L.lib().tante_block_fused_train(_p(x), _p(block_stream), C_, n_head, hidden, C.byref(seq), int(causal), eps, C.byref(tr),
                                            _stream()), "tante_block_fused_train")
    return t


# ---- fused derivative head (bf16) ---------------------------------------------------------------------------------------
def head_fused_supported(C_: int, D: int) -> bool:
    return bool(L.lib().tante_head_fused_supported(C_, D))


def pack_head(params: Sequence[torch.Tensor], C_: int, D: int) -> torch.Tensor:
    """params = (deconv1.w, deconv1.b, deconv2.w, deconv2.b, deconv3.w, deconv3.b) -> the head's weight stream."""
    ps = [p.detach() for p in params]
    _dev(*ps)
    st = torch.empty(L.lib().tante_head_stream_bytes(C_), dtype=torch.uint8, device=ps[0].device)
    L.check(L.lib().tante_pack_head(*[_p(p) for p in ps], C_, D, _p(st), _stream()), "tante_pack_head")
    return st


def head_fused(x: torch.Tensor, a_n0: int, a_s1: int, a_s0: int, a_off: int, n_img: int, Hp: int, Wp: int, C_: int, D: int,
               head_stream: torch.Tensor, out: torch.Tensor, out_bstride: int, coefs: Sequence[float], last: Optional[torch.Tensor],
               last_elem_off: int = 0, last_bstride: int = 0):
    """out_i (+)= coefs[i] * head(x rows); `out` / `last` are base tensors addressed by data_ptr (+ offset) and a batch stride."""
    _dev(x, head_stream)
    if not out.is_cuda:
        raise RuntimeError("tante_amd kernels need CUDA/HIP tensors (no CPU fallback)")
    n_out = len(coefs)
    arr = (C.c_float * n_out)(*[float(c) for c in coefs])
    lp = None if last is None else last.data_ptr() + 4 * last_elem_off
    L.check(L.lib().tante_head_fused(_p(x), a_n0, a_s1, a_s0, a_off, n_img, Hp, Wp, C_, D, _p(head_stream), out.data_ptr(), out_bstride,
                                     n_out, arr, lp, last_bstride, _stream()), "tante_head_fused")
    return out


def head_fused_multi(rows: Sequence[torch.Tensor], a_n0: int, a_s1: int, a_s0: int, a_off: int, n_img: int, Hp: int, Wp: int, C_: int, D: int,
                     head_streams: Sequence[torch.Tensor], coefs: Sequence[float], out: torch.Tensor, out_bstride: int, last: torch.Tensor,
                     last_elem_off: int, last_bstride: int, streams: bool = False):
    """out = last + sum_k coefs[k] * head_k(rows[k]) in ONE launch (one prediction frame).  rows[:-1]: dense (n_img * Hp * Wp, C_) fp32 copies of
    the last-slot token rows after each earlier backbone; rows[-1]: the stream itself, addressed by (a_n0, a_s1, a_s0, a_off)."""
    n = len(rows)
    _dev(*rows, *head_streams)
    if not (out.is_cuda and last.is_cuda):      # `out` / `last` are base pointers + strides (views of a rollout buffer): not required dense
        raise RuntimeError("tante_amd kernels need CUDA/HIP tensors (no CPU fallback)")
    rp = (C.c_void_p * n)(*[r.data_ptr() for r in rows])
    sp = (C.c_void_p * n)(*[h.data_ptr() for h in head_streams])
    cf = (C.c_float * n)(*[float(c) for c in coefs])
    fn = L.lib().tante_head_fused_multi_streams if streams else L.lib().tante_head_fused_multi      # streams: every rows[k] is a whole stream
    L.check(fn(n, rp, sp, cf, a_n0, a_s1, a_s0, a_off, n_img, Hp, Wp, C_, D, out.data_ptr(), out_bstride,
               last.data_ptr() + 4 * last_elem_off, last_bstride, _stream()), "tante_head_fused_multi")
    return out


def head_enc_supported(C_: int, D: int, Hp: int, Wp: int) -> bool:
    """The one-launch tail of a rollout call (heads + Taylor sum + re-encoding of the predicted frame): C = 256, whole 16-token tiles per image."""
    return bool(L.lib().tante_head_enc_supported(C_, D)) and (Hp * Wp) % 16 == 0


def pack_head_enc(params: Sequence[torch.Tensor], C_: int, D: int) -> torch.Tensor:
    """params = (conv1.w, conv1.b, conv2.w, conv2.b, conv3.w, conv3.b) of enc_CNN -> the encoder stream of tante_head_enc_fused."""
    ps = [p.detach() for p in params]
    _dev(*ps)
    st = torch.empty(L.lib().tante_head_enc_stream_bytes(C_), dtype=torch.uint8, device=ps[0].device)
    L.check(L.lib().tante_pack_head_enc(*[_p(p) for p in ps], C_, D, _p(st), _stream()), "tante_pack_head_enc")
    return st


_HE_WS = {}


def head_enc_workspace(rows: int, device: torch.device) -> torch.Tensor:
    """The zeroed workspace of tante_head_enc_fused for `rows` tokens, one per (device, stream, size): the kernel leaves its arrival
    counters at zero, so it is cleared once."""
    # (the group size is part of the key: the arrival counters sit behind the fragments, whose size depends on it -- flipping
    # TANTE_HEAD_WAVES between calls would otherwise move the counters onto bytes that held fragments: ADVICE round 4)
    key = (device.index, _stream(), rows, L.get_option("TANTE_HEAD_WAVES", 0))
    ws = _HE_WS.get(key)
    if ws is None:
        ws = torch.zeros(L.lib().tante_head_enc_ws_bytes(rows), dtype=torch.uint8, device=device)
        _HE_WS[key] = ws
    return ws


def head_enc_fused(rows: Sequence[torch.Tensor], a_n0: int, a_s1: int, a_s0: int, a_off: int, n_img: int, Hp: int, Wp: int, C_: int, D: int,
                   head_streams: Sequence[torch.Tensor], coefs: Sequence[float], out: torch.Tensor, out_bstride: int, last: torch.Tensor,
                   last_elem_off: int, last_bstride: int, enc_stream: Optional[torch.Tensor] = None, z: Optional[torch.Tensor] = None):
    """out = last + sum_k coefs[k] * head_k(rows[k]) and (enc_stream given) z = enc_CNN(out) before FiLM, in ONE launch.
    rows[k]: the whole residual stream after backbone k, all addressed by (a_n0, a_s1, a_s0, a_off)."""
    n = len(rows)
    _dev(*rows, *head_streams, enc_stream, z)
    if not (out.is_cuda and last.is_cuda):
        raise RuntimeError("tante_amd kernels need CUDA/HIP tensors (no CPU fallback)")
    if a_n0 % 16 or (Hp * Wp) % 16:
        raise RuntimeError("head_enc_fused: whole 16-token tiles per image expected")
    rp = (C.c_void_p * n)(*[r.data_ptr() for r in rows])
    sp = (C.c_void_p * n)(*[h.data_ptr() for h in head_streams])
    cf = (C.c_float * n)(*[float(c) for c in coefs])
    ws, wsb = None, 0
    if enc_stream is not None:
        if z is None or z.dtype != torch.float32 or z.numel() != n_img * Hp * Wp * C_:
            raise RuntimeError("head_enc_fused: z must be a contiguous fp32 (n_img * Hp * Wp, C) tensor")
        wst = head_enc_workspace(n_img * Hp * Wp, out.device)
        ws, wsb = wst.data_ptr(), wst.numel()
    L.check(L.lib().tante_head_enc_fused(n, rp, sp, cf, a_n0, a_s1, a_s0, a_off, n_img, Hp, Wp, C_, D, out.data_ptr(), out_bstride,
                                         last.data_ptr() + 4 * last_elem_off, last_bstride, _p(enc_stream), _p(z), ws, wsb, _stream()),
            "tante_head_enc_fused")
    return out


def enc23_supported(C_: int) -> bool:
    return bool(L.lib().tante_enc23_supported(C_))


def pack_enc23(params: Sequence[torch.Tensor], C_: int) -> torch.Tensor:
    """params = (conv2.w, conv2.b, conv3.w, conv3.b) -> the fused stage-2+3 weight stream."""
    ps = [p.detach() for p in params]
    _dev(*ps)
    st = torch.empty(L.lib().tante_enc23_stream_bytes(C_), dtype=torch.uint8, device=ps[0].device)
    L.check(L.lib().tante_pack_enc23(*[_p(p) for p in ps], C_, _p(st), _stream()), "tante_pack_enc23")
    return st


def enc23_fused(h1: torch.Tensor, n_img: int, Hp: int, Wp: int, C_: int, enc_stream: torch.Tensor, film: tuple, out: torch.Tensor):
    fa, fb, se, T, HW = film
    _dev(h1, enc_stream, fa, fb, se, out)
    if h1.dtype != torch.bfloat16 or HW != Hp * Wp:
        raise RuntimeError("enc23_fused: bf16 stage-1 image and a (Hp*Wp, C) spatial embedding expected")
    L.check(L.lib().tante_enc23_fused(_p(h1), n_img, Hp, Wp, C_, _p(enc_stream), _p(fa), _p(fb), _p(se), T, _p(out), _stream()),
            "tante_enc23_fused")
    return out


def enc23_frames(h1: torch.Tensor, n_img: int, frames: int, Hp: int, Wp: int, C_: int, enc_stream: torch.Tensor, out: torch.Tensor):
    """Stages 2 + 3 without FiLM into the frame-major pre-FiLM cache: image (b, f) -> out[f, b]."""
    _dev(h1, enc_stream, out)
    if h1.dtype != torch.bfloat16 or out.dtype != torch.float32:
        raise RuntimeError("enc23_frames: bf16 stage-1 image, fp32 cache expected")
    L.check(L.lib().tante_enc23_frames(_p(h1), n_img, frames, Hp, Wp, C_, _p(enc_stream), _p(out), _stream()), "tante_enc23_frames")
    return out


# ---- general conv stages, spectral layer, CViT operators (operators.hip) ---------------------------------------------------------
def im2col(x: torch.Tensor, nchw: bool, n_img: int, C_: int, H: int, W: int, kh: int, kw: int, sh: int, sw: int, ph: int, pw: int,
           korder: int, out_dtype: torch.dtype) -> torch.Tensor:
    """-> (n_img * Ho * Wo, C * kh * kw) patch matrix (zero padded); korder 0 = (c, kh, kw), 1 = (kh, kw, c)."""
    _dev(x)
    Ho, Wo = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
    cols = torch.empty(n_img * Ho * Wo, C_ * kh * kw, dtype=out_dtype, device=x.device)
    L.check(L.lib().tante_im2col(_p(x), _DT[x.dtype], int(nchw), n_img, C_, H, W, kh, kw, sh, sw, ph, pw, korder, _p(cols), _DT[out_dtype],
                                 _stream()), "tante_im2col")
    return cols


def avgpool_nhwc(x: torch.Tensor, n_img: int, H: int, W: int, C_: int, Ht: int, Wt: int, act: int, out_dtype: torch.dtype) -> torch.Tensor:
    _dev(x)
    y = torch.empty(n_img * Ht * Wt, C_, dtype=out_dtype, device=x.device)
    L.check(L.lib().tante_avgpool_nhwc(_p(x), _DT[x.dtype], n_img, H, W, C_, Ht, Wt, act, _p(y), _DT[out_dtype], _stream()), "tante_avgpool_nhwc")
    return y


def col2im_nhwc(cols: torch.Tensor, n_img: int, Hi: int, Wi: int, P: int, stride: int, pad: int, Cout: int, bias: Optional[torch.Tensor],
                out_dtype: torch.dtype) -> torch.Tensor:
    _dev(cols, bias)
    Hf, Wf = (Hi - 1) * stride - 2 * pad + P, (Wi - 1) * stride - 2 * pad + P
    out = torch.empty(n_img, Hf, Wf, Cout, dtype=out_dtype, device=cols.device)
    L.check(L.lib().tante_col2im_nhwc(_p(cols), _DT[cols.dtype], n_img, Hi, Wi, P, stride, pad, Cout, _p(bias), _p(out), _DT[out_dtype],
                                      _stream()), "tante_col2im_nhwc")
    return out


def resize_bilinear(x: torch.Tensor, n_img: int, C_: int, Hi: int, Wi: int, crop: Tuple[int, int], in_strides, Ho: int, Wo: int,
                    out: torch.Tensor, out_strides, act: int):
    """Bilinear resize (align_corners=False) of the (Hi, Wi) window at `crop` of x into out; strides = (sn, sc, sh, sw) in elements."""
    _dev(x, out)
    L.check(L.lib().tante_resize_bilinear(_p(x), _DT[x.dtype], n_img, C_, Hi, Wi, crop[0], crop[1], *in_strides, Ho, Wo, *out_strides, act,
                                          _p(out), _DT[out.dtype], _stream()), "tante_resize_bilinear")
    return out


def layernorm_affine(x: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor], eps: float,
                     out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    _dev(x, gamma, beta)
    x = x.contiguous()
    Cc = x.shape[-1]
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    L.check(L.lib().tante_layernorm_affine(_p(x), _DT[x.dtype], x.numel() // Cc, Cc, eps, _p(gamma), _p(beta), _p(y), _DT[out_dtype], _stream()),
            "tante_layernorm_affine")
    return y


def spectral_layer(x: torch.Tensor, w_re: torch.Tensor, w_im: torch.Tensor, modes1: int, modes2: int, w0: torch.Tensor, b0: torch.Tensor,
                   act: int, compute: int = L.F32, bf16_out: bool = False, nhwc_out: bool = False) -> torch.Tensor:
    """x (n, Cin, H, W) fp32 -> act(SpectralLayer(x)) (n, Cout, H, W) fp32.  compute = L.BF16 (a bf16 model): the inverse row transform may
    use split-operand products on the bf16 matrix pipe (~1e-5 relative to the fp32 result).  bf16_out (bf16 mode, a consumer that rounds to
    bf16 anyway): the image is written as bf16 where the shape has that form (tante_spectral_layer_bf16out), fp32 otherwise.
    x may be a batch-strided view of contiguous images (a frame of every item of a rollout buffer): read in place where
    tante_spectral_layer_x serves the shape, copied otherwise.  nhwc_out: -> channels-last rows (n * H * W, Cout) fp32 (a row GEMM follows):
    written that way by the layer's last kernel where it can, by a layout copy otherwise."""
    n, Cin, H, W = x.shape
    Cout = w_re.shape[1]
    dense = x.is_contiguous()
    img_ok = x.is_cuda and x.dtype == torch.float32 and x[0].is_contiguous() and x.stride(0) >= Cin * H * W and x.stride(0) % 4 == 0
    if (compute == L.BF16 and img_ok and (not dense or nhwc_out)
            and L.lib().tante_spectral_layer_x_supported(n, Cin, Cout, H, W, modes1, modes2, int(not dense), 2 if nhwc_out else int(bool(bf16_out)))
            and x.data_ptr() % 16 == 0 and w0.data_ptr() % 16 == 0 and (b0 is None or b0.data_ptr() % 16 == 0)):
        _dev(w_re, w_im, w0, b0)
        mode = 2 if nhwc_out else int(bool(bf16_out))
        nbytes = L.lib().tante_spectral_workspace_bytes(n, Cin, Cout, H, W)
        work = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        out = (torch.empty(n * H * W, Cout, dtype=torch.float32, device=x.device) if nhwc_out
               else torch.empty(n, Cout, H, W, dtype=torch.bfloat16 if bf16_out else torch.float32, device=x.device))
        L.check(L.lib().tante_spectral_layer_x(x.data_ptr(), x.stride(0), n, Cin, H, W, _p(w_re), _p(w_im), w_re.shape[2], w_re.shape[3], modes1, modes2,
                                               _p(w0), _p(b0), Cout, act, _p(out), mode, _p(work), nbytes, _stream()), "tante_spectral_layer_x")
        return out
    if not dense:
        x = x.contiguous()
    if nhwc_out:
        y = spectral_layer(x, w_re, w_im, modes1, modes2, w0, b0, act, compute)
        return y.permute(0, 2, 3, 1).contiguous().view(n * H * W, Cout)
    _dev(x, w_re, w_im, w0, b0)
    nbytes = L.lib().tante_spectral_workspace_bytes(n, Cin, Cout, H, W)
    work = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    if bf16_out and compute == L.BF16 and L.lib().tante_spectral_bf16out_supported(n, Cin, Cout, H, W, modes1, modes2):
        out = torch.empty(n, Cout, H, W, dtype=torch.bfloat16, device=x.device)
        L.check(L.lib().tante_spectral_layer_bf16out(_p(x), n, Cin, H, W, _p(w_re), _p(w_im), w_re.shape[2], w_re.shape[3], modes1, modes2, _p(w0),
                                                     _p(b0), Cout, act, _p(out), _p(work), nbytes, _stream()), "tante_spectral_layer_bf16out")
        return out
    out = torch.empty(n, Cout, H, W, dtype=torch.float32, device=x.device)
    L.check(L.lib().tante_spectral_layer_c(_p(x), n, Cin, H, W, _p(w_re), _p(w_im), w_re.shape[2], w_re.shape[3], modes1, modes2, _p(w0), _p(b0),
                                           Cout, act, _p(out), _p(work), nbytes, compute, _stream()), "tante_spectral_layer")
    return out


def cross_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, o: torch.Tensor, n_batch: int, n_head: int, D: int, Lq: int, Lk: int,
                    ldq: int, ldkv: int, ldo: int, shared_q: bool = False):
    """k and v may be views into one packed (…, 2C) buffer: rows are addressed by data_ptr + strides.  shared_q: q holds Lq rows that every
    sample attends with (the decoder's coordinate queries)."""
    if not (q.is_cuda and k.is_cuda and v.is_cuda and o.is_cuda):
        raise RuntimeError("tante_amd kernels need CUDA/HIP tensors (no CPU fallback)")
    if not (q.dtype == k.dtype == v.dtype == o.dtype):
        raise RuntimeError("q, k, v, o must share a dtype")
    L.check(L.lib().tante_cross_attention_q(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), _DT[q.dtype], n_batch, n_head, D, Lq, Lk, ldq,
                                            ldkv, ldo, 0 if shared_q else Lq, _stream()), "tante_cross_attention")
    return o


# ---- CViT blocks at width 512 in one launch (cvit_fused.hip) ------------------------------------------------------------------------
def pack_chain_matrix(w: torch.Tensor, b: torch.Tensor, gamma: torch.Tensor = None, beta: torch.Tensor = None):
    """(512, 512) Linear weight (+ a LayerNorm affine folded into its input side) -> (bf16 operand-fragment stream, fp32 bias)
    in the order tante_cvit_chain512 reads: [wave 8][k-step 16][row tile 4][kk 4][l15 16][8 values]."""
    w = w.detach().float()
    b = b.detach().float()
    if gamma is not None:
        b = b + w @ beta.detach().float()
        w = w * gamma.detach().float()[None, :]
    f = w.to(torch.bfloat16).view(8, 4, 16, 16, 4, 8).permute(0, 3, 1, 4, 2, 5).contiguous()      # [w][j][l15][ks][kk][e] -> [w][ks][j][kk][l15][e]
    return f.view(-1), b.contiguous()


def pack_chain_output(w: torch.Tensor, b: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor):
    """(out_dim <= 16, 512) output layer with the preceding LayerNorm's affine folded in -> 16 KiB of fragments in the k order of the
    accumulator tiles that feed it (lane (row, kk) of fragment (wave, pair): columns 64 wave + 32 pair + 4 kk + 0..3, then the same + 16)."""
    w = w.detach().float()
    b = b.detach().float() + w @ beta.detach().float()
    w = w * gamma.detach().float()[None, :]
    n = w.shape[0]
    wp = torch.zeros(16, 512, dtype=torch.float32, device=w.device)
    wp[:n] = w
    bp = torch.zeros(16, dtype=torch.float32, device=w.device)
    bp[:n] = b
    f = wp.to(torch.bfloat16).view(16, 8, 2, 2, 4, 4).permute(1, 2, 4, 0, 3, 5).contiguous()        # [row][w][p][h][kk][e] -> [w][p][kk][row][h][e]
    return f.view(-1), bp


def cvit_chain512(a: torch.Tensor, resid: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, eps_ln2: float, M: int, out: torch.Tensor,
                  tail=None) -> torch.Tensor:
    """tante_cvit_chain512: a (M, 512) bf16, resid (period, 512) fp32 (row t % period), w / bias the packed matrices.  tail = None: mode 0,
    out (M, 512); tail = (g2, b2, eps_norm2, eps_mlp, wout, bout, out_dim): mode 1, out (M, out_dim)."""
    for t in (a, resid, w, bias, out):
        _dev(t)
    if a.dtype != torch.bfloat16 or resid.dtype != torch.float32 or out.dtype != torch.float32 or not (a.is_contiguous() and resid.is_contiguous() and out.is_contiguous()):
        raise RuntimeError("cvit_chain512: a bf16, resid / out fp32, all contiguous")
    if tail is None:
        L.check(L.lib().tante_cvit_chain512(_p(a), _p(resid), resid.shape[0], _p(w), _p(bias), None, None, eps_ln2, 0.0, 0.0, None, None, 0, M, 0,
                                            _p(out), _stream()), "tante_cvit_chain512")
    else:
        g2, b2, eps_n2, eps_mlp, wout, bout, out_dim = tail
        L.check(L.lib().tante_cvit_chain512(_p(a), _p(resid), resid.shape[0], _p(w), _p(bias), _p(g2), _p(b2), eps_ln2, eps_n2, eps_mlp, _p(wout),
                                            _p(bout), out_dim, M, 1, _p(out), _stream()), "tante_cvit_chain512")
    return out


def grid_embed(coords: torch.Tensor, grid: torch.Tensor, latents: torch.Tensor, eps: float) -> torch.Tensor:
    _dev(coords, grid, latents)
    N, G, LD = coords.shape[0], grid.shape[0], latents.shape[1]
    out = torch.empty(N, LD, dtype=torch.float32, device=coords.device)
    L.check(L.lib().tante_grid_embed(_p(coords), _p(grid), _p(latents), N, G, LD, eps, _p(out), _stream()), "tante_grid_embed")
    return out


def fourier_embed(coords: torch.Tensor, kernel: torch.Tensor) -> torch.Tensor:
    _dev(coords, kernel)
    N, E = coords.shape[0], 2 * kernel.shape[1]
    out = torch.empty(N, E, dtype=torch.float32, device=coords.device)
    L.check(L.lib().tante_fourier_embed(_p(coords), _p(kernel), N, E, _p(out), _stream()), "tante_fourier_embed")
    return out


# ---- the training form of the rollout tail (csrc/tail_chain.hip): decoder stages -> Taylor sum -> re-encoding, forward and backward --------
def tail_supported(C_: int, D: int, Hp: int, Wp: int) -> bool:
    return bool(L.lib().tante_tail_supported(C_, D, Hp, Wp))


def pack_tail(params: Sequence[torch.Tensor], D: int, decoder: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """(forward stream, backward stream) of one decoder (conv-transpose weights (Cin, Cout, 2, 2)) or of the encoder (conv weights
    (Cout, Cin, 2, 2)); params = (w1, b1, w2, b2, w3, b3) in stage order."""
    _dev(*params)
    dev = params[0].device
    fwd = torch.empty(L.lib().tante_tail_stream_bytes(0 if decoder else 2), dtype=torch.uint8, device=dev)
    bwd = torch.empty(L.lib().tante_tail_stream_bytes(1 if decoder else 3), dtype=torch.uint8, device=dev)
    fn = L.lib().tante_tail_pack_dec if decoder else L.lib().tante_tail_pack_enc
    L.check(fn(*[_p(q.detach()) for q in params], D, fwd.data_ptr(), bwd.data_ptr(), _stream()), "tante_tail_pack")
    return fwd, bwd
