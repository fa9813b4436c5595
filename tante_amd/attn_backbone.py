"""Axis-factorised transformer backbone on MI355X -- host side.

Mirrors the reference operator surface (models/attn_backbone.py): ``TransformerBlock(embed_dim, n_head,
mlp_ratio, dropout).forward(x, key_padding_mask, attn_mask, causal)`` and ``Attn_Backbone(tensor_shape,
attn_axes, expanded_channel, n_head, mlp_ratio, dropout).forward(x)``, with identical parameter names,
shapes and default initialisation (the torch.nn modules below are PARAMETER CONTAINERS only -- their
forward() is never called; all arithmetic goes through libtante_hip.so).

MI355X-first differences from the reference's execution (results identical to rounding):
  * the residual stream stays a flat (tokens, C) fp32 array for the whole backbone; the per-letter
    ``rearrange`` copies (attn_backbone.py:150-182) do not exist -- only the attention kernel sees the
    regrouping, as index arithmetic (kernels.make_seq);
  * LayerNorm is fused into the projection that consumes it (gamma/beta folded into the packed weight);
  * bias, GELU and the residual add are GEMM epilogues.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch
import torch.nn as nn

# The temporal propagator inside the first (T-letter) block's launch (tante_block_fused_tprop).  Round 2's form exchanged the four time
# steps between lane groups (32 ds_bpermute per 16 bytes): the block launch grew by the 12 us the propagator kernel took -- no gain, off.
# Round 3: a lane owns 4 channels of all four time steps and the two 4 x 4 contractions run as fp32 v_mfma_f32_4x4x1 (every lane its own
# column, no cross-lane traffic) in the kernel's LayerNorm1 phase, where the matrix pipe is idle: ON by default; TANTE_FUSE_TPROP=0
# brings the separate launch back (A/B timing, tests).
from . import _lib as L
from . import kernels as K
from . import options as _O

FUSE_TPROP = _O.register("TANTE_FUSE_TPROP", True, __name__, "FUSE_TPROP")


def resolve_compute(module_default: Optional[str] = None) -> int:
    """fp32 unless the caller runs under torch.autocast(bfloat16) (the reference's AMP switch,
    trainer/trainer.py:183) or the module was pinned with set_compute()."""
    if module_default is not None:
        return K.COMPUTE[module_default]
    if torch.is_autocast_enabled("cuda"):
        dt = torch.get_autocast_dtype("cuda")
        if dt == torch.bfloat16:
            return L.BF16
        if dt == torch.float16:
            # The reference Trainer's default amp_type is "float16" with a GradScaler (trainer/trainer.py:86-104).  There is one 16-bit
            # MFMA path here and its operand format is bf16: the same operand width and matrix rate as v_mfma_f32_16x16x32_f16, 8 instead
            # of 11 mantissa bits, and fp32's exponent range -- nothing overflows, so the scaler's loss scaling is accepted and is
            # arithmetically a power-of-two no-op (train.train_step(scaler=...), FlatAdamW.param_groups).  Said once, not silently.
            global _FP16_NOTED
            if not _FP16_NOTED:
                _FP16_NOTED = True
                import warnings
                warnings.warn("tante_amd: torch.autocast(float16) runs the bf16 MFMA kernels (16-bit operands rounded to bfloat16, fp32 "
                              "accumulation); results meet the same 1e-2 bar, GradScaler is supported but never has to skip a step")
            return L.BF16
        raise NotImplementedError(f"autocast dtype {dt} has no MFMA path here (use bfloat16, float16 or fp32)")
    return L.F32


_FP16_NOTED = False


def _no_autograd(module: nn.Module):
    """Sub-modules that have no differentiable path of their own yet (the spectral enc/dec and CViT's internals: their models train through
    TANTE.forward / CViT.forward / FNO.forward)."""
    if torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters()):
        raise NotImplementedError(
            "this sub-module has no stand-alone differentiable path: call it under torch.no_grad(), or train through the model's forward()")


def _gpu_only(x: torch.Tensor):
    if not x.is_cuda:
        raise RuntimeError("tante_amd modules run on the GPU only (no CPU fallback); move the module and its input to cuda")


def _wants_grad(module: nn.Module, *inputs) -> bool:
    """The reference's modules are differentiable anywhere (attn_backbone.py:59-83, 134-191; enc_dec_cnn.py:217-229, 263-277): under
    autograd a sub-module's forward() takes the differentiable path (train_forward.py: HIP forward that saves what the HIP backward
    kernels read), exactly like TANTE.forward; without it the inference kernels run."""
    return torch.is_grad_enabled() and (any(p.requires_grad for p in module.parameters())
                                        or any(isinstance(t, torch.Tensor) and t.requires_grad for t in inputs))


_WEIGHT_EPOCH = [0]


def bump_weight_epoch():
    """Called by optimisers that update parameters through raw pointers (optim.FlatAdamW): invalidates every packed copy."""
    _WEIGHT_EPOCH[0] += 1


class _PackCache:
    """Packed (bf16 / fp32, swizzled, LayerNorm-folded) copies of a module's weights, rebuilt when any
    source parameter changed (optimizer step, load_state_dict, .to())."""

    def __init__(self):
        self._store = {}

    def get(self, compute: int, params, build):
        key = (_WEIGHT_EPOCH[0],) + tuple((p.data_ptr(), p._version) for p in params)
        hit = self._store.get(compute)
        if hit is None or hit[0] != key:
            hit = (key, build())
            self._store[compute] = hit
        return hit[1]


class TransformerBlock(nn.Module):
    """Pre-LN block: x += MHA(LN1 x); x += W2 gelu_tanh(W1 LN2 x)   (attn_backbone.py:38-83)."""

    def __init__(self, embed_dim: int, n_head: int, mlp_ratio: float = 4.0, dropout: float = 0.1):
        super().__init__()
        self.embed_dim, self.n_head, self.p_drop = embed_dim, n_head, dropout
        self.ln1 = nn.LayerNorm(embed_dim)
        self.attn = nn.MultiheadAttention(embed_dim, n_head, batch_first=True, dropout=dropout, bias=True)
        self.ln2 = nn.LayerNorm(embed_dim)
        hidden = int(embed_dim * mlp_ratio)
        self.hidden = hidden
        self.mlp = nn.Sequential(nn.Linear(embed_dim, hidden, bias=True), nn.GELU(approximate="tanh"),
                                 nn.Linear(hidden, embed_dim, bias=True))
        self.drop = nn.Dropout(dropout)
        self._cache = _PackCache()
        self.compute: Optional[str] = None
        self.fused = True        # bf16: use the fused block kernels whenever the shape allows

    def _packed(self, compute: int):
        a, m = self.attn, self.mlp
        params = [self.ln1.weight, self.ln1.bias, a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias,
                  self.ln2.weight, self.ln2.bias, m[0].weight, m[0].bias, m[2].weight, m[2].bias]

        def build():
            return dict(
                qkv=K.pack_weight(a.in_proj_weight, a.in_proj_bias, compute, gamma=self.ln1.weight, beta=self.ln1.bias),
                out=K.pack_weight(a.out_proj.weight, a.out_proj.bias, compute),
                fc1=K.pack_weight(m[0].weight, m[0].bias, compute, gamma=self.ln2.weight, beta=self.ln2.bias),
                fc2=K.pack_weight(m[2].weight, m[2].bias, compute))
        return self._cache.get(compute, params, build)

    def _packed_fused(self):
        params = self.__dict__.get("_fused_params")      # nn.Module attribute walks cost ~20 us per call, 72 calls per rollout
        if params is None:
            a, m = self.attn, self.mlp
            params = [self.ln1.weight, self.ln1.bias, a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias,
                      self.ln2.weight, self.ln2.bias, m[0].weight, m[0].bias, m[2].weight, m[2].bias]
            self.__dict__["_fused_params"] = params
        return self._cache.get(-1, params, lambda: K.pack_block(params, self.embed_dim, self.hidden))

    def _apply(self, fn, *a, **kw):      # .to() / .half() may replace Parameter objects: drop the cached list
        self.__dict__.pop("_fused_params", None)
        return super()._apply(fn, *a, **kw)

    def takes_tprop(self, seq_L: int, compute: int) -> bool:
        """Can this block apply the temporal propagator itself (tante_block_fused_tprop: the feature-sliced kernel at L = 4)?"""
        return (FUSE_TPROP and compute == L.BF16 and self.fused and self.ln1.eps == self.ln2.eps and seq_L == 4 and self.embed_dim == 256
                and self.n_head == 8 and self.hidden == 256 and not (self.training and self.p_drop > 0.0))

    def forward_tokens(self, x: torch.Tensor, seq: L.Seq, causal: bool, compute: int, tprop: Optional[torch.Tensor] = None) -> torch.Tensor:
        """In place on the flat fp32 residual stream x (tokens, C); `seq` says which tokens attend to which.  tprop: the temporal
        propagator's packed weights, applied to the rows inside the fused launch (only when takes_tprop says so)."""
        if self.training and self.p_drop > 0.0:
            # train() mode without autograd (a validation pass that forgot eval(), MC dropout): nn.Dropout and the attention's dropout are
            # active whatever the grad mode (attn_backbone.py:47,56,81-82) -- the training forward's kernels, their saved tensors dropped
            if tprop is not None:
                raise RuntimeError("TransformerBlock.forward_tokens: tprop is an inference-path fusion (takes_tprop is False under dropout)")
            from .train_forward import block_train
            with torch.no_grad():
                y = block_train(self, x.view(-1, self.embed_dim), seq, causal, compute)
            x.view(-1, self.embed_dim).copy_(y)
            return x
        C_ = self.embed_dim
        n_tok = x.numel() // C_
        if (compute == L.BF16 and self.fused and self.ln1.eps == self.ln2.eps
                and K.block_fused_supported(C_, self.n_head, self.hidden, seq.L)):
            # one launch per block: the residual rows are read once and written once
            return K.block_fused(x, self._packed_fused(), C_, self.n_head, self.hidden, seq, causal, self.ln1.eps, tprop)
        if tprop is not None:
            raise RuntimeError("TransformerBlock.forward_tokens: tprop needs the fused block kernel (takes_tprop)")
        pk = self._packed(compute)
        adt = K.act_torch_dtype(compute)
        qkv = torch.empty(n_tok, 3 * C_, dtype=adt, device=x.device)
        K.linear(x, pk["qkv"], qkv, M=n_tok, ln=True, ln_eps=self.ln1.eps)
        o = torch.empty(n_tok, C_, dtype=adt, device=x.device)
        K.attention(qkv, o, C_, self.n_head, seq, causal)
        K.linear(o, pk["out"], x, M=n_tok, residual=x)
        h = qkv.view(-1)[: n_tok * self.hidden].view(n_tok, self.hidden) if self.hidden <= 3 * C_ else \
            torch.empty(n_tok, self.hidden, dtype=adt, device=x.device)
        K.linear(x, pk["fc1"], h, M=n_tok, ln=True, ln_eps=self.ln2.eps, act=L.ACT_GELU_TANH)
        K.linear(h, pk["fc2"], x, M=n_tok, residual=x)
        return x

    def forward(self, x: torch.Tensor, key_padding_mask=None, attn_mask=None, causal: bool = False) -> torch.Tensor:
        """attn_backbone.py:59-83.  `causal` is what the TANTE path uses (l.148-189); `attn_mask` (bool: True = blocked, or additive float;
        (L, L) or (B * n_head, L, L)) and `key_padding_mask` ((B, L) bool: True = ignored, or additive float) follow
        nn.MultiheadAttention's semantics and run the masked attention kernels of the unfused path (forward, and the recomputing backward
        under autograd; no attention dropout there).  `causal` next to an attn_mask follows what the reference's call computes, which
        depends on the mode (see below)."""
        _gpu_only(x)
        Bp, Lq, C_ = x.shape
        compute = resolve_compute(self.compute)
        if causal and attn_mask is not None and key_padding_mask is None and (self.training or _wants_grad(self, x)):
            # What the reference computes here is PLAIN causal attention: it hands nn.MultiheadAttention `attn_mask.bool() | causal_mask`
            # together with is_causal=True, need_weights=False (attn_backbone.py:70-80), and torch's multi_head_attention_forward -- the path
            # taken in train() mode or when anything requires grad -- treats is_causal without a key_padding_mask as "the mask IS the causal
            # mask" and drops the tensor.  Only eval() without autograd (torch's fused fast path) applies the combined mask.  Fixture
            # g17_block_masked_grad_ambool_causal pins the reference's actual output and gradients.
            attn_mask = None
        if key_padding_mask is not None or attn_mask is not None:
            if _wants_grad(self, x):
                from .train_forward import block_train
                masks = self._masks(x.device, Bp, Lq, causal, key_padding_mask, attn_mask)
                y = block_train(self, x.to(torch.float32).reshape(Bp * Lq, C_).contiguous(), K.dense_seq(Bp, Lq), causal, compute, masks)
                return y.view(Bp, Lq, C_)
            y = x.detach().to(torch.float32).contiguous().clone()
            self._forward_masked(y.view(Bp * Lq, C_), Bp, Lq, causal, compute, key_padding_mask, attn_mask)
            return y
        if _wants_grad(self, x):
            from .train_forward import block_train
            y = block_train(self, x.to(torch.float32).reshape(Bp * Lq, C_).contiguous(), K.dense_seq(Bp, Lq), causal, compute)
            return y.view(Bp, Lq, C_)
        y = x.detach().to(torch.float32).contiguous().clone()
        self.forward_tokens(y.view(Bp * Lq, C_), K.dense_seq(Bp, Lq), causal, compute)
        return y

    def _masks(self, dev, Bp: int, Lq: int, causal: bool, key_padding_mask, attn_mask):
        """nn.MultiheadAttention's masks as ONE additive fp32 tensor pair (attn_mask (1, L, L) | (Bp n_head, L, L) | None, kpm (Bp, L) | None)."""
        nh = self.n_head
        ninf = float("-inf")
        am = None
        if attn_mask is not None:
            am = attn_mask.to(dev)
            if causal and am.dtype != torch.bool:
                # attn_backbone.py:70-72: under `causal` the reference combines `attn_mask.bool() | causal_mask`, i.e. ANY non-zero entry of a
                # float mask blocks (it is not added); without `causal` a float mask stays additive, as nn.MultiheadAttention takes it
                am = am != 0
            if am.dtype == torch.bool:
                am = torch.zeros(am.shape, dtype=torch.float32, device=dev).masked_fill_(am, ninf)
            am = am.to(torch.float32).contiguous()
            if am.dim() == 2 and tuple(am.shape) == (Lq, Lq):
                am = am.view(1, Lq, Lq)
            elif not (am.dim() == 3 and tuple(am.shape) == (Bp * nh, Lq, Lq)):
                raise ValueError(f"attn_mask must be (L, L) or (B * n_head, L, L) = ({Lq}, {Lq}) / ({Bp * nh}, {Lq}, {Lq}), got {tuple(attn_mask.shape)}")
        kp = None
        if key_padding_mask is not None:
            kp = key_padding_mask.to(dev)
            if tuple(kp.shape) != (Bp, Lq):
                raise ValueError(f"key_padding_mask must be (B, L) = ({Bp}, {Lq}), got {tuple(key_padding_mask.shape)}")
            if kp.dtype == torch.bool:
                kp = torch.zeros(kp.shape, dtype=torch.float32, device=dev).masked_fill_(kp, ninf)
            kp = kp.to(torch.float32).contiguous()
        return am, kp

    def _forward_masked(self, x: torch.Tensor, Bp: int, Lq: int, causal: bool, compute: int, key_padding_mask, attn_mask) -> torch.Tensor:
        """The unfused block with the masked attention kernel (tante_attention_masked)."""
        C_ = self.embed_dim
        dev = x.device
        am, kp = self._masks(dev, Bp, Lq, causal, key_padding_mask, attn_mask)
        pk = self._packed(compute)
        adt = K.act_torch_dtype(compute)
        n_tok = Bp * Lq
        qkv = torch.empty(n_tok, 3 * C_, dtype=adt, device=dev)
        K.linear(x, pk["qkv"], qkv, M=n_tok, ln=True, ln_eps=self.ln1.eps)
        o = torch.empty(n_tok, C_, dtype=adt, device=dev)
        K.attention_masked(qkv, o, C_, self.n_head, Bp, Lq, causal, am, kp)
        K.linear(o, pk["out"], x, M=n_tok, residual=x)
        h = torch.empty(n_tok, self.hidden, dtype=adt, device=dev)
        K.linear(x, pk["fc1"], h, M=n_tok, ln=True, ln_eps=self.ln2.eps, act=L.ACT_GELU_TANH)
        K.linear(h, pk["fc2"], x, M=n_tok, residual=x)
        return x


class Attn_Backbone(nn.Module):
    """Three residual axis propagators, then one TransformerBlock per axis letter (attn_backbone.py:88-191)."""

    def __init__(self, tensor_shape=(10, 8, 4, 256), attn_axes: str = "L TT TT TT L", expanded_channel: int = 128,
                 n_head: int = 8, mlp_ratio: float = 1.0, dropout: float = 0.0):
        super().__init__()
        self.T, self.H, self.W, self.C = tensor_shape
        self.L = self.H * self.W
        self.expanded_channel = expanded_channel
        if attn_axes == "":
            raise ValueError("Invalid block: empty segment.")
        self.attn_axes = attn_axes
        self.n_head = n_head
        self.blocks = nn.ModuleList()

        def prop(n):
            return nn.Sequential(nn.Linear(n, n), nn.GELU(), nn.Linear(n, n))
        self.vertical_propagator = prop(self.H)
        self.horizontal_propagator = prop(self.W)
        self.temporal_propagator = prop(self.T)
        self.channel_blocks = nn.ModuleList()
        for axis in self.attn_axes:
            if axis in "LTHWAXY":
                embed_dim = self.C
            elif axis == "C":
                embed_dim = self.expanded_channel
                self.channel_blocks.append(nn.Sequential(nn.Linear(1, embed_dim // 4), nn.GELU(),
                                                         nn.Linear(embed_dim // 4, embed_dim)))
            else:
                raise ValueError(f"invalid axis letter {axis!r}")
            self.blocks.append(TransformerBlock(embed_dim=embed_dim, n_head=n_head, mlp_ratio=mlp_ratio, dropout=dropout))
        self._cache = _PackCache()
        self._tp_cache = _PackCache()
        self.compute: Optional[str] = None

    def _packed_tprop(self) -> torch.Tensor:
        """w1 (4 x 4), b1, w2 (4 x 4), b2 of the temporal propagator as 40 contiguous floats (tante_block_fused_tprop)."""
        tp = self.temporal_propagator
        params = [tp[0].weight, tp[0].bias, tp[2].weight, tp[2].bias]
        return self._tp_cache.get(0, params, lambda: torch.cat([p.detach().float().reshape(-1) for p in params]).contiguous())

    def _packed_channel(self, compute: int):
        params = [p for cb in self.channel_blocks for p in cb.parameters()]

        def build():
            return [(K.pack_weight(cb[0].weight, cb[0].bias, compute), K.pack_weight(cb[2].weight, cb[2].bias, compute))
                    for cb in self.channel_blocks]
        return self._cache.get(compute, params, build)

    def takes_x_in(self, compute: int) -> bool:
        """forward_tokens(x, ..., x_in=other) can read its input from another buffer (the first propagator launch runs out of place)."""
        return K.axis_hw_train_supported(self.H, self.W, self.C, compute)

    def forward_tokens(self, x: torch.Tensor, B: int, compute: int, film_src: Optional[tuple] = None, x_in: Optional[torch.Tensor] = None,
                       film_frames: Optional[tuple] = None) -> torch.Tensor:
        """In place on x = (B,T,H,W,C) fp32 contiguous.  film_src = (z, t_stride, b_stride, film): x is not read but produced from the
        frame-major pre-FiLM encoder cache z while the first propagator kernel loads its planes (TANTE.forward(enc_cache=...)).
        x_in (takes_x_in): the input stream, left intact -- x is only written.
        film_frames = (TanteFrames, the frame tensors, a, b, s_emb) (planes too large for the whole-plane kernel: the spectral path at
        512 x 512): the same, by the vertical propagator's launch (tante_axis_mlp_film); the caller asked _supported."""
        T, H, W, C_ = self.T, self.H, self.W, self.C
        vp, hp, tp = self.vertical_propagator, self.horizontal_propagator, self.temporal_propagator
        if film_frames is not None:
            import ctypes as C
            fr, _keep, fa, fb, se = film_frames
            L.check(L.lib().tante_axis_mlp_film(x.data_ptr(), C.byref(fr), fa.data_ptr(), fb.data_ptr(), se.data_ptr(), B, T, H, W * C_, C_,
                                                vp[0].weight.data_ptr(), vp[0].bias.data_ptr(), vp[2].weight.data_ptr(), vp[2].bias.data_ptr(),
                                                K._stream()), "tante_axis_mlp_film")
            K.axis_mlp(x, B * T * H, W, C_, hp[0].weight, hp[0].bias, hp[2].weight, hp[2].bias, compute)      # l.142-143
        elif x_in is not None:
            K.axis_hw_oop(x_in, x, B * T, H, W, C_, (vp[0].weight, vp[0].bias, vp[2].weight, vp[2].bias),
                          (hp[0].weight, hp[0].bias, hp[2].weight, hp[2].bias), compute)
        elif film_src is not None:
            z, ts, bs, film = film_src
            K.axis_hw_film(x, z, ts, bs, film, B * T, H, W, C_, (vp[0].weight, vp[0].bias, vp[2].weight, vp[2].bias),
                           (hp[0].weight, hp[0].bias, hp[2].weight, hp[2].bias), compute)
        elif K.axis_hw_supported(H, W, C_, compute):      # both axes in one pass over x, contractions on MFMA     l.140-143
            K.axis_hw(x, B * T, H, W, C_, (vp[0].weight, vp[0].bias, vp[2].weight, vp[2].bias),
                      (hp[0].weight, hp[0].bias, hp[2].weight, hp[2].bias), compute)
        else:
            K.axis_mlp(x, B * T, H, W * C_, vp[0].weight, vp[0].bias, vp[2].weight, vp[2].bias, compute)      # l.140-141
            K.axis_mlp(x, B * T * H, W, C_, hp[0].weight, hp[0].bias, hp[2].weight, hp[2].bias, compute)      # l.142-143
        # l.144-145: the temporal propagator -- inside the first block's launch when that block is a T letter on the fused kernel (the four
        # time steps of a sequence are the four lane groups of its row loads), a launch of its own otherwise
        tprop = None
        if (T == 4 and len(self.attn_axes) > 0 and self.attn_axes[0] == "T" and not torch.is_grad_enabled()
                and self.blocks[0].takes_tprop(T, compute)):
            tprop = self._packed_tprop()
        else:
            K.axis_mlp(x, B, T, H * W * C_, tp[0].weight, tp[0].bias, tp[2].weight, tp[2].bias, compute)
        ci = 0
        for i, axis in enumerate(self.attn_axes):
            blk = self.blocks[i]
            if axis == "C":                                                                          # l.184-189
                E = self.expanded_channel
                lift1, lift2 = self._packed_channel(compute)[ci]
                ci += 1
                n = x.numel()
                adt = K.act_torch_dtype(compute)
                z1 = torch.empty(n, E // 4, dtype=adt, device=x.device)
                K.linear(x, lift1, z1, M=n, act=L.ACT_GELU_ERF, a_s0=1)
                z = torch.empty(n, E, dtype=torch.float32, device=x.device)
                K.linear(z1, lift2, z, M=n)
                blk.forward_tokens(z, K.dense_seq(n // C_, C_), False, compute)
                K.gather_last(z, n, E, x)
            else:
                blk.forward_tokens(x, K.make_seq(axis, B, T, H, W), axis == "T", compute, tprop if i == 0 else None)
        return x

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _gpu_only(x)
        B, T, H, W, C_ = x.shape
        if (T, H, W, C_) != (self.T, self.H, self.W, self.C):
            raise ValueError(f"expected (B,{self.T},{self.H},{self.W},{self.C}), got {tuple(x.shape)}")
        if _wants_grad(self, x):
            from .train_forward import backbone_train
            y = backbone_train(self, x.to(torch.float32).reshape(B * T * H * W, C_).contiguous(), B, resolve_compute(self.compute))
            return y.view(B, T, H, W, C_)
        y = x.detach().to(torch.float32).contiguous().clone()
        return self.forward_tokens(y, B, resolve_compute(self.compute))
