"""Differentiable (training) forward of TANTE, composed from tante_amd.autograd ops.

Same arithmetic as the inference path, one HIP launch per layer instead of the fused block kernel, because the
backward pass needs the intermediate activations (normalised tokens, packed q/k/v, attention output, MLP pre-activation).
The rollout harness back-propagates through every re-fed frame exactly like the reference (trainer/trainer.py:154: no
detach), the tape being torch.autograd's.
"""
from __future__ import annotations

import torch

from . import _lib as L
from . import options as _O
from . import kernels as K
from . import stages as S
from .autograd import (EncTailFn, TailCfg, TailFn, FilmTableFn, ActFn, AttentionFn, BlockFn, BlockTailFn, block_tail_ready, AxisHWFn, AxisMlpFn, BranchOutFn, DeconvFn, DropoutAddFn, FilmPosFn, FilmPosFramesFn, FoldFn, LayerNormFn, LayerNormSkipFn, LinearFn, PatchEmbedFn, RtReduceFn,
                       TaylorFn)


_FOLDS = None   # (id(ln), id(W)) -> (W diag(gamma), b + W beta) while a fold_scope is open


class fold_scope:
    """One autograd graph's worth of folded LayerNorm affines.  A BPTT rollout calls every block n_steps times with the same weights;
    folding (two torch ops), re-packing the folded weight for the forward and the dgrad GEMM, and back-propagating through the fold
    once per CALL was ~10 % of the train step in ~2000 tiny launches.  Inside a scope the fold is built once and shared: autograd sums
    the per-call gradients of the folded weight and runs the fold's backward once.  The scope must not outlive its graph (the fold's
    saved tensors are freed by backward), so the rollouts open one per call."""

    def __enter__(self):
        global _FOLDS
        self.prev, _FOLDS = _FOLDS, {}
        return self

    def __exit__(self, *exc):
        global _FOLDS
        _FOLDS = self.prev
        return False


def _folded(lin_w, lin_b, ln, pre=None):
    """LayerNorm affine folded into the consumer: (W diag(gamma), b + W beta) -- parameter-sized torch expressions whose
    autograd distributes the gradients back to W, b, gamma, beta.  pre: the pair already computed (prepare_blocks)."""
    if _FOLDS is None or not torch.is_grad_enabled():
        return lin_w * ln.weight[None, :], lin_b + lin_w @ ln.bias
    key = (id(ln), id(lin_w))
    hit = _FOLDS.get(key)
    if hit is None:
        we, be, gw, gb = FoldFn.apply(lin_w, lin_b, ln.weight, ln.bias, pre)
        we._tante_grad, be._tante_grad = gw, gb       # the weight-gradient kernels of every use accumulate here
        hit = _FOLDS[key] = (we, be)
    return hit


FUSED_TRAIN_FORWARD = _O.register("TANTE_TRAIN_FUSED", True, __name__, "FUSED_TRAIN_FORWARD")
FUSED_TAIL_BACKWARD = _O.register("TANTE_TRAIN_FUSED_BWD", True, __name__, "FUSED_TAIL_BACKWARD")
BLOCK_RECORDS = _O.register("TANTE_TRAIN_BLOCK_RECORDS", True, __name__, "BLOCK_RECORDS")   # per-scope prepared record of a block (host time)
FUSED_ENC_ACT = _O.register("TANTE_TRAIN_FUSED_ENC_ACT", True, __name__, "FUSED_ENC_ACT")   # encoder GELUs inside the next stage's node
FUSED_AXIS_HW = _O.register("TANTE_TRAIN_FUSED_AXIS", True, __name__, "FUSED_AXIS_HW")     # H + W propagators' training forward in one launch
FUSED_HEAD_BACKWARD = _O.register("TANTE_TRAIN_FUSED_HEAD_BWD", True, __name__, "FUSED_HEAD_BACKWARD")   # q|k|v dgrad + LayerNorm1 backward in one launch
# the WHOLE backward of a block in one launch (tante_block_bwd_fused: tail + attention backward + head, q | k | v recomputed); off: three launches
FUSED_BLOCK_BACKWARD = _O.register("TANTE_TRAIN_FUSED_BLOCK_BWD", True, __name__, "FUSED_BLOCK_BACKWARD")


# decoder stages -> Taylor sum -> re-encoding of the predicted frame, forward and backward, as one launch each (csrc/tail_chain.hip)
FUSED_TAIL = _O.register("TANTE_TRAIN_FUSED_TAIL", True, __name__, "FUSED_TAIL")


def tail_train_cfg(model, B: int, compute: int, want_z: bool):
    """TailCfg for this model and batch when the one-launch training tail applies (shipped shapes: C = 256, three 2 x 2 stages without
    overlap, D <= 12, Wp % 16 == 0, bf16 compute, fixed dt with one output frame, every conv parameter with a gradient slot), else None.
    The packed streams live in the fold scope: one set per rollout graph."""
    if not (FUSED_TAIL and compute == L.BF16 and torch.is_grad_enabled() and _FOLDS is not None and model.deg and model.output_length == 1
            and 1 <= model.taylor_order <= 3 and type(model.encoder).__name__ == "enc_CNN" and model.C == 256):
        return None
    enc, decs = model.encoder, list(model.decoders)
    D = enc.chans[0]
    if any(type(d).__name__ != "dec_CNN" or tuple(d.P) != (2, 2, 2) or d.overlap != 0.0 or list(d.chans) != [256, 128, 64, D] for d in decs):
        return None
    if tuple(enc.P) != (2, 2, 2) or enc.overlap != 0.0 or list(enc.chans) != [D, 64, 128, 256] or not K.tail_supported(model.C, D, model.H_p, model.W_p):
        return None
    key = ("tail", id(model))
    rec = _FOLDS.get(key)
    if rec is None:
        from .autograd import _grad_slot
        ep = [q for i in range(3) for q in (getattr(enc, f"enc_conv_{i + 1}").conv.weight, getattr(enc, f"enc_conv_{i + 1}").conv.bias)]
        dps = [[q for i in range(3) for q in (getattr(d, f"dec_conv_{i + 1}").deconv.weight, getattr(d, f"dec_conv_{i + 1}").deconv.bias)]
               for d in decs]
        ok = all(q is not None and q.requires_grad and _grad_slot(q) is not None for q in ep + [q for dp in dps for q in dp])
        rec = _FOLDS[key] = (ep, dps, K.pack_tail(ep, D, False), [K.pack_tail(dp, D, True) for dp in dps]) if ok else False
    if rec is False:
        return None
    ep, dps, es, dss = rec
    import math
    coefs = [float(model.frame_interval) ** (k + 1) / math.factorial(k + 1) for k in range(model.taylor_order)]
    return TailCfg(B, model.T, model.H_p, model.W_p, model.C, D, coefs, dps, dss, ep, es, want_z)


BLOCK_CALLS = [0, 0]      # block_train calls / those that took the fused one-node path (GraphedTrainStep checks them at capture)
FRAME_FILM = _O.register("TANTE_TRAIN_FRAME_FILM", True, __name__, "FRAME_FILM")     # FiLM reads the window's frames where they are (no stack per call)
BATCH_PREP = _O.register("TANTE_TRAIN_BATCH_PREP", True, __name__, "BATCH_PREP")
FILM_TABLE_HIP = _O.register("TANTE_TRAIN_FILM_TABLE_HIP", True, __name__, "FILM_TABLE_HIP")   # FiLM tables + their backward as two HIP launches     # folds and backward fragment streams of all blocks in two launches


def prepare_blocks(model, compute: int):
    """Once per fold scope (= per rollout graph), before the first block runs: the LayerNorm folds of EVERY block in one launch
    (tante_fold_fwd_multi), their two backward fragment streams in another (tante_pack_block_tail_bwd_multi) and the forward kernel's weight
    streams in a third (tante_pack_block_train_multi), left in the scope under the keys block_train looks up -- per block these were five
    launches of ~4.7 us of latency each, 45 per train step.  Blocks that will not
    take the fused one-node path are left to block_train."""
    if _FOLDS is None or not BATCH_PREP or not torch.is_grad_enabled():
        return
    key = ("prepared", id(model))
    if key in _FOLDS:
        return
    _FOLDS[key] = True
    if not (FUSED_TRAIN_FORWARD and FUSED_TAIL_BACKWARD and compute == L.BF16):
        return
    todo = []
    for bb in getattr(model, "blocks", ()):
        if not hasattr(bb, "attn_axes"):
            return
        for i, axis in enumerate(bb.attn_axes):
            if axis == "C":
                continue
            blk = bb.blocks[i]
            a, m = blk.attn, blk.mlp
            if (id(blk.ln1), id(a.in_proj_weight)) in _FOLDS or (id(blk.ln2), id(m[0].weight)) in _FOLDS:
                continue
            if not (blk.fused and blk.ln1.eps == blk.ln2.eps and blk.hidden == blk.embed_dim and a.in_proj_bias is not None
                    and K.block_fused_train_supported(blk.embed_dim, blk.n_head, blk.hidden, K.make_seq(axis, 1, bb.T, bb.H, bb.W).L)
                    and block_tail_ready(a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias, m[0].weight, m[0].bias,
                                         m[2].weight, m[2].bias, blk.ln1.weight, blk.ln1.bias, blk.ln2.weight, blk.ln2.bias)):
                continue
            todo.append(blk)
    if len(todo) < 2:
        return
    import ctypes as C
    dev = todo[0].attn.in_proj_weight.device
    pairs = []
    for blk in todo:
        pairs.append((blk.attn.in_proj_weight, blk.attn.in_proj_bias, blk.ln1))
        pairs.append((blk.mlp[0].weight, blk.mlp[0].bias, blk.ln2))
    arr = (L.FoldFwd * len(pairs))()
    outs = []
    for f, (W, b, ln) in zip(arr, pairs):
        N, Kk = W.shape
        we, be = torch.empty_like(W), torch.empty(N, dtype=torch.float32, device=dev)
        f.W, f.b, f.gamma, f.beta, f.We, f.be, f.N, f.K = W.data_ptr(), b.data_ptr(), ln.weight.data_ptr(), ln.bias.data_ptr(), we.data_ptr(), be.data_ptr(), N, Kk
        outs.append((we, be))
    L.check(L.lib().tante_fold_fwd_multi(C.byref(arr), len(pairs), K._stream()), "tante_fold_fwd_multi")
    folded = [_folded(W, b, ln, pre=o) for (W, b, ln), o in zip(pairs, outs)]
    nbytes = L.lib().tante_block_tail_bwd_stream_bytes(todo[0].embed_dim, todo[0].hidden)
    mats = (L.Mat3 * (2 * len(todo)))()
    for i, blk in enumerate(todo):
        E = blk.embed_dim
        w_in, w1 = folded[2 * i][0].detach(), folded[2 * i + 1][0].detach()
        bst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        hst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        t, h = mats[2 * i], mats[2 * i + 1]
        t.a, t.b, t.c, t.dst = blk.mlp[2].weight.data_ptr(), w1.data_ptr(), blk.attn.out_proj.weight.data_ptr(), bst.data_ptr()
        h.a, h.b, h.c, h.dst = w_in[0:E].data_ptr(), w_in[E:2 * E].data_ptr(), w_in[2 * E:3 * E].data_ptr(), hst.data_ptr()
        _FOLDS[("bt_stream", id(blk))] = bst
        _FOLDS[("bh_stream", id(blk))] = hst
    L.check(L.lib().tante_pack_block_tail_bwd_multi(C.byref(mats), 2 * len(todo), todo[0].embed_dim, todo[0].hidden, K._stream()),
            "tante_pack_block_tail_bwd_multi")
    # ... and the forward kernel's weight streams in a third
    sbytes = L.lib().tante_block_stream_bytes(todo[0].embed_dim, todo[0].hidden)
    bw = (L.BlockWeights * len(todo))()
    for i, blk in enumerate(todo):
        (w_in, b_in), (w1, b1) = folded[2 * i], folded[2 * i + 1]
        st = torch.empty(sbytes, dtype=torch.uint8, device=dev)
        e = bw[i]
        e.in_w, e.in_b, e.out_w, e.out_b = w_in.data_ptr(), b_in.data_ptr(), blk.attn.out_proj.weight.data_ptr(), blk.attn.out_proj.bias.data_ptr()
        e.fc1_w, e.fc1_b, e.fc2_w, e.fc2_b = w1.data_ptr(), b1.data_ptr(), blk.mlp[2].weight.data_ptr(), blk.mlp[2].bias.data_ptr()
        e.block_stream = st.data_ptr()
        _FOLDS[("fs_stream", id(blk))] = st
    L.check(L.lib().tante_pack_block_train_multi(C.byref(bw), len(todo), todo[0].embed_dim, todo[0].hidden, K._stream()), "tante_pack_block_train_multi")


def block_train(blk, x: torch.Tensor, seq, causal: bool, compute: int, masks=None) -> torch.Tensor:
    """masks: None, or (attn_mask, key_padding_mask) as the additive fp32 tensors TransformerBlock._masks builds (dense sequences only):
    the per-operator chain with the masked attention node."""
    BLOCK_CALLS[0] += 1
    p = blk.p_drop if blk.training else 0.0     # nn.Dropout / MHA dropout are active in train() mode only
    if masks is not None:
        if p > 0.0:
            raise NotImplementedError("attn_mask / key_padding_mask under autograd with dropout > 0 in train() mode: the masked attention "
                                      "kernels have no attention-probability dropout (the TANTE path passes `causal` only)")
        from .autograd import MaskedAttentionFn
        adt = K.act_torch_dtype(compute)
        a, m = blk.attn, blk.mlp
        w_in, b_in = _folded(a.in_proj_weight, a.in_proj_bias, blk.ln1)
        xh, x = LayerNormSkipFn.apply(x, blk.ln1.eps, adt)
        qkv = LinearFn.apply(xh, w_in, b_in, None, compute, adt)
        o = MaskedAttentionFn.apply(qkv, blk.embed_dim, blk.n_head, seq.nseq, seq.L, causal, masks[0], masks[1])
        x = BranchOutFn.apply(o, a.out_proj.weight, a.out_proj.bias, x, L.ACT_NONE, 0.0, compute)
        w1, b1 = _folded(m[0].weight, m[0].bias, blk.ln2)
        xh2, x = LayerNormSkipFn.apply(x, blk.ln2.eps, adt)
        hpre = LinearFn.apply(xh2, w1, b1, None, compute, adt)
        return BranchOutFn.apply(hpre, m[2].weight, m[2].bias, x, L.ACT_GELU_TANH, 0.0, compute)
    # Every later call of this block inside one rollout graph (the BPTT steps) takes the record its first call left in the fold scope:
    # folded weights, the three fragment streams, the decision for the one-node path: the module attribute chains, fold / stream look-ups
    # and accumulator checks below are ~60 us of host time per block call (the step is GPU-bound on the bench box, with ~2 ms of margin).
    rec = _FOLDS.get(("blk_rec", id(blk), seq.L, compute)) if (_FOLDS is not None and BLOCK_RECORDS and torch.is_grad_enabled()) else None
    if rec is not None and x.dtype == torch.float32 and x.is_contiguous():
        from .autograd import next_seed
        seeds = (next_seed(), next_seed(), next_seed()) if p > 0.0 else (0, 0, 0)
        t = K.block_fused_train(x.detach(), rec[8], blk.embed_dim, blk.n_head, blk.hidden, seq, causal, rec[11], p, seeds, need_x1=False,
                                need_qkv=not rec[12])
        BLOCK_CALLS[1] += 1
        return BlockFn.apply(x, *rec[:8], t, rec[9], seq, blk.n_head, causal, p, seeds, compute, rec[10], rec[8] if rec[12] else None)
    adt = K.act_torch_dtype(compute)
    a, m = blk.attn, blk.mlp
    w_in, b_in = _folded(a.in_proj_weight, a.in_proj_bias, blk.ln1)
    if (FUSED_TRAIN_FORWARD and compute == L.BF16 and blk.fused and blk.ln1.eps == blk.ln2.eps and x.dtype == torch.float32 and x.is_contiguous()
            and K.block_fused_train_supported(blk.embed_dim, blk.n_head, blk.hidden, seq.L)):
        # ONE launch computes the block and stores what the backward pass reads; the autograd nodes below launch nothing in forward
        from .autograd import next_seed
        seeds = (next_seed(), next_seed(), next_seed()) if p > 0.0 else (0, 0, 0)
        w1, b1 = _folded(m[0].weight, m[0].bias, blk.ln2)
        # the kernel's weight stream is packed from the folded tensors, once per fold scope (= once per rollout graph: BPTT calls share it)
        key = ("fs_stream", id(blk))
        stream = _FOLDS.get(key) if _FOLDS is not None else None
        if stream is None:
            stream = K.pack_block_train((w_in, b_in, a.out_proj.weight, a.out_proj.bias, w1, b1, m[2].weight, m[2].bias), blk.embed_dim, blk.hidden)
            if _FOLDS is not None:
                _FOLDS[key] = stream
        fused_tail = (FUSED_TAIL_BACKWARD and torch.is_grad_enabled()
                      and block_tail_ready(a.out_proj.weight, a.out_proj.bias, w1, b1, m[2].weight, m[2].bias))
        # the whole backward in one launch: needs both transposed-fragment streams and every gradient slot (decided here: the forward then
        # skips the packed projection, which only the three-launch attention backward reads)
        fused_bwd = (fused_tail and FUSED_BLOCK_BACKWARD and FUSED_HEAD_BACKWARD and _FOLDS is not None and block_tail_ready(w_in, b_in)
                     and w_in.shape == (3 * blk.embed_dim, blk.embed_dim)
                     and K.block_bwd_fused_supported(blk.embed_dim, blk.n_head, blk.hidden, seq.L, causal))
        t = K.block_fused_train(x.detach(), stream, blk.embed_dim, blk.n_head, blk.hidden, seq, causal, blk.ln1.eps, p, seeds,
                                need_x1=not fused_tail, need_qkv=not fused_bwd)
        if fused_tail:
            # the block behind its attention as ONE autograd node whose backward is ONE launch (tante_block_tail_bwd)
            key = ("bt_stream", id(blk))
            bstream = _FOLDS.get(key) if _FOLDS is not None else None
            if bstream is None:
                bstream = K.pack_block_tail_bwd(m[2].weight, w1, a.out_proj.weight, blk.embed_dim, blk.hidden)
                if _FOLDS is not None:
                    _FOLDS[key] = bstream
            if block_tail_ready(w_in, b_in):
                hstream = None
                if FUSED_HEAD_BACKWARD and w_in.shape == (3 * blk.embed_dim, blk.embed_dim) and blk.hidden == blk.embed_dim:
                    key = ("bh_stream", id(blk))
                    hstream = _FOLDS.get(key) if _FOLDS is not None else None
                    if hstream is None:      # the three 256-row blocks of the folded in-projection weight, transposed into fragments
                        E = blk.embed_dim
                        wd = w_in.detach()
                        hstream = K.pack_block_tail_bwd(wd[0:E], wd[E:2 * E], wd[2 * E:3 * E], E, blk.hidden)
                        if _FOLDS is not None:
                            _FOLDS[key] = hstream
                if _FOLDS is not None:
                    _FOLDS[("blk_rec", id(blk), seq.L, compute)] = (w_in, b_in, a.out_proj.weight, a.out_proj.bias, w1, b1, m[2].weight, m[2].bias,
                                                                    stream, bstream, hstream, blk.ln1.eps, fused_bwd and hstream is not None)
                BLOCK_CALLS[1] += 1
                return BlockFn.apply(x, w_in, b_in, a.out_proj.weight, a.out_proj.bias, w1, b1, m[2].weight, m[2].bias, t, bstream, seq,
                                     blk.n_head, causal, p, seeds, compute, hstream, stream if (fused_bwd and hstream is not None) else None)
            xh, xs = LayerNormSkipFn.apply(x, blk.ln1.eps, adt, (t["xh1"], t["st1"]))
            qkv = LinearFn.apply(xh, w_in, b_in, None, compute, adt, t["qkv"])
            o = AttentionFn.apply(qkv, seq, blk.embed_dim, blk.n_head, causal, p, (t["o"], seeds[0]))
            return BlockTailFn.apply(o, xs, a.out_proj.weight, a.out_proj.bias, w1, b1, m[2].weight, m[2].bias, t, bstream, p, seeds)
        xh, xs = LayerNormSkipFn.apply(x, blk.ln1.eps, adt, (t["xh1"], t["st1"]))
        qkv = LinearFn.apply(xh, w_in, b_in, None, compute, adt, t["qkv"])
        o = AttentionFn.apply(qkv, seq, blk.embed_dim, blk.n_head, causal, p, (t["o"], seeds[0]))
        x1 = BranchOutFn.apply(o, a.out_proj.weight, a.out_proj.bias, xs, L.ACT_NONE, p, compute, (t["x1"], None, seeds[1]))
        xh2, x1s = LayerNormSkipFn.apply(x1, blk.ln2.eps, adt, (t["xh2"], t["st2"]))
        hpre = LinearFn.apply(xh2, w1, b1, None, compute, adt, t["hpre"])
        return BranchOutFn.apply(hpre, m[2].weight, m[2].bias, x1s, L.ACT_GELU_TANH, p, compute, (t["out"], t["act"], seeds[2]))
    xh, x = LayerNormSkipFn.apply(x, blk.ln1.eps, adt)      # x: the same tokens, as the skip operand whose gradient LN's backward adds
    qkv = LinearFn.apply(xh, w_in, b_in, None, compute, adt)
    o = AttentionFn.apply(qkv, seq, blk.embed_dim, blk.n_head, causal, p)
    x = BranchOutFn.apply(o, a.out_proj.weight, a.out_proj.bias, x, L.ACT_NONE, p, compute)       # x + drop(out_proj(o))
    w1, b1 = _folded(m[0].weight, m[0].bias, blk.ln2)
    xh2, x = LayerNormSkipFn.apply(x, blk.ln2.eps, adt)
    hpre = LinearFn.apply(xh2, w1, b1, None, compute, adt)
    return BranchOutFn.apply(hpre, m[2].weight, m[2].bias, x, L.ACT_GELU_TANH, p, compute)           # x + drop(fc2(gelu(hpre)))


def backbone_train(bb, x: torch.Tensor, B: int, compute: int) -> torch.Tensor:
    T, H, W, C_ = bb.T, bb.H, bb.W, bb.C
    vp, hp, tp = bb.vertical_propagator, bb.horizontal_propagator, bb.temporal_propagator
    if FUSED_AXIS_HW and K.axis_hw_train_supported(H, W, C_, compute) and x.dtype == torch.float32:
        x = AxisHWFn.apply(x, vp[0].weight, vp[0].bias, vp[2].weight, vp[2].bias, hp[0].weight, hp[0].bias, hp[2].weight, hp[2].bias,
                           B * T, H, W, C_, compute)
    else:
        x = AxisMlpFn.apply(x, vp[0].weight, vp[0].bias, vp[2].weight, vp[2].bias, B * T, H, W * C_, compute)
        x = AxisMlpFn.apply(x, hp[0].weight, hp[0].bias, hp[2].weight, hp[2].bias, B * T * H, W, C_, compute)
    x = AxisMlpFn.apply(x, tp[0].weight, tp[0].bias, tp[2].weight, tp[2].bias, B, T, H * W * C_, compute)
    ci = 0
    for i, axis in enumerate(bb.attn_axes):
        if axis == "C":
            # attn_backbone.py:184-189: every scalar channel value is lifted 1 -> E/4 -> E (channel_blocks), the block attends over the C
            # axis of each token (sequences of C "tokens" of width E), and feature E - 1 comes back as the channel's new value
            cb = bb.channel_blocks[ci]
            ci += 1
            n, E = x.shape[0], bb.expanded_channel
            adt = K.act_torch_dtype(compute)
            # the first lift is an outer product with a (E/4, 1) weight, not a contraction: an elementwise torch expression
            z1 = ActFn.apply((x.reshape(n * C_, 1) * cb[0].weight.view(1, -1) + cb[0].bias).contiguous(), L.ACT_GELU_ERF, adt)
            z = LinearFn.apply(z1, cb[2].weight, cb[2].bias, None, compute, torch.float32)
            z = block_train(bb.blocks[i], z, K.dense_seq(n, C_), False, compute)
            x = z.view(n, C_, E)[:, :, -1].contiguous()
            continue
        x = block_train(bb.blocks[i], x, K.make_seq(axis, B, T, H, W), axis == "T", compute)
    return x


def encoder_train(enc, inp: torch.Tensor, compute: int) -> torch.Tensor:
    if type(enc).__name__ == "enc_FNO":
        from .spectral import enc_fno_train
        return enc_fno_train(enc, inp, compute)
    B, T, D, H, W = inp.shape
    adt = K.act_torch_dtype(compute)
    n_img, h, w = B * T, H, W
    z = inp.reshape(n_img, D, H, W)
    if any(S.stride_pad(p, enc.overlap) != (p, 0) for p in enc.P):
        # 'same'-padded stages (kernel 4: patch_scale 16 / 32 / 64, enc_dec_cnn.py:66-81) and overlapping ones (overlap_ratio > 0: stride <
        # kernel, adaptive average pool behind the conv, enc_dec_cnn.py:97-110): the general route, stage by stage -- im2col with the padding /
        # stride (its backward: the gather-sum col2im) + the dense GEMM (+ AvgPoolFn), GELU between the stages as an op of its own
        from .spectral import _conv_stage_train
        for i in range(3):
            conv = getattr(enc, f"enc_conv_{i + 1}").conv
            p, ci = enc.P[i], enc.chans[i]
            last = i == 2
            z, h, w = _conv_stage_train(z, conv, p, enc.overlap, compute, torch.float32 if last else adt, None if i == 0 else (n_img, ci, h, w))
            if not last:
                z = ActFn.apply(z, L.ACT_GELU_ERF, adt)
        return z
    for i in range(3):
        conv = getattr(enc, f"enc_conv_{i + 1}").conv
        p, ci = enc.P[i], enc.chans[i]
        last = i == 2
        # stages 2 and 3 take the previous stage's PRE-activation and apply the GELU themselves (their backward folds GELU' into the
        # data-gradient scatter: no stand-alone activation backward over the two largest images of the step)
        fold = FUSED_ENC_ACT and i > 0 and ci % 4 == 0
        z = PatchEmbedFn.apply(z, conv.weight, conv.bias, n_img, h, w, ci, p, i == 0, compute, torch.float32 if last else adt,
                               L.ACT_GELU_ERF if fold else L.ACT_NONE)
        if not last and not (FUSED_ENC_ACT and enc.chans[i + 1] % 4 == 0):
            z = ActFn.apply(z, L.ACT_GELU_ERF, adt)
        h, w = h // p, w // p
    return z                                   # (B*T*Hp*Wp, C) fp32, before FiLM / positional embeddings


def decoder_train(dec, a: torch.Tensor, n_img: int, compute: int) -> torch.Tensor:
    if type(dec).__name__ == "dec_FNO":
        from .spectral import dec_fno_train
        return dec_fno_train(dec, a.reshape(-1, dec.chans[0]), n_img, compute)
    adt = K.act_torch_dtype(compute)
    h, w = dec.patch_shape
    x = a
    if any(S.stride_pad(p, dec.overlap) != (p, 0) for p in dec.P):
        # 'same'-padded stages (enc_dec_cnn.py:128-143, 164-184): the padding crops the transposed convolution's result and the reference
        # resizes it back (bilinear) -- DeconvFn + CropResizeFn per stage, channels-last rows between the stages; overlapping stages
        # (overlap_ratio > 0): tap GEMM + Col2imFn (summed taps) + CropResizeFn
        from .spectral import _deconv_stage_train
        for i in range(3):
            dc = getattr(dec, f"dec_conv_{i + 1}").deconv
            p = dec.P[i]
            last = i == 2
            x = _deconv_stage_train(x.reshape(-1, dec.chans[i]), dc, n_img, h, w, p, dec.overlap, compute, last, torch.float32 if last else adt)
            if not last:
                x = ActFn.apply(x, L.ACT_GELU_ERF, adt)
            h, w = h * p, w * p
        return x
    for i in range(3):
        dc = getattr(dec, f"dec_conv_{i + 1}").deconv
        p, co = dec.P[i], dec.chans[i + 1]
        last = i == 2
        x = DeconvFn.apply(x.reshape(-1, dec.chans[i]), dc.weight, dc.bias, n_img, h, w, p, last, compute, adt)
        if not last:
            x = ActFn.apply(x, L.ACT_GELU_ERF, adt)
        h, w = h * p, w * p
    return x                                   # (n_img, D, H, W) fp32


def interprator_train(it, d3: torch.Tensor, B: int, out_T: float, compute: int) -> torch.Tensor:
    """tante.py:191-201 on (B*L, C) tokens -> rt (B,)."""
    adt = K.act_torch_dtype(compute)
    lin = it.interprete
    h = ActFn.apply(LinearFn.apply(d3, lin[0].weight, lin[0].bias, None, compute, adt), L.ACT_RELU, adt)
    h = ActFn.apply(LinearFn.apply(h, lin[2].weight, lin[2].bias, None, compute, adt), L.ACT_RELU, adt)
    t = LinearFn.apply(h, lin[4].weight, lin[4].bias, None, compute, torch.float32)
    return RtReduceFn.apply(t, B, it.sp_dim, float(out_T), float(it.ep))


TRAIN_ENC_CACHE = _O.register("TANTE_TRAIN_ENC_CACHE", True, __name__, "TRAIN_ENC_CACHE")


def train_enc_cache_ok(model) -> bool:
    """The BPTT rollout may encode every frame once (see rollout_model): the patch-embed encoder is frame-wise (no time mixing before
    FiLM), so the encoding of a frame that sits in several windows is the same tensor, and autograd sums its gradients."""
    return (TRAIN_ENC_CACHE and getattr(model, "deg", False) and type(model.encoder).__name__ == "enc_CNN"
            and all(S.stride_pad(p, model.encoder.overlap) == (p, 0) for p in model.encoder.P))


def encode_frames_train(model, frames: torch.Tensor, compute: int) -> torch.Tensor:
    """frames (B, k, D, H, W) -> their pre-FiLM token rows (B, k, HW, C) fp32, differentiable (encoder_train on the k frames of every item)."""
    B, k = frames.shape[:2]
    if not frames.requires_grad and frames.dtype == torch.float32:
        tail = tail_train_cfg(model, B, compute, True)       # input frames: the tail kernels' encoder half, one launch each way
        if tail is not None:
            z = EncTailFn.apply(frames.reshape(B * k, *frames.shape[2:]), tail, *tail.enc_params)
            return z.view(B, k, model.H_p * model.W_p, model.C)
    z = encoder_train(model.encoder, frames.to(torch.float32).contiguous(), compute)
    return z.view(B, k, model.H_p * model.W_p, model.C)


def tante_train_forward(model, inp: torch.Tensor, compute: int, out_T=1, z_win=None, per_sample_counts: bool = False, next_z=None):
    """z_win (optional): the window's frames already encoded, (B, T, HW, C) fp32 contiguous (encode_frames_train); `inp` then only
    supplies its last frame (the Taylor sum's base) and may be that frame alone, (B, 1, D, H, W).
    next_z (optional): a list; True / False in next_z[0] on entry = whether the caller wants the predicted frame's encoding for the
    next call's window.  When the one-launch tail runs (tail_train_cfg), next_z[0] is that encoding (B, HW, C) -- or None for False --
    on return; when it does not, next_z is left alone and the caller encodes the frame itself."""
    B, _, D, H, W = inp.shape
    T = model.T
    Hp, Wp, C_ = model.H_p, model.W_p, model.C
    HW = Hp * Wp
    prepare_blocks(model, compute)
    z_frames = None
    if isinstance(z_win, (list, tuple)):       # the window as T separate frame encodings, each (B, HW, C)
        if FRAME_FILM and FilmPosFramesFn.supported(z_win, C_):
            z_frames = z_win
        else:
            z_win = torch.stack(list(z_win), dim=1)
    z = None if z_frames is not None else (encoder_train(model.encoder, inp, compute) if z_win is None else z_win.reshape(B * T * HW, C_))
    # film(x, t) = x * (1 + scale(t)) + shift(t) with t = the window's fixed time stamps: the two tables are the same for every call of a
    # rollout graph, so they are built once per fold scope (like the folded LayerNorm weights) -- four tiny torch Linear layers, their
    # activations and their backward were ~150 launches of ~4.5 us per train step when rebuilt in each of the four BPTT calls
    key = ("film_tables", id(model))
    tabs = _FOLDS.get(key) if _FOLDS is not None else None
    if tabs is None:
        te = model.t_encode
        sc, sh = te.condition_to_scale, te.condition_to_shift
        if FILM_TABLE_HIP and T <= 8 and C_ % 16 == 0:
            # one launch forward, one backward (autograd.FilmTableFn); the tables' gradients of the rollout's calls accumulate on the node
            tsec = model.t_seq.to(inp.device, torch.float32).contiguous()
            fa, fb, acc = FilmTableFn.apply(tsec, model.t_emb.view(T, C_), sc[0].weight, sc[0].bias, sc[2].weight, sc[2].bias,
                                            sh[0].weight, sh[0].bias, sh[2].weight, sh[2].bias)
            fa._tante_grad, fb._tante_grad = acc[0], acc[1]
        else:
            t = model.t_seq.to(inp.device, torch.float32)[:, None]
            fa = (1.0 + sc(t)).contiguous()                     # (T, C)
            fb = (sh(t) + model.t_emb.view(T, C_)).contiguous()
        tabs = (fa, fb)
        if _FOLDS is not None:
            _FOLDS[key] = tabs
    fa, fb = tabs
    s_view = model.s_emb.view(HW, C_)
    if FILM_TABLE_HIP:      # s_emb's gradient slot, for the kernel to add into (the rollout's calls share the parameter)
        from .autograd import _grad_slot
        gs = _grad_slot(model.s_emb)
        if gs is not None:
            s_view._tante_grad = gs.view(HW, C_)
    if z_frames is not None:
        x = FilmPosFramesFn.apply(fa, fb, s_view, *z_frames)
    else:
        x = FilmPosFn.apply(z, fa, fb, s_view, T, HW)
    derivs, rts = [], []
    tail = None
    if next_z is not None and inp.shape[1] >= 1 and inp.dtype == torch.float32:
        tail = tail_train_cfg(model, B, compute, bool(next_z[0]))
    if tail is not None:
        xs = []
        for i in range(model.taylor_order):
            x = backbone_train(model.blocks[i], x, B, compute)
            xs.append(x)
        y, zn = TailFn.apply(inp[:, -1:], tail, *xs)
        next_z[0] = zn
        next_z.append("tail")
        return y
    for i in range(model.taylor_order):
        x = backbone_train(model.blocks[i], x, B, compute)
        last = x.view(B, T, HW, C_)[:, -1].reshape(B * HW, C_)
        if not model.deg:
            # intended semantics of tante.py:148-152 (the shipped glue raises): rt from the last slot, 3-D film with rt, decode
            rt = interprator_train(model.interprators[i], last, B, out_T, compute)
            rts.append(rt)
            mod = model.modifiers[i]
            ma = 1.0 + mod.condition_to_scale(rt[:, None])                     # (B, C): film(x, rt) = x * (1 + scale) + shift
            mb = mod.condition_to_shift(rt[:, None])
            zero_pos = torch.zeros(HW, C_, dtype=torch.float32, device=inp.device)
            last = FilmPosFn.apply(last, ma.contiguous(), mb.contiguous(), zero_pos, B, HW)    # rows (b, hw): table row = b
        d = decoder_train(model.decoders[i], last, B, compute)
        derivs.append(d.view(B, 1, D, H, W))
    if model.deg:
        return TaylorFn.apply(inp, model.frame_interval, model.output_length, *derivs)
    R_t = torch.stack(rts, dim=1).mean(dim=1)
    # tante.py:163 -- sample 0 decides (host sync, as in the reference); per_sample_counts: frames for the largest count (TANTE.forward)
    n_out = int(torch.floor(R_t.detach()).max()) if per_sample_counts else int(torch.floor(R_t[0].detach()))
    if n_out < 1:
        return torch.empty(B, 0, D, H, W, dtype=torch.float32, device=inp.device), R_t
    return TaylorFn.apply(inp, model.frame_interval, n_out, *derivs), R_t
