"""Autoregressive rollout harness (host side) with the reference's semantics.

Trainer.rollout_model / Evaler.rollout_model  (trainer/trainer.py:144-159, trainer/evaler.py:121-138)
R_Trainer.rollout_model / R_Evaler.rollout_model (trainer/r_trainer.py:112-133, trainer/r_evaler.py:87-105)
DefaultChannelsFirstFormatter / ...LastFormatter   (data/datamodule.py:184-202)
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from . import options as _O

# A/B switches (tante_amd.set_option): each falls back to the form the default replaced
NO_ENC_CACHE = _O.register("TANTE_NO_ENC_CACHE", False, __name__, "NO_ENC_CACHE")          # window-by-window encoder
NO_TAIL_ENC = _O.register("TANTE_NO_TAIL_ENC", False, __name__, "NO_TAIL_ENC")            # predicted frames re-encoded by the encoder launches
NO_FUSED_FORMAT = _O.register("TANTE_NO_FUSED_FORMAT", False, __name__, "NO_FUSED_FORMAT")  # formatter.process_input as torch ops
# the reference frames' nan_to_num on a SECOND stream under the rollout: the default of round 3, when it hid 57 us; measured again at the
# end of round 4 (tools/_r4_side.sh, same box, three rounds) it COSTS 3.4 % at B = 8 (16.79 -> 17.36 k frames/s without it), 4.1 % at
# B = 4 and 8 % at B = 1 -- its workgroups take CU slots from block launches that need the whole chip for their one resident round, and
# the fork / join is tens of microseconds of a small-batch rollout.  Opt-in now.
SIDE_STREAM = _O.register("TANTE_SIDE_STREAM", False, __name__, "SIDE_STREAM")


class DefaultChannelsFirstFormatter:
    """'b t h w c -> b t c h w' (+ nan_to_num) on the way in, the inverse on the way out."""

    def __init__(self, metadata=None):
        self.metadata = metadata

    def process_input(self, data: Dict) -> Tuple:
        x = data["input"]
        x = x.permute(0, 1, x.dim() - 1, *range(2, x.dim() - 1))
        return (torch.nan_to_num(x),), torch.nan_to_num(data["output"])

    def process_output(self, output: torch.Tensor) -> torch.Tensor:
        return output.permute(0, 1, *range(3, output.dim()), 2)


class DefaultChannelsLastFormatter:
    def __init__(self, metadata=None):
        self.metadata = metadata

    def process_input(self, data: Dict) -> Tuple:
        return (torch.nan_to_num(data["input"]),), torch.nan_to_num(data["output"])

    def process_output(self, output: torch.Tensor) -> torch.Tensor:
        return output


def _nan_to_num(t: torch.Tensor) -> torch.Tensor:
    """torch.nan_to_num(t) -- the formatter's pass over the reference frames -- as one 16-byte-per-lane HIP pass where it applies."""
    if t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0 and t.numel() > 0:
        from . import _lib as L
        from . import kernels as K
        y = torch.empty_like(t)
        L.check(L.lib().tante_nan_to_num(t.data_ptr(), y.data_ptr(), t.numel(), K._stream()), "tante_nan_to_num")
        return y
    return torch.nan_to_num(t)


_SIDE = {}


def _side_stream(device: torch.device) -> "torch.cuda.Stream":
    """One helper stream per device for work that is independent of the model calls of a rollout (the reference frames' nan_to_num)."""
    st = _SIDE.get(device.index)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _SIDE[device.index] = st
    return st


def _rollout_in_place(model, x: torch.Tensor, n_steps: int, raw_input: torch.Tensor = None) -> torch.Tensor:
    """The reference's loop without its copies: one (B, T + frames, D, H, W) buffer holds the input window and every
    predicted frame; each model call reads its window in place (strided view) and writes its prediction into the next
    slots, so `torch.cat([moving[:, k:], y])` and the per-step output concatenation disappear."""
    if raw_input is not None:     # (B, T, H, W, D) channels-last, as the datamodule yields it
        B, T = raw_input.shape[:2]
        frame_shape, dev = (raw_input.shape[4], raw_input.shape[2], raw_input.shape[3]), raw_input.device
    else:
        B, T = x.shape[:2]
        frame_shape, dev = tuple(x.shape[2:]), x.device
    ol = model.output_length
    n_calls = -(-n_steps // ol)
    buf = torch.empty(B, T + n_calls * ol, *frame_shape, dtype=torch.float32, device=dev)
    if raw_input is not None:     # the default formatter's permute + nan_to_num, fused into the one copy that fills the buffer
        from . import _lib as L
        from . import kernels as K
        Bq, Tq, H, W, D = raw_input.shape
        L.check(L.lib().tante_format_input(raw_input.data_ptr(), Bq * Tq, Tq, H * W, D, buf.data_ptr(), buf.stride(0), K._stream()),
                "tante_format_input")
    else:
        buf[:, :T].copy_(x)       # the formatter's 'b t h w c -> b t c h w' is materialised here, once
    if model.enc_cache_supported() and not NO_ENC_CACHE:
        # every frame (input or predicted) is encoded once, when it first enters a window; the windows read the frame-major cache
        HW, C_ = model.H_p * model.W_p, model.C
        z = torch.empty(T + n_calls * ol, B, HW, C_, dtype=torch.float32, device=dev)
        encoded = 0
        # one output frame per call: the launch that writes the predicted frame also writes its encoding for the next call's window
        # (csrc/head_enc.hip) -- only the initial window goes through the encoder kernels
        tail = ol == 1 and model.tail_fused_supported() and not NO_TAIL_ENC
        for s in range(n_calls):
            need = s * ol + T
            if need > encoded:                       # the first call encodes the whole window, later calls the `ol` new frames
                model.encode_frames(buf[:, encoded: need], z[encoded: need])
            encoded = need
            nxt = z[T + s] if (tail and s + 1 < n_calls) else None
            model(buf[:, s * ol: s * ol + T], out=buf[:, T + s * ol: T + (s + 1) * ol], enc_cache=(z[s * ol:], B * HW * C_, HW * C_),
                  **({"enc_next": nxt} if nxt is not None else {}))
            if nxt is not None:
                encoded = need + 1
        return buf[:, T: T + n_steps]
    for s in range(n_calls):
        model(buf[:, s * ol: s * ol + T], out=buf[:, T + s * ol: T + (s + 1) * ol])
    return buf[:, T: T + n_steps]


def rollout_model(model, batch: Dict, formatter, n_steps: int, device=None):
    """Sliding-window re-feed until n_steps frames exist; returns (y_pred channels-last [:, :n_steps], y_ref)."""
    device = device or next(model.parameters()).device
    from .tante import TANTE
    raw = batch["input"]
    if (not NO_FUSED_FORMAT and type(formatter) is DefaultChannelsFirstFormatter and isinstance(model, TANTE) and model.deg and not torch.is_grad_enabled()
            and raw.dim() == 5 and raw.shape[1] == model.T and raw.dtype == torch.float32 and raw.is_cuda and raw.is_contiguous()):
        # same result as formatter.process_input + the in-place rollout below, without the two extra passes over the window
        out_ref = batch["output"]
        if out_ref.is_cuda and SIDE_STREAM:
            # (opt-in, see SIDE_STREAM) the formatter's nan_to_num of the REFERENCE frames (370 MB through HBM at cfg2, 57 us) depends on
            # nothing the rollout computes: on a second stream, joined before returning (inside a graph capture the fork and the join
            # are captured with it; the capture's private pool needs no record_stream)
            main = torch.cuda.current_stream(out_ref.device)
            side = _side_stream(out_ref.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                y_ref = _nan_to_num(out_ref)
            if not torch.cuda.is_current_stream_capturing():
                y_ref.record_stream(main)
            y = _rollout_in_place(model, None, n_steps, raw_input=raw)
            main.wait_stream(side)
            return formatter.process_output(y), y_ref.to(device)
        y_ref = _nan_to_num(out_ref)
        return formatter.process_output(_rollout_in_place(model, None, n_steps, raw_input=raw)), y_ref.to(device)
    if (not NO_FUSED_FORMAT and type(formatter) is DefaultChannelsFirstFormatter and raw.dim() == 5 and raw.dtype == torch.float32 and raw.is_cuda
            and raw.is_contiguous() and raw.data_ptr() % 16 == 0 and not raw.requires_grad and not batch["output"].requires_grad):
        # the default formatter's 'b t h w c -> b t c h w' + nan_to_num as ONE HIP pass, also on the training path (as torch ops: a
        # permute, two nan_to_num kernels and the copy that makes the window contiguous)
        from . import _lib as L
        from . import kernels as K
        Bq, Tq, H, W, D = raw.shape
        moving = torch.empty(Bq, Tq, D, H, W, dtype=torch.float32, device=raw.device)
        L.check(L.lib().tante_format_input(raw.data_ptr(), Bq * Tq, Tq, H * W, D, moving.data_ptr(), moving.stride(0), K._stream()), "tante_format_input")
        y_ref = _nan_to_num(batch["output"])
    else:
        moving, y_ref = formatter.process_input(batch)
        moving = moving[0].to(device)
    if isinstance(model, TANTE) and model.deg and not torch.is_grad_enabled() and moving.shape[1] == model.T \
            and moving.dtype == torch.float32:
        return formatter.process_output(_rollout_in_place(model, moving, n_steps)), y_ref.to(device)
    from .train_forward import fold_scope
    preds, produced = [], 0
    if isinstance(model, TANTE) and torch.is_grad_enabled() and moving.shape[1] == model.T and moving.dtype == torch.float32:
        from .train_forward import encode_frames_train, tante_train_forward, train_enc_cache_ok
        from .attn_backbone import resolve_compute
        if train_enc_cache_ok(model) and any(p.requires_grad for p in model.parameters()):
            # BPTT with every frame encoded ONCE: the windows of consecutive calls share T - output_length frames and the encoder is
            # frame-wise, so a window is assembled from cached per-frame encodings (autograd sums the gradients a frame's encoding
            # receives from every window it sits in, and the encoder's backward runs once per frame).  Same values and gradients as
            # re-encoding every window -- the reference's graph with its common subexpressions shared -- for 7 / 16 of the encoder work.
            compute = resolve_compute(model.compute)
            T = model.T
            def frames_of(z):      # (B, k, HW, C) -> k frame tensors without unbind's zero-fill + copy + sum backward
                if z.shape[1] == 1:
                    return [z.view(z.shape[0], z.shape[2], z.shape[3])]
                from .autograd import SplitFramesFn
                return list(SplitFramesFn.apply(z))
            from .autograd import guard_frame_gradient
            with fold_scope():
                zf = frames_of(encode_frames_train(model, moving, compute))
                for f in zf:
                    guard_frame_gradient(f)
                last = moving[:, -1:].contiguous()
                while produced < n_steps:
                    # (the one-launch tail, where it applies, also returns the predicted frame's encoding: train_forward.tail_train_cfg)
                    nz = [produced + model.output_length < n_steps]
                    y = tante_train_forward(model, last, compute, 1, z_win=zf[-T:], next_z=nz)      # the window's frames where they are
                    produced += y.shape[1]
                    preds.append(formatter.process_output(y))
                    if produced < n_steps:
                        if len(nz) > 1:
                            zf.append(nz[0])
                        else:
                            zf.extend(frames_of(encode_frames_train(model, y, compute)))
                        guard_frame_gradient(zf[-1])
                        last = y[:, -1:].contiguous()
            return torch.cat(preds, dim=1)[:, :n_steps], y_ref.to(device)
    with fold_scope():      # the re-fed calls of one rollout share one autograd graph (and one folded copy of every LayerNorm affine)
        while produced < n_steps:
            y = model(moving)
            produced += y.shape[1]
            if produced < n_steps:
                moving = torch.cat([moving[:, y.shape[1]:], y], dim=1)
            preds.append(formatter.process_output(y))
    return torch.cat(preds, dim=1)[:, :n_steps], y_ref.to(device)


class GraphedRollout:
    """rollout_model(model, batch, formatter, n_steps) replayed as ONE captured HIP graph -- for small batches, where the hundred-odd
    launches of a rollout are issue-bound on the host (cfg2, B = 1: 3.3 k -> 4.4 k frames/s).  Same kernels, same bits as the eager call.

        roll = tante_amd.GraphedRollout(model, batch, formatter, n_steps)     # warms up, captures; `batch`'s tensors become its inputs
        y_pred, y_ref = roll(other_batch)                                      # copies the batch in (nothing if it IS `batch`), replays

    The returned tensors are the graph's own output buffers: they are overwritten by the next call (clone what must be kept).  The
    graph holds the packed weights by address, so it is re-captured when a parameter changed (optimizer step, load_state_dict, .to()).
    Inference only (the model must be in eval mode; no autograd)."""

    def __init__(self, model, batch: Dict, formatter, n_steps: int, device=None):
        self.model, self.formatter, self.n_steps = model, formatter, n_steps
        self.device = device or next(model.parameters()).device
        if self.device.type != "cuda":
            raise RuntimeError("GraphedRollout needs the model on the GPU (no CPU fallback)")
        if model.training:
            raise RuntimeError("GraphedRollout replays an inference rollout: call model.eval() first")
        # the graph reads THESE tensors: the batch given here is adopted as its input buffers (a call with the same tensors copies nothing;
        # any other batch is copied into them first)
        self._in = {k: batch[k].to(self.device) for k in ("input", "output")}
        self._graph = self._out = self._key = None
        self._capture()

    def _weights_key(self):
        # what the captured launches depend on besides the batch: the parameters (by address and version), the compute mode the model
        # resolves to (set_compute / an enclosing autocast) and the A/B switches (options.EPOCH): any change re-captures (ADVICE round 4:
        # set_compute or a set_option flip kept replaying the stale graph)
        from .attn_backbone import _WEIGHT_EPOCH, resolve_compute
        from .options import EPOCH
        return (_WEIGHT_EPOCH[0], EPOCH[0], resolve_compute(getattr(self.model, "compute", None))) + tuple(
            (p.data_ptr(), p._version) for p in self.model.parameters())

    def _run(self):
        with torch.inference_mode():
            return rollout_model(self.model, self._in, self.formatter, self.n_steps, device=self.device)

    def _capture(self):
        dev = self.device
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=dev)      # one warm-up stream for every (re-)capture: per-stream workspaces are keyed to it
        side = self._side
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                self._run()                      # packs, tables, workspaces and allocator pools exist before the capture
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        # thread-local capture: another thread's HIP calls (RCCL's watchdog / proxy threads in a data-parallel job) must not invalidate it
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            out = self._run()
        self._graph, self._out, self._key = g, out, self._weights_key()

    def __call__(self, batch: Dict):
        for k in ("input", "output"):
            src = batch[k]
            if src.shape != self._in[k].shape or src.dtype != self._in[k].dtype:
                raise ValueError(f"GraphedRollout was captured for batch[{k!r}] of shape {tuple(self._in[k].shape)} {self._in[k].dtype}, got "
                                 f"{tuple(src.shape)} {src.dtype}")
            if src.data_ptr() != self._in[k].data_ptr():
                self._in[k].copy_(src, non_blocking=True)
        if self._weights_key() != self._key:
            self._capture()
        self._graph.replay()
        return self._out


def rollout_adaptive(model, batch: Dict, formatter, n_steps: int, out_T: float, per_sample: bool, device=None,
                     batch_when_equivalent: bool = True):
    """deg=False rollouts: per_sample=True, out_T=1.5 is R_Trainer's loop; per_sample=False,
    out_T=n_steps_rollout is R_Evaler's.  Returns (y_pred, y_ref, Rts).

    R_Trainer serialises the batch (`for i in range(batch)`, r_trainer.py:118) because `floor(R_t[0])` lets sample 0 decide the frame
    count of a call.  With out_T < 1.999 the clamp bounds every r_t to [1.001, out_T + 0.001) (tante.py:191-201), so EVERY sample
    produces exactly one frame per call whatever sample 0 says: the per-sample loop and one batched rollout compute the same frames,
    and `eval_rt` takes a mean over all R_t (order-free).  Then the batch runs as one.
    For larger out_T the samples advance at their OWN rates (floor(R_t[i]) frames per call): _rollout_adaptive_batched keeps them in one
    batch all the same (batch_when_equivalent=False brings the reference's serial loop back)."""
    device = device or next(model.parameters()).device
    xs, y_ref = formatter.process_input(batch)
    xs = xs[0].to(device)
    if per_sample and batch_when_equivalent and float(out_T) < 1.999:
        per_sample = False
    if per_sample and batch_when_equivalent and xs.shape[0] > 1:
        return _rollout_adaptive_batched(model, xs, y_ref.to(device), formatter, n_steps, out_T)
    chunks = [xs[i:i + 1] for i in range(xs.shape[0])] if per_sample else [xs]
    from .train_forward import fold_scope
    rts, outs = [], []
    with fold_scope():
        for moving in chunks:
            preds, produced = [], 0
            while produced < n_steps:
                y, rt = model(moving, out_T)
                produced += y.shape[1]
                if produced < n_steps:
                    moving = torch.cat([moving[:, y.shape[1]:], y], dim=1)
                preds.append(formatter.process_output(y))
                rts.append(rt)
            outs.append(torch.cat(preds, dim=1)[:, :n_steps])
    return torch.cat(outs, dim=0), y_ref.to(device), torch.cat(rts, dim=0)


def _rollout_adaptive_batched(model, xs: torch.Tensor, y_ref: torch.Tensor, formatter, n_steps: int, out_T: float):
    """R_Trainer.rollout_model's per-sample loop (r_trainer.py:112-133) WITHOUT its serialisation, for any out_T: every call runs all
    samples that still need frames as one batch with per_sample_counts (TANTE.forward); sample i keeps floor(R_t[i]) frames of the call,
    shifts ITS window by that many, and leaves the batch when it has n_steps frames.  Every operator of the path is per sample, so the
    frames are those of the serial loop, and R_t comes back in that loop's order (all calls of sample 0, then sample 1, ...).
    Differentiable: slicing and re-stacking are torch views / copies that autograd tracks."""
    from .train_forward import fold_scope
    B, T = xs.shape[0], model.T
    win = [xs[i, -T:] for i in range(B)]                     # (T, D, H, W) each
    preds = [[] for _ in range(B)]
    rts = [[] for _ in range(B)]
    produced = [0] * B
    with fold_scope():
        while True:
            active = [i for i in range(B) if produced[i] < n_steps]
            if not active:
                break
            y, rt = model(torch.stack([win[i] for i in active], dim=0), out_T, per_sample_counts=True)
            counts = torch.floor(rt.detach()).to(torch.int64).tolist()          # one host read per call (the reference: one per sample)
            for j, i in enumerate(active):
                n = int(counts[j])
                rts[i].append(rt[j:j + 1])
                if n < 1:
                    continue
                yi = y[j, :n]
                preds[i].append(yi)
                produced[i] += n
                if produced[i] < n_steps:
                    win[i] = torch.cat([win[i][n:], yi], dim=0)[-T:]
    out = torch.stack([formatter.process_output(torch.cat(p, dim=0)[:n_steps].unsqueeze(0))[0] for p in preds], dim=0)
    return out, y_ref, torch.cat([torch.cat(r, dim=0) for r in rts], dim=0)
